#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round6.py -m gpu -q --tb=short -k "glue_mode" > gpurun_out/r6l_pytest.log 2>&1; tail -12 gpurun_out/r6l_pytest.log
for G in 0 1 0 1; do
  echo "[S2F_GLUE_MODE=$G] $(S2F_GLUE_MODE=$G python bench.py --no-cpu-baseline --no-kernel-events 2> gpurun_out/r6l_bench_$G.err | grep -o '"ms_per_step": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6l_ab_glue_mode.txt
export S2F_GLUE_MODE=1
bash tools/prof_bench.sh r6l > /dev/null 2>&1
cat gpurun_out/prof_r6l_categories.txt
DB=$(find /tmp/prof_r6l -name "*.db" | head -1)
python3 tools/rocpd_step_kernels.py $DB 7 > gpurun_out/r6l_step_kernels.txt 2>&1
grep -n "ew_\|reduce_\|fill_kernel\|at::native" gpurun_out/r6l_step_kernels.txt | head -30
