#!/bin/bash
# GlueMode: which aten calls a C2 step still makes, the routed step against the ATen step, and the bench with / without it
mkdir -p gpurun_out
timeout 600 python tools/glue_census.py > gpurun_out/r6k_glue_census.txt 2> gpurun_out/r6k_glue_census.err; tail -3 gpurun_out/r6k_glue_census.err; wc -l gpurun_out/r6k_glue_census.txt
python -m pytest tests/test_gpu_round6.py -m gpu -q --tb=short -k "glue_mode" > gpurun_out/r6k_pytest.log 2>&1; tail -25 gpurun_out/r6k_pytest.log
for G in 0 1 0 1; do
  echo "[S2F_GLUE_MODE=$G] $(S2F_GLUE_MODE=$G python bench.py --no-cpu-baseline --no-kernel-events 2> gpurun_out/r6k_bench_$G.err | grep -o '"ms_per_step": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6k_ab_glue_mode.txt
tail -5 gpurun_out/r6k_bench_1.err
