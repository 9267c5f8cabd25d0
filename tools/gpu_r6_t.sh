#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round6.py tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short -s > gpurun_out/r6t_pytest.log 2>&1; tail -8 gpurun_out/r6t_pytest.log | cut -c1-300; grep "ports on vs off" gpurun_out/r6t_pytest.log
timeout 600 python tools/glue_fanout.py > gpurun_out/r6t_glue_fanout.txt 2> gpurun_out/r6t_glue_fanout.err; cat gpurun_out/r6t_glue_fanout.txt
for G in 0 1 0 1; do
  echo "[S2F_FANOUT_PORTS=$G] $(S2F_FANOUT_PORTS=$G python bench.py --no-cpu-baseline --no-kernel-events 2> gpurun_out/r6t_bench_$G.err | grep -o '"ms_per_step": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6t_ab_fanout_ports.txt
