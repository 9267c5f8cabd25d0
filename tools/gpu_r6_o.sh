#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round6.py tests/test_gpu_model.py tests/test_gpu_full_size.py -m gpu -q -x --tb=short -s -k "fanout or decoder or e2e or blocks or c2_stages or glue or stateful" > gpurun_out/r6o_pytest.log 2>&1; tail -15 gpurun_out/r6o_pytest.log; grep "ports on vs off" gpurun_out/r6o_pytest.log
for G in 0 1 0 1; do
  echo "[S2F_FANOUT_PORTS=$G] $(S2F_FANOUT_PORTS=$G python bench.py --no-cpu-baseline --no-kernel-events 2> gpurun_out/r6o_bench_$G.err | grep -o '"ms_per_step": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6o_ab_fanout_ports.txt
timeout 600 python tools/glue_fanout.py > gpurun_out/r6o_glue_fanout.txt 2> gpurun_out/r6o_glue_fanout.err; cat gpurun_out/r6o_glue_fanout.txt
