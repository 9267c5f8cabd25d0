#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --tb=short > gpurun_out/r6v_pytest.log 2>&1; tail -6 gpurun_out/r6v_pytest.log | cut -c1-300
timeout 600 python tools/glue_fanout.py > gpurun_out/r6v_glue_fanout.txt 2> gpurun_out/r6v_glue_fanout.err; cat gpurun_out/r6v_glue_fanout.txt
for G in 0 1 0 1; do
  echo "[S2F_FANOUT_PORTS=$G] $(S2F_FANOUT_PORTS=$G python bench.py --no-cpu-baseline --no-kernel-events 2> gpurun_out/r6v_bench_$G.err | grep -o '"ms_per_step": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6v_ab_fanout_ports.txt
