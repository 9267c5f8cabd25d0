#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --tb=short > gpurun_out/r6z_pytest.log 2>&1; tail -6 gpurun_out/r6z_pytest.log | cut -c1-300
for G in 0 1 0 1; do
  echo "[S2F_ZERO_ARENA=$G] $(S2F_ZERO_ARENA=$G python bench.py --no-cpu-baseline --no-kernel-events 2> gpurun_out/r6z_bench_$G.err | grep -o '"ms_per_step": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6z_ab_zero_arena.txt
