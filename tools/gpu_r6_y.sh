#!/bin/bash
mkdir -p gpurun_out
for G in 0 1 0 1; do
  echo "[S2F_CONV_DW_DIRECT=$G] $(S2F_CONV_DW_DIRECT=$G python bench.py --no-cpu-baseline --no-kernel-events 2> gpurun_out/r6y_bench_$G.err | grep -o '"ms_per_step": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6y_ab_conv_dw_direct.txt
