#!/bin/bash
# one GPU-box call of round 2: parity suite + default bench line
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r2_pytest_gpu.log
tail -40 gpurun_out/r2_pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
cut -c1-1500 gpurun_out/r2_bench.json; tail -5 gpurun_out/r2_bench.err
