#!/bin/bash
# round 6, call A: the suite as it stands on this round's box (with the per-test fall-back census) and the default bench line
mkdir -p gpurun_out
S2F_FALLBACK_CENSUS=1 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r6a_pytest_gpu.log
python bench.py > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err
tail -3 gpurun_out/r6a_pytest_gpu.log
cut -c1-400 gpurun_out/r6a_bench.json
cat gpurun_out/fallbacks_by_test.json | head -80
