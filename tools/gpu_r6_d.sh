#!/bin/bash
mkdir -p gpurun_out
python tools/debug_dcn_gap.py > gpurun_out/r6d_dcn_gap.txt 2>&1
cat gpurun_out/r6d_dcn_gap.txt | tail -60
python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -15
