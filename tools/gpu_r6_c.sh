#!/bin/bash
mkdir -p gpurun_out
python tools/debug_grad_gap.py > gpurun_out/r6c_grad_gap.txt 2>&1
python tools/debug_grad_gap.py SPIKES_BF16=0 > gpurun_out/r6c_grad_gap_fp32spikes.txt 2>&1
grep -c "<<<" gpurun_out/r6c_grad_gap.txt gpurun_out/r6c_grad_gap_fp32spikes.txt
head -3 gpurun_out/r6c_grad_gap.txt
python -m pytest tests/test_gpu_round6.py -q -x 2>&1 | tail -15
python __graft_entry__.py smoke 2>&1 | tail -3
