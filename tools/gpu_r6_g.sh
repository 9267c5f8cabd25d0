#!/bin/bash
mkdir -p gpurun_out
python tools/debug_dcn_core.py > gpurun_out/r6g_dcn_core.txt 2>&1; tail -5 gpurun_out/r6g_dcn_core.txt
python tools/debug_grad_gap.py > gpurun_out/r6g_grad_gap.txt 2>&1
grep -c "<<<" gpurun_out/r6g_grad_gap.txt; head -3 gpurun_out/r6g_grad_gap.txt
python -m pytest tests -m gpu -q --tb=short --deselect tests/test_gpu_full_size.py > gpurun_out/r6g_pytest_fast.log 2>&1
tail -8 gpurun_out/r6g_pytest_fast.log
python -m pytest tests/test_gpu_full_size.py -m gpu -q --tb=short > gpurun_out/r6g_pytest_full.log 2>&1
tail -6 gpurun_out/r6g_pytest_full.log
