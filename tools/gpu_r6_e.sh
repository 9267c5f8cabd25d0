#!/bin/bash
mkdir -p gpurun_out
python tools/debug_grad_gap.py > gpurun_out/r6e_grad_gap.txt 2>&1
grep -c "<<<" gpurun_out/r6e_grad_gap.txt; head -3 gpurun_out/r6e_grad_gap.txt
python -m pytest tests -m gpu -q --deselect tests/test_gpu_full_size.py 2>&1 | tail -40 > gpurun_out/r6e_pytest_fast.log
tail -12 gpurun_out/r6e_pytest_fast.log
python -m pytest tests/test_gpu_full_size.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r6e_pytest_full.log
tail -8 gpurun_out/r6e_pytest_full.log
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > gpurun_out/r6e_bench.json 2> gpurun_out/r6e_bench.err; cut -c1-300 gpurun_out/r6e_bench.json; tail -3 gpurun_out/r6e_bench.err
