#!/bin/bash
# round 6, call B: the whole GPU suite under STRICT (every test) + smoke()
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_full_size.py 2>&1 | tail -40 > gpurun_out/r6b_pytest_fast.log
tail -5 gpurun_out/r6b_pytest_fast.log
python -m pytest tests/test_gpu_full_size.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r6b_pytest_full.log
tail -5 gpurun_out/r6b_pytest_full.log
python __graft_entry__.py smoke 2>&1 | tail -3 > gpurun_out/r6b_smoke.log
cat gpurun_out/r6b_smoke.log
cat gpurun_out/fallbacks_by_test.json
