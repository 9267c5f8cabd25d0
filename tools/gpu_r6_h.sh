#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_model.py tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -q --tb=short > gpurun_out/r6h_pytest_fast.log 2>&1
tail -4 gpurun_out/r6h_pytest_fast.log
python -m pytest tests/test_gpu_full_size.py -m gpu -q --tb=short -k "bench_batch or c3_stage or c5_size" > gpurun_out/r6h_pytest_full.log 2>&1
tail -4 gpurun_out/r6h_pytest_full.log; grep "re-seeded stage gaps" gpurun_out/r6h_pytest_full.log | cut -c1-1500
python bench.py > gpurun_out/r6h_bench.json 2> gpurun_out/r6h_bench.err; cut -c1-200 gpurun_out/r6h_bench.json
bash tools/prof_bench.sh r6h > /dev/null 2>&1
cat gpurun_out/prof_r6h_categories.txt
