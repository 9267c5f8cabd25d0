#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_model.py tests/test_gpu_round6.py tests/test_gpu_kernels.py -m gpu -q --tb=short > gpurun_out/r6j_pytest.log 2>&1
tail -8 gpurun_out/r6j_pytest.log
