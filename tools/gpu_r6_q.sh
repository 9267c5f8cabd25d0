#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round6.py -m gpu -q --tb=short -s -k fanout > gpurun_out/r6q_pytest.log 2>&1; tail -15 gpurun_out/r6q_pytest.log | cut -c1-300; grep "ports on vs off" gpurun_out/r6q_pytest.log
