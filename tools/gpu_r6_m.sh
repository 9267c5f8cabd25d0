#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round6.py -m gpu -q --tb=short -k "glue_mode" > gpurun_out/r6m_pytest.log 2>&1; tail -12 gpurun_out/r6m_pytest.log
