#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -s > gpurun_out/r6p_pytest.log 2>&1; tail -15 gpurun_out/r6p_pytest.log; grep "ports on vs off" gpurun_out/r6p_pytest.log
python __graft_entry__.py smoke 2>&1 | tail -2
