#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/r6u_pytest.log 2>&1; tail -8 gpurun_out/r6u_pytest.log | cut -c1-300
python __graft_entry__.py smoke 2>&1 | tail -1
