#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/final_pytest.log 2>&1; tail -3 gpurun_out/final_pytest.log | cut -c1-200
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py 2> gpurun_out/final_bench.err | cut -c1-400
