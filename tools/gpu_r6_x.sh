#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --tb=short > gpurun_out/r6x_pytest.log 2>&1; tail -6 gpurun_out/r6x_pytest.log | cut -c1-300
for i in 1 2; do echo "$(python bench.py --no-cpu-baseline --no-kernel-events 2> gpurun_out/r6x_bench.err | grep -o '"ms_per_step": [0-9.]*')"; done
