#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/glue_census.py > gpurun_out/r6w_glue_census.txt 2> gpurun_out/r6w_glue_census.err
bash tools/prof_bench.sh r6w > /dev/null 2>&1
cat gpurun_out/prof_r6w_categories.txt
DB=$(find /tmp/prof_r6w -name "*.db" | head -1)
python3 tools/rocpd_step_kernels.py $DB 7 > gpurun_out/r6w_step_kernels.txt 2>&1
