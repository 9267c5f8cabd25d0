#!/bin/bash
mkdir -p gpurun_out
python tools/debug_dcn_core.py > gpurun_out/r6f_dcn_core.txt 2>&1; tail -30 gpurun_out/r6f_dcn_core.txt
