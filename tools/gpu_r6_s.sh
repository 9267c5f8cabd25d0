#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/debug_determinism2.py backbone.block3.0 3 > gpurun_out/r6s_determinism2.txt 2>&1; head -40 gpurun_out/r6s_determinism2.txt | cut -c1-260
