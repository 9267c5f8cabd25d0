#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/debug_determinism.py C2 3 > gpurun_out/r6r_determinism.txt 2>&1; head -60 gpurun_out/r6r_determinism.txt | cut -c1-200
