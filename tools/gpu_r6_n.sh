#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/glue_fanout.py > gpurun_out/r6n_glue_fanout.txt 2> gpurun_out/r6n_glue_fanout.err; tail -3 gpurun_out/r6n_glue_fanout.err; cat gpurun_out/r6n_glue_fanout.txt
