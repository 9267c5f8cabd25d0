#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -k "pixel_decoder_teacher or c2_step_properties or other_baseline or graph or resplit or batched_qkv" 2>&1 | grep -v Warning | tail -150 > gpurun_out/r2_failed.log
cat gpurun_out/r2_failed.log | cut -c1-400
