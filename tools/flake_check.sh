#!/bin/bash
# the GPU suite several times over: lists every test that failed in any run (tolerances next to run-to-run atomics noise)
mkdir -p gpurun_out
for i in 1 2 3 4; do
  python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/flake_$i.log 2>&1
  tail -1 gpurun_out/flake_$i.log
  grep -h "^FAILED\|Error" gpurun_out/flake_$i.log | head -5
done
