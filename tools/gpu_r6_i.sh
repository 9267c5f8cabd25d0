#!/bin/bash
# same-box A/B of the shifted-sum BatchNorm statistics (libs2f_hip.so) against the plain form (libs2f_noshift.so), then the whole suite
mkdir -p gpurun_out
for V in hip noshift hip noshift hip noshift; do
  echo "[$V] $(S2F_LIB=$GRAFT_REPO_ROOT/spike2former_amd/libs2f_$V.so python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done 2>&1 | tee gpurun_out/r6i_ab_bn_shifted.txt
python -m pytest tests -m gpu -q --tb=short > gpurun_out/r6i_pytest_all.log 2>&1
tail -6 gpurun_out/r6i_pytest_all.log
python __graft_entry__.py smoke 2>&1 | tail -1
cat gpurun_out/fallbacks_by_test.json
