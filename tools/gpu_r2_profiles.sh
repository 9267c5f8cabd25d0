#!/bin/bash
# Round-2 evidence in one GPU-box call: bench lines of every BASELINE config, kernel trace, PMC traffic + MFMA passes.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
cut -c1-300 gpurun_out/r02_bench_default.json
for wl in C3 C4 C5; do
  timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/r02_bench_$wl.json 2> gpurun_out/r02_bench_$wl.err
  cut -c1-200 gpurun_out/r02_bench_$wl.json
done
timeout 600 python bench.py --loss hungarian --steps 5 --warmup 2 > gpurun_out/r02_bench_hungarian_loss.json 2> gpurun_out/r02_bench_hungarian.err
cut -c1-260 gpurun_out/r02_bench_hungarian_loss.json
bash tools/prof_bench.sh r02 > /dev/null 2>&1
cp gpurun_out/prof_r02_categories.txt gpurun_out/r02_categories.txt
cat gpurun_out/r02_categories.txt
bash tools/pmc_traffic.sh > gpurun_out/r02_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/r02_pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o s2f -- python3 $R/bench.py --steps 1 --warmup 1 --no-graph --no-kernel-events --no-cpu-baseline > /tmp/pmc_mfma.log 2>&1
tail -2 /tmp/pmc_mfma.log | cut -c1-200
python3 $R/tools/pmc_mfma.py $(find /tmp/pmc_mfma -name "*.db" | head -1) > $R/gpurun_out/r02_pmc_mfma.txt 2>&1
cat $R/gpurun_out/r02_pmc_mfma.txt
