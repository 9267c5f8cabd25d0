#!/bin/bash
# Round-6 evidence in one GPU-box call: bench lines (default, training iteration, predict, C3-C5, Hungarian loss), kernel trace of the
# default bench, PMC traffic + MFMA passes, GEMM census, glue attribution (the round-5 probes are unchanged kernels: not re-run).   bash tools/gpu_r5_profiles.sh
# FAILS (exit 1) when the PMC traffic profile does not cover the kernel families of the bench line's roofline keys, so that
# `roofline.traffic` cannot silently come from a profile of other kernels.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
T=r06
python bench.py > gpurun_out/${T}_bench_default_pre.json 2> gpurun_out/${T}_bench_default.err
timeout 600 python bench.py --optimizer > gpurun_out/${T}_bench_iteration.json 2>> gpurun_out/${T}_bench_default.err
cut -c1-260 gpurun_out/${T}_bench_iteration.json
timeout 600 python bench.py --mode predict > gpurun_out/${T}_bench_predict.json 2> gpurun_out/${T}_bench_predict.err
cut -c1-260 gpurun_out/${T}_bench_predict.json
timeout 600 python bench.py --mode predict --no-eval-fusion > gpurun_out/${T}_bench_predict_unfused.json 2>> gpurun_out/${T}_bench_predict.err
timeout 600 python bench.py --mode predict --predict-all-layers > gpurun_out/${T}_bench_predict_all_layers.json 2>> gpurun_out/${T}_bench_predict.err
cut -c1-260 gpurun_out/${T}_bench_predict_all_layers.json
for wl in C3 C4 C5; do
  timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/${T}_bench_$wl.json 2> gpurun_out/${T}_bench_$wl.err
  cut -c1-200 gpurun_out/${T}_bench_$wl.json
done
for wl in C3 C5; do          # category + per-kernel tables of the other configurations
  bash tools/prof_workload.sh ${T}_$wl --workload $wl > /dev/null 2>&1
  cd $R
  cp gpurun_out/prof_${T}_${wl}_categories.txt gpurun_out/${T}_categories_$wl.txt
  cp gpurun_out/prof_${T}_${wl}_kernels.txt gpurun_out/${T}_step_kernels_$wl.txt
done
timeout 600 python bench.py --loss hungarian > gpurun_out/${T}_bench_hungarian_loss.json 2> gpurun_out/${T}_bench_hungarian.err
timeout 600 python bench.py --loss hungarian --optimizer > gpurun_out/${T}_bench_iteration_hungarian_loss.json 2>> gpurun_out/${T}_bench_hungarian.err
timeout 600 python bench.py --loss hungarian --gt noise > gpurun_out/${T}_bench_hungarian_loss_noise_maps.json 2>> gpurun_out/${T}_bench_hungarian.err
cut -c1-320 gpurun_out/${T}_bench_hungarian_loss.json
bash tools/prof_bench.sh $T > /dev/null 2>&1
cp gpurun_out/prof_${T}_categories.txt gpurun_out/${T}_categories.txt
cp gpurun_out/prof_${T}_stats.txt gpurun_out/${T}_kernel_stats_graph_replay.txt
cp gpurun_out/prof_${T}_top.txt gpurun_out/${T}_top_kernels_by_grid.txt
cp gpurun_out/prof_${T}_glue.txt gpurun_out/${T}_glue_kernels.txt
cat gpurun_out/${T}_categories.txt
DB=$(find /tmp/prof_$T -name "*.db" | head -1)
python3 $R/tools/rocpd_step_kernels.py $DB 7 > $R/gpurun_out/${T}_step_kernels.txt 2>&1
cd $R
bash tools/pmc_traffic.sh > gpurun_out/${T}_pmc_traffic.log 2>&1
cd $R
cp gpurun_out/pmc_traffic.json gpurun_out/${T}_pmc_traffic.json
cp gpurun_out/pmc_FETCH_SIZE.txt gpurun_out/${T}_pmc_FETCH_SIZE.txt
cp gpurun_out/pmc_WRITE_SIZE.txt gpurun_out/${T}_pmc_WRITE_SIZE.txt
mkdir -p profiles && cp gpurun_out/${T}_pmc_traffic.json profiles/${T}_pmc_traffic.json      # the bench line below reads it
python bench.py > gpurun_out/${T}_bench_default.json 2>> gpurun_out/${T}_bench_default.err
cut -c1-300 gpurun_out/${T}_bench_default.json
python3 - <<'PY' || { echo "STALE OR INCOMPLETE PMC TRAFFIC PROFILE"; exit 1; }
import json
line = json.load(open("gpurun_out/r06_bench_default.json"))
prof = json.load(open("profiles/r06_pmc_traffic.json"))
fams = [v["kernel"] for k, v in line.items() if k.startswith("roofline") and isinstance(v, dict) and v.get("bound") == "hbm" and "min_algorithmic_bytes" not in v]
missing = [f for f in fams if f not in prof["kernels"] or not prof["kernels"][f].get("kernel_names")]
bad = [k for k, v in line.items() if k.startswith("roofline") and isinstance(v, dict) and v.get("bound") == "hbm" and "min_algorithmic_bytes" not in v and v.get("traffic") is None]
assert not missing and not bad, (missing, bad)
print("pmc traffic profile covers", fams)
PY
bash tools/pmc_mfma.sh $T > /dev/null 2>&1
cat gpurun_out/${T}_pmc_mfma.txt
cd $R
timeout 300 python tools/gemm_census.py > gpurun_out/${T}_gemm_census.txt 2> gpurun_out/${T}_gemm_census.err
timeout 300 python tools/glue_sites.py > gpurun_out/${T}_glue_sites.txt 2> gpurun_out/${T}_glue_sites.err
timeout 300 python tools/predict_census.py > gpurun_out/${T}_predict_census.txt 2>&1
timeout 300 python tools/glue_sites.py C2 predict > gpurun_out/${T}_glue_sites_predict.txt 2> /dev/null
bash tools/prof_predict.sh ${T}p > /dev/null 2>&1
cp gpurun_out/prof_${T}p_categories.txt gpurun_out/${T}_categories_predict.txt
cp gpurun_out/prof_${T}p_kernels.txt gpurun_out/${T}_step_kernels_predict.txt
cd $R
