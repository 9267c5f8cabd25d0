#!/bin/bash
# Round-4 evidence in one GPU-box call: bench lines (default, predict, C3-C5, Hungarian loss with two kinds of synthetic maps), kernel
# trace of the default bench, PMC traffic + MFMA passes, the GEMM census and the glue attribution.   bash tools/gpu_r4_profiles.sh
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
cut -c1-300 gpurun_out/r04_bench_default.json
timeout 600 python bench.py --mode predict > gpurun_out/r04_bench_predict.json 2> gpurun_out/r04_bench_predict.err
cut -c1-260 gpurun_out/r04_bench_predict.json
timeout 600 python bench.py --mode predict --no-eval-fusion > gpurun_out/r04_bench_predict_unfused.json 2> gpurun_out/r04_bench_predict_unfused.err
cut -c1-260 gpurun_out/r04_bench_predict_unfused.json
for wl in C3 C4 C5; do
  timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/r04_bench_$wl.json 2> gpurun_out/r04_bench_$wl.err
  cut -c1-200 gpurun_out/r04_bench_$wl.json
done
timeout 600 python bench.py --loss hungarian > gpurun_out/r04_bench_hungarian_loss.json 2> gpurun_out/r04_bench_hungarian.err
cut -c1-320 gpurun_out/r04_bench_hungarian_loss.json
timeout 600 python bench.py --loss hungarian --gt noise > gpurun_out/r04_bench_hungarian_loss_noise_maps.json 2> gpurun_out/r04_bench_hungarian_noise.err
cut -c1-320 gpurun_out/r04_bench_hungarian_loss_noise_maps.json
timeout 600 python bench.py --loss hungarian --hungarian-graphs split --gt noise > gpurun_out/r04_bench_hungarian_loss_eager_loss.json 2> gpurun_out/r04_bench_hungarian_split.err
cut -c1-320 gpurun_out/r04_bench_hungarian_loss_eager_loss.json
bash tools/prof_bench.sh r04 > /dev/null 2>&1
cp gpurun_out/prof_r04_categories.txt gpurun_out/r04_categories.txt
cp gpurun_out/prof_r04_stats.txt gpurun_out/r04_kernel_stats_graph_replay.txt
cp gpurun_out/prof_r04_top.txt gpurun_out/r04_top_kernels_by_grid.txt
cp gpurun_out/prof_r04_glue.txt gpurun_out/r04_glue_kernels.txt
cat gpurun_out/r04_categories.txt
bash tools/prof_hungarian.sh r04h > /dev/null 2>&1
cp gpurun_out/prof_r04h_categories.txt gpurun_out/r04_categories_hungarian_step.txt
bash tools/pmc_traffic.sh > gpurun_out/r04_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/r04_pmc_traffic.json
cp gpurun_out/pmc_FETCH_SIZE.txt gpurun_out/r04_pmc_FETCH_SIZE.txt
cp gpurun_out/pmc_WRITE_SIZE.txt gpurun_out/r04_pmc_WRITE_SIZE.txt
bash tools/pmc_mfma.sh r04 > /dev/null 2>&1
cat gpurun_out/r04_pmc_mfma.txt
cd $R
timeout 300 python tools/gemm_census.py > gpurun_out/r04_gemm_census.txt 2> gpurun_out/r04_gemm_census.err
timeout 300 python tools/glue_sites.py > gpurun_out/r04_glue_sites.txt 2> gpurun_out/r04_glue_sites.err

DB=$(find /tmp/prof_r04 -name "*.db" | head -1)
python3 $R/tools/rocpd_step_kernels.py $DB 7 > $R/gpurun_out/r04_step_kernels.txt 2>&1
bash tools/prof_predict.sh r04p > /dev/null 2>&1
cp gpurun_out/prof_r04p_categories.txt gpurun_out/r04_categories_predict.txt
cp gpurun_out/prof_r04p_kernels.txt gpurun_out/r04_step_kernels_predict.txt
cd $R
bash tools/prof_ab_env.sh r04_partials S2F_BN_PARTIALS=0 S2F_BN_PARTIALS=1 > /dev/null 2>&1
cd $R
bash tools/prof_ab_env.sh r04_bn2 S2F_BN2_FUSED=0 S2F_BN2_FUSED=1 > /dev/null 2>&1
cd $R
bash tools/prof_ab_env.sh r04_rows S2F_BN_ROWS_CHUNK=0 S2F_BN_ROWS_CHUNK=8,4 > /dev/null 2>&1
cd $R
python tools/probe_bn_stream.py > gpurun_out/r04_bn_stream_probe.txt 2>&1
S2F_BN_ROWS_CHUNK=0 python tools/probe_bn_stream.py > gpurun_out/r04_bn_stream_probe_resident_grid.txt 2>&1
