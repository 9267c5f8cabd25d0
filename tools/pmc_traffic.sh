# HBM traffic of the streaming kernels (two PMC passes) -> gpurun_out/pmc_traffic.json      bash tools/pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o s2f -- python3 $R/bench.py --steps 1 --warmup 1 --no-graph --no-kernel-events --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  tail -1 /tmp/pmc_$c.log | cut -c1-120
done
python3 $R/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) > $R/gpurun_out/pmc_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  python3 $R/tools/pmc_summary.py $(find /tmp/pmc_$c -name "*.db" | head -1) $c | head -40 | cut -c1-200 > $R/gpurun_out/pmc_$c.txt
done
cat $R/gpurun_out/pmc_traffic.json | head -60
