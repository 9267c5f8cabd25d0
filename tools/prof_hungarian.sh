# rocprofv3 kernel trace of the Hungarian-loss bench: bash tools/prof_hungarian.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$1
rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o s2f -- python3 $R/bench.py --loss hungarian --steps 5 --warmup 2 > $R/gpurun_out/prof_$1.log 2>&1
DB=$(find /tmp/prof_$1 -name "*.db" | head -1)
python3 $R/tools/rocpd_categories.py $DB 7 > $R/gpurun_out/prof_$1_categories.txt 2>&1
python3 $R/tools/rocpd_stats.py $DB > $R/gpurun_out/prof_$1_stats.txt 2>&1
python3 $R/tools/rocpd_glue.py $DB 7 > $R/gpurun_out/prof_$1_glue.txt 2>&1
tail -1 $R/gpurun_out/prof_$1.log | cut -c1-300
