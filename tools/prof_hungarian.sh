# rocprofv3 kernel trace of the Hungarian-loss bench: bash tools/prof_hungarian.sh <tag> [extra bench flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o s2f -- python3 $R/bench.py --loss hungarian --steps 5 --warmup 2 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 $R/tools/rocpd_categories.py $DB 7 > $R/gpurun_out/prof_${tag}_categories.txt 2>&1
python3 $R/tools/rocpd_stats.py $DB > $R/gpurun_out/prof_${tag}_stats.txt 2>&1
python3 $R/tools/rocpd_glue.py $DB 7 > $R/gpurun_out/prof_${tag}_glue.txt 2>&1
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-300
