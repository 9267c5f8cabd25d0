# every (kernel, grid) of one family in the default bench's graph replays: bash tools/prof_family.sh <tag> <substring> [...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o s2f -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /tmp/prof_$tag.log 2>&1
DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 $R/tools/rocpd_top.py $DB 7 "$@" > $R/gpurun_out/family_$tag.txt 2>&1
cat $R/gpurun_out/family_$tag.txt
