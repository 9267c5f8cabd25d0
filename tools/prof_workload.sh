# rocprofv3 kernel trace of a bench line other than the default (C3 / C4 / C5, graph replay): categories + per-kernel table
#   bash tools/prof_workload.sh <tag> <bench.py arguments ...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; shift
rm -rf /tmp/prof_$T
rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o s2f -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events "$@" > $R/gpurun_out/prof_$T.log 2>&1
DB=$(find /tmp/prof_$T -name "*.db" | head -1)
python3 $R/tools/rocpd_categories.py $DB 6 > $R/gpurun_out/prof_${T}_categories.txt 2>&1
python3 $R/tools/rocpd_step_kernels.py $DB 6 > $R/gpurun_out/prof_${T}_kernels.txt 2>&1
tail -1 $R/gpurun_out/prof_$T.log | cut -c1-300
cat $R/gpurun_out/prof_${T}_categories.txt
head -45 $R/gpurun_out/prof_${T}_kernels.txt | cut -c1-150
