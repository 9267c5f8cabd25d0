# Kernel traces of the default bench under two environments on ONE box, and the per-kernel difference:
#   bash tools/prof_ab_env.sh <tag> "ENV_A=.." "ENV_B=.."      -> gpurun_out/<tag>_{A,B}_kernels.txt, <tag>_diff.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
for V in A B; do
  E="$2"; [ $V = B ] && E="$3"
  rm -rf /tmp/prof_${TAG}_$V
  export $E
  rocprofv3 --kernel-trace -d /tmp/prof_${TAG}_$V -o s2f -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/${TAG}_$V.log 2>&1
  unset ${E%%=*}
  DB=$(find /tmp/prof_${TAG}_$V -name "*.db" | head -1)
  python3 $R/tools/rocpd_step_kernels.py $DB 7 > $R/gpurun_out/${TAG}_${V}_kernels.txt 2>&1
  python3 $R/tools/rocpd_categories.py $DB 7 > $R/gpurun_out/${TAG}_${V}_categories.txt 2>&1
  eval DB_$V=$DB
done
python3 $R/tools/rocpd_step_kernels.py $DB_A 7 $DB_B > $R/gpurun_out/${TAG}_diff.txt 2>&1
head -60 $R/gpurun_out/${TAG}_diff.txt
