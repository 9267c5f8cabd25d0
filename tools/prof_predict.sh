# rocprofv3 kernel trace of the inference bench (eval mode, hipGraph replay): bash tools/prof_predict.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$1
rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o s2f -- python3 $R/bench.py --mode predict --steps 5 --warmup 2 > $R/gpurun_out/prof_$1.log 2>&1
DB=$(find /tmp/prof_$1 -name "*.db" | head -1)
export S2F_STEP_MARKER=dcn_fwd
python3 $R/tools/rocpd_categories.py $DB 7 > $R/gpurun_out/prof_$1_categories.txt 2>&1
python3 $R/tools/rocpd_step_kernels.py $DB 7 > $R/gpurun_out/prof_$1_kernels.txt 2>&1
tail -1 $R/gpurun_out/prof_$1.log | cut -c1-300
cat $R/gpurun_out/prof_$1_categories.txt
head -50 $R/gpurun_out/prof_$1_kernels.txt | cut -c1-150
