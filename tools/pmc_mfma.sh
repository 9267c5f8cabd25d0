# In-step MFMA utilisation of the GEMM families (one eager C2 step under rocprofv3 --pmc): bash tools/pmc_mfma.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o s2f -- python3 $R/bench.py --steps 1 --warmup 1 --no-graph --no-kernel-events --no-cpu-baseline > /tmp/pmc_mfma.log 2>&1
tail -2 /tmp/pmc_mfma.log | cut -c1-200
python3 $R/tools/pmc_mfma.py $(find /tmp/pmc_mfma -name "*.db" | head -1) > $R/gpurun_out/$1_pmc_mfma.txt 2>&1
cat $R/gpurun_out/$1_pmc_mfma.txt
