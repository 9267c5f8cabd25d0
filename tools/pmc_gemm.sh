# SQ counters of the spike GEMM kernels on one shape (run on the GPU box): bash tools/pmc_gemm.sh M K L
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcg
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmcg -o s2f -- python3 $R/tools/probe_gemm_one.py $1 $2 $3 > /tmp/pmcg.log 2>&1
DB=$(find /tmp/pmcg -name "*.db" | head -1)
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  python3 $R/tools/pmc_summary.py $DB $c spike_gemm | grep -v "^#" | awk -v c=$c '{print c, $0}' | cut -c1-150
done
rm -rf /tmp/pmcg2
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD -d /tmp/pmcg2 -o s2f -- python3 $R/tools/probe_gemm_one.py $1 $2 $3 > /tmp/pmcg2.log 2>&1
DB=$(find /tmp/pmcg2 -name "*.db" | head -1)
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD; do
  python3 $R/tools/pmc_summary.py $DB $c spike_gemm | grep -v "^#" | awk -v c=$c '{print c, $0}' | cut -c1-150
done
tail -3 /tmp/pmcg2.log
