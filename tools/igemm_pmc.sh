#!/bin/bash
# SQ counters of single implicit-GEMM launches (own --pmc passes, kernel trace only).  usage: tools/igemm_pmc.sh <outdir>
OUT=${1:-gpurun_out/igemm_pmc}; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/ip$i
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/ip$i -o a -- python $GRAFT_REPO_ROOT/tools/igemm_pmc_run.py > $GRAFT_REPO_ROOT/$OUT/pass$i.log 2>&1)
done
grep "us" $OUT/pass1.log
python tools/igemm_pmc_summarise.py $(find /tmp/ip1 /tmp/ip2 /tmp/ip3 -name '*counter_collection*') | tee $OUT/igemm_pmc.txt
