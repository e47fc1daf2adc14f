#!/bin/bash
# SQ counters of the f16 self-attention kernels (own --pmc passes, kernel trace only).  usage: tools/attn_pmc.sh <outdir>
OUT=${1:-gpurun_out/attn_pmc}; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); rm -rf /tmp/ap$i
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/ap$i -o a -- python $GRAFT_REPO_ROOT/tools/attn_pmc_run.py > $GRAFT_REPO_ROOT/$OUT/pass$i.log 2>&1)
  tail -2 $OUT/pass$i.log
done
python tools/attn_pmc_summarise.py $(find /tmp/ap1 /tmp/ap2 /tmp/ap3 -name '*counter_collection*') | tee $OUT/attn_pmc.txt
