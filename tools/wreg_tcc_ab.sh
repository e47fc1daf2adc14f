#!/bin/bash
# L2 counters of the weights-in-registers GEMM inside one eager UNet step, XCD ownership of tiles: whole row-tile runs (production) vs 2-D patches
# (knob wreg_xcd2d=1).  Separate --pmc passes, kernel trace only.  usage: tools/wreg_tcc_ab.sh <outdir>
OUT=${1:-gpurun_out/wreg_tcc}; mkdir -p $OUT
export TMPDIR=/tmp
for knob in 0 1; do
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1)); d=/tmp/tcc_${knob}_$i; rm -rf $d
    (cd /tmp && SDXL_DEBUG_SET="wreg_xcd2d=$knob" timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o t -- python $GRAFT_REPO_ROOT/tools/profile_step.py > $GRAFT_REPO_ROOT/$OUT/pass_${knob}_$i.log 2>&1)
  done
done
python tools/tcc_summarise.py $(for knob in 0 1; do for i in 1 2 3 4; do find /tmp/tcc_${knob}_$i -name '*counter_collection*' | head -1 | sed "s/^/$knob:/"; done; done) | tee $OUT/wreg_tcc_ab.txt
