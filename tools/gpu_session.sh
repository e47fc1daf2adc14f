#!/bin/bash
# One GPU-box session: parity tests, igemm sweep, per-class step profile, bench line, rocprofv3 kernel stats.
# usage: tools/gpu_session.sh <tag> [parts...]   parts: tests sweep prof bench rocprof
set -u
TAG=${1:-s}; shift || true
PARTS=${*:-tests sweep prof bench rocprof}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for p in $PARTS; do
  case $p in
    vtests) timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "igemm_variants" > $OUT/vtests.log 2>&1; echo "vtests rc=$?" >> $OUT/vtests.log; tail -15 $OUT/vtests.log;;
    counters) rocprofv3 -L > $OUT/counters.txt 2>&1; grep -c . $OUT/counters.txt;;
    pmc) # PMC_ARGS="variant B H W Cin Cout k geglu iters"; one pass per counter group, kernel-trace only
      i=0
      IFS=';' read -ra GRPS <<< "${PMC_GROUPS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum;GRBM_GUI_ACTIVE GRBM_COUNT}"
      for grp in "${GRPS[@]}"; do
        i=$((i+1))
        (cd /tmp && timeout 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- ${PMC_CMD:-python $GRAFT_REPO_ROOT/tools/igemm_one.py $PMC_ARGS} > $GRAFT_REPO_ROOT/$OUT/pmc_${PMC_TAG:-x}_$i.log 2>&1)
        find /tmp/pmc$i -name '*counter_collection*' -exec cp {} $OUT/pmc_${PMC_TAG:-x}_$i.csv \;
      done; ls $OUT;;
    ksweep) timeout 600 python tools/igemm_ksweep.py ${KSWEEP_VARIANTS:-0,11} > $OUT/ksweep.txt 2>&1; cat $OUT/ksweep.txt;;
    attn) timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention or geglu" > $OUT/attn_tests.log 2>&1; tail -3 $OUT/attn_tests.log; timeout 300 python tools/attn_bench.py ${ATTN_VARIANTS:-1,2} > $OUT/attn_bench.txt 2>&1; cat $OUT/attn_bench.txt;;
    traffic) # HBM-side counters of the dominant GEMM shapes (one shape per pass; only the small summary is kept)
      rm -f $OUT/traffic_raw.txt
      for shp in "0 2 32 32 1280 10240 1 1 5" "0 2 32 32 1280 3840 1 0 5" "0 2 32 32 5120 1280 1 0 5" "0 2 32 32 1280 1280 1 0 5" "0 2 128 128 320 320 3 0 5"; do
        for c in FETCH_SIZE WRITE_SIZE; do
          rm -rf /tmp/tr; (cd /tmp && timeout 60 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/igemm_one.py $shp > /tmp/tr.log 2>&1)
          echo "== $shp $c" >> $OUT/traffic_raw.txt
          for f in $(find /tmp/tr -name '*counter_collection*'); do python tools/pmc_parse.py $f >> $OUT/traffic_raw.txt 2>&1; done
          tail -2 /tmp/tr.log | head -1 >> $OUT/traffic_raw.txt
        done
      done; cut -c1-400 $OUT/traffic_raw.txt;;
    tests) timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log;;
    sweep) timeout 600 python tools/igemm_sweep.py ${SWEEP_VARIANTS:--1,0,4,6,1,8} > $OUT/igemm_sweep.txt 2>&1; tail -25 $OUT/igemm_sweep.txt;;
    prof)  SDXL_PROFILE_DUMP=$OUT/step_launches.csv timeout 600 python tools/profile_step.py > $OUT/profile_step.txt 2>&1; tail -3 $OUT/profile_step.txt;;
    bench) timeout 900 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json;;
    rocprof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err); find /tmp/rp -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \; ; head -25 $OUT/kernel_stats.csv;;
  esac
done
