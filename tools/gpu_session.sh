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
    tests) timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log;;
    sweep) timeout 600 python tools/igemm_sweep.py ${SWEEP_VARIANTS:--1,0,4,6,1,8} > $OUT/igemm_sweep.txt 2>&1; tail -25 $OUT/igemm_sweep.txt;;
    prof)  SDXL_PROFILE_DUMP=$OUT/step_launches.csv timeout 600 python tools/profile_step.py > $OUT/profile_step.txt 2>&1; tail -3 $OUT/profile_step.txt;;
    bench) timeout 900 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json;;
    rocprof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err); find /tmp/rp -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \; ; head -25 $OUT/kernel_stats.csv;;
  esac
done
