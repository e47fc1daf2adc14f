#!/bin/bash
# Round-3 GPU-box session.  usage: tools/gpu_session_r03.sh <tag> [parts...]
# parts: tests ptests optests bench bench16 cfg1 cfg4 cfg5 rocprof pmc traffic trace sweep timeline vae custom
set -u
TAG=${1:-s}; shift || true
PARTS=${*:-tests bench}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for p in $PARTS; do
  case $p in
    tests) timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=12 > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log; grep -E "passed|failed|error" $OUT/tests.log | tail -5;;
    ptests) timeout 1200 python -m pytest tests/test_gpu_baseline_parity.py -m gpu -q -s > $OUT/ptests.log 2>&1; echo "ptests rc=$?" >> $OUT/ptests.log; grep -E "vs oracle|drift|passed|failed|Error" $OUT/ptests.log | tail -30;;
    optests) timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -s --maxfail=12 > $OUT/optests.log 2>&1; echo "optests rc=$?" >> $OUT/optests.log; grep -E "passed|failed|error" $OUT/optests.log | tail -5;;
    bench) SDXL_PROFILE_DUMP=$OUT/step_launches.csv timeout 900 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json;;
    bench16) timeout 600 python bench.py --steps 2 --warmup 1 --vae-dtype f16 --no-cpu-baseline > $OUT/bench_vae16.json 2> $OUT/bench_vae16.err; tail -c 600 $OUT/bench_vae16.json;;
    cfg1) timeout 900 python bench.py --config 1 --steps 3 --warmup 1 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err; tail -c 1200 $OUT/bench_cfg1.json;;
    cfg4) timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; tail -c 800 $OUT/bench_cfg4.json;;
    cfg5) timeout 900 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; tail -c 800 $OUT/bench_cfg5.json;;
    rocprof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err); find /tmp/rp -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \; ; head -30 $OUT/kernel_stats.csv;;
    pmc) # MFMA-busy / SQ-busy / GRBM counters of the whole bench command (own passes, kernel-trace only)
      i=0
      for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
        i=$((i+1)); rm -rf /tmp/pm$i
        (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pm$i -o p -- python $GRAFT_REPO_ROOT/tools/profile_step.py > $GRAFT_REPO_ROOT/$OUT/pmc_step_$i.log 2>&1)
      done
      python tools/pmc_summarise.py $(find /tmp/pm1 /tmp/pm2 -name '*counter_collection*') > $OUT/pmc_step.json 2>&1; tail -40 $OUT/pmc_step.json;;
    traffic) # HBM-side bytes of one UNet step (separate --pmc passes, kernel trace only; FETCH_SIZE doubled per MI355X_MICROARCH.md)
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/tr_$c; (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $GRAFT_REPO_ROOT/tools/profile_step.py > $GRAFT_REPO_ROOT/$OUT/traffic_$c.log 2>&1)
      done
      python tools/pmc_traffic.py $OUT/pmc_traffic.json $(find /tmp/tr_FETCH_SIZE -name '*counter_collection*' | head -1) $(find /tmp/tr_WRITE_SIZE -name '*counter_collection*' | head -1) 3;;   # 2 trajectory iterations + 1 profiled step
    trace) # per-kernel durations of ONE replayed UNet step + kernel-to-kernel gaps (kernel trace only)
      rm -rf /tmp/kt; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > /dev/null 2>&1)
      f=$(find /tmp/kt -name '*kernel_trace*' | head -1)
      python tools/trace_step_summary.py $f > $OUT/step_kernels.txt 2>&1; python tools/trace_gaps.py $f $OUT/trace_gaps.json > /dev/null 2>&1; head -20 $OUT/step_kernels.txt;;
    sweep) timeout 600 python tools/igemm_sweep.py ${SWEEP_VARIANTS:-0} > $OUT/igemm_sweep.txt 2>&1; tail -25 $OUT/igemm_sweep.txt;;
    timeline) SDXL_MEASURE_LIB=1 timeout 300 python tools/timeline_probe.py $OUT/timeline_probe.json > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
      echo "--- LDS-staged epilogue"; EPI_STAGED=1 SDXL_MEASURE_LIB=1 timeout 300 python tools/timeline_probe.py $OUT/timeline_probe_staged.json 2>&1 | tee $OUT/timeline_staged.txt | grep -E "production|prologue:";;
    timeline_kernarg) for kv in 0 1; do echo "--- HIP_FORCE_DEV_KERNARG=$kv"; HIP_FORCE_DEV_KERNARG=$kv SDXL_MEASURE_LIB=1 timeout 300 python tools/timeline_probe.py $OUT/timeline_probe_kernarg$kv.json 2>&1 | tee $OUT/timeline_kernarg$kv.txt | grep -E "production|prologue:"; done;;
    vae) timeout 600 python tools/vae_bench.py > $OUT/vae_bench.txt 2>&1; tail -30 $OUT/vae_bench.txt;;
    benchab) # A/B of a debug knob on the bench line: AB_KNOB="igemm_epilogue_staged=1"
      for kn in "" "${AB_KNOB:-}" "" "${AB_KNOB:-}"; do SDXL_DEBUG_SET="$kn" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('knob[%s]' % '$kn', d['value'], d['unet_step_ms_p50'], d['roofline']['class_ms_per_unet_step'], d['roofline']['frac'])"; done | tee $OUT/benchab.txt;;
    custom) bash -c "${CUSTOM_CMD}" > $OUT/custom.log 2>&1; tail -40 $OUT/custom.log;;
  esac
done
