#!/bin/bash
# Round-4 GPU-box session.  usage: tools/gpu_session_r04.sh <tag> [parts...]
set -u
TAG=${1:-s}; shift || true
PARTS=${*:-tests bench}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for p in $PARTS; do
  case $p in
    wregtests) timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "wreg or test_linear or layer_norm_linear or ln_query or igemm_variants_linear" --maxfail=8 > $OUT/wregtests.log 2>&1; echo "rc=$?" >> $OUT/wregtests.log; grep -E "passed|failed|error|Error|wrong|differ" $OUT/wregtests.log | tail -15;;
    knock) SDXL_MEASURE_LIB=1 timeout 600 python tools/wreg_knockout.py > $OUT/wreg_knockout.txt 2> $OUT/wreg_knockout.err; tail -3 $OUT/wreg_knockout.err; cat $OUT/wreg_knockout.txt | tail -12;;
    hazard) SDXL_MEASURE_LIB=1 timeout 900 python tools/hazard_xa_probe.py ${HAZ_N:-500} > $OUT/hazard_xa.txt 2>&1; cat $OUT/hazard_xa.txt | tail -10;;
    newtests) timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -s -k "outside_the_f16_range or token_count or wreg or vae or split_operand" --maxfail=8 > $OUT/newtests.log 2>&1; echo "rc=$?" >> $OUT/newtests.log; grep -E "passed|failed|rel err|Error" $OUT/newtests.log | tail -25;;
    widetl) SDXL_MEASURE_LIB=1 timeout 300 python tools/wide_timeline.py > $OUT/wide_timeline.txt 2>&1; cat $OUT/wide_timeline.txt;;
    attnko) SDXL_MEASURE_LIB=1 timeout 600 python tools/attn_knockout.py > $OUT/attn_knockout.txt 2>&1; cat $OUT/attn_knockout.txt;;
    attntl) SDXL_MEASURE_LIB=1 timeout 300 python tools/attn_timeline.py > $OUT/attn_timeline.txt 2>&1; cat $OUT/attn_timeline.txt;;
    attntests) timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "attention" --maxfail=8 > $OUT/attntests.log 2>&1; echo "rc=$?" >> $OUT/attntests.log; grep -E "passed|failed|pipelined vs serial|Error" $OUT/attntests.log | tail -14;;
    attnab) for v in 10 6 10 6; do SDXL_DEBUG_SET=attn_variant=$v python -c "import os,sys; sys.path.insert(0, \".\"); import __graft_entry__ as ge; pkg=ge.load_package(); ctx=pkg.Context(0); pkg.debug_set(\"attn_variant\", $v); print(\"attn_variant $v: 32^2 self-attention\", round(min(pkg.bench_attention(ctx,2,20,1024,1024,50) for _ in range(3))*1e3,2), \"us\")"; done 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_ab.txt;;
    wregtl) SDXL_MEASURE_LIB=1 timeout 300 python tools/wreg_timeline.py > $OUT/wreg_timeline.txt 2>&1; cat $OUT/wreg_timeline.txt;;
    wregdepth) SDXL_MEASURE_LIB=1 timeout 300 python tools/wreg_depth_ab.py > $OUT/wreg_depth.txt 2>&1; cat $OUT/wreg_depth.txt;;
    libab) bash tools/lib_ab.sh ${LIB_AB} 2>&1 | tee $OUT/lib_ab.txt;;
    fstests) timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -x > $OUT/fstests.log 2>&1; echo "rc=$?" >> $OUT/fstests.log; grep -E "passed|failed|Error|assert" $OUT/fstests.log | tail -8;;
    icache) SDXL_MEASURE_LIB=1 timeout 300 python tools/icache_probe.py > $OUT/icache_probe.txt 2>&1; cat $OUT/icache_probe.txt;;
    wregab) SDXL_MEASURE_LIB=1 timeout 600 python tools/wreg_ab.py ${WREG_ITERS:-20} > $OUT/wreg_ab.txt 2>&1; cat $OUT/wreg_ab.txt | tail -14;;
    tests) timeout 1800 python -m pytest tests -m gpu -q -s --maxfail=12 > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log; grep -E "passed|failed|error" $OUT/tests.log | tail -5;;
    ptests) timeout 1200 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fullsize.py -m gpu -q -s > $OUT/ptests.log 2>&1; echo "ptests rc=$?" >> $OUT/ptests.log; grep -E "vs oracle|drift|passed|failed|Error" $OUT/ptests.log | tail -30;;
    bench) SDXL_PROFILE_DUMP=$OUT/step_launches.csv timeout 900 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1800 $OUT/bench.json;;
    benchab) # A/B of a debug knob on the bench line: AB_KNOB="igemm_wreg=0"
      for kn in "" "${AB_KNOB:-}" "" "${AB_KNOB:-}"; do SDXL_PROFILE_DUMP=$OUT/launches_${kn:-default}.csv SDXL_DEBUG_SET="$kn" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('knob[%s]' % '$kn', d['value'], d['unet_step_ms_p50'], d['roofline']['class_ms_per_unet_step'], d['roofline']['frac'], d['outputs_finite'])"; done | tee $OUT/benchab.txt
      [ -n "${AB_KNOB:-}" ] && python tools/launch_ab.py "$OUT/launches_${AB_KNOB}.csv" $OUT/launches_default.csv > $OUT/launch_ab.txt 2>&1 && tail -30 $OUT/launch_ab.txt;;
    clock) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o /tmp/clock_probe > $OUT/clock.txt 2>&1; timeout 60 /tmp/clock_probe >> $OUT/clock.txt 2>&1; cat $OUT/clock.txt;;
    rocprof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err); find /tmp/rp -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \; ; head -30 $OUT/kernel_stats.csv;;
    rocprof4) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp4 -o r -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > $GRAFT_REPO_ROOT/$OUT/rocprof_cfg4.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof_cfg4.err); find /tmp/rp4 -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_cfg4.csv \; ; head -24 $OUT/kernel_stats_cfg4.csv;;
    trace) rm -rf /tmp/kt; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > /dev/null 2>&1)
      f=$(find /tmp/kt -name '*kernel_trace*' | head -1)
      python tools/trace_step_summary.py $f > $OUT/step_kernels.txt 2>&1; python tools/trace_gaps.py $f $OUT/trace_gaps.json > /dev/null 2>&1; head -24 $OUT/step_kernels.txt;;
    pmc) i=0
      for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
        i=$((i+1)); rm -rf /tmp/pm$i
        (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pm$i -o p -- python $GRAFT_REPO_ROOT/tools/profile_step.py > $GRAFT_REPO_ROOT/$OUT/pmc_step_$i.log 2>&1)
      done
      python tools/pmc_summarise.py $(find /tmp/pm1 /tmp/pm2 -name '*counter_collection*') > $OUT/pmc_step.json 2>&1; tail -40 $OUT/pmc_step.json;;
    traffic) for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/tr_$c; (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $GRAFT_REPO_ROOT/tools/profile_step.py > $GRAFT_REPO_ROOT/$OUT/traffic_$c.log 2>&1)
      done
      python tools/pmc_traffic.py $OUT/pmc_traffic.json $(find /tmp/tr_FETCH_SIZE -name '*counter_collection*' | head -1) $(find /tmp/tr_WRITE_SIZE -name '*counter_collection*' | head -1) 3;;
    cfg1) timeout 900 python bench.py --config 1 --steps 3 --warmup 1 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err; tail -c 1200 $OUT/bench_cfg1.json;;
    cfg4) timeout 900 python bench.py --config 4 --steps 2 --warmup 1 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; tail -c 1200 $OUT/bench_cfg4.json;;
    cfg5) timeout 900 python bench.py --config 5 --steps 1 --warmup 1 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; tail -c 1200 $OUT/bench_cfg5.json;;
    custom) bash -c "${CUSTOM_CMD}" > $OUT/custom.log 2>&1; tail -40 $OUT/custom.log;;
  esac
done
