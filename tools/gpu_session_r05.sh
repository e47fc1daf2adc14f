#!/bin/bash
# Round-5 GPU-box session.  usage: tools/gpu_session_r05.sh <tag> [parts...]
set -u
TAG=${1:-s}; shift || true
PARTS=${*:-tests bench}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
benchline() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['unet_step_ms_p50'], r['class_ms_per_unet_step'], 'sum', r.get('class_sum_ms'), 'frac', r['frac'], d['outputs_finite'])"; }
for p in $PARTS; do
  case $p in
    tests) timeout 1800 python -m pytest tests -m gpu -q -s --maxfail=12 > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log; grep -E "passed|failed|error" $OUT/tests.log | tail -5;;
    ktests) timeout 1200 python -m pytest tests -m gpu -q -s --maxfail=12 -k "${KEXPR}" > $OUT/ktests.log 2>&1; echo "ktests rc=$?" >> $OUT/ktests.log; grep -E "passed|failed|error|Error|assert" $OUT/ktests.log | tail -12;;
    ptests) timeout 1200 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fullsize.py -m gpu -q -s > $OUT/ptests.log 2>&1; echo "ptests rc=$?" >> $OUT/ptests.log; grep -E "vs oracle|drift|passed|failed|Error" $OUT/ptests.log | tail -30;;
    smoke) timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log;;
    bench) SDXL_PROFILE_DUMP=$OUT/step_launches.csv timeout 900 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json;;
    splitcfg) for fl in "" "--split-cfg" "" "--split-cfg"; do timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-parity $fl 2>/dev/null | tee -a $OUT/splitcfg_lines.json | benchline "cfg[$fl]"; done | tee $OUT/splitcfg.txt;;
    frontier) timeout 1200 python tools/precision_frontier.py $OUT/precision_frontier.json > $OUT/frontier.log 2>&1; grep frontier $OUT/frontier.log | tail -60; tail -3 $OUT/frontier.log;;
    soak) timeout 1200 python tools/ticket_soak.py ${SOAK_N:-500} > $OUT/ticket_soak.txt 2>&1; cat $OUT/ticket_soak.txt | tail -24;;
    benchab) # A/B of a debug knob on the bench line: AB_KNOB="igemm_wreg=0"
      for kn in "" "${AB_KNOB:-}" "" "${AB_KNOB:-}"; do SDXL_DEBUG_SET="$kn" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-parity 2>/dev/null | benchline "knob[$kn]"; done | tee $OUT/benchab.txt;;
    libab) bash tools/lib_ab.sh ${LIB_AB} 2>&1 | tee $OUT/lib_ab.txt;;
    rocprof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err); find /tmp/rp -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \; ; head -30 $OUT/kernel_stats.csv;;
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
    ppc) for n in 2 4; do timeout 900 python bench.py --prompts-per-call $n --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_ppc$n.json 2> $OUT/bench_ppc$n.err; python -c "import json; d=json.load(open('$OUT/bench_ppc$n.json')); print('prompts per call $n:', d['value'], d['unet_step_ms_p50'], d['roofline']['frac'])"; done;;
    rocprofmix) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpm -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype f32_split_mix --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_mix.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof_mix.err); find /tmp/rpm -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_mix.csv \; ; head -16 $OUT/kernel_stats_mix.csv | cut -c1-160;;
    custom) bash -c "${CUSTOM_CMD}" > $OUT/custom.log 2>&1; tail -60 $OUT/custom.log;;
  esac
done
