#!/bin/bash
# Round-6 GPU-box session.  usage: tools/gpu_session_r06.sh <tag> [parts...]   (parts: tests smoke bench benchf16w benchg2 bench7 benchmix cfg1 cfg4 cfg5 cfg5f16w trace rocprof rocproff16w pmc traffic)
set -u
TAG=${1:-s}; shift || true
PARTS=${*:-tests smoke bench}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
line() { python -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'img/s step', d['unet_step_ms_p50'], 'ms', r['class_ms_per_unet_step'], 'frac', r['frac'], r.get('frac_event_overhead_removed'), 'finite', d['outputs_finite'])"; }
for p in $PARTS; do
  case $p in
    tests) timeout 2400 python -m pytest tests -m gpu -q -s --maxfail=12 > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log; grep -E "passed|failed|error" $OUT/tests.log | tail -5;;
    smoke) timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log;;
    bench) SDXL_PROFILE_DUMP=$OUT/step_launches.csv timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; line $OUT/bench.json f16; python -c "import json; d=json.load(open('$OUT/bench.json')); [print(' strict', k, v.get('images_per_sec'), v.get('unet_step_ms'), v.get('config2_final_latent_max_abs_vs_oracle'), v.get('inside_lat_bound_scaled')) for k, v in (d.get('strict_f32') or {}).items()]";;
    benchf16w) SDXL_PROFILE_DUMP=$OUT/step_launches_f16w.csv timeout 900 python bench.py --dtype f32_split_mix_f16w --weights f16 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_f16w.json 2> $OUT/bench_f16w.err; line $OUT/bench_f16w.json f16w; python -c "import json; d=json.load(open('$OUT/bench_f16w.json')); print(' timed engine parity', d['parity']['live'].get('timed_engine'))";;
    benchg2) timeout 900 python bench.py --dtype f32_split_mix_f16w_geglu2 --weights f16 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_geglu2.json 2> $OUT/bench_geglu2.err; line $OUT/bench_geglu2.json f16w-geglu2; python -c "import json; d=json.load(open('$OUT/bench_geglu2.json')); print(' timed engine parity', d['parity']['live'].get('timed_engine'))";;
    bench7) timeout 900 python bench.py --dtype f32_split_f16w --weights f16 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_f32_split_f16w.json 2> $OUT/bench_f32_split_f16w.err; line $OUT/bench_f32_split_f16w.json f32_split_f16w; python -c "import json; d=json.load(open('$OUT/bench_f32_split_f16w.json')); print(' timed engine parity', d['parity']['live'].get('timed_engine'))";;
    benchmix) timeout 900 python bench.py --dtype f32_split_mix --steps 3 --warmup 1 --no-cpu-baseline --no-live-parity > $OUT/bench_mix.json 2> $OUT/bench_mix.err; line $OUT/bench_mix.json f32_split_mix;;
    cfg1) timeout 900 python bench.py --config 1 --steps 3 --warmup 1 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err; line $OUT/bench_cfg1.json cfg1;;
    cfg4) timeout 900 python bench.py --config 4 --steps 2 --warmup 1 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; line $OUT/bench_cfg4.json cfg4;;
    cfg5) timeout 900 python bench.py --config 5 --steps 1 --warmup 1 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; line $OUT/bench_cfg5.json cfg5;;
    cfg4f16w) timeout 900 python bench.py --config 4 --dtype f32_split_mix_f16w --weights f16 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg4_f16w.json 2> $OUT/bench_cfg4_f16w.err; line $OUT/bench_cfg4_f16w.json cfg4-f16w;;
    cfg5f16w) timeout 900 python bench.py --config 5 --dtype f32_split_mix_f16w --weights f16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg5_f16w.json 2> $OUT/bench_cfg5_f16w.err; line $OUT/bench_cfg5_f16w.json cfg5-f16w;;
    trace) rm -rf /tmp/kt; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > /dev/null 2>&1)
      f=$(find /tmp/kt -name '*kernel_trace*' | head -1)
      python tools/trace_step_summary.py $f > $OUT/step_kernels.txt 2>&1; python tools/trace_gaps.py $f $OUT/trace_gaps.json > /dev/null 2>&1; head -24 $OUT/step_kernels.txt;;
    rocprof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err); find /tmp/rp -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \; ; head -16 $OUT/kernel_stats.csv | cut -c1-170;;
    rocproff16w) rm -rf /tmp/rpw; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpw -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype f32_split_mix_f16w --weights f16 --steps 1 --warmup 1 --no-cpu-baseline --no-live-parity > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_f16w.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof_f16w.err); find /tmp/rpw -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats_f16w.csv \; ; f=$(find /tmp/rpw -name '*kernel_trace*' | head -1); python tools/trace_step_summary.py $f > $OUT/step_kernels_f16w.txt 2>&1; head -14 $OUT/step_kernels_f16w.txt | cut -c1-170;;
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
    custom) bash -c "${CUSTOM_CMD}" > $OUT/custom.log 2>&1; tail -60 $OUT/custom.log;;
  esac
done
