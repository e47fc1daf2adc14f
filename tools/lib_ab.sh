# same-box A/B of library builds on the bench line: tools/lib_ab.sh "" path/to/a.so path/to/b.so ...   ("" = the in-tree library)
for round in 1 2; do
for l in "$@"; do
  if [ -n "$l" ]; then export SDXL_LIB_PATH=$PWD/$l; else unset SDXL_LIB_PATH; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib[%s]' % '$l', d['value'], d['unet_step_ms_p50'], d['roofline']['class_ms_per_unet_step'], d['roofline']['frac'], d['outputs_finite'])"
done; done
