# same-box A/B of library builds on the attention micro-benchmark: tools/attn_lib_ab.sh "B H N variants" a.so b.so ...
SHAPES="$1"; shift
for round in 1 2 3; do
for l in "$@"; do
  export SDXL_LIB_PATH=$PWD/$l
  echo "lib[$l] $(python tools/attn_variant_times.py $SHAPES 2>&1 | grep -v amdgpu.ids | tail -1)"
done; done
