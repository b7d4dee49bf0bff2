#!/bin/bash
# Several probe builds of the library against the product build inside ONE gpurun call (boxes of the pool differ by a few per cent, so
# only numbers of one call compare): per-shape table tails of tools/layer_profile.py, two interleaved passes.
# usage: [MPX_TILE_RULES=...] tools/ab_variants.sh "<flags of build 1>" "<flags of build 2>" ...     (build 0 = the product library)
# Probe builds live in /tmp and are bound per process by tools/with_lib.py: the product libmpx.so and its stamp are never overwritten.
set -e
ARCH=${ARCH:-resnet101}; B=${BATCH:-2340}; REPS=${REPS:-3}; PAT=${PAT:-"256->1024|conv total"}
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
mkdir -p gpurun_out
LIBS=(network_interpretation_imagenet_amd/libmpx.so)
n=1
for FLAGS in "$@"; do
  ( cd network_interpretation_imagenet_amd/csrc && ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $FLAGS -o /tmp/libmpx_v$n.so mpx_api.hip ) &
  LIBS+=(/tmp/libmpx_v$n.so); n=$((n+1))
done
wait
for pass in 1 2; do
  for i in "${!LIBS[@]}"; do
    python tools/with_lib.py ${LIBS[$i]} tools/layer_profile.py $ARCH $B $REPS > gpurun_out/abv_${i}_$pass.txt 2>&1
    if [ $i = 0 ]; then echo "== build 0 (product), pass $pass"; else j=$((i-1)); a=("$@"); echo "== build $i (${a[$j]}), pass $pass"; fi
    grep -E "$PAT" gpurun_out/abv_${i}_$pass.txt | cut -c1-100
  done
done
