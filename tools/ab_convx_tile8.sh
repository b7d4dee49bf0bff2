#!/bin/bash
# VERDICT r3 item 5, the one decisive experiment on the expanding 1x1 layers (256->1024 x22, 128->512 x3): two co-resident 4-wave
# workgroups per CU on a 128 cout x 128 pixel PERSISTENT tile with a register epilogue (tools/probes/experimental/mpx_convp.h, tile 8:
# fixed grid of 2 workgroups per CU, next tile's prologue issued before the epilogue) against the default tile 10, in the network,
# two interleaved passes in ONE call.  Both sides run the same probe build (tools/probes/libmpx_experimental.so); only the rule differs.
set -e
ARCH=${1:-resnet101}; B=${2:-2340}; REPS=${3:-3}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/probes/build_experimental.sh > /dev/null
L=tools/probes/libmpx_experimental.so
for pass in 1 2; do
  python tools/with_lib.py $L tools/layer_profile.py $ARCH $B $REPS > gpurun_out/abx_A_$pass.txt 2>&1
  MPX_TILE_RULES="k1x:8" python tools/with_lib.py $L tools/layer_profile.py $ARCH $B $REPS > gpurun_out/abx_B_$pass.txt 2>&1
  MPX_TILE_RULES="k1x:7" python tools/with_lib.py $L tools/layer_profile.py $ARCH $B $REPS > gpurun_out/abx_C_$pass.txt 2>&1
  for v in A B C; do echo "== $v (pass $pass; A = tile 10, B = tile 8, C = tile 7)"; grep -E "^rule|256->1024|128->512|conv total" gpurun_out/abx_${v}_$pass.txt | cut -c1-110; done
done
