#!/bin/bash
# A/B of a tile rule against the defaults in the network, inside ONE gpurun call (two interleaved passes).
# usage: tools/ab_rules.sh "<class:tile,...>" [arch] [batch] [reps]     classes: tools/layer_profile.py (MPX_TILE_RULES)
set -e
RULES="$1"; ARCH=${2:-resnet101}; B=${3:-2340}; REPS=${4:-3}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for pass in 1 2; do
  python tools/layer_profile.py $ARCH $B $REPS > gpurun_out/abr_A_$pass.txt 2>&1
  MPX_TILE_RULES="$RULES" python tools/layer_profile.py $ARCH $B $REPS > gpurun_out/abr_B_$pass.txt 2>&1
  for v in A B; do echo "== $v (pass $pass)"; grep -E "^rule| k[13] s[12] out|conv total" gpurun_out/abr_${v}_$pass.txt | cut -c1-100; done
done
