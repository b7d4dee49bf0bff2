#!/bin/bash
# Timing-only ablations of the persistent 3x3 patch kernel (mpx_conv3pp.h, -DP3_ABL=<mask>, wrong results), one isolated layer with fixed
# random inputs (the input planes do not depend on what the ablated kernel writes), product build first and last.
# usage: tools/ablate_conv3pp.sh "<prebuilt probe libs>" [layer] [batch] [reps]
LIBS="$1"; LAYER=${2:-layer3.5.conv2}; B=${3:-2340}; REPS=${4:-200}
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
for lib in network_interpretation_imagenet_amd/libmpx.so $LIBS network_interpretation_imagenet_amd/libmpx.so; do
  echo "== $(basename $lib .so): $(python tools/with_lib.py $lib tools/conv_bench.py resnet101 $LAYER $B $REPS 2>/dev/null | tail -1)"
done
