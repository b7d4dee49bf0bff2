#!/bin/bash
# A/B/C... of PREBUILT libraries inside ONE gpurun call: the product build (A) against probe builds that travel with the snapshot.
# usage: AB_GREP="..." tools/ab_libs.sh "<lib1.so> <lib2.so> ..." [arch] [batch] [reps] [passes]   (per-layer table rows of A, lib1, lib2, ..., repeated `passes` times)
#        AB_BENCH=1 ... runs `python bench.py --steps 2 --warmup 1 --cpu-masks 0` instead
set -e
LIBS="$1"; ARCH=${2:-resnet101}; B=${3:-2340}; REPS=${4:-3}; PASSES=${5:-2}
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
mkdir -p gpurun_out
for pass in $(seq 1 $PASSES); do
  for lib in network_interpretation_imagenet_amd/libmpx.so $LIBS; do
    tag=$(basename $lib .so)
    if [ -n "$AB_BENCH" ]; then
      python tools/with_lib.py $lib bench.py --steps 2 --warmup 1 --cpu-masks 0 > gpurun_out/abb_${tag}_$pass.json 2> /dev/null
      python -c "import json; j=json.load(open('gpurun_out/abb_${tag}_$pass.json')); print('== $tag (pass $pass): %.0f fwd/s, conv %.2f ms per batch of %d' % (j['value'], j['roofline']['conv_ms_per_batch'], j['config']['forward_batch']))"
    else
      python tools/with_lib.py $lib tools/layer_profile.py $ARCH $B $REPS > gpurun_out/ab_${tag}_$pass.txt 2>&1
      echo "== $tag (pass $pass)"; grep -E "${AB_GREP:-k3 s1 out14|256->1024|1024->256|128->512|conv total}" gpurun_out/ab_${tag}_$pass.txt | cut -c1-90
    fi
  done
done
