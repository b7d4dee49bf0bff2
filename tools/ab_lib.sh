#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call: the product build against a probe build with extra compile flags.
# usage: tools/ab_lib.sh "<extra hipcc flags of build B>" [arch] [batch] [reps]      (prints the per-shape table tails of A, B, A, B)
#        AB_BENCH=1 tools/ab_lib.sh "<flags>"      (runs `python bench.py --steps 2 --warmup 1 --cpu-masks 0` on A, B, A, B instead)
#        AB_GREP="64->64    k3|layer1|conv total" tools/ab_lib.sh "<flags>"      (other rows of the per-layer table)
# Build B lives in /tmp and is bound per process by tools/with_lib.py: the product libmpx.so and its stamp are never overwritten.
set -e
FLAGS="$1"; ARCH=${2:-resnet101}; B=${3:-2340}; REPS=${4:-3}
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
A=network_interpretation_imagenet_amd/libmpx.so
( cd network_interpretation_imagenet_amd/csrc && ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $FLAGS -o /tmp/libmpx_B.so mpx_api.hip )
lib() { if [ "$1" = A ]; then echo $A; else echo /tmp/libmpx_B.so; fi; }
bench() { python tools/with_lib.py $(lib $1) bench.py --steps 2 --warmup 1 --cpu-masks 0 > gpurun_out/abb_$1_$2.json 2> /dev/null
          python -c "import json,sys; j=json.load(open('gpurun_out/abb_$1_$2.json')); print('== build $1 (run $2): %.0f fwd/s, conv %.2f ms per batch of %d' % (j['value'], j['roofline']['conv_ms_per_batch'], j['config']['forward_batch']))"; }
run() { python tools/with_lib.py $(lib $1) tools/layer_profile.py $ARCH $B $REPS > gpurun_out/ab_$1_$2.txt 2>&1; echo "== build $1 (run $2)"; grep -E "${AB_GREP:-k3 s1 out14|256->1024|1024->256|128->512|conv total}" gpurun_out/ab_$1_$2.txt | cut -c1-90; }
mkdir -p gpurun_out
if [ -n "$AB_BENCH" ]; then bench A 1; bench B 1; bench A 2; bench B 2; else run A 1; run B 1; run A 2; run B 2; fi
