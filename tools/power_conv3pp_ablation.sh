#!/bin/bash
# Package power and shader clock of the persistent 3x3 patch kernel and of its timing-only ablations (tools/ablate_conv3pp.sh): does an
# ablation run faster because it waits less, or because it toggles less and the governor gives the clock back?
# usage: tools/power_conv3pp_ablation.sh "<prebuilt probe libs>" [batch=2340] [seconds=6]
LIBS="$1"; B=${2:-2340}; SECS=${3:-6}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/power
python -c "import __graft_entry__ as g; g.build()"
sample() { ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ';'; echo; sleep 0.5; done ) > "$1" & echo $!; }
median() { python - "$1" <<'PY'
import re, sys, statistics
p, c = [], []
for line in open(sys.argv[1]):
    m = re.search(r"Power \(W\): ([0-9.]+)", line); k = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", line)
    if m and k and float(m.group(1)) > 500:
        p.append(float(m.group(1))); c.append(int(k.group(1)))
print("   power median %.0f W (min %.0f, max %.0f), sclk median %d MHz, %d samples under load" % (statistics.median(p), min(p), max(p), statistics.median(c), len(p)) if p else "   no sample under load")
PY
}
for lib in network_interpretation_imagenet_amd/libmpx.so $LIBS; do
  tag=$(basename $lib .so)
  reps=$(python -c "print(int($SECS * 1000 / 0.9))")
  S=$(sample gpurun_out/power/$tag.txt); sleep 1
  echo "== $tag: $(python tools/with_lib.py $lib tools/conv_bench.py resnet101 layer3.5.conv2 $B $reps 2>/dev/null | tail -1)"
  sleep 0.5; kill $S; wait $S 2>/dev/null || true
  median gpurun_out/power/$tag.txt
done
