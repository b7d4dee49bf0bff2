#!/bin/bash
# Package power and shader clock while ONE kernel class runs in a loop (isolated layers at the benched batch): is a kernel that sits
# far below its MFMA / HBM bound nevertheless AT the package power limit?  (If every class is, the forward is energy-bound as a whole and
# overlapping an HBM-bound with an MFMA-bound layer cannot buy time.)  One rocm-smi sample per 0.5 s next to each loop; prints the
# median of the samples taken while the loop ran.   usage: tools/power_by_kernel.sh [batch=2340] [seconds=6]
set -e
B=${1:-2340}; SECS=${2:-6}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/power
python -c "import __graft_entry__ as g; g.build()"
sample() { ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ';'; echo; sleep 0.5; done ) > "$1" & echo $!; }
median() { python - "$1" <<'PY'
import re, sys, statistics
p, c = [], []
for line in open(sys.argv[1]):
    m = re.search(r"Power \(W\): ([0-9.]+)", line); k = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", line)
    if m and k and float(m.group(1)) > 500:        # samples taken under load only
        p.append(float(m.group(1))); c.append(int(k.group(1)))
print("   power median %.0f W (min %.0f, max %.0f), sclk median %d MHz, %d samples under load" % (statistics.median(p), min(p), max(p), statistics.median(c), len(p)) if p else "   no sample under load")
PY
}
one() {   # name layer tile ms-per-launch-estimate
  local reps=$(python -c "print(max(20, int($SECS * 1000 / $4)))")
  S=$(sample gpurun_out/power/$1.txt); sleep 1
  python tools/conv_bench.py resnet101 $2 $B $reps $3 2>&1 | grep TFLOP
  sleep 0.5; kill $S; wait $S 2>/dev/null || true
  median gpurun_out/power/$1.txt
}
echo "# tools/power_by_kernel.sh $B $SECS   (idle package power ~260 W, limit 1400 W)"
one convx_256_1024     layer3.5.conv3  10 1.05
one convw_256_1024     layer3.5.conv3  14 0.80
one patch_3x3_256_256  layer3.5.conv2  12 1.06
one tile256_1024_256   layer3.5.conv1  13 0.60
one convx_128_512      layer2.1.conv3  10 1.80
one tile7_512_2048     layer4.1.conv3   7 1.20
one tile0_3x3_s2       layer3.0.conv2   0 1.30
for K in 1; do
  S=$(sample gpurun_out/power/btail$K.txt); sleep 1
  python tools/tail_bench.py $K $B $(python -c "print(int($SECS * 1000 / 4.3))") 2>&1 | grep -E "ms per launch" | tail -1
  sleep 0.5; kill $S; wait $S 2>/dev/null || true
  median gpurun_out/power/btail$K.txt
done
