#!/bin/bash
# Conv time per masked image against the forward batch, in ONE gpurun call (tile rounds: 256-pixel tiles on 14x14 maps fill the
# 256 CUs in whole rounds at 2006 images, not at 2048).   usage: tools/batch_sweep.sh "2006 2048 ..." [arch] [reps]
set -e
BATCHES=${1:-"1920 2006 2048 2173 2340"}; ARCH=${2:-resnet101}; REPS=${3:-3}
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
mkdir -p gpurun_out
for pass in 1 2; do for b in $BATCHES; do
  python tools/layer_profile.py $ARCH $b $REPS > gpurun_out/bs_${b}_$pass.txt 2>&1
  python - $b gpurun_out/bs_${b}_$pass.txt <<'PY'
import re, sys
b = int(sys.argv[1]); t = open(sys.argv[2]).read()
tot = float(re.search(r"conv total ([0-9.]+) ms", t).group(1))
rows = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^\s*(\d+->\d+\s+k\d s\d out\d+)\s+x\d+\s+([0-9.]+) ms", t, re.M)}
pick = ["256->256   k3 s1 out14", "256->1024  k1 s1 out14", "1024->256   k1 s1 out14", "64->64    k3 s1 out56", "128->512   k1 s1 out28"]
print("batch %5d  conv %.3f ms  %.3f us/image  | " % (b, tot, 1e3 * tot / b) + "  ".join("%s %.2f" % (k.split()[0], 1e3 * rows[k] / b) for k in pick if k in rows))
PY
done; done
