#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of every conv kernel in the library (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
A non-zero scratch size inside a K loop would break the counted s_waitcnt vmcnt bookkeeping: check after every kernel edit."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + [f for f in g.HIPCC_FLAGS if f != "-shared"] + [
    "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/tmp/mpx_resources.o", os.path.join(g.CSRC, "mpx_api.hip")]
out = subprocess.run(cmd, capture_output=True, text=True, cwd=g.CSRC).stderr
rows, cur = {}, None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur:
            rows[cur][key.split()[0]] = int(m.group(1))
for name, r in rows.items():
    if r:
        print("%4d VGPR %3d AGPR %3d scratch  occ %d  %s" % (r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("ScratchSize", 0), r.get("Occupancy", 0),
                                                          re.sub(r"^void mpx::", "", name)[:110]))
