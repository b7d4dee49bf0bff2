#!/usr/bin/env python3
"""Effective shader clock and MFMA duty of a kernel from one rocprofv3 --pmc pass (MI355X_MICROARCH.md 'DVFS give-back':
clock ~ GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time; MFMA duty = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles): the counter adds 16 per v_mfma_f32_16x16x32_f16).
usage: python tools/pmc_clock.py <rocprofv3 output dir> <kernel-substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

path, sub = sys.argv[1], sys.argv[2]
rows = defaultdict(dict)
for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if sub in r.get("Kernel_Name", ""):
                d = rows[(f, r["Dispatch_Id"])]
                d[r["Counter_Name"]] = float(r["Counter_Value"])
                d["name"] = r["Kernel_Name"].split("(")[0][-44:]
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
if rows and not any("ns" in d for d in rows.values()):          # older layouts keep the timestamps in kernel_trace.csv
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                for (cf, did), d in rows.items():
                    if did == r.get("Dispatch_Id") and os.path.dirname(cf) == os.path.dirname(f):
                        d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ds = [d for d in rows.values() if "ns" in d and "GRBM_GUI_ACTIVE" in d]
ds = ds[len(ds) // 4:]                                           # drop the clock ramp of the first launches
if not ds:
    sys.exit("no dispatch of *%s* with timestamps and GRBM_GUI_ACTIVE under %s" % (sub, path))
n = len(ds)
ns = sum(d["ns"] for d in ds) / n
clk = sum(d["GRBM_GUI_ACTIVE"] / 8.0 / d["ns"] for d in ds) / n      # GHz
line = "%s: %d dispatches, %.1f us under the counters, effective clock %.2f GHz" % (ds[0]["name"], n, ns / 1e3, clk)
if "SQ_VALU_MFMA_BUSY_CYCLES" in ds[0]:
    duty = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8.0) for d in ds) / n
    line += ", MFMA pipe busy %.0f %% of the SIMD-cycles" % (100 * duty)
if "SQ_BUSY_CYCLES" in ds[0]:
    line += ", SQ_BUSY_CYCLES %.3g" % (sum(d["SQ_BUSY_CYCLES"] for d in ds) / n)
print(line)
