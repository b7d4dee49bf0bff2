#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc CSV (counter_collection.csv): per (kernel, grid) mean counter values.
usage: python tools/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "conv_f16x3"
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if sub not in name:
                continue
            key = (name.split("(")[0][-48:], row.get("Grid_Size", "?"), row.get("LDS_Block_Size", "?"))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key in sorted(acc, key=lambda k: (k[0], int(k[1]) if k[1].isdigit() else 0)):
    c = acc[key]
    n = max(len(v) for v in c.values())
    print("%s grid=%s lds=%s dispatches=%d" % (key[0], key[1], key[2], n))
    print("   " + "  ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
