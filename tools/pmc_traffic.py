#!/usr/bin/env python3
"""HBM bytes of the conv launches per forward batch from two rocprofv3 --pmc passes over
`tools/layer_profile.py <arch> <batch> <reps>` (FETCH_SIZE and WRITE_SIZE do not fit one pass).
usage: python tools/pmc_traffic.py <fetch-pass-dir> <write-pass-dir> <arch> <batch> <forward-batches-profiled> > profiles/rNN_pmc_traffic.json
Units and corrections as MI355X_MICROARCH.md (HBM) prescribes: the counters are KiB; WRITE_SIZE is exact for
16-B-per-lane streaming stores; FETCH_SIZE tallies 128-B requests at 64 B and is doubled."""
import csv
import glob
import json
import os
import sys


def total(path, counter, sub="_f16x3_kernel"):      # conv_f16x3_kernel and conv3x3p_f16x3_kernel
    s, n = 0.0, 0
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if (sub in name or "stem_apply_kernel" in name) and row["Counter_Name"] == counter:
                    s += float(row["Counter_Value"])
                    n += 1
    return s, n


fetch_dir, write_dir, arch, batch, nb = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
f, nf = total(fetch_dir, "FETCH_SIZE")
w, nw = total(write_dir, "WRITE_SIZE")
assert nf and nf == nw and nf % nb == 0, (nf, nw, nb)
fetch = f * 1024 / nb
write = w * 1024 / nb
print(json.dumps({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) --kernel-trace -- python3 tools/layer_profile.py "
              "%s %d; every kernel whose name contains _f16x3_kernel = all conv launches (conv_f16x3_kernel, conv3x3p_f16x3_kernel, conv3x3pp_f16x3_kernel, "
              "conv256_f16x3_kernel, conv256p_f16x3_kernel, convx_f16x3_kernel, convw_f16x3_kernel, btail_f16x3_kernel) + stem_apply_kernel (the stem by superposition, when the profile staged that way), per forward batch of %d masked images" % (arch, batch, batch),
    "arch": arch,
    "forward_batch": batch,
    # kernel dispatches the counters were summed over: a layer whose last round of tiles is split off runs as two dispatches, so this
    # is larger than the engine's launch count; bench.py divides the bytes by ITS OWN conv-launch count (one denominator, stated there)
    "dispatches_per_batch": nf // nb,
    "write_bytes_per_batch": write,
    "fetch_raw_bytes_per_batch": fetch,
    "fetch_corrected_bytes_per_batch_guide_x2": 2 * fetch,
    "total_bytes_per_batch": 2 * fetch + write,
}, indent=1))
