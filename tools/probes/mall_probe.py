#!/usr/bin/env python3
"""Does the 256-MB memory-side cache give back lines a kernel has just WRITTEN?  For N MiB: (hot) write the buffer, then time a
read of it; (cold) write it, push 3 GiB of other traffic through, then time the read.  Also read-after-read.  torch kernels only
(fill_ / sum): what matters is the ratio hot : cold per size, not the absolute rate."""
import torch

dev = torch.device("cuda", 0)
flush = torch.empty(3 << 30, dtype=torch.uint8, device=dev)


def timed(fn, reps=1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("%8s %14s %14s %14s %14s" % ("MiB", "read-after-write", "cold read", "read-after-read", "write GB/s"))
for mib in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    n = mib << 20
    buf = torch.empty(n // 4, dtype=torch.float32, device=dev)
    res = {}
    for mode in ("hot", "cold", "rar"):
        best = 1e9
        for _ in range(5):
            w = timed(lambda: buf.fill_(1.0))
            if mode != "hot":
                flush.fill_(3)
            if mode == "rar":
                buf.sum()
            torch.cuda.synchronize()
            best = min(best, timed(lambda: buf.sum()))
        res[mode] = n / best / 1e6
    print("%8d %11.0f GB/s %9.0f GB/s %9.0f GB/s %11.0f" % (mib, res["hot"], res["cold"], res["rar"], n / w / 1e6))
