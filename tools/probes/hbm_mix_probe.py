#!/usr/bin/env python3
"""What does HBM sustain for the epilogue's traffic mix (two streams read, one written) with a trivial kernel?  torch.add on
fp16 tensors far larger than the caches; also copy (1:1) and fill (write only) and sum (read only)."""
import torch

dev = torch.device("cuda", 0)
n = 3 << 30                                    # 3 Gi elements fp16 = 6 GiB per tensor
a = torch.ones(n, dtype=torch.float16, device=dev)
b = torch.ones(n, dtype=torch.float16, device=dev)
c = torch.empty(n, dtype=torch.float16, device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = n * 2 / 1e9
for name, fn, streams in (("add  (2 read : 1 write)", lambda: torch.add(a, b, out=c), 3), ("copy (1 : 1)", lambda: c.copy_(a), 2),
                          ("fill (write only)", lambda: c.fill_(2.0), 1), ("sum  (read only)", lambda: a.sum(), 1),
                          ("relu_ in place (1 : 1 same lines)", lambda: a.relu_(), 2)):
    ms = timed(fn)
    print("%-36s %7.3f ms  %6.2f TB/s" % (name, ms, gb * streams / ms))
