#!/usr/bin/env python3
"""What does this MI355X actually sustain for pure reads, pure writes and copies (GB/s)?  The roofline fractions of the HBM-bound kernels
 are quoted against 8 TB/s; this is the achievable ceiling they live under.  16-byte accesses, 2 GiB buffers (beyond L2 +
Infinity Cache), HIP events.    python tools/micro/hbm_rates.py"""
import torch

n = 1 << 29                        # 2 GiB of fp32
a = torch.empty(n, device="cuda")
b = torch.empty(n, device="cuda")
a.normal_()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 4 / 1e9
print("write only (fill_)     %7.0f GB/s" % (gb / timed(lambda: b.fill_(1.5))))
print("read only  (sum)       %7.0f GB/s" % (gb / timed(lambda: a.sum())))
print("copy       (copy_)     %7.0f GB/s of read + write" % (2 * gb / timed(lambda: b.copy_(a))))
print("read 2, write 1 (add)  %7.0f GB/s" % (3 * gb / timed(lambda: torch.add(a, b, out=b))))
h = a.half()
g = torch.empty_like(h)
print("copy fp16  (copy_)     %7.0f GB/s" % (2 * gb / 2 / timed(lambda: g.copy_(h))))
