#!/usr/bin/env python3
"""What the memory system delivers to plain streaming kernels on this box (context for the roofline fractions, which are quoted against the
8 TB/s spec peak): device-to-device copy, read-only reduction, write-only fill, on buffers far larger than the 256 MB of MALL + L2."""
import json
import torch

dev = torch.device("cuda:0")
n = 1 << 30
a = torch.empty(n // 4, dtype=torch.int32, device=dev).random_(0, 100)
b = torch.empty_like(a)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


res = {}
t = timed(lambda: b.copy_(a))
res["copy_1GiB"] = {"us": round(t * 1e6, 1), "GBps_read_plus_write": round(2 * n / t / 1e9, 1)}
t = timed(lambda: a.sum())
res["read_sum_1GiB"] = {"us": round(t * 1e6, 1), "GBps_read": round(n / t / 1e9, 1)}
t = timed(lambda: b.fill_(7))
res["fill_1GiB"] = {"us": round(t * 1e6, 1), "GBps_write": round(n / t / 1e9, 1)}
print(json.dumps(res))
