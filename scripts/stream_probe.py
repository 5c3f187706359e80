#!/usr/bin/env python3
"""C2 one frame per launch at the C ABI, round robin over N HIP streams with nothing between the launches (no events): what overlapping
launches can give a per-buffer element.  python scripts/stream_probe.py"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
wl = bench.VideoWorkload("c2", 1)
dev = torch.device("cuda:0")
wl.setup(dev, 0)
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    hs = [s.cuda_stream for s in streams]
    for rep in range(2):
        torch.cuda.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for i in range(n):
            wl.conv.frame(wl.in_ptrs[i % wl.pool_in], wl.out_ptrs[i % wl.pool_out], hs[i % ns])
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    print("streams", ns, "us/frame", round(t * 1e6 / n, 3), "host issue us/frame", round(t_host * 1e6 / n, 3), "frac", round(wl.alg_bytes / (t / n) / 8e12, 3))
