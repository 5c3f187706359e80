import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import cases, test_compositor as T
from gstreamer_amd import video as V
from oracle import ref
gpu = torch.device("cuda:0")
dw, dh = 3840, 2160
layout = [(1920, 1080, 960, 540, "cubic", (i % 4) * 960, (i // 4) * 540, 1.0 if i % 2 else 0.8, 1) for i in range(16)]
frames = [cases.frame_bytes(1920 * 1080 * 4, "random", 5100 + i) for i in range(16)]
exp = T.scaled_expected(ref, "BGRA", 1, layout, frames, dw, dh)
got = T.hip_scaled(gpu, "BGRA", 1, layout, frames, dw, dh)
bad = np.flatnonzero(exp != got)
print(len(bad))
px = np.unique(bad // 4)
ys, xs = px // dw, px % dw
print(sorted(set(ys.tolist()))[:20], sorted(set(xs.tolist()))[:40])
for q in px[:10]:
    print(q // dw, q % dw, exp[4*q:4*q+4], got[4*q:4*q+4])
