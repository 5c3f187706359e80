#!/usr/bin/env python3
"""Stage timeline of k_scale420_fused from a -DGSTAMD_TUNING build (GSTAMD_TUNING_LIB=1): per-wave s_memtime stamps of one C3 launch
taken in steady state (400 untraced frames first).  Prints, per stage, the median / max time since the workgroup's first stamp."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GSTAMD_TUNING_LIB"] = "1"
import torch
import cases
from gstreamer_amd import video as V

dev = torch.device("cuda:0")
ii, oi = V.video_info("I420", 7680, 4320), V.video_info("RGBA", 1920, 1080)
conv = V.VideoConverter(ii, oi, V.converter_config(**cases.LAN))
base = torch.from_numpy(cases.frame_bytes(int(ii.size), "random", 5)).to(dev)
ins = [torch.roll(base, shifts=i * 4099) for i in range(10)]
out = torch.zeros(int(oi.size), dtype=torch.uint8, device=dev)
for i in range(400):
    conv.frame(ins[i % 10], out)
path = "/tmp/fused_trace.bin"
os.environ["GSTAMD_FUSED_TRACE"] = path
conv.frame(ins[3], out)
torch.cuda.synchronize()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 16, 32).astype(np.int64)
nw = int(os.environ.get("GSTAMD_FUSED_WAVES", "8"))
t = t[:, :nw, :]
valid = t > 0
t0 = np.where(valid, t, np.iinfo(np.int64).max).min()
print("workgroups", t.shape[0], "waves", nw, "kernel span (cycles of s_memtime)", int(t[valid].max() - t0))
start = t[:, :, 0] - t0
print("wave start after kernel start: median %d max %d" % (np.median(start), start.max()))
names = {0: "start", 1: "taps+first loads issued"}
for k in range(1, 32):
    col = t[:, :, k]
    ok = col > 0
    if not ok.any():
        continue
    d = (col - t[:, :, 0])[ok]
    print("stamp %2d: n=%5d  since wave start: median %6d  p90 %6d  max %6d   | since kernel start: median %6d max %6d" %
          (k, ok.sum(), np.median(d), np.percentile(d, 90), d.max(), np.median((col - t0)[ok]), (col - t0)[ok].max()))
