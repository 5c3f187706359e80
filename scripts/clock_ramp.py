#!/usr/bin/env python3
"""Per-launch time of the C2 kernel from a cold start (one 32-frame launch per step): shows how long the device needs
under sustained load before it reaches its steady rate.  Used to size bench.py's pre-heat."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases
from gstreamer_amd import video as V
W, H, N = 3840, 2160, 32
dev = torch.device("cuda:0")
ii, oi = V.video_info("NV12", W, H), V.video_info("BGRA", W, H)
conv = V.VideoConverter(ii, oi)
pin = torch.empty((32, int(ii.size)), dtype=torch.uint8, device=dev)
base = torch.from_numpy(cases.frame_bytes(int(ii.size), "random", 1)).to(dev)
for i in range(32):
    pin[i] = torch.roll(base, shifts=i * 4099)
pout = torch.zeros((16, int(oi.size)), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ip = [pin[i].data_ptr() for i in range(32)]; op = [pout[i % 16].data_ptr() for i in range(32)]
torch.cuda.synchronize(); time.sleep(float(os.environ.get("IDLE_S", "2.0")))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for s in range(steps):
    conv.frames(ip, op, stream)
    ev[s + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 / N for i in range(steps)]
acc = 0.0
for i in range(0, steps, max(1, steps // 40)):
    chunk = t[i:i + max(1, steps // 40)]
    print("launch %4d  t=%7.2f ms  us/frame=%6.3f  TB/s=%5.2f" % (i, sum(t[:i]) * N / 1e3, sum(chunk) / len(chunk), 45619200 / (sum(chunk) / len(chunk)) / 1e6))
