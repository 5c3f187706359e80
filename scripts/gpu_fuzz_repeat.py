"""One GPU fuzz draw N times in one process (fresh converter and buffers every time): nondeterminism on the device.
python scripts/gpu_fuzz_repeat.py SEED INDEX [N]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np                        # noqa: E402
import torch                              # noqa: E402
import cases                              # noqa: E402
import fuzz_video                         # noqa: E402
from gstreamer_amd import video as V      # noqa: E402
from oracle import ref                    # noqa: E402

seed, target = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rnd = random.Random(seed)
rects = random.Random(seed + 77) if seed >= 700 else None
case = [fuzz_video.random_case(rnd, rects) for _ in range(target + 1)][target]
ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
gpu = torch.device("cuda:0")
ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site)
oi = V.video_info(ofmt, ow, oh)
src = cases.frame_bytes(int(ii.size), "random", seed * 1000 + target, w)
want = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, config=cases.ref_config_string(ref, cfg)).frame(src)
print(case)
fails = 0
for k in range(n):
    junk = torch.full((4 << 20,), (37 * k + 11) & 0xff, dtype=torch.uint8, device=gpu)        # recycled memory holds something else every time
    del junk
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    d_src = torch.from_numpy(src).to(gpu)
    d_dst = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
    if os.environ.get("SYNC_BEFORE"):
        torch.cuda.synchronize()
    conv.frame(d_src, d_dst)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    conv.free()
    d = np.nonzero(got != want)[0]
    if len(d):
        fails += 1
        st = int(oi.stride[0])
        if fails <= 3:
            print("run", k, len(d), "bytes; rows", sorted(set(int(i) // st for i in d))[:30], "cols", sorted(set(int(i) % st for i in d))[:40])
            print("   ", [(int(i) // st, int(i) % st, int(got[i]), int(want[i])) for i in d[:10]])
print("failures", fails, "of", n)
