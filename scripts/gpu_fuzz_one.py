"""One draw of the GPU fuzz (tests/test_video_fuzz.py) on the device: python scripts/gpu_fuzz_one.py SEED INDEX"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch                              # noqa: E402
import cases                              # noqa: E402
import fuzz_video                         # noqa: E402
from gstreamer_amd import video as V      # noqa: E402
from oracle import ref                    # noqa: E402

seed, target = int(sys.argv[1]), int(sys.argv[2])
rnd = random.Random(seed)
rects = random.Random(seed + 77) if seed >= 700 else None
gpu = torch.device("cuda:0")
for it in range(target + 1):
    case = fuzz_video.random_case(rnd, rects)
    if it != target:
        continue
    ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
    ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site)
    oi = V.video_info(ofmt, ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    print(case, conv.describe(), "divergence:", conv.divergence(), flush=True)
    src = cases.frame_bytes(int(ii.size), "random", seed * 1000 + it, w)
    d_src = torch.from_numpy(src).to(gpu)
    d_dst = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
    conv.frame(d_src, d_dst)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    print(fuzz_video.matches_reference(ref, case, src, got, oi), flush=True)
    want = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, config=cases.ref_config_string(ref, cfg)).frame(src)
    import numpy as np
    d = np.nonzero(got != want)[0]
    if len(d):
        st = int(oi.stride[0])
        print("stride", st, "first differing bytes (row, byte, got, want):", [(int(i) // st, int(i) % st, int(got[i]), int(want[i])) for i in d[:24]])
        rows = sorted(set(int(i) // st for i in d)); print("rows", rows[:40], "cols", sorted(set(int(i) % st for i in d))[:60])
