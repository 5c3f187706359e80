"""k_scale_col against the reference on the GPU box, with a description of where the bytes differ (development aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from gstreamer_amd import video as V  # noqa: E402
from oracle import ref  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [("I420", 640, 360, "BGRA", 320, 180), ("I420", 1280, 720, "BGRA", 320, 180), ("NV12", 1920, 1080, "BGRx", 640, 360), ("NV21", 1920, 1080, "BGRx", 1280, 720), ("I420", 3840, 2160, "RGBA", 960, 540)]
KNOBS = [dict(GSTAMD_COL_OPL=1, GSTAMD_COL_WAVES=1), dict(GSTAMD_COL_OPL=1, GSTAMD_COL_WAVES=1, GSTAMD_COL_CHUNKS=1)]
for (ifmt, w, h, ofmt, ow, oh) in SHAPES:
    src = cases.frame_bytes(ref.video_info(ifmt, w, h)["size"], "random", 555)
    exp = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cases.LAN)).frame(src)
    for kn in KNOBS:
        with V.tuning(**kn):
            conv = V.VideoConverter(V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh), V.converter_config(**cases.LAN))
            d_src = torch.from_numpy(src).to(dev)
            d_dst = torch.zeros(ow * oh * 4, dtype=torch.uint8, device=dev)
            conv.frame(d_src, d_dst)
            torch.cuda.synchronize()
            out = d_dst.cpu().numpy()
            conv.free()
        d = np.nonzero(out != exp)[0]
        print(ifmt, w, h, ow, oh, kn, "differing bytes", len(d), "of", len(out))
        if len(d):
            ys, xs, cs = d // (ow * 4), (d % (ow * 4)) // 4, d % 4
            print("   rows:", len(np.unique(ys)), np.unique(ys)[:24], " cols:", len(np.unique(xs)), np.unique(xs)[:24], " bytes:", np.bincount(cs, minlength=4))
            for i in d[:6]:
                print("   y %d x %d b %d got %d exp %d" % (i // (ow * 4), (i % (ow * 4)) // 4, i % 4, out[i], exp[i]))
        sys.stdout.flush()
