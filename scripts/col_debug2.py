"""impulse lines through k_scale_col vs the reference (development aid)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from gstreamer_amd import video as V
from oracle import ref
dev = torch.device("cuda:0")
ifmt, w, h, ofmt, ow, oh = "NV12", 640, 360, "BGRA", 320, 180
size = ref.video_info(ifmt, w, h)["size"]
with V.tuning(GSTAMD_COL_OPL=1, GSTAMD_COL_WAVES=1):
    conv = V.VideoConverter(V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh), V.converter_config(**cases.LAN))
rc = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cases.LAN))
for L in range(16, 24):
    src = np.full(size, 128, np.uint8)
    src[: w * h] = 16
    src[L * w:(L + 1) * w] = 235
    exp = rc.frame(src).reshape(oh, ow, 4)
    d_src = torch.from_numpy(src).to(dev); d_dst = torch.zeros(ow * oh * 4, dtype=torch.uint8, device=dev)
    conv.frame(d_src, d_dst); torch.cuda.synchronize()
    out = d_dst.cpu().numpy().reshape(oh, ow, 4)
    r0 = L // 2 - 4
    print("line", L, "rows", r0, "..: got", list(out[r0:r0 + 9, 100, 1]), "exp", list(exp[r0:r0 + 9, 100, 1]))
