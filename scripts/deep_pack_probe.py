"""k_deep_scale_pack (video_deep_pack.h) on the device: byte-exact against oracle/_ref, then us per frame single and in lists of 8.
   python scripts/deep_pack_probe.py            GSTAMD_NO_DEEP_SCALE_PACK=1 python scripts/deep_pack_probe.py  (the multi-launch composite)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np                        # noqa: E402
import torch                              # noqa: E402
import cases                              # noqa: E402
from gstreamer_amd import video as V      # noqa: E402
from oracle import ref                    # noqa: E402

BIL, LIN, CUB = dict(resampler_method="linear", max_taps=2), dict(resampler_method="linear"), dict(resampler_method="cubic")
CASES = [("P010_10LE", 3840, 2160, "NV12", 1920, 1080, BIL), ("P010_10LE", 3840, 2160, "I420", 1920, 1080, BIL), ("I420_10LE", 3840, 2160, "NV12", 1920, 1080, BIL),
         ("P010_10LE", 3840, 2160, "NV12", 1280, 720, BIL), ("P010_10LE", 3840, 2160, "NV12", 1920, 1080, LIN), ("P010_10LE", 3840, 2160, "NV12", 1920, 1080, CUB),
         ("P010_10LE", 1920, 1080, "NV12", 960, 540, BIL), ("P010_10LE", 2560, 1440, "NV12", 1920, 1080, BIL), ("P010_10LE", 3840, 2160, "YUY2", 1920, 1080, BIL),
         ("P010_10LE", 7680, 4320, "NV12", 3840, 2160, BIL), ("P010_10LE", 3840, 2160, "BGRA", 1920, 1080, BIL), ("I420_10LE", 3840, 2160, "RGBA", 1920, 1080, BIL),
         ("P010_10LE", 7680, 4320, "BGRA", 3840, 2160, BIL)]
if len(sys.argv) > 1:
    CASES = [CASES[int(a)] for a in sys.argv[1:]]
dev = torch.device("cuda:0")
for ifmt, w, h, ofmt, ow, oh, cfg in CASES:
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    host = cases.frame_bytes(int(ii.size), "random", 7, w)
    n_in = 8
    src = torch.from_numpy(host).to(dev).repeat(n_in, 1)
    dst = torch.zeros((n_in, int(oi.size)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    conv.frame(src[0], dst[0], st)
    torch.cuda.synchronize()
    verdict = "unchecked"
    if w * h <= 3840 * 2160 and ref.available():
        want = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cfg)).frame(host)
        got = dst[0].cpu().numpy()
        bad = int((got != want).sum())
        verdict = "exact" if bad == 0 else "MISMATCH %d first %d" % (bad, int(np.argmax(got != want)))
    for i in range(5):
        conv.frame(src[i % n_in], dst[i % n_in], st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for i in range(n):
        conv.frame(src[i % n_in], dst[i % n_in], st)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
    for i in range(3):
        conv.frames(srcs, dsts, st)
    torch.cuda.synchronize()
    lists_exact = all(bool((dst[i] == dst[0]).all()) for i in range(8))
    t0 = time.perf_counter()
    for i in range(20):
        conv.frames(srcs, dsts, st)
    torch.cuda.synchronize()
    usl = (time.perf_counter() - t0) / (20 * 8) * 1e6
    alg = conv.algorithmic_bytes()
    name = "bilinear" if cfg.get("max_taps") == 2 else cfg.get("resampler_method", "")
    print("%-10s %4dx%-4d -> %-5s %4dx%-4d %-8s %s | single %6.1f us frac %.3f | lists of 8 %6.1f us frac %.3f (list launches %s, frames equal %s) | %5.1f MB" % (
        ifmt, w, h, ofmt, ow, oh, name, verdict, us, alg / (us * 1e-6) / 8e12, usl, alg / (usl * 1e-6) / 8e12, conv.list_launches(), lists_exact, alg / 1e6), flush=True)
    conv.free()
