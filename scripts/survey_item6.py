"""VERDICT r04 item 6 pairs: one frame per call and lists of 8, us per frame and fraction of the 8 TB/s peak; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.   python scripts/survey_item6.py [case-index ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch                              # noqa: E402
from gstreamer_amd import video as V      # noqa: E402

BIL, LIN, CUB, LAN = dict(resampler_method="linear", max_taps=2), dict(resampler_method="linear"), dict(resampler_method="cubic"), dict(resampler_method="lanczos")
CASES = [("NV12", 3840, 2160, "BGRA", 1920, 1080, BIL), ("NV12", 3840, 2160, "BGRA", 2560, 1440, BIL), ("NV12", 1920, 1080, "BGRA", 3840, 2160, BIL),
         ("I420", 3840, 2160, "RGBA", 1280, 720, BIL),
         ("BGRA", 3840, 2160, "NV12", 1920, 1080, BIL), ("BGRA", 3840, 2160, "NV12", 1920, 1080, LIN), ("BGRA", 3840, 2160, "I420", 1920, 1080, CUB),
         ("BGRA", 1920, 1080, "NV12", 1280, 720, BIL),
         ("P010_10LE", 3840, 2160, "NV12", 1920, 1080, BIL), ("P010_10LE", 3840, 2160, "BGRA", 1920, 1080, BIL), ("P010_10LE", 3840, 2160, "BGRA", 1920, 1080, LIN),
         ("UYVY", 3840, 2160, "I420", 1920, 1080, LIN), ("YUY2", 3840, 2160, "NV12", 1920, 1080, BIL), ("YUY2", 3840, 2160, "NV12", 3840, 2160, {}),
         ("BGRA", 3840, 2160, "BGRA", 1920, 1080, CUB), ("BGRA", 3840, 2160, "BGRA", 1920, 1080, LIN),
         ("AYUV", 3840, 2160, "I420", 1920, 1080, LIN)]
if len(sys.argv) > 1:
    CASES = [CASES[int(a)] for a in sys.argv[1:]]
dev = torch.device("cuda:0")
for ifmt, w, h, ofmt, ow, oh, cfg in CASES:
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    try:
        conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    except Exception as e:
        print("%s %dx%d -> %s %dx%d %s: %s" % (ifmt, w, h, ofmt, ow, oh, cfg, e))
        continue
    n_in = 16
    src = torch.randint(0, 255, (n_in, int(ii.size)), dtype=torch.uint8, device=dev)
    dst = torch.zeros((n_in, int(oi.size)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(10):
        conv.frame(src[i % n_in], dst[i % n_in], st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for i in range(n):
        conv.frame(src[i % n_in], dst[i % n_in], st)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
    for i in range(3):
        conv.frames(srcs, dsts, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        conv.frames(srcs, dsts, st)
    torch.cuda.synchronize()
    usl = (time.perf_counter() - t0) / (40 * 8) * 1e6
    alg = conv.algorithmic_bytes()
    name = "bilinear" if cfg.get("max_taps") == 2 else cfg.get("resampler_method", "")
    print("%-10s %4dx%-4d -> %-10s %4dx%-4d %-8s single %6.1f us frac %.3f | lists of 8 %6.1f us frac %.3f (list launches %s) | %5.1f MB  %s" % (
        ifmt, w, h, ofmt, ow, oh, name, us, alg / (us * 1e-6) / 8e12, usl, alg / (usl * 1e-6) / 8e12, conv.list_launches(), alg / 1e6, conv.describe()), flush=True)
    conv.free()
