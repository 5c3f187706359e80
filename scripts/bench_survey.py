"""Common conversions at 4K, one frame per launch: us per frame and the fraction of the 8 TB/s HBM peak their algorithmic bytes give -
where the generic kernels stand (a survey, not a bench.py line).   python scripts/bench_survey.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                              # noqa: E402
from gstreamer_amd import video as V      # noqa: E402

# BIL = the elements' default method (bilinear: linear with max-taps 2); LIN = the library's linear (taps grow with the ratio)
BIL, LIN, CUB, LAN = dict(resampler_method="linear", max_taps=2), dict(resampler_method="linear"), dict(resampler_method="cubic"), dict(resampler_method="lanczos")
CASES = [("BGRA", 3840, 2160, "BGRA", 1920, 1080, BIL), ("NV12", 3840, 2160, "BGRA", 1920, 1080, BIL), ("NV12", 3840, 2160, "NV12", 1920, 1080, BIL),
         ("BGRA", 3840, 2160, "NV12", 1920, 1080, BIL), ("NV12", 1920, 1080, "BGRA", 3840, 2160, BIL), ("I420", 1920, 1080, "I420", 1280, 720, BIL),
         ("BGRA", 3840, 2160, "BGRA", 1920, 1080, LIN), ("BGRA", 3840, 2160, "BGRA", 1920, 1080, CUB), ("BGRA", 3840, 2160, "BGRA", 1280, 720, LAN),
         ("BGRA", 1920, 1080, "BGRA", 3840, 2160, LIN), ("BGRA", 3840, 2160, "NV12", 3840, 2160, {}), ("BGRA", 3840, 2160, "NV12", 1920, 1080, LIN),
         ("BGRA", 3840, 2160, "I420", 1920, 1080, CUB), ("NV12", 3840, 2160, "BGRA", 1920, 1080, LIN), ("NV12", 3840, 2160, "BGRA", 1920, 1080, LAN),
         ("NV12", 1920, 1080, "BGRA", 3840, 2160, LIN), ("I420", 3840, 2160, "BGRA", 3840, 2160, {}), ("I420", 3840, 2160, "NV12", 3840, 2160, {}),
         ("NV12", 3840, 2160, "I420", 1920, 1080, LIN), ("YUY2", 3840, 2160, "BGRA", 3840, 2160, {}), ("YUY2", 3840, 2160, "NV12", 3840, 2160, {}),
         ("UYVY", 3840, 2160, "I420", 1920, 1080, LIN), ("RGB", 3840, 2160, "BGRA", 3840, 2160, {}), ("BGRA", 3840, 2160, "RGB", 3840, 2160, {}),
         ("AYUV", 3840, 2160, "BGRA", 3840, 2160, {}), ("BGRA", 3840, 2160, "AYUV", 3840, 2160, {}), ("NV12", 3840, 2160, "RGB", 3840, 2160, {}),
         ("P010_10LE", 3840, 2160, "BGRA", 3840, 2160, {}), ("P010_10LE", 3840, 2160, "BGRA", 1920, 1080, LIN), ("BGRA", 3840, 2160, "P010_10LE", 3840, 2160, {})]
if len(sys.argv) > 1:          # python scripts/bench_survey.py P010 I420_10LE ...: only the cases naming one of these formats
    CASES = [c for c in CASES if c[0] in sys.argv[1:] or c[3] in sys.argv[1:]]
    CASES += [("I420_10LE", 3840, 2160, "BGRA", 3840, 2160, {}), ("P010_10LE", 3840, 2160, "NV12", 1920, 1080, BIL), ("NV12", 3840, 2160, "P010_10LE", 3840, 2160, {}),
              ("P010_10LE", 3840, 2160, "BGRA", 1920, 1080, BIL), ("P010_10LE", 1920, 1080, "BGRA", 3840, 2160, BIL),
              ("AYUV", 3840, 2160, "NV12", 3840, 2160, {}), ("BGRA", 3840, 2160, "Y444", 3840, 2160, {}), ("BGRA", 3840, 2160, "I420", 3840, 2160, dict(matrix_mode="none"))] if "P010_10LE" in sys.argv[1:] else []
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
    alg = conv.algorithmic_bytes()
    print("%-10s %4dx%-4d -> %-10s %4dx%-4d %-8s %7.1f us  %6.1f MB  frac %.3f  %s" % (ifmt, w, h, ofmt, ow, oh, ("bilinear" if cfg.get("max_taps") == 2 else cfg.get("resampler_method", "")), us, alg / 1e6,
                                                                                 alg / (us * 1e-6) / 8e12, conv.describe()))
    conv.free()
