import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gstreamer_amd import video as V
BIL = dict(resampler_method="linear", max_taps=2)
CASES = [("P010_10LE", 3840, 2160, "P010_10LE", 1920, 1080, BIL), ("NV12", 3840, 2160, "I420", 1920, 1080, BIL), ("I420", 3840, 2160, "NV12", 1920, 1080, BIL),
         ("NV12", 3840, 2160, "NV12", 1920, 1080, BIL), ("I420", 3840, 2160, "I420", 1920, 1080, BIL), ("P010_10LE", 3840, 2160, "I420_10LE", 1920, 1080, BIL),
         ("NV12", 1920, 1080, "I420", 1280, 720, BIL), ("NV12", 3840, 2160, "I420", 3840, 2160, {}), ("P010_10LE", 3840, 2160, "I420", 3840, 2160, {}),
         ("P010_10LE", 1920, 1080, "NV12", 1280, 720, BIL), ("P010_10LE", 3840, 2160, "NV12", 1280, 720, BIL)]
LAN, CUB, L4 = dict(resampler_method="lanczos"), dict(resampler_method="cubic"), dict(resampler_method="linear")
if len(sys.argv) > 1 and sys.argv[1] == "ntap":
    CASES = [("NV12", 3840, 2160, "I420", 1920, 1080, LAN), ("NV12", 3840, 2160, "I420", 1920, 1080, CUB), ("NV12", 3840, 2160, "I420", 1920, 1080, L4), ("NV12", 1920, 1080, "I420", 1280, 720, LAN),
             ("NV12", 3840, 2160, "BGRA", 1920, 1080, LAN), ("NV12", 3840, 2160, "NV12", 1920, 1080, LAN), ("I420", 3840, 2160, "NV12", 1280, 720, CUB), ("NV12", 1920, 1080, "I420", 3840, 2160, BIL)]
if len(sys.argv) > 1 and sys.argv[1] == "deep":
    CASES = [("P010_10LE", 3840, 2160, "I420_10LE", 3840, 2160, {}), ("I420_10LE", 3840, 2160, "P010_10LE", 3840, 2160, {}), ("NV12", 3840, 2160, "I420_10LE", 3840, 2160, {}),
             ("NV12", 3840, 2160, "P010_10LE", 3840, 2160, {}), ("I420", 3840, 2160, "P010_10LE", 3840, 2160, {}), ("P010_10LE", 3840, 2160, "P016_LE", 3840, 2160, {}),
             ("P010_10LE", 3840, 2160, "NV12", 3840, 2160, {}), ("I420_10LE", 3840, 2160, "I420", 3840, 2160, {}), ("P010_10LE", 1920, 1080, "I420_10LE", 1920, 1080, {})]
dev = torch.device("cuda:0")
for ifmt, w, h, ofmt, ow, oh, cfg in CASES:
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    src = torch.randint(0, 255, (8, int(ii.size)), dtype=torch.uint8, device=dev)
    dst = torch.zeros((8, int(oi.size)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(5): conv.frame(src[i % 8], dst[i % 8], st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): conv.frame(src[i % 8], dst[i % 8], st)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 100 * 1e6
    srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
    for i in range(3): conv.frames(srcs, dsts, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): conv.frames(srcs, dsts, st)
    torch.cuda.synchronize(); usl = (time.perf_counter() - t0) / 160 * 1e6
    alg = conv.algorithmic_bytes()
    print("%-10s %4dx%-4d -> %-10s %4dx%-4d single %6.1f us %.3f | lists %6.1f us %.3f | %s" % (ifmt, w, h, ofmt, ow, oh, us, alg / (us * 1e-6) / 8e12, usl, alg / (usl * 1e-6) / 8e12, conv.describe()[:110]), flush=True)
    conv.free()
