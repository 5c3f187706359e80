"""k_scale_col on planar vs semi-planar sources (the semi-planar form reads its raw chroma windows with 4-byte LDS loads on 2-byte alignment): lists of 8"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gstreamer_amd import video as V
dev = torch.device("cuda:0")
for ifmt, w, h, ow, oh, m in (("I420", 7680, 4320, 1920, 1080, "lanczos"), ("NV12", 7680, 4320, 1920, 1080, "lanczos"), ("I420", 3840, 2160, 1920, 1080, "lanczos"), ("NV12", 3840, 2160, 1920, 1080, "lanczos"),
                              ("I420", 3840, 2160, 2560, 1440, "lanczos"), ("NV12", 3840, 2160, 2560, 1440, "lanczos"), ("NV12", 3840, 2160, 1920, 1080, "cubic"), ("I420", 3840, 2160, 1920, 1080, "cubic")):
    ii, oi = V.video_info(ifmt, w, h), V.video_info("RGBA", ow, oh)
    conv = V.VideoConverter(ii, oi, V.converter_config(resampler_method=m))
    src = torch.randint(0, 255, (8, int(ii.size)), dtype=torch.uint8, device=dev)
    dst = torch.zeros((8, int(oi.size)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
    for i in range(3): conv.frames(srcs, dsts, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): conv.frames(srcs, dsts, st)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 160 * 1e6
    alg = conv.algorithmic_bytes()
    print("%s %dx%d -> RGBA %dx%d %s: %.1f us per frame in lists of 8, frac %.3f, list launches %s  %s" % (ifmt, w, h, ow, oh, m, us, alg / (us * 1e-6) / 8e12, conv.list_launches(), conv.describe()[:60]), flush=True)
