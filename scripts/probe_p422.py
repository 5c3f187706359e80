"""where the time of packed 4:2:2 -> scaled 4:2:0 goes: the same 2 x 2 scaler from sources of different fronts (us per 4K frame, one frame per call)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gstreamer_amd import video as V
BIL = dict(resampler_method="linear", max_taps=2)
LIN = dict(resampler_method="linear")
dev = torch.device("cuda:0")
CASES = [("BGRA", "NV12", BIL, None), ("AYUV", "NV12", BIL, None), ("YUY2", "NV12", BIL, None), ("YUY2", "NV12", dict(BIL, chroma_mode="none"), None),
         ("YUY2", "NV12", BIL, "mpeg2"), ("UYVY", "NV12", BIL, None), ("YUY2", "AYUV", BIL, None), ("YUY2", "BGRA", BIL, None), ("Y42B", "NV12", BIL, None),
         ("UYVY", "I420", LIN, None), ("AYUV", "I420", LIN, None), ("BGRA", "NV12", LIN, None), ("BGRA", "I420", dict(resampler_method="cubic"), None), ("BGRA", "BGRA", LIN, None), ("YUY2", "NV12", {}, None), ("AYUV", "NV12", {}, None)]
for ifmt, ofmt, cfg, site in CASES:
    ii, oi = V.video_info(ifmt, 3840, 2160, chroma_site=site), V.video_info(ofmt, 1920, 1080)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    src = torch.randint(0, 255, (8, int(ii.size)), dtype=torch.uint8, device=dev)
    dst = torch.zeros((8, int(oi.size)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(10):
        conv.frame(src[i % 8], dst[i % 8], st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        conv.frame(src[i % 8], dst[i % 8], st)
    torch.cuda.synchronize()
    print("%-5s -> %-5s %-40s site %-6s %6.1f us  %s" % (ifmt, ofmt, cfg, site, (time.perf_counter() - t0) / 200 * 1e6, conv.describe()), flush=True)
    conv.free()
