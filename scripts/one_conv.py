"""One conversion in a loop (for rocprofv3 --kernel-trace --stats):  python scripts/one_conv.py IN w h OUT ow oh [method [max_taps]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                              # noqa: E402
from gstreamer_amd import video as V      # noqa: E402

ifmt, w, h, ofmt, ow, oh = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
cfg = {}
if len(sys.argv) > 7:
    cfg["resampler_method"] = sys.argv[7]
if len(sys.argv) > 8:
    cfg["max_taps"] = int(sys.argv[8])
ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
dev = torch.device("cuda:0")
src = torch.randint(0, 255, (4, int(ii.size)), dtype=torch.uint8, device=dev)
dst = torch.zeros((4, int(oi.size)), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(60):
    conv.frame(src[i % 4], dst[i % 4], st)
torch.cuda.synchronize()
print(conv.describe())
