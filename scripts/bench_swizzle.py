import sys, os, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
from gstreamer_amd import video as V
dev = torch.device("cuda:0")
for (a, b) in (("BGRA", "RGBA"), ("ARGB", "BGRx"), ("BGRA", "AYUV"), ("AYUV", "BGRA")):
    ii, oi = V.video_info(a, 3840, 2160), V.video_info(b, 3840, 2160)
    c = V.VideoConverter(ii, oi)
    src = torch.randint(0, 255, (8, int(ii.size)), dtype=torch.uint8, device=dev)
    dst = torch.zeros((8, int(oi.size)), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for i in range(16): c.frame(src[i % 8].data_ptr(), dst[i % 8].data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): c.frame(src[i % 8].data_ptr(), dst[i % 8].data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(a, b, c.describe(), round(us, 2), "us", round(66.36e6 / us / 1e6, 2), "TB/s")
