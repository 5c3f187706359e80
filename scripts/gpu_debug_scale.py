import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, cases
from gstreamer_amd import video as V
E = C.CDLL(os.path.join(ROOT, "tests/emu/libgstamdemu.so"))
E.emu_video_convert.argtypes = [C.POINTER(V.VideoInfo), C.POINTER(V.VideoInfo), C.POINTER(V.ConverterConfig), C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
dev = torch.device("cuda:0")
def run(ifmt, w, h, ofmt, ow, oh, cfg):
    ii = V.video_info(ifmt, w, h); oi = V.video_info(ofmt, ow, oh)
    src = cases.frame_bytes(int(ii.size), "random", 5)
    c = V.converter_config(**cfg)
    exp = np.zeros(int(oi.size), np.uint8); desc = C.create_string_buffer(256)
    r = E.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(c), src.ctypes.data, exp.ctypes.data, 1, desc, 256)
    conv = V.VideoConverter(ii, oi, c)
    d_src = torch.from_numpy(src).to(dev); d_dst = torch.zeros(int(oi.size), dtype=torch.uint8, device=dev)
    conv.frame(d_src, d_dst); torch.cuda.synchronize()
    out = d_dst.cpu().numpy()
    nd = int((out != exp).sum())
    print(ifmt, w, h, ofmt, ow, oh, desc.value.decode(), "mismatch", nd, "/", out.size)
    if nd:
        idx = np.nonzero(out != exp)[0]
        print("  first idx", idx[:8], "exp", exp[idx[:8]], "got", out[idx[:8]], "rows with diffs:", np.unique(idx // (ow * 4))[:10])
        print("  head exp", exp[:16], "got", out[:16])
LIN = dict(resampler_method="linear", max_taps=2); LAN = dict(resampler_method="lanczos"); NEAR = dict(resampler_method="nearest")
run("BGRA", 64, 32, "RGBA", 32, 32, NEAR)
run("BGRA", 64, 32, "RGBA", 64, 16, NEAR)
run("BGRA", 64, 32, "RGBA", 32, 32, LIN)
run("BGRA", 64, 32, "RGBA", 64, 16, LIN)
run("BGRA", 64, 32, "RGBA", 32, 32, LAN)
run("BGRA", 64, 32, "RGBA", 64, 16, LAN)
run("BGRA", 64, 32, "RGBA", 32, 16, LAN)
run("NV12", 64, 32, "BGRA", 32, 32, LAN)
run("Y42B", 64, 32, "BGRA", 64, 16, LAN)
run("NV12", 640, 360, "BGRA", 320, 180, {})
