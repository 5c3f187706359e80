#!/usr/bin/env python3
"""Development helper: kernel bodies on the host emulator vs the reference (oracle/_ref) for ad-hoc conversions.
   python scripts/cmp_emu_ref.py cases.py   where cases.py holds lines like  run("NV12", 64, 48, "RGB", 64, 48, cases.LIN)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from gstreamer_amd import video as V
from oracle import ref
emu = C.CDLL(os.path.join(ROOT, "tests", "emu", "libgstamdemu.so"))
emu.emu_video_convert.argtypes = [C.POINTER(V.VideoInfo), C.POINTER(V.VideoInfo), C.POINTER(V.ConverterConfig), C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_char_p, C.c_int]


def run(ifmt, w, h, ofmt, ow, oh, cfg={}, col=None, site=None, seed=1):
    col, ocol = cases.split_colorimetry(col)
    ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site)
    oi = V.video_info(ofmt, ow, oh, colorimetry=ocol)
    src = cases.frame_bytes(ii.size, "random", seed, w)
    c = V.converter_config(**cfg)
    dst = np.zeros(oi.size, np.uint8)
    desc = C.create_string_buffer(256)
    r = emu.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(c), src.ctypes.data, dst.ctypes.data, 1, desc, 256)
    tag = "%-6s %4dx%-4d -> %-6s %4dx%-4d %s" % (ifmt, w, h, ofmt, ow, oh, cfg)
    if r != 0:
        print("%s: REFUSED %s" % (tag, desc.value.decode()))
        return
    want = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, out_colorimetry=ocol,
                               config=cases.ref_config_string(ref, cfg)).frame(src)
    bad = int((dst != want).sum())
    if bad:
        vb = lambda b: cases.visible_bytes(ofmt, ow, oh, list(oi.stride), list(oi.offset), b)
        if (vb(dst) == vb(want)).all():
            print("%s: OK(visible) [%s]" % (tag, desc.value.decode()))
            return
    emu.emu_video_last_divergence.restype = C.c_char_p
    defined = " (DEFINED: the plan announces a divergence)" if emu.emu_video_last_divergence() else ""
    print("%s: %s%s  [%s]" % (tag, "OK" if bad == 0 else "MISMATCH %d of %d, first at %d" % (bad, dst.size, int(np.argmax(dst != want))), defined, desc.value.decode()))


if __name__ == "__main__":
    exec(open(sys.argv[1]).read())
