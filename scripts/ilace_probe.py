#!/usr/bin/env python3
"""Interlaced conversions on the host emulator (tests/emu) against the reference run with interlace-mode=interleaved (oracle/_ref): a sweep of format
pairs, sizes and methods; one line per case - ok / BAD (bytes that differ) / refused (why).  python scripts/ilace_probe.py [filter]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from gstreamer_amd import video as V  # noqa: E402
from oracle import ref  # noqa: E402

emu = C.CDLL(os.path.join(ROOT, "tests", "emu", "libgstamdemu.so"))
emu.emu_video_convert.argtypes = [C.POINTER(V.VideoInfo), C.POINTER(V.VideoInfo), C.POINTER(V.ConverterConfig), C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]


def run(ifmt, w, h, ofmt, ow, oh, cfg, site=None, verbose=True):
    ii = V.video_info(ifmt, w, h, chroma_site=site)
    oi = V.video_info(ofmt, ow, oh)
    ii.interlace_mode = oi.interlace_mode = 1
    c = V.converter_config(**cfg)
    src = cases.frame_bytes(int(ii.size), "random", 1234 + w * 7 + h, w)
    dst = np.zeros(int(oi.size), np.uint8)
    desc = C.create_string_buffer(512)
    r = emu.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(c), src.ctypes.data, dst.ctypes.data, 1, desc, 512)
    name = "%s %dx%d -> %s %dx%d %s%s" % (ifmt, w, h, ofmt, ow, oh, cfg, " site=" + site if site else "")
    if r != 0:
        if verbose:
            print("refused  ", name, "|", desc.value.decode()[:110])
        return "refused"
    try:
        want = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_chroma_site=site, config=cases.ref_config_string(ref, dict(cfg, threads=1)), interlaced=True).frame(src)
    except Exception as e:
        print("ref-fail ", name, e)
        return "ref-fail"
    ri = ref.video_info(ofmt, ow, oh)
    a = cases.visible_bytes(ofmt, ow, oh, list(ri["stride"]), list(ri["offset"]), dst)
    b = cases.visible_bytes(ofmt, ow, oh, list(ri["stride"]), list(ri["offset"]), want)
    bad = int((a != b).sum())
    if bad:
        rows = sorted(set(np.nonzero(a != b)[0] // max(1, (a.size // max(1, oh)))))[:12]
        print("BAD      ", name, "|", bad, "of", a.size, "| first rows ~", rows, "|", desc.value.decode()[:150])
        return "bad"
    if verbose:
        print("ok       ", name, "|", desc.value.decode()[:150])
    return "ok"


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    fmts = ["I420", "YV12", "NV12", "NV21", "YUY2", "UYVY", "AYUV", "Y42B", "Y444", "BGRA", "RGBx", "RGB", "NV16", "GRAY8", "v308", "A420"]
    tally = {}
    for a in fmts:
        for b in fmts:
            for (w, h, ow, oh) in ((32, 16, 32, 16), (32, 16, 24, 8), (22, 12, 40, 20)):
                for cfg in ({}, {"resampler_method": 1}):          # cubic (library default), linear
                    if cfg and (w, h) == (ow, oh):
                        continue
                    tag = "%s>%s" % (a, b)
                    if flt and flt not in tag:
                        continue
                    r = run(a, w, h, b, ow, oh, cfg, verbose=bool(flt))
                    tally[r] = tally.get(r, 0) + 1
    print(tally)


def diff_ayuv(ifmt, w, h, ow, oh, cfg, site=None):
    """-> per component, the rows of an AYUV destination that differ between the emulator and the reference"""
    ii = V.video_info(ifmt, w, h, chroma_site=site)
    oi = V.video_info("AYUV", ow, oh)
    ii.interlace_mode = oi.interlace_mode = 1
    c = V.converter_config(**cfg)
    src = cases.frame_bytes(int(ii.size), "random", 1234 + w * 7 + h, w)
    dst = np.zeros(int(oi.size), np.uint8)
    desc = C.create_string_buffer(512)
    r = emu.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(c), src.ctypes.data, dst.ctypes.data, 1, desc, 512)
    assert r == 0, desc.value
    want = ref.VideoConverter(ifmt, w, h, "AYUV", ow, oh, in_chroma_site=site, config=cases.ref_config_string(ref, dict(cfg, threads=1)), interlaced=True).frame(src)
    a, b = dst.reshape(oh, ow, 4), want.reshape(oh, ow, 4)
    return {n: [int(r) for r in np.nonzero((a[:, :, k] != b[:, :, k]).any(axis=1))[0]] for k, n in enumerate("AYUV")}, desc.value.decode()
