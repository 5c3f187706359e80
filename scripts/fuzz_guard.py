"""The fuzz draws of the GPU seeds through the host emulator with the source frame's END (or START: --front) against an unmapped page: a kernel body
that reads outside the frame dies here with a fault instead of reading a neighbour's bytes.   python scripts/fuzz_guard.py SEED [N] [--front]"""
import ctypes as C
import mmap
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cases                              # noqa: E402
import fuzz_video                         # noqa: E402
from gstreamer_amd import video as V      # noqa: E402

libc = C.CDLL(None, use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
PAGE = 4096


def guarded(size, front):
    """size bytes whose end (front: start) touches a PROT_NONE page -> (address, total mapping)"""
    body = (size + PAGE - 1) // PAGE * PAGE
    base = libc.mmap(None, body + 2 * PAGE, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    assert base and base != C.c_void_p(-1).value
    libc.mprotect(base, PAGE, 0)
    libc.mprotect(base + PAGE + body, PAGE, 0)
    return (base + PAGE) if front else (base + PAGE + body - size)


def main():
    seed, n = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 150
    front = "--front" in sys.argv
    emu = fuzz_video.load_emu()
    rnd = random.Random(seed)
    rects = random.Random(seed + 77) if seed >= 700 else None
    for it in range(n):
        case = fuzz_video.random_case(rnd, rects)
        ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
        ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site)
        oi = V.video_info(ofmt, ow, oh)
        src = cases.frame_bytes(ii.size, "random", seed * 1000 + it, w)
        # 16-byte aligned like a device allocation: the size rounded up, the frame at the end (the start) of it
        size = (int(ii.size) + 15) & ~15 if not front else int(ii.size)
        sa = guarded(size, front)
        C.memmove(sa, src.ctypes.data, int(ii.size))
        da = guarded((int(oi.size) + 15) & ~15, front)
        print(seed, it, case, flush=True)
        desc = C.create_string_buffer(256)
        emu.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(V.converter_config(**cfg)), sa, da, 1, desc, 256)


main()
