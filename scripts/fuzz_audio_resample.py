#!/usr/bin/env python3
"""Randomised differential test of the resampler: the FIR kernel bodies on the host emulator against the reference's
gst_audio_resampler_resample over random formats, channel counts, rate pairs, methods, qualities and buffer sizes; integer formats
byte-exact, floats within 1 ULP (north_star).  python scripts/fuzz_audio_resample.py <seed> <count>"""
import ctypes as C
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from gstreamer_amd import audio as A  # noqa: E402
from oracle import ref  # noqa: E402

E = C.CDLL(os.path.join(ROOT, "tests", "emu", "libgstamdemu.so"))
E.emu_audio_new.restype = C.c_void_p
E.emu_audio_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(A.ResamplerOptions), C.POINTER(C.c_int), C.c_char_p, C.c_int]
E.emu_audio_get_out_frames.restype = C.c_size_t
E.emu_audio_get_out_frames.argtypes = [C.c_void_p, C.c_size_t]
E.emu_audio_resample.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
E.emu_audio_free.argtypes = [C.c_void_p]


def ulps(a, b):
    if a.dtype == np.float32:
        ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    else:
        ia, ib = a.view(np.int64), b.view(np.int64)
    ia = np.where(ia < 0, np.int64(-(2 ** 31 if a.dtype == np.float32 else 2 ** 63)) - ia, ia)
    ib = np.where(ib < 0, np.int64(-(2 ** 31 if a.dtype == np.float32 else 2 ** 63)) - ib, ib)
    return int(np.abs(ia - ib).max()) if a.size else 0


def main():
    seed, n = int(sys.argv[1]), int(sys.argv[2])
    rnd = random.Random(seed)
    rates = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000]
    ok = bad = refused = 0
    for it in range(n):
        fmt = rnd.choice(["F32LE", "F64LE", "S16LE", "S32LE"])
        ch = rnd.choice([1, 2, 2, 3, 6])
        ir, orr = rnd.choice(rates), rnd.choice(rates)
        method = rnd.choice(["nearest", "linear", "cubic", "blackman-nuttall", "kaiser", "kaiser"])
        quality = rnd.randint(0, 10)
        bufs = [rnd.choice([1, 37, 256, 1024, 4096]) for _ in range(rnd.randint(1, 4))]
        o = A.options(method, quality, ir, orr)
        st = C.c_int(0)
        h = E.emu_audio_new(A.METHODS[method], 0, A.FORMATS[fmt], ch, ir, orr, C.byref(o), C.byref(st), None, 0)
        tag = (fmt, ch, ir, orr, method, quality, bufs)
        if not h:
            refused += 1
            continue
        rr = ref.AudioResampler(fmt, ch, ir, orr, method=method, quality=quality)
        dt = cases.AUDIO_DTYPES[fmt]
        good = True
        for i, nb in enumerate(bufs):
            data = cases.audio_buffer(fmt, ch, nb, seed * 100 + it * 7 + i)
            no = E.emu_audio_get_out_frames(h, nb)
            if no != rr.get_out_frames(nb):
                good = False
                print("OUT_FRAMES", tag, i, no, rr.get_out_frames(nb))
                break
            got = np.zeros((no, ch), dt)
            E.emu_audio_resample(h, data.ctypes.data, nb, got.ctypes.data, no)
            want = rr.resample(data, in_frames=nb, out_frames=no).reshape(no, ch)
            if np.issubdtype(dt, np.floating):
                u = ulps(got.reshape(-1), want.reshape(-1).astype(dt))
                if u > 1:
                    good = False
                    print("ULP", tag, i, u)
                    break
            elif not (got == want).all():
                good = False
                print("MISMATCH", tag, i, int((got != want).sum()))
                break
        E.emu_audio_free(h)
        ok += good
        bad += not good
    print("seed %d: ok %d refused %d bad %d" % (seed, ok, refused, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
