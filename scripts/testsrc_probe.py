#!/usr/bin/env python3
"""GstVideoTestSrc's painters on the host emulator against the reference element itself (the hand-built 1.29 runtime's videotestsrc, run through
plugins/tests/launch129): python scripts/testsrc_probe.py [pattern ...]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from gstreamer_amd import video as V  # noqa: E402
from oracle import ref  # noqa: E402

emu = C.CDLL(os.path.join(ROOT, "tests", "emu", "libgstamdemu.so"))
emu.emu_video_test_pattern.argtypes = [C.POINTER(V.VideoInfo), C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_char_p, C.c_int]


def env129():
    """plugins/tests/launch129 loads the plugin files it is told about (no registry in the hand-built runtime)"""
    rt = os.path.join(ROOT, "oracle", "_ref", "rt129")
    e = dict(os.environ)
    plugs = [os.path.join(rt, "plugins", f) for f in ("libgstcoreelements.so", "libgstvideotestsrc.so")]
    e.update(GSTAMD_LAUNCH_PLUGINS=":".join(plugs), LD_LIBRARY_PATH=os.path.join(rt, "lib") + ":" + e.get("LD_LIBRARY_PATH", ""))
    return e


def reference_frames(pattern, fmt, w, h, n, extra="", colorimetry=None):
    out = os.path.join(tempfile.gettempdir(), "vts_%s_%s_%dx%d.raw" % (pattern, fmt, w, h))
    caps = "video/x-raw,format=%s,width=%d,height=%d,framerate=30/1" % (fmt, w, h) + (",colorimetry=%s" % colorimetry if colorimetry else "")
    cmd = [os.path.join(ROOT, "plugins", "tests", "launch129"), "-q"] + ("videotestsrc num-buffers=%d pattern=%s %s ! %s ! filesink location=%s" % (n, pattern, extra, caps, out)).split()
    r = subprocess.run(cmd, env=env129(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:]
    return np.fromfile(out, np.uint8).reshape(n, -1)


def run(pattern, fmt, w, h, n=3, fg=0xffffffff, bg=0xff000000, extra="", colorimetry=None, verbose=True):
    want = reference_frames(pattern, fmt, w, h, n, extra, colorimetry)
    info = V.video_info(fmt, w, h, colorimetry=colorimetry)
    ri = ref.video_info(fmt, w, h)
    bad = 0
    desc = C.create_string_buffer(512)
    for k in range(n):
        dst = np.zeros(int(info.size), np.uint8)
        r = emu.emu_video_test_pattern(C.byref(info), V.TEST_PATTERNS[pattern], fg, bg, k, dst.ctypes.data, desc, 512)
        if r != 0:
            if verbose:
                print("refused  ", pattern, fmt, w, h, desc.value.decode()[:80])
            return "refused"
        a = cases.visible_bytes(fmt, w, h, list(ri["stride"]), list(ri["offset"]), dst)
        b = cases.visible_bytes(fmt, w, h, list(ri["stride"]), list(ri["offset"]), want[k])
        bad += int((a != b).sum())
    if bad:
        print("BAD      ", pattern, fmt, w, h, "|", bad, "bytes over", n, "frames |", desc.value.decode()[:100])
        return "bad"
    if verbose:
        print("ok       ", pattern, fmt, w, h, "|", desc.value.decode()[:100])
    return "ok"


if __name__ == "__main__":
    pats = sys.argv[1:] or ["smpte", "snow", "black", "white", "red", "green", "blue", "checkers-1", "checkers-2", "checkers-4", "checkers-8", "blink", "smpte75",
                            "smpte100", "solid-color", "bar", "gradient", "colors", "ball"]
    for pat in pats:
        for fmt in ("I420", "BGRA", "NV12", "AYUV", "UYVY", "RGB"):
            run(pat, fmt, 320, 240)
            run(pat, fmt, 70, 46, verbose=False)
