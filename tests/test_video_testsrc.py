"""GstVideoTestSrc's frames painted by this library (gstamd_video_test_pattern_*, `amdhipvideotestsrc`) against the reference ELEMENT itself: the hand-built 1.29
runtime's videotestsrc (oracle/_ref/rt129, test infrastructure) run through plugins/tests/launch129 into a file - gst/videotestsrc/videotestsrc.c's painters,
the caps' chroma downsampler and the format's pack function, byte for byte.  Host: the painter bodies on the emulator.  Device (-m gpu): through the C ABI."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import cases
from gstreamer_amd import video as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

BUILT = ["smpte", "snow", "black", "white", "red", "green", "blue", "checkers-1", "checkers-2", "checkers-4", "checkers-8", "blink", "smpte75", "smpte100",
         "solid-color", "bar", "gradient", "colors", "ball"]
NOT_BUILT = ["circular", "zone-plate", "gamut", "chroma-zone-plate", "pinwheel", "spokes", "smpte-rp-219"]


@pytest.fixture(scope="module")
def probe(ref):
    if not os.path.exists(os.path.join(ROOT, "plugins", "tests", "launch129")) or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "rt129", "plugins", "libgstvideotestsrc.so")):
        pytest.skip("the 1.29 runtime with the reference's videotestsrc is not built (oracle/rt129_build.py needs /root/reference)")
    import testsrc_probe
    return testsrc_probe


@pytest.mark.parametrize("pattern", BUILT)
def test_test_patterns_match_the_reference_element_on_host(native_lib, emu_lib, ref, probe, pattern):
    """three consecutive frames (the random generator of snow / smpte, blink and the ball run on from frame to frame) in planar, semi-planar, packed and RGB formats"""
    for fmt, w, h in (("I420", 320, 240), ("NV12", 70, 46), ("BGRA", 64, 48), ("UYVY", 53, 31), ("AYUV", 32, 24), ("RGB", 33, 17), ("I420_10LE", 64, 32), ("v210", 48, 16),
                      ("GRAY8", 40, 20), ("Y444", 21, 9)):
        assert probe.run(pattern, fmt, w, h, n=3, verbose=False) == "ok", (pattern, fmt, w, h)


def test_test_pattern_colours_follow_the_caps_and_the_properties_on_host(native_lib, emu_lib, ref, probe):
    """bt601 caps take the other colour table (videotestsrc_setup_paintinfo :205-214), foreground-color / background-color are converted with the caps' matrix"""
    for pattern in ("smpte", "ball", "bar", "snow"):
        assert probe.run(pattern, "I420", 96, 64, n=2, colorimetry="bt601", verbose=False) == "ok"
        assert probe.run(pattern, "NV12", 96, 64, n=2, fg=0x80ff2010, bg=0xff102040, extra="foreground-color=0x80ff2010 background-color=0xff102040", verbose=False) == "ok"
        assert probe.run(pattern, "BGRA", 96, 64, n=2, fg=0x80ff2010, bg=0xff102040, extra="foreground-color=0x80ff2010 background-color=0xff102040", verbose=False) == "ok"


def test_patterns_that_are_not_built_are_refused(native_lib):
    info = V.video_info("I420", 64, 48)
    for pattern in NOT_BUILT:
        with pytest.raises(V.GstAmdError) as e:
            V.VideoTestPattern(info, pattern)
        assert e.value.code == V.ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", BUILT)
def test_hip_test_patterns_match_the_reference_element(native_lib, gpu, ref, probe, pattern):
    import torch
    for fmt, w, h in (("NV12", 1920, 1080), ("I420", 320, 240), ("BGRA", 640, 360), ("UYVY", 53, 31), ("P010_10LE", 64, 32), ("RGB", 33, 17)):
        n = 2 if w > 1000 else 3
        want = probe.reference_frames(pattern, fmt, w, h, n)
        info = V.video_info(fmt, w, h)
        ri = ref.video_info(fmt, w, h)
        tp = V.VideoTestPattern(info, pattern)
        for k in range(n):
            d = torch.zeros(int(info.size), dtype=torch.uint8, device=gpu)
            tp.frame(k, d)
            torch.cuda.synchronize()
            got = d.cpu().numpy()
            a = cases.visible_bytes(fmt, w, h, list(ri["stride"]), list(ri["offset"]), got)
            b = cases.visible_bytes(fmt, w, h, list(ri["stride"]), list(ri["offset"]), want[k])
            assert (a == b).all(), (pattern, fmt, w, h, k, int((a != b).sum()), tp.describe())
        tp.free()
