"""The GStreamer elements build against the runtime of this image (1.14) AND type-check against the reference's own headers (1.29)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_elements_compile_against_the_reference_headers(native_lib):
    if not os.path.isdir("/opt/conda/include/gstreamer-1.0"):
        pytest.skip("no GStreamer development files in this image")
    sys.path.insert(0, os.path.join(ROOT, "plugins"))
    import build as plugin_build
    plugin_build.build()
    checked = plugin_build.check_against_reference_headers()
    if checked is None:
        pytest.skip("/root/reference (and oracle/_ref/gen) not present here")
    assert "gstamdcompositor.c" in checked and "gstamdvideoconvertscale.c" in checked and len(checked) >= 8
