"""GStreamer-level drop-in test (-m gpu): the C elements of plugins/ run inside real pipelines (GStreamer 1.14
runtime from /opt/conda driving gst-launch-1.0) and their output files are compared byte for byte with the
reference library (oracle/_ref) fed the very frames the pipeline saw."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GST = "/opt/conda/bin/gst-launch-1.0"


@pytest.fixture(scope="module")
def gst_env(native_lib, tmp_path_factory):
    if not os.path.exists(GST):
        pytest.skip("no GStreamer runtime in this image")
    sys.path.insert(0, os.path.join(ROOT, "plugins"))
    import build as plugin_build
    so = plugin_build.build()
    assert os.path.exists(so)
    tmp = tmp_path_factory.mktemp("gst")
    env = dict(os.environ)
    env.update(GST_PLUGIN_PATH=os.path.join(ROOT, "plugins") + ":/opt/conda/lib/gstreamer-1.0", GST_PLUGIN_SYSTEM_PATH="/nonexistent",
               GST_REGISTRY=str(tmp / "registry.bin"), GST_REGISTRY_FORK="no",
               LD_LIBRARY_PATH=os.path.join(ROOT, "gstreamer_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", ""))
    # the conda runtime ships an older libstdc++ than the one hipcc links against: load the system one first
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sys_stdcpp):
        env["LD_PRELOAD"] = sys_stdcpp
    return env, tmp


def launch(env, pipeline):
    r = subprocess.run([GST, "-q"] + pipeline.split(), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_videoconvertscale_element_matches_reference(gst_env, ref):
    env, tmp = gst_env
    w, h, n = 1280, 720, 4
    fin, fout = tmp / "in.nv12", tmp / "out.bgra"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=BGRA ! filesink location=%s"
           % (n, w, h, fin, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt709", in_chroma_site="mpeg2")
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), i


def test_chained_elements_keep_frames_in_hbm_and_scale(gst_env, ref):
    """NV12 -> BGRA in HBM (memory:AMDHIPMemory between the two elements) -> RGBA; then a Lanczos downscale."""
    env, tmp = gst_env
    w, h, n = 640, 360, 3
    fin, fout, fsc = tmp / "in2.nv12", tmp / "out2.rgba", tmp / "out3.rgba"
    launch(env, "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw(memory:AMDHIPMemory),format=BGRA "
                "! videoconvertscale ! video/x-raw,format=RGBA ! filesink location=%s" % (n, w, h, fin, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    a = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt601", in_chroma_site="jpeg")
    b = ref.VideoConverter("BGRA", w, h, "RGBA", w, h)
    for i in range(n):
        assert (b.frame(a.frame(src[i])) == out[i]).all(), i
    launch(env, "filesrc location=%s blocksize=%d ! video/x-raw,format=NV12,width=%d,height=%d,framerate=30/1,colorimetry=bt601,chroma-site=jpeg "
                "! videoconvertscale method=lanczos ! video/x-raw,format=RGBA,width=160,height=90 ! filesink location=%s"
           % (fin, src.shape[1], w, h, fsc))
    sc = np.fromfile(fsc, np.uint8).reshape(n, -1)
    c = ref.VideoConverter("NV12", w, h, "RGBA", 160, 90, in_colorimetry="bt601", in_chroma_site="jpeg",
                           config=cases.ref_config_string(ref, cases.LAN))
    for i in range(n):
        assert (c.frame(src[i]) == sc[i]).all(), i


def test_audioresample_element_matches_reference(gst_env, ref):
    env, tmp = gst_env
    fin, fout = tmp / "in.f32", tmp / "out.f32"
    launch(env, "audiotestsrc num-buffers=40 wave=white-noise samplesperbuffer=1024 ! audio/x-raw,format=F32LE,rate=48000,channels=2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdaudioresample quality=4 ! audio/x-raw,rate=44100 ! filesink location=%s"
           % (fin, fout))
    src = np.fromfile(fin, np.float32).reshape(-1, 2)
    out = np.fromfile(fout, np.float32).reshape(-1, 2)
    rr = ref.AudioResampler("F32LE", 2, 48000, 44100, quality=4)
    exp = []
    for off in range(0, len(src), 1024):
        blk = src[off:off + 1024]
        exp.append(rr.resample(blk, in_frames=len(blk), out_frames=rr.get_out_frames(len(blk))))
    lat = rr.get_max_latency()
    exp.append(rr.resample(None, in_frames=lat, out_frames=rr.get_out_frames(lat)))       # EOS drain
    exp = np.concatenate(exp)
    assert out.shape == exp.shape and (out == exp).all()
