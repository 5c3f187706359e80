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


RT129 = os.path.join(ROOT, "oracle", "_ref", "rt129")
LAUNCH129 = os.path.join(ROOT, "plugins", "tests", "launch129")


# Two runtimes: the GStreamer 1.14 of this image (conda: gst-launch-1.0, a registry, GstHarness tools) and - round 5 - the reference's OWN
# version, 1.29, hand-built from /root/reference by oracle/rt129_build.py (core, libgstbase, libgstvideo with GstVideoAggregator, libgstaudio,
# coreelements, videotestsrc, audiotestsrc; no gst_parse and no registry: plugins/tests/launch129 builds the pipelines and loads the plugins by
# path) with the elements of plugins/ compiled and linked against it (plugins/rt129/libgstamdhipdsp.so).
@pytest.fixture(scope="module", params=["1.14", "1.29"])
def gst_env(request, native_lib, tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "plugins"))
    import build as plugin_build
    env = dict(os.environ)
    tmp = tmp_path_factory.mktemp("gst" + request.param.replace(".", ""))
    if request.param == "1.29":
        so = plugin_build.build129()
        if not so or not os.path.exists(LAUNCH129) or not os.path.exists(os.path.join(RT129, "lib", "libgstvideo-1.0.so.0")):
            pytest.skip("the 1.29 runtime is not built (oracle/rt129_build.py needs /root/reference)")
        plugs = [os.path.join(RT129, "plugins", f) for f in ("libgstcoreelements.so", "libgstvideotestsrc.so", "libgstaudiotestsrc.so")] + [so]
        env.update(GSTAMD_LAUNCH_PLUGINS=":".join(plugs), GSTAMD_RUNTIME="1.29", GSTAMD_LAUNCH_BIN=LAUNCH129,
                   LD_LIBRARY_PATH=os.path.join(RT129, "lib") + ":" + os.path.join(ROOT, "gstreamer_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", ""))
        return env, tmp
    if not os.path.exists(GST):
        pytest.skip("no GStreamer runtime in this image")
    so = plugin_build.build()
    assert os.path.exists(so)
    env.update(GST_PLUGIN_PATH=os.path.join(ROOT, "plugins") + ":/opt/conda/lib/gstreamer-1.0", GST_PLUGIN_SYSTEM_PATH="/nonexistent",
               GST_REGISTRY=str(tmp / "registry.bin"), GST_REGISTRY_FORK="no", GSTAMD_RUNTIME="1.14", GSTAMD_LAUNCH_BIN=GST,
               LD_LIBRARY_PATH=os.path.join(ROOT, "gstreamer_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", ""))
    # the conda runtime ships an older libstdc++ than the one hipcc links against: load the system one first
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sys_stdcpp):
        env["LD_PRELOAD"] = sys_stdcpp
    return env, tmp


def only_on_114(env, why):
    """tools that exist for the conda runtime only (GstHarness programs need libgstcheck, gst-inspect a registry)"""
    if env.get("GSTAMD_RUNTIME") != "1.14":
        pytest.skip("1.29 runtime: " + why)


def launch(env, pipeline):
    r = subprocess.run([env["GSTAMD_LAUNCH_BIN"], "-q"] + pipeline.split(), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
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


def test_videoconvertscale_converter_config_replaces_the_element_options(gst_env, ref):
    """converter-config (gstvideoconvertscale.c:378-396, 962-967): the structure is the ONLY configuration the converter gets - here
    Lanczos with a destination rectangle and a border colour although `method` stays bilinear - and a structure that names nothing
    gives the library defaults (cubic), not the element's bilinear."""
    env, tmp = gst_env
    w, h, ow, oh, n = 640, 360, 400, 300, 2
    fin, fout, fout2 = tmp / "cc_in.nv12", tmp / "cc_out.bgra", tmp / "cc_out2.bgra"
    src_desc = ("videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! " % (n, w, h, fin))
    cc = ("cfg,GstVideoConverter.resampler-method=4,GstVideoConverter.dest-x=40,GstVideoConverter.dest-y=30,GstVideoConverter.dest-width=320,"
          "GstVideoConverter.dest-height=240,GstVideoConverter.border-argb=(uint)4280303680")
    r = subprocess.run([env["GSTAMD_LAUNCH_BIN"], "-q"] + src_desc.split() + ["videoconvertscale", "converter-config=" + cc, "!"] +
                       ("video/x-raw,format=BGRA,width=%d,height=%d,pixel-aspect-ratio=1/1 ! filesink location=%s" % (ow, oh, fout)).split(),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    cfg = dict(resampler_method="lanczos", dest_x=40, dest_y=30, dest_width=320, dest_height=240, border_argb=4280303680)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", ow, oh, config=cases.ref_config_string(ref, cfg), in_colorimetry="bt601", in_chroma_site="jpeg")
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), (i, int((rc.frame(src[i]) != out[i]).sum()))
    r = subprocess.run([env["GSTAMD_LAUNCH_BIN"], "-q"] + src_desc.split() + ["videoconvertscale", "converter-config=cfg", "!"] +
                       ("video/x-raw,format=BGRA,width=%d,height=%d,pixel-aspect-ratio=1/1 ! filesink location=%s" % (ow, oh, fout2)).split(),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    out = np.fromfile(fout2, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", ow, oh, in_colorimetry="bt601", in_chroma_site="jpeg")
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), (i, int((rc.frame(src[i]) != out[i]).sum()))


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


def test_videoconvertscale_element_packed_444_formats(gst_env, ref):
    """v308 and IYU2 (3-byte packed 4:4:4 YUV) on either side of the element"""
    env, tmp = gst_env
    w, h, n = 322, 242, 2
    fin, fmid, fout = tmp / "p.nv12", tmp / "p.v308", tmp / "p.iyu2"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=v308 ! tee name=u "
                "u. ! queue ! filesink location=%s u. ! queue ! videoconvertscale ! video/x-raw,format=IYU2,width=160,height=120 ! filesink location=%s"
           % (n, w, h, fin, fmid, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    mid = np.fromfile(fmid, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    c0 = ref.VideoConverter("NV12", w, h, "v308", w, h, in_colorimetry="bt601", in_chroma_site="jpeg", out_colorimetry="bt601")
    c1 = ref.VideoConverter("v308", w, h, "IYU2", 160, 120, in_colorimetry="bt601", out_colorimetry="bt601",
                            config=cases.ref_config_string(ref, cases.LIN))
    for f in range(n):
        assert (c0.frame(src[f]) == mid[f]).all()
        assert (c1.frame(mid[f]) == out[f]).all()


def test_videoconvertscale_element_gray8(gst_env, ref):
    """GRAY8 on either side: NV12 -> GRAY8 half size (the luma plane scaled alone), GRAY8 -> BGRA"""
    env, tmp = gst_env
    w, h, n = 322, 242, 2
    fin, fmid, fout = tmp / "g.nv12", tmp / "g.gray", tmp / "g.bgra"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=GRAY8,width=160,height=120 ! tee name=u "
                "u. ! queue ! filesink location=%s u. ! queue ! videoconvertscale ! video/x-raw,format=BGRA,width=160,height=120 ! filesink location=%s"
           % (n, w, h, fin, fmid, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    mid = np.fromfile(fmid, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    c0 = ref.VideoConverter("NV12", w, h, "GRAY8", 160, 120, in_colorimetry="bt601", in_chroma_site="jpeg", config=cases.ref_config_string(ref, cases.LIN))
    c1 = ref.VideoConverter("GRAY8", 160, 120, "BGRA", 160, 120)
    for f in range(n):
        assert (c0.frame(src[f]) == mid[f]).all()
        assert (c1.frame(mid[f]) == out[f]).all()


def test_videoconvertscale_element_round5_formats(gst_env, ref):
    """the formats of round 5 through the element, as a destination and as a source: GRAY16_LE on both runtimes; RGB10A2_LE (1.18) and
    RGBA64_BE / AV12 (1.20) where the runtime knows them (the reference's own version)"""
    env, tmp = gst_env
    fmts = ["GRAY16_LE", "RGB16", "A420", "Y41B"] + (["RGB10A2_LE", "RGBA64_BE", "AV12"] if env.get("GSTAMD_RUNTIME") == "1.29" else [])
    w, h, n = 322, 242, 2
    for fmt in fmts:
        fin, fmid, fout = tmp / ("r5_%s.bgra" % fmt), tmp / ("r5_%s.mid" % fmt), tmp / ("r5_%s.out" % fmt)
        launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=%d,height=%d "
                    "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=%s,width=192,height=120 ! tee name=u "
                    "u. ! queue ! filesink location=%s u. ! queue ! videoconvertscale ! video/x-raw,format=BGRA,width=192,height=120 ! filesink location=%s"
               % (n, w, h, fin, fmt, fmid, fout))
        src = np.fromfile(fin, np.uint8).reshape(n, -1)
        mid = np.fromfile(fmid, np.uint8).reshape(n, -1)
        out = np.fromfile(fout, np.uint8).reshape(n, -1)
        c0 = ref.VideoConverter("BGRA", w, h, fmt, 192, 120, config=cases.ref_config_string(ref, cases.LIN))
        c1 = ref.VideoConverter(fmt, 192, 120, "BGRA", 192, 120)
        for f in range(n):
            assert (c0.frame(src[f]) == mid[f]).all(), fmt
            assert (c1.frame(mid[f]) == out[f]).all(), fmt


def test_videoconvertscale_element_v210(gst_env, ref):
    """v210 (SDI capture / playout) on either side through the generic 16-bit chain: BGRA -> v210 at another size, v210 -> BGRA"""
    env, tmp = gst_env
    w, h, n = 322, 242, 2
    fin, fmid, fout = tmp / "v.bgra", tmp / "v.v210", tmp / "v.out"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=%d,height=%d "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=v210,width=192,height=120,colorimetry=bt709 ! tee name=u "
                "u. ! queue ! filesink location=%s u. ! queue ! videoconvertscale ! video/x-raw,format=BGRA,width=192,height=120 ! filesink location=%s"
           % (n, w, h, fin, fmid, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    mid = np.fromfile(fmid, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    c0 = ref.VideoConverter("BGRA", w, h, "v210", 192, 120, out_colorimetry="bt709", config=cases.ref_config_string(ref, cases.LIN))
    c1 = ref.VideoConverter("v210", 192, 120, "BGRA", 192, 120, in_colorimetry="bt709")
    for f in range(n):
        assert (c0.frame(src[f]) == mid[f]).all()
        assert (c1.frame(mid[f]) == out[f]).all()


def test_videoconvertscale_chroma_resampler_property(gst_env, ref):
    """`chroma-resampler` (gstvideoconvertscale.c:137, 345, 1076): the method of the chroma planes when a planar frame is scaled in its
    own format (convert_scale_planes) - cubic luma with nearest / lanczos chroma against the reference's chroma-resampler-method"""
    env, tmp = gst_env
    w, h, ow, oh, n = 320, 240, 200, 136, 2
    fin = tmp / "cr.nv12"
    outs = {m: tmp / ("cr_%s.nv12" % m) for m in ("linear", "nearest", "lanczos")}
    branches = " ".join("t. ! queue ! videoconvertscale method=catrom chroma-resampler=%s ! video/x-raw,format=NV12,width=%d,height=%d ! filesink location=%s"
                        % (m, ow, oh, f) for m, f in outs.items())
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg "
                "! tee name=t t. ! queue ! filesink location=%s %s" % (n, w, h, fin, branches))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    results = {}
    for m, f in outs.items():
        out = np.fromfile(f, np.uint8).reshape(n, -1)
        cfg = ref.config_string(GstVideoConverter__resampler_method="cubic", GstVideoResampler__cubic_b=0.0, GstVideoResampler__cubic_c=0.5,
                                GstVideoConverter__chroma_resampler_method=m)
        c = ref.VideoConverter("NV12", w, h, "NV12", ow, oh, in_colorimetry="bt601", in_chroma_site="jpeg", out_colorimetry="bt601", out_chroma_site="jpeg",
                               config=cfg)
        for k in range(n):
            assert (c.frame(src[k]) == out[k]).all(), m
        results[m] = out
    assert (results["linear"] != results["nearest"]).any() and (results["linear"] != results["lanczos"]).any()


def test_videoconvertscale_element_12_bit_formats(gst_env, ref):
    """I420_12LE out of the element and back into it as Y444_10LE -> BGRA"""
    env, tmp = gst_env
    w, h, n = 320, 240, 2
    fin, fmid, fout = tmp / "d.nv12", tmp / "d.i42012", tmp / "d.bgra"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=I420_12LE ! tee name=u "
                "u. ! queue ! filesink location=%s u. ! queue ! videoconvertscale ! video/x-raw,format=Y444_10LE ! videoconvertscale ! video/x-raw,format=BGRA "
                "! filesink location=%s" % (n, w, h, fin, fmid, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    mid = np.fromfile(fmid, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    kw = dict(in_colorimetry="bt601", in_chroma_site="jpeg", out_colorimetry="bt601")
    c0 = ref.VideoConverter("NV12", w, h, "I420_12LE", w, h, out_chroma_site="jpeg", **kw)
    c1 = ref.VideoConverter("I420_12LE", w, h, "Y444_10LE", w, h, out_chroma_site="jpeg", **kw)
    c2 = ref.VideoConverter("Y444_10LE", w, h, "BGRA", w, h, in_colorimetry="bt601", in_chroma_site="jpeg")
    for f in range(n):
        m = c0.frame(src[f])
        assert (m == mid[f]).all()
        assert (c2.frame(c1.frame(m)) == out[f]).all()


def test_compositor_element_matches_reference(gst_env, ref):
    """`compositor` element (GstAggregator subclass, plugins/gstamdcompositor.c): three BGRA pads with positions, pad
    alpha, zorder and the `source` operator over the checker background, then a transparent background whose first pad
    is a videoconvertscale output that stays in HBM.  Expected frames: the reference's own fill + blend functions
    applied pad by pad in zorder (compositor.c:1619-1697)."""
    env, tmp = gst_env
    n, dw, dh = 3, 320, 240
    pads = [("smpte", 320, 240, 0, 0, 1.0, 1), ("ball", 160, 120, 100, 80, 0.6, 1), ("snow", 64, 48, -10, 200, 0.8, 0)]
    files = [tmp / ("cin%d.bgra" % i) for i in range(len(pads))]
    fout = tmp / "cout.bgra"
    desc = "compositor name=c background=checker"
    for i, (pat, w, h, x, y, a, op) in enumerate(pads):
        desc += " sink_%d::xpos=%d sink_%d::ypos=%d sink_%d::alpha=%s sink_%d::operator=%d" % (i, x, i, y, i, a, i, op)
    # the canvas leaves the compositor as HBM buffers of its GstAmdHipBufferPool; videoconvertscale brings it to RGBA
    desc += (" ! video/x-raw(memory:AMDHIPMemory),format=BGRA,width=%d,height=%d ! videoconvertscale ! video/x-raw,format=RGBA "
             "! filesink location=%s" % (dw, dh, fout))
    for i, (pat, w, h, x, y, a, op) in enumerate(pads):
        desc += (" videotestsrc num-buffers=%d pattern=%s foreground-color=0x80ff4020 ! video/x-raw,format=BGRA,width=%d,height=%d,framerate=30/1 "
                 "! tee name=t%d t%d. ! queue ! filesink location=%s t%d. ! queue ! c.sink_%d" % (n, pat, w, h, i, i, files[i], i, i))
    launch(env, desc)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    ins = [np.fromfile(f, np.uint8).reshape(n, -1) for f in files]
    for f in range(n):
        canvas = np.zeros(dw * dh * 4, np.uint8)
        ref.compositor_fill(0, "bgra", "BGRA", canvas, dw, dh, 0, dh)
        for i, (pat, w, h, x, y, a, op) in enumerate(pads):
            ref.compositor_blend("blend_bgra", "BGRA", ins[i][f], w, h, x, y, a, canvas, dw, dh, 0, dh, op)
        exp = canvas.reshape(-1, 4)[:, [2, 1, 0, 3]].reshape(-1)      # BGRA -> RGBA
        assert (exp == out[f]).all(), (f, int((exp != out[f]).sum()))

    # transparent background -> overlay functions; pad 0 arrives as memory:AMDHIPMemory from videoconvertscale
    fin0, fin1, fout2 = tmp / "o0.nv12", tmp / "o1.argb", tmp / "o.argb"
    launch(env, "compositor name=c background=transparent sink_1::xpos=40 sink_1::ypos=30 sink_1::alpha=0.5 ! video/x-raw,format=ARGB ! filesink location=%s "
                "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=256,height=144,framerate=30/1,colorimetry=bt601,chroma-site=jpeg "
                "! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! videoconvertscale ! video/x-raw(memory:AMDHIPMemory),format=ARGB ! c.sink_0 "
                "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=ARGB,width=128,height=72,framerate=30/1 "
                "! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1" % (fout2, n, fin0, n, fin1))
    out = np.fromfile(fout2, np.uint8).reshape(n, -1)
    s0 = np.fromfile(fin0, np.uint8).reshape(n, -1)
    s1 = np.fromfile(fin1, np.uint8).reshape(n, -1)
    cv = ref.VideoConverter("NV12", 256, 144, "ARGB", 256, 144, in_colorimetry="bt601", in_chroma_site="jpeg")
    for f in range(n):
        canvas = np.zeros(256 * 144 * 4, np.uint8)        # transparent background = zero fill (compositor.c:1650)
        ref.compositor_blend("overlay_argb", "ARGB", cv.frame(s0[f]), 256, 144, 0, 0, 1.0, canvas, 256, 144, 0, 144, 1)
        ref.compositor_blend("overlay_argb", "ARGB", s1[f], 128, 72, 40, 30, 0.5, canvas, 256, 144, 0, 144, 1)
        assert (canvas == out[f]).all(), (f, int((canvas != out[f]).sum()))


def test_videoconvertscale_element_planar_output(gst_env, ref):
    """RGB -> NV12 (the encoder-feeding direction) with a bilinear downscale, and NV12 -> I420, through the element."""
    env, tmp = gst_env
    n = 3
    fin, fout, fout2 = tmp / "p_in.bgra", tmp / "p_out.nv12", tmp / "p_out.i420"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=640,height=360 ! tee name=t "
                "t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=NV12,width=320,height=180,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=u u. ! queue ! filesink location=%s u. ! queue ! videoconvertscale ! video/x-raw,format=I420,colorimetry=bt709,chroma-site=mpeg2 ! filesink location=%s"
           % (n, fin, fout, fout2))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    nv12 = np.fromfile(fout, np.uint8).reshape(n, -1)
    i420 = np.fromfile(fout2, np.uint8).reshape(n, -1)
    a = ref.VideoConverter("BGRA", 640, 360, "NV12", 320, 180, out_colorimetry="bt709", out_chroma_site="mpeg2",
                           config=cases.ref_config_string(ref, cases.LIN))
    b = ref.VideoConverter("NV12", 320, 180, "I420", 320, 180, in_colorimetry="bt709", in_chroma_site="mpeg2",
                           out_colorimetry="bt709", out_chroma_site="mpeg2")
    for i in range(n):
        assert (a.frame(src[i]) == nv12[i]).all(), i
        assert (b.frame(nv12[i]) == i420[i]).all(), i


def test_videoconvertscale_element_add_borders(gst_env, ref):
    """16:9 into a square frame with add-borders (default): the element letterboxes through dest-x/-y/-width/-height."""
    env, tmp = gst_env
    n, w, h, ow, oh = 2, 640, 360, 400, 400
    fin, fout = tmp / "b_in.nv12", tmp / "b_out.bgra"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,pixel-aspect-ratio=1/1,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=BGRA,width=%d,height=%d,pixel-aspect-ratio=1/1 "
                "! filesink location=%s" % (n, w, h, fin, ow, oh, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    to_h = ow * h // w                       # 225: borders_h = 175, dest_y = 87
    cfg = dict(cases.LIN, dest_x=0, dest_y=(oh - to_h) // 2, dest_width=ow, dest_height=to_h)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", ow, oh, in_colorimetry="bt709", in_chroma_site="mpeg2",
                            config=cases.ref_config_string(ref, cfg))
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), i


def test_compositor_element_scales_and_converts_pads(gst_env, ref):
    """BASELINE C4 variant A in small: pads with width / height properties are scaled, an NV12 pad is converted, each by
    its own converter with the library defaults (what GstVideoAggregatorConvertPad does), then blended."""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    f0, f1, fout = tmp / "s0.bgra", tmp / "s1.nv12", tmp / "s_out.bgra"
    log = launch(dict(env, GSTAMD_ELEMENT_STATS="1"), "compositor name=c background=black sink_0::width=160 sink_0::height=120 sink_0::xpos=10 sink_0::ypos=20 "
                "sink_1::xpos=150 sink_1::ypos=100 sink_1::alpha=0.7 sink_1::width=128 sink_1::height=96 "
                "! video/x-raw,format=BGRA,width=%d,height=%d ! filesink location=%s "
                "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=320,height=240,framerate=30/1 ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=NV12,width=256,height=192,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
           % (dw, dh, fout, n, f0, n, f1))
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
    s1 = np.fromfile(f1, np.uint8).reshape(n, -1)
    c0 = ref.VideoConverter("BGRA", 320, 240, "BGRA", 160, 120)
    c1 = ref.VideoConverter("NV12", 256, 192, "BGRA", 128, 96, in_colorimetry="bt601", in_chroma_site="jpeg")
    for f in range(n):
        canvas = np.zeros(dw * dh * 4, np.uint8)
        ref.compositor_fill(1, "bgra", "BGRA", canvas, dw, dh, 0, dh, 0, 0, 0)
        ref.compositor_blend("blend_bgra", "BGRA", c0.frame(s0[f]), 160, 120, 10, 20, 1.0, canvas, dw, dh, 0, dh, 1)
        ref.compositor_blend("blend_bgra", "BGRA", c1.frame(s1[f]), 128, 96, 150, 100, 0.7, canvas, dw, dh, 0, dh, 1)
        assert (canvas == out[f]).all(), (f, int((canvas != out[f]).sum()))
    # the BGRA pad only changes size: it is sampled inside the blend kernel; the NV12 pad goes through its converter first
    assert "inline-scaled %d" % n in log, log[-500:]


def test_compositor_element_picture_in_picture_of_opaque_pads(gst_env, ref):
    """three I420 / NV12 pads at pad alpha 1.0 piled on a 640-wide BGRA canvas (whole 256-pixel strips of the lower pads hidden) and a translucent one on
    top: the element hands the converted pads to gstamd_compositor_aggregate_opaque as all_opaque, the canvas is blend_pads' over every pad"""
    env, tmp = gst_env
    n, dw, dh = 2, 640, 360
    f0, f1, f2, fout = tmp / "p0.i420", tmp / "p1.nv12", tmp / "p2.i420", tmp / "p_out.bgra"
    launch(env, "compositor name=c background=checker sink_1::xpos=100 sink_1::ypos=40 sink_2::xpos=180 sink_2::ypos=90 sink_2::alpha=0.6 "
                "! video/x-raw,format=BGRA,width=%d,height=%d ! filesink location=%s "
                "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=I420,width=640,height=360,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=NV12,width=480,height=270,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1 "
                "videotestsrc num-buffers=%d pattern=snow ! video/x-raw,format=I420,width=320,height=200,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t2 t2. ! queue ! filesink location=%s t2. ! queue ! c.sink_2"
           % (dw, dh, fout, n, f0, n, f1, n, f2))
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    srcs = [np.fromfile(f, np.uint8).reshape(n, -1) for f in (f0, f1, f2)]
    convs = [ref.VideoConverter("I420", 640, 360, "BGRA", 640, 360, in_colorimetry="bt601", in_chroma_site="jpeg"),
             ref.VideoConverter("NV12", 480, 270, "BGRA", 480, 270, in_colorimetry="bt601", in_chroma_site="jpeg"),
             ref.VideoConverter("I420", 320, 200, "BGRA", 320, 200, in_colorimetry="bt601", in_chroma_site="jpeg")]
    geo = [(640, 360, 0, 0, 1.0), (480, 270, 100, 40, 1.0), (320, 200, 180, 90, 0.6)]
    for f in range(n):
        canvas = np.zeros(dw * dh * 4, np.uint8)
        ref.compositor_fill(0, "bgra", "BGRA", canvas, dw, dh, 0, dh)
        for k, (w, h, x, y, a) in enumerate(geo):
            ref.compositor_blend("blend_bgra", "BGRA", convs[k].frame(srcs[k][f]), w, h, x, y, a, canvas, dw, dh, 0, dh, 1)
        assert (canvas == out[f]).all(), (f, int((canvas != out[f]).sum()))


def test_compositor_element_takes_an_a420_pad(gst_env, ref):
    """a pad in A420 (round 5: I420 plus an alpha plane): its converter brings it to the BGRA canvas with the alpha plane copied - the reference's
    convert_A420_BGRA fastpath at the pad's own size - and the blend uses that alpha"""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    f0, f1, fout = tmp / "a0.bgra", tmp / "a1.a420", tmp / "a_out.bgra"
    launch(env, "compositor name=c background=black sink_1::xpos=40 sink_1::ypos=30 sink_1::alpha=0.8 "
                "! video/x-raw,format=BGRA,width=%d,height=%d ! filesink location=%s "
                "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=320,height=240,framerate=30/1 ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                "videotestsrc num-buffers=%d pattern=ball foreground-color=0x80ffffff ! video/x-raw,format=A420,width=256,height=192,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
           % (dw, dh, fout, n, f0, n, f1))
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
    s1 = np.fromfile(f1, np.uint8).reshape(n, -1)
    c1 = ref.VideoConverter("A420", 256, 192, "BGRA", 256, 192, in_colorimetry="bt601", in_chroma_site="jpeg")
    for f in range(n):
        canvas = np.zeros(dw * dh * 4, np.uint8)
        ref.compositor_fill(1, "bgra", "BGRA", canvas, dw, dh, 0, dh, 0, 0, 0)
        ref.compositor_blend("blend_bgra", "BGRA", s0[f], 320, 240, 0, 0, 1.0, canvas, dw, dh, 0, dh, 1)
        ref.compositor_blend("blend_bgra", "BGRA", c1.frame(s1[f]), 256, 192, 40, 30, 0.8, canvas, dw, dh, 0, dh, 1)
        assert (canvas == out[f]).all(), (f, int((canvas != out[f]).sum()))


def test_compositor_element_takes_y41b_and_av12_pads(gst_env, ref):
    """pads in the last two formats of round 5: Y41B (4:1:1; its converter upsamples the chroma 4 x) and, where the runtime knows it, AV12 (NV12 + an alpha
    plane: the blend uses that alpha)"""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    fmts = ["Y41B"] + (["AV12"] if env.get("GSTAMD_RUNTIME") == "1.29" else [])
    for fmt in fmts:
        f0, f1, fout = tmp / ("q0_%s.bgra" % fmt), tmp / ("q1_%s.raw" % fmt), tmp / ("q_out_%s.bgra" % fmt)
        launch(env, "compositor name=c background=black sink_1::xpos=40 sink_1::ypos=30 sink_1::alpha=0.8 "
                    "! video/x-raw,format=BGRA,width=%d,height=%d ! filesink location=%s "
                    "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=320,height=240,framerate=30/1 ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                    "videotestsrc num-buffers=%d pattern=ball foreground-color=0x80ffffff ! video/x-raw,format=%s,width=256,height=192,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
               % (dw, dh, fout, n, f0, n, fmt, f1))
        out = np.fromfile(fout, np.uint8).reshape(n, -1)
        s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
        s1 = np.fromfile(f1, np.uint8).reshape(n, -1)
        c1 = ref.VideoConverter(fmt, 256, 192, "BGRA", 256, 192, in_colorimetry="bt601", in_chroma_site="jpeg")
        for f in range(n):
            canvas = np.zeros(dw * dh * 4, np.uint8)
            ref.compositor_fill(1, "bgra", "BGRA", canvas, dw, dh, 0, dh, 0, 0, 0)
            ref.compositor_blend("blend_bgra", "BGRA", s0[f], 320, 240, 0, 0, 1.0, canvas, dw, dh, 0, dh, 1)
            ref.compositor_blend("blend_bgra", "BGRA", c1.frame(s1[f]), 256, 192, 40, 30, 0.8, canvas, dw, dh, 0, dh, 1)
            assert (canvas == out[f]).all(), (fmt, f, int((canvas != out[f]).sum()))


def test_compositor_pad_converter_config(gst_env, ref):
    """GstVideoAggregatorConvertPad::converter-config (gstvideoaggregator.c:444-488): the pad's converter takes its options from the
    structure - nearest scaling on one pad, lanczos + alpha-mode=set on a converted NV12 pad - instead of the library defaults"""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    f0, f1, fout = tmp / "cc0.bgra", tmp / "cc1.nv12", tmp / "cc_out.bgra"
    cc0 = "GstVideoConverter,GstVideoConverter.resampler-method=(GstVideoResamplerMethod)nearest"
    cc1 = "GstVideoConverter,GstVideoConverter.resampler-method=(GstVideoResamplerMethod)lanczos,GstVideoConverter.alpha-mode=(GstVideoAlphaMode)set,GstVideoConverter.alpha-value=(double)0.5"
    r = subprocess.run([env["GSTAMD_LAUNCH_BIN"], "-q", "compositor", "name=c", "background=black", "sink_0::width=160", "sink_0::height=120", "sink_0::xpos=10", "sink_0::ypos=20",
                        "sink_0::converter-config=" + cc0, "sink_1::xpos=150", "sink_1::ypos=100", "sink_1::width=128", "sink_1::height=96",
                        "sink_1::converter-config=" + cc1, "!", "video/x-raw,format=BGRA,width=%d,height=%d" % (dw, dh), "!", "filesink", "location=%s" % fout]
                       + ("videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=320,height=240,framerate=30/1 ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                          "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=NV12,width=256,height=192,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
                          % (n, f0, n, f1)).split(), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
    s1 = np.fromfile(f1, np.uint8).reshape(n, -1)
    c0 = ref.VideoConverter("BGRA", 320, 240, "BGRA", 160, 120, config=ref.config_string(GstVideoConverter__resampler_method="nearest"))
    c1 = ref.VideoConverter("NV12", 256, 192, "BGRA", 128, 96, in_colorimetry="bt601", in_chroma_site="jpeg",
                            config=ref.config_string(GstVideoConverter__resampler_method="lanczos", GstVideoConverter__alpha_mode="set",
                                                     GstVideoConverter__alpha_value=0.5))
    for f in range(n):
        canvas = np.zeros(dw * dh * 4, np.uint8)
        ref.compositor_fill(1, "bgra", "BGRA", canvas, dw, dh, 0, dh, 0, 0, 0)
        ref.compositor_blend("blend_bgra", "BGRA", c0.frame(s0[f]), 160, 120, 10, 20, 1.0, canvas, dw, dh, 0, dh, 1)
        ref.compositor_blend("blend_bgra", "BGRA", c1.frame(s1[f]), 128, 96, 150, 100, 1.0, canvas, dw, dh, 0, dh, 1)
        assert (canvas == out[f]).all(), (f, int((canvas != out[f]).sum()))


def test_compositor_element_on_a_64_bit_canvas(gst_env, ref):
    """ARGB64 output (compositor.c:1048-1053: blend_argb64 / overlay_argb64, 16-bit fills): an ARGB64 pad as it is and a BGRA
    pad brought to ARGB64 by the pad's converter, over the checker and over the transparent background."""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    f0, f1 = tmp / "w0.argb64", tmp / "w1.bgra"
    for background, func in (("checker", "blend_argb64"), ("transparent", "overlay_argb64")):
        fout = tmp / ("w_out_%s.argb64" % background)
        launch(env, "compositor name=c background=%s sink_1::xpos=100 sink_1::ypos=80 sink_1::alpha=0.6 sink_0::alpha=0.9 "
                    "! video/x-raw,format=ARGB64,width=%d,height=%d ! filesink location=%s "
                    "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=ARGB64,width=320,height=240,framerate=30/1 ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                    "videotestsrc num-buffers=%d pattern=ball foreground-color=0x80ff4020 ! video/x-raw,format=BGRA,width=160,height=120,framerate=30/1 ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
               % (background, dw, dh, fout, n, f0, n, f1))
        out = np.fromfile(fout, np.uint8).reshape(n, -1)
        s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
        s1 = np.fromfile(f1, np.uint8).reshape(n, -1)
        c1 = ref.VideoConverter("BGRA", 160, 120, "ARGB64", 160, 120)
        for f in range(n):
            canvas = np.zeros(dw * dh * 8, np.uint8)
            if background == "checker":
                ref.compositor_fill(0, "argb64", "ARGB64", canvas, dw, dh, 0, dh)
            ref.compositor_blend(func, "ARGB64", s0[f], 320, 240, 0, 0, 0.9, canvas, dw, dh, 0, dh, 1)
            ref.compositor_blend(func, "ARGB64", c1.frame(s1[f]), 160, 120, 100, 80, 0.6, canvas, dw, dh, 0, dh, 1)
            assert (canvas == out[f]).all(), (background, f, int((canvas != out[f]).sum()))


def test_compositor_element_selects_frames_by_time_and_repeats_a_slower_pad(gst_env, ref):
    """GstVideoAggregator's frame selection (gstvideoaggregator.c:1753-2000): a 15 fps pad under a 30 fps pad - the output runs at
    30 fps (the best framerate) and every frame of the slower pad is shown twice."""
    env, tmp = gst_env
    n, dw, dh = 6, 320, 240
    f0, f1, fout = tmp / "t0.bgra", tmp / "t1.bgra", tmp / "t_out.bgra"
    launch(env, "compositor name=c background=black sink_1::xpos=100 sink_1::ypos=60 sink_1::alpha=0.75 ! video/x-raw,format=BGRA,width=%d,height=%d,framerate=30/1 ! filesink location=%s "
                "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=BGRA,width=320,height=240,framerate=30/1 ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                "videotestsrc num-buffers=%d pattern=ball foreground-color=0xff20c040 ! video/x-raw,format=BGRA,width=160,height=120,framerate=15/1 ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
           % (dw, dh, fout, n, f0, n // 2, f1))
    out = np.fromfile(fout, np.uint8).reshape(-1, dw * dh * 4)
    s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
    s1 = np.fromfile(f1, np.uint8).reshape(n // 2, -1)
    assert out.shape[0] == n
    assert not (s1[0] == s1[1]).all()           # the ball moves: a wrong frame choice shows
    for f in range(n):
        canvas = np.zeros(dw * dh * 4, np.uint8)
        ref.compositor_fill(1, "bgra", "BGRA", canvas, dw, dh, 0, dh, 0, 0, 0)
        ref.compositor_blend("blend_bgra", "BGRA", s0[f], 320, 240, 0, 0, 1.0, canvas, dw, dh, 0, dh, 1)
        ref.compositor_blend("blend_bgra", "BGRA", s1[f // 2], 160, 120, 100, 60, 0.75, canvas, dw, dh, 0, dh, 1)
        assert (canvas == out[f]).all(), (f, int((canvas != out[f]).sum()))


def test_compositor_element_leaves_out_frames_nobody_can_see(gst_env, ref):
    """prepare_frame_start's visibility rules (compositor.c:464-601): alpha 0, off-canvas and fully obscured pads never reach the
    GPU - the picture equals blending all of them, and the element counts what it left out."""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    names = ["v0.bgra", "v1.bgra", "v2.bgrx", "v3.bgra", "v4.bgra"]
    f = [tmp / x for x in names]
    fout = tmp / "v_out.bgra"
    env2 = dict(env, GSTAMD_ELEMENT_STATS="1")
    log = launch(env2,
        "compositor name=c background=checker sink_0::zorder=0 sink_1::zorder=1 sink_1::xpos=20 sink_1::ypos=20 "
        "sink_2::zorder=2 sink_2::xpos=-8 sink_2::ypos=-8 sink_3::zorder=3 sink_3::alpha=0 sink_4::zorder=4 sink_4::xpos=400 sink_4::alpha=0.5 "
        "! video/x-raw,format=BGRA,width=%d,height=%d ! filesink location=%s "
        "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=320,height=240,framerate=30/1 ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
        "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=BGRA,width=160,height=120,framerate=30/1 ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1 "
        "videotestsrc num-buffers=%d pattern=snow ! video/x-raw,format=BGRx,width=336,height=256,framerate=30/1 ! tee name=t2 t2. ! queue ! filesink location=%s t2. ! queue ! c.sink_2 "
        "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=BGRA,width=64,height=64,framerate=30/1 ! tee name=t3 t3. ! queue ! filesink location=%s t3. ! queue ! c.sink_3 "
        "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=BGRA,width=64,height=64,framerate=30/1 ! tee name=t4 t4. ! queue ! filesink location=%s t4. ! queue ! c.sink_4"
        % (dw, dh, fout, n, f[0], n, f[1], n, f[2], n, f[3], n, f[4]))
    assert "culled-frames %d" % (4 * n) in log, log[-500:]
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    src = [np.fromfile(x, np.uint8).reshape(n, -1) for x in f]
    cx = ref.VideoConverter("BGRx", 336, 256, "BGRA", 336, 256)
    geo = [(320, 240, 0, 0, 1.0), (160, 120, 20, 20, 1.0), (336, 256, -8, -8, 1.0), (64, 64, 0, 0, 0.0), (64, 64, 400, 0, 0.5)]
    for k in range(n):
        canvas = np.zeros(dw * dh * 4, np.uint8)
        ref.compositor_fill(0, "bgra", "BGRA", canvas, dw, dh, 0, dh)
        for i, (w, h, x, y, a) in enumerate(geo):
            frame = cx.frame(src[i][k]) if i == 2 else src[i][k]
            ref.compositor_blend("blend_bgra", "BGRA", frame, w, h, x, y, a, canvas, dw, dh, 0, dh, 1)
        assert (canvas == out[k]).all(), (k, int((canvas != out[k]).sum()))


def test_compositor_element_sizing_policy_keeps_the_aspect_ratio(gst_env, ref):
    """sizing-policy=keep-aspect-ratio (_mixer_pad_get_output_size, compositor.c:290-412): a 4:3 picture asked into 200 x 200 is
    scaled to 200 x 150 and centred - 25 lines down from ypos."""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    f0, fout = tmp / "k0.bgra", tmp / "k_out.bgra"
    launch(env, "compositor name=c background=black sink_0::width=200 sink_0::height=200 sink_0::xpos=30 sink_0::ypos=10 sink_0::sizing-policy=keep-aspect-ratio "
                "! video/x-raw,format=BGRA,width=%d,height=%d ! filesink location=%s "
                "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=320,height=240,framerate=30/1 ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0"
           % (dw, dh, fout, n, f0))
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
    c0 = ref.VideoConverter("BGRA", 320, 240, "BGRA", 200, 150)
    for k in range(n):
        canvas = np.zeros(dw * dh * 4, np.uint8)
        ref.compositor_fill(1, "bgra", "BGRA", canvas, dw, dh, 0, dh, 0, 0, 0)
        ref.compositor_blend("blend_bgra", "BGRA", c0.frame(s0[k]), 200, 150, 30, 35, 1.0, canvas, dw, dh, 0, dh, 1)
        assert (canvas == out[k]).all(), (k, int((canvas != out[k]).sum()))


def test_audioresample_element_interpolated_filter(gst_env, ref):
    """sinc-filter-mode=interpolated sinc-filter-interpolation=linear through the element, S16."""
    env, tmp = gst_env
    fin, fout = tmp / "in.s16", tmp / "out.s16"
    launch(env, "audiotestsrc num-buffers=20 wave=white-noise samplesperbuffer=1024 ! audio/x-raw,format=S16LE,rate=48000,channels=2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdaudioresample quality=4 sinc-filter-mode=interpolated "
                "sinc-filter-interpolation=linear ! audio/x-raw,rate=44100 ! filesink location=%s" % (fin, fout))
    src = np.fromfile(fin, np.int16).reshape(-1, 2)
    out = np.fromfile(fout, np.int16).reshape(-1, 2)
    rr = ref.AudioResampler("S16LE", 2, 48000, 44100, quality=4, filter_mode="interpolated", filter_interpolation="linear")
    exp = []
    for off in range(0, len(src), 1024):
        blk = src[off:off + 1024]
        exp.append(rr.resample(blk, in_frames=len(blk), out_frames=rr.get_out_frames(len(blk))))
    lat = rr.get_max_latency()
    exp.append(rr.resample(None, in_frames=lat, out_frames=rr.get_out_frames(lat)))
    exp = np.concatenate(exp)
    assert out.shape == exp.shape and (out == exp).all()


def test_videoconvertscale_element_packed422_and_rgb24(gst_env, ref):
    """Capture-style formats: YUY2 720p -> RGB 24-bit 360p (bilinear, the element default) and UYVY -> NV12 same size."""
    env, tmp = gst_env
    w, h, n = 1280, 720, 3
    fin, frgb = tmp / "in.yuy2", tmp / "out.rgb"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=YUY2,width=%d,height=%d,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=RGB,width=640,height=360 "
                "! filesink location=%s" % (n, w, h, fin, frgb))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(frgb, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("YUY2", w, h, "RGB", 640, 360, in_colorimetry="bt709", in_chroma_site="mpeg2",
                            config=cases.ref_config_string(ref, cases.LIN))
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), i
    fin2, fnv = tmp / "in.uyvy", tmp / "out.nv12"
    launch(env, "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=UYVY,width=640,height=360,colorimetry=bt601,chroma-site=jpeg "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale "
                "! video/x-raw,format=NV12,colorimetry=bt601,chroma-site=jpeg ! filesink location=%s" % (n, fin2, fnv))
    src = np.fromfile(fin2, np.uint8).reshape(n, -1)
    out = np.fromfile(fnv, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("UYVY", 640, 360, "NV12", 640, 360, in_colorimetry="bt601", in_chroma_site="jpeg", out_colorimetry="bt601",
                            out_chroma_site="jpeg")
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), i


def test_audioresample_element_non_interleaved_layout(gst_env, ref):
    """layout=non-interleaved: every buffer is [channels][frames]; a caps filter labels 1024-frame blocks of a planar file
    (rawaudioparse 1.14 ignores its interleaved property)."""
    env, tmp = gst_env
    ch, blk, nblk = 2, 1024, 24
    src = (cases.audio_buffer("F32LE", ch, blk * nblk, 777)).reshape(nblk, blk, ch).transpose(0, 2, 1).copy()      # [block][channel][frame]
    fin, fout = tmp / "in_planar.f32", tmp / "out_planar.f32"
    src.tofile(fin)
    launch(env, "filesrc location=%s blocksize=%d ! audio/x-raw,format=F32LE,layout=non-interleaved,rate=48000,channels=%d "
                "! amdaudioresample quality=4 ! audio/x-raw,rate=44100 ! filesink location=%s" % (fin, blk * ch * 4, ch, fout))
    out = np.fromfile(fout, np.float32)
    rr = ref.AudioResampler("F32LE", ch, 48000, 44100, quality=4, in_planar=True, out_planar=True)
    exp = [rr.resample(src[i], in_frames=blk, out_frames=rr.get_out_frames(blk)).reshape(-1) for i in range(nblk)]
    lat = rr.get_max_latency()
    exp.append(rr.resample(None, in_frames=lat, out_frames=rr.get_out_frames(lat)).reshape(-1))       # EOS drain
    exp = np.concatenate(exp)
    assert out.shape == exp.shape, (out.shape, exp.shape)
    assert (out == exp).all(), int(np.argmax(out != exp))


def test_compositor_element_i420_and_nv12_output(gst_env, ref):
    """Outputs without per-pixel alpha are aggregated plane by plane: an I420 canvas from an I420 pad (odd position: rounded up to
    even) and a BGRA pad converted to I420 by its pad converter; then an NV12 canvas over the checker background."""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    for ofmt, bg, bgkind in (("I420", "black", 1), ("NV12", "checker", 0)):
        f0, f1, fout = tmp / ("p0_%s.yuv" % ofmt), tmp / ("p1_%s.bgra" % ofmt), tmp / ("pout_%s.yuv" % ofmt)
        launch(env, "compositor name=c background=%s sink_0::xpos=11 sink_0::ypos=21 sink_0::alpha=0.6 sink_1::xpos=150 sink_1::ypos=100 sink_1::alpha=0.7 "
                    "! video/x-raw,format=%s,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg ! filesink location=%s "
                    "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=%s,width=160,height=120,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                    "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=BGRA,width=128,height=96,framerate=30/1 ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
               % (bg, ofmt, dw, dh, fout, n, ofmt, f0, n, f1))
        out = np.fromfile(fout, np.uint8).reshape(n, -1)
        s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
        s1 = np.fromfile(f1, np.uint8).reshape(n, -1)
        c1 = ref.VideoConverter("BGRA", 128, 96, ofmt, 128, 96, out_colorimetry="bt601", out_chroma_site="jpeg")
        low = ofmt.lower()
        for f in range(n):
            canvas = np.zeros(out.shape[1], np.uint8)
            if bgkind == 0:
                ref.compositor_fill(0, low, ofmt, canvas, dw, dh, 0, dh)
            else:
                ref.compositor_fill(1, low, ofmt, canvas, dw, dh, 0, dh, 16, 128, 128)
            ref.compositor_blend("blend_" + low, ofmt, s0[f], 160, 120, 11, 21, 0.6, canvas, dw, dh, 0, dh, 1)
            ref.compositor_blend("blend_" + low, ofmt, c1.frame(s1[f]), 128, 96, 150, 100, 0.7, canvas, dw, dh, 0, dh, 1)
            assert (canvas == out[f]).all(), (ofmt, f, int((canvas != out[f]).sum()))


def test_compositor_element_10_and_12_bit_planar_canvases(gst_env, ref):
    """Planar canvases of more than 8 bits (blend.c:609-697): an I420_10LE canvas over black from an I420_10LE pad and a BGRA pad (its pad
    converter goes through k_encode16), an I422_12LE canvas over the checker; the reference's blend_i420_10le / blend_i422_12le and
    fill functions pad by pad, black = the limited range's offsets at the format's depth (compositor.c:1131-1149)."""
    env, tmp = gst_env
    n, dw, dh = 2, 320, 240
    for ofmt, bits, bg, bgkind in (("I420_10LE", 10, "black", 1), ("I422_12LE", 12, "checker", 0)):
        f0, f1, fout = tmp / ("q0_%s.yuv" % ofmt), tmp / ("q1_%s.bgra" % ofmt), tmp / ("qout_%s.yuv" % ofmt)
        launch(env, "compositor name=c background=%s sink_0::xpos=11 sink_0::ypos=21 sink_0::alpha=0.6 sink_1::xpos=150 sink_1::ypos=100 sink_1::alpha=0.7 "
                    "! video/x-raw,format=%s,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg ! filesink location=%s "
                    "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=%s,width=160,height=120,framerate=30/1,colorimetry=bt601,chroma-site=jpeg ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                    "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=BGRA,width=128,height=96,framerate=30/1 ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
               % (bg, ofmt, dw, dh, fout, n, ofmt, f0, n, f1))
        out = np.fromfile(fout, np.uint8).reshape(n, -1)
        s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
        s1 = np.fromfile(f1, np.uint8).reshape(n, -1)
        c1 = ref.VideoConverter("BGRA", 128, 96, ofmt, 128, 96, out_colorimetry="bt601", out_chroma_site="jpeg")
        low = ofmt.lower()
        for f in range(n):
            canvas = np.zeros(out.shape[1], np.uint8)
            if bgkind == 0:
                ref.compositor_fill(0, low, ofmt, canvas, dw, dh, 0, dh)
            else:
                ref.compositor_fill(1, low, ofmt, canvas, dw, dh, 0, dh, 16 << (bits - 8), 128 << (bits - 8), 128 << (bits - 8))
            ref.compositor_blend("blend_" + low, ofmt, s0[f], 160, 120, 11, 21, 0.6, canvas, dw, dh, 0, dh, 1)
            ref.compositor_blend("blend_" + low, ofmt, c1.frame(s1[f]), 128, 96, 150, 100, 0.7, canvas, dw, dh, 0, dh, 1)
            assert (canvas == out[f]).all(), (ofmt, f, int((canvas != out[f]).sum()))


def test_compositor_element_packed_canvases_without_alpha(gst_env, ref):
    """xRGB / BGRx (RGB_BLEND with four bytes per pixel) and YUY2 / UYVY (PACKED_422_BLEND) canvases: a pad of the canvas format at an odd
    xpos (rounded up to an even pixel on 4:2:2) and a BGRA pad through its converter, over white (MEMSET_XRGB's byte order for xRGB), black
    and the 4:2:2 checker - whole output buffers against the reference's functions"""
    env, tmp = gst_env
    n, dw, dh = 2, 322, 240
    for ofmt, bg, bgkind in (("xRGB", "white", 2), ("BGRx", "black", 1), ("YUY2", "checker", 0), ("UYVY", "white", 2)):
        yuv = ofmt in ("YUY2", "UYVY")
        extra = ",colorimetry=bt601,chroma-site=jpeg" if yuv else ""
        f0, f1, fout = tmp / ("k0_%s.raw" % ofmt), tmp / ("k1_%s.bgra" % ofmt), tmp / ("kout_%s.raw" % ofmt)
        launch(env, "compositor name=c background=%s sink_0::xpos=11 sink_0::ypos=21 sink_0::alpha=0.6 sink_1::xpos=150 sink_1::ypos=100 sink_1::alpha=0.7 "
                    "! video/x-raw,format=%s,width=%d,height=%d%s ! filesink location=%s "
                    "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=%s,width=160,height=120,framerate=30/1%s ! tee name=t0 t0. ! queue ! filesink location=%s t0. ! queue ! c.sink_0 "
                    "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=BGRA,width=128,height=96,framerate=30/1 ! tee name=t1 t1. ! queue ! filesink location=%s t1. ! queue ! c.sink_1"
               % (bg, ofmt, dw, dh, extra, fout, n, ofmt, extra, f0, n, f1))
        out = np.fromfile(fout, np.uint8).reshape(n, -1)
        s0 = np.fromfile(f0, np.uint8).reshape(n, -1)
        s1 = np.fromfile(f1, np.uint8).reshape(n, -1)
        c1 = ref.VideoConverter("BGRA", 128, 96, ofmt, 128, 96, **(dict(out_colorimetry="bt601", out_chroma_site="jpeg") if yuv else {}))
        low = ofmt.lower()
        func = "blend_yuy2" if yuv else "blend_xrgb"
        for f in range(n):
            canvas = np.zeros(out.shape[1], np.uint8)
            if bgkind == 0:
                ref.compositor_fill(0, low, ofmt, canvas, dw, dh, 0, dh)
            else:
                c = ((16, 128, 128) if bgkind == 1 else (235, 128, 128)) if yuv else ((0, 0, 0) if bgkind == 1 else (255, 255, 255))
                ref.compositor_fill(1, low, ofmt, canvas, dw, dh, 0, dh, *c)
            ref.compositor_blend(func, ofmt, s0[f], 160, 120, 11, 21, 0.6, canvas, dw, dh, 0, dh, 1)
            ref.compositor_blend(func, ofmt, c1.frame(s1[f]), 128, 96, 150, 100, 0.7, canvas, dw, dh, 0, dh, 1)
            assert (canvas == out[f]).all(), (ofmt, f, int((canvas != out[f]).sum()))


def _bench_element(env, args, keep_stderr=False):
    only_on_114(env, "plugins/tests/bench_element drives the element through GstHarness (libgstcheck)")
    exe = os.path.join(ROOT, "plugins", "tests", "bench_element")
    assert os.path.exists(exe), "plugins/build.py builds it"
    r = subprocess.run([exe] + [str(a) for a in args], env=dict(env, GSTAMD_ELEMENT_STATS="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    print(r.stderr.strip()[-600:])
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return dict(json.loads(line), stderr=r.stderr) if keep_stderr else json.loads(line)


def _list_stats(res):
    import re
    m = re.search(r"buffer lists: (\d+) converter calls, (\d+) list launches", res["stderr"])
    assert m, res["stderr"][-600:]
    return int(m.group(1)), int(m.group(2))


@pytest.mark.parametrize("list_n", [4, 32])
def test_element_buffer_lists_take_one_launch(gst_env, list_n):
    """GstBufferLists through the element's chain_list path (one converter call per list): frames/s at 4K and 1080p, and the
    element's own count of converter calls against the launches that each served a whole list (one per call)."""
    env, tmp = gst_env
    out = []
    for (w, h, n) in ((3840, 2160, 640), (1920, 1080, 1600)):
        res = _bench_element(env, ["NV12", w, h, "BGRA", w, h, n, 3, "bilinear", list_n], keep_stderr=True)
        calls, launches = _list_stats(res)
        assert calls > 0 and launches == calls, (calls, launches)
        res.pop("stderr")
        out.append(res)
        print(res)
    with open(os.path.join(ROOT, "gpurun_out", "element_bench.jsonl"), "a") as f:
        for res in out:
            f.write(__import__("json").dumps(res) + "\n")
    assert out[0]["frames_per_s"] > 40000, out


LIST_ELEMENT_PLANS = [      # single-kernel plans whose kernels take the list as the grid's third dimension (video_kernels.hip: frame lists)
    ("plane_scaler", "NV12", 1920, 1080, "NV12", 1280, 720, 1),          # (sizes of one default-colorimetry class: no matrix between them)
    ("p010_out", "NV12", 3840, 2160, "P010_10LE", 3840, 2160, 1),
    ("p010_in", "P010_10LE", 3840, 2160, "NV12", 3840, 2160, 1),
    ("planar_pack", "YUY2", 3840, 2160, "I420", 3840, 2160, 1),
    ("swizzle", "BGRA", 3840, 2160, "RGBA", 3840, 2160, 1),
    ("column_scaler", "I420", 3840, 2160, "RGBA", 1920, 1080, 1),
]


@pytest.mark.parametrize("plan", LIST_ELEMENT_PLANS, ids=lambda p: p[0])
def test_element_buffer_lists_of_single_kernel_plans_take_one_launch(gst_env, plan):
    """the plane scaler, the P010 plans, the planar packer, the swizzle and the column scaler behind chain_list: every converter call
    of a list is served by launches that take the whole list (the plane scaler: one per kernel form of its planes)"""
    env, tmp = gst_env
    _, ifmt, w, h, ofmt, ow, oh, per_call = plan
    res = _bench_element(env, [ifmt, w, h, ofmt, ow, oh, 256, 1, "lanczos" if plan[0] == "column_scaler" else "bilinear", 8], keep_stderr=True)
    calls, launches = _list_stats(res)
    res.pop("stderr")
    print(res)
    assert calls >= 256 // 8 and launches >= calls * per_call and launches <= calls * 3, (calls, launches)
    with open(os.path.join(ROOT, "gpurun_out", "element_bench.jsonl"), "a") as f:
        f.write(__import__("json").dumps(dict(res, plan=plan[0])) + "\n")


def test_buffer_list_output_equals_single_buffers(gst_env, ref):
    """A pipeline whose source hands over buffer lists... the harness tool is the only list producer here, so parity of the
    chain_list path is pinned at the library level (tests/test_video_gpu.py: frames == n x frame) and by this run's checksum:
    the list path and the per-buffer path of the element must produce the same last frame."""
    env, tmp = gst_env
    a = _bench_element(dict(env, GSTAMD_BENCH_SUM="1"), ["NV12", 1280, 720, "BGRA", 1280, 720, 64, 3, "bilinear", 1])
    b = _bench_element(dict(env, GSTAMD_BENCH_SUM="1"), ["NV12", 1280, 720, "BGRA", 1280, 720, 64, 3, "bilinear", 8])
    assert a["last_frame_sum"] == b["last_frame_sum"] and a["last_frame_sum"] != 0


@pytest.mark.parametrize("streams", [1, 3])
def test_element_throughput_hbm_resident(gst_env, streams):
    """Frames/s THROUGH the videoconvertscale element (GstHarness, buffers stay in HBM, one kernel launch per buffer as
    gst_base_transform_chain delivers them): BASELINE C2 at 4K and the C1 size.  The numbers go to gpurun_out/element_bench.jsonl;
    the assertion is only that the element path is not pathologically slower than the library's per-frame launch."""
    env, tmp = gst_env
    out = []
    for (w, h, n) in ((3840, 2160, 600), (1920, 1080, 1500)):
        res = _bench_element(env, ["NV12", w, h, "BGRA", w, h, n, streams])
        out.append(res)
        print(res)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "element_bench.jsonl"), "a") as f:
        for res in out:
            f.write(__import__("json").dumps(res) + "\n")
    assert out[0]["frames_per_s"] > 30000 and out[1]["frames_per_s"] > 45000, out


def test_upload_download_elements_bracket_an_hbm_pipeline(gst_env, ref):
    """amdhipupload ! videoconvertscale (HBM in, HBM out) ! videoconvertscale ! amdhipdownload: the frames enter and leave HBM at
    the edges only; bytes equal the reference's two conversions."""
    env, tmp = gst_env
    w, h, n = 640, 360, 3
    fin, fout = tmp / "ud_in.nv12", tmp / "ud_out.rgba"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdhipupload ! video/x-raw(memory:AMDHIPMemory),format=NV12 "
                "! videoconvertscale ! video/x-raw(memory:AMDHIPMemory),format=BGRA ! videoconvertscale ! video/x-raw(memory:AMDHIPMemory),format=RGBA "
                "! amdhipdownload ! video/x-raw,format=RGBA ! filesink location=%s" % (n, w, h, fin, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    a = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt709", in_chroma_site="mpeg2")
    b = ref.VideoConverter("BGRA", w, h, "RGBA", w, h)
    for i in range(n):
        assert (b.frame(a.frame(src[i])) == out[i]).all(), i


@pytest.mark.parametrize("b1,b2,queue", [(4, 3, True), (4, 4, False), (8, 1, True), (0, 0, True)])
def test_deferred_launches_give_the_same_frames(gst_env, ref, b1, b2, queue):
    """batch-buffers: HBM -> HBM frames of consecutive buffers filed and launched together (one gstamd_video_converter_frames call), the
    launch forced by whoever needs a frame first (the next element's stream, the download's map), by EOS (11 frames: the last batch is
    short) or by the 2 ms watcher.  Two batching converters in a row, with and without a thread boundary; 0 = automatic (videotestsrc is
    not live: 4).  Bytes equal the reference's two conversions."""
    env, tmp = gst_env
    w, h, n = 640, 360, 11
    fin, fout = tmp / "bb_in.nv12", tmp / "bb_out.rgba"
    log = launch(dict(env, GSTAMD_ELEMENT_STATS="1"),
                 "videotestsrc num-buffers=%d pattern=ball ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt709,chroma-site=mpeg2 "
                 "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdhipupload ! video/x-raw(memory:AMDHIPMemory),format=NV12 "
                 "! videoconvertscale batch-buffers=%d ! video/x-raw(memory:AMDHIPMemory),format=BGRA %s! videoconvertscale batch-buffers=%d "
                 "! video/x-raw(memory:AMDHIPMemory),format=RGBA ! amdhipdownload ! video/x-raw,format=RGBA ! filesink location=%s"
                 % (n, w, h, fin, b1, "! queue " if queue else "", b2, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    a = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt709", in_chroma_site="mpeg2")
    b = ref.VideoConverter("BGRA", w, h, "RGBA", w, h)
    for i in range(n):
        assert (b.frame(a.frame(src[i])) == out[i]).all(), i
    assert "deferred launches: %d frames" % n in log, log[-800:]


def test_deferred_launches_through_the_harness(gst_env):
    """The element alone on HBM buffers (plugins/tests/bench_element): the last output frame of a per-buffer run with batch-buffers=4
    equals the one without, and the per-buffer rates with deferred launches go to gpurun_out/element_bench.jsonl."""
    env, tmp = gst_env
    a = _bench_element(dict(env, GSTAMD_BENCH_SUM="1"), ["NV12", 1280, 720, "BGRA", 1280, 720, 66, 1, "bilinear", 1, 1])
    b = _bench_element(dict(env, GSTAMD_BENCH_SUM="1"), ["NV12", 1280, 720, "BGRA", 1280, 720, 66, 1, "bilinear", 1, 4])
    assert a["last_frame_sum"] == b["last_frame_sum"] and a["last_frame_sum"] != 0
    out = []
    for (w, h, n) in ((3840, 2160, 600), (1920, 1080, 1500)):
        for streams, batch in ((1, 4), (2, 4), (1, 8)):
            res = _bench_element(env, ["NV12", w, h, "BGRA", w, h, n, streams, "bilinear", 1, batch])
            out.append(res)
            print(res)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "element_bench.jsonl"), "a") as f:
        for res in out:
            f.write(__import__("json").dumps(res) + "\n")
    assert max(r["frames_per_s"] for r in out[:3]) > 90000, out


def test_reference_factory_names_are_ours_in_a_registry_without_the_stock_elements(gst_env, ref):
    """north_star: "register under the same factory names".  A registry keeps one feature per name, so the drop-in deployment is a
    plugin directory that ships this plugin INSTEAD of gst-plugins-base's videoconvert / videoscale / audioresample plugins: there
    `videoconvert`, `videoscale` and `audioresample` resolve to the MI355X elements and run unchanged pipelines."""
    env, tmp = gst_env
    only_on_114(env, "no registry to resolve factory names in")
    pdir = tmp / "dropin_plugins"
    pdir.mkdir()
    stock = "/opt/conda/lib/gstreamer-1.0"
    for f in os.listdir(stock):
        if f.endswith(".so") and f not in ("libgstvideoconvert.so", "libgstvideoscale.so", "libgstaudioresample.so"):
            os.symlink(os.path.join(stock, f), pdir / f)
    os.symlink(os.path.join(ROOT, "plugins", "libgstamdhipdsp.so"), pdir / "libgstamdhipdsp.so")
    env2 = dict(env, GST_PLUGIN_PATH=str(pdir), GST_REGISTRY=str(tmp / "registry_names.bin"))
    insp = os.path.join(os.path.dirname(GST), "gst-inspect-1.0")
    for name in ("videoconvert", "videoscale", "audioresample", "videoconvertscale", "compositor"):
        r = subprocess.run([insp, name], env=env2, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert r.returncode == 0 and "MI355X" in r.stdout, (name, r.stdout[:600])
    w, h, n = 320, 240, 2
    fin, fout, fsc = tmp / "nm_in.nv12", tmp / "nm_out.bgra", tmp / "nm_sc.nv12"
    launch(env2, "videotestsrc num-buffers=%d ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt601,chroma-site=jpeg ! tee name=t "
                 "t. ! queue ! filesink location=%s t. ! queue ! videoconvert ! video/x-raw,format=BGRA ! filesink location=%s "
                 "t. ! queue ! videoscale ! video/x-raw,width=160,height=120 ! filesink location=%s" % (n, w, h, fin, fout, fsc))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    sc = np.fromfile(fsc, np.uint8).reshape(n, -1)
    a = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt601", in_chroma_site="jpeg")
    b = ref.VideoConverter("NV12", w, h, "NV12", 160, 120, in_colorimetry="bt601", in_chroma_site="jpeg", out_colorimetry="bt601",
                           out_chroma_site="jpeg", config=cases.ref_config_string(ref, cases.LIN))
    for i in range(n):
        assert (a.frame(src[i]) == out[i]).all(), i
        assert (b.frame(src[i]) == sc[i]).all(), i


# the table of tests/check/elements/videoscale.c:601-648 (test_negotiation): input caps, downstream restriction, expected width, height, PAR
NEGOTIATION = [
    ("width=720,height=576,pixel-aspect-ratio=16/15", "width=768,height=576", 768, 576, "1/1"),
    ("width=320,height=240", "width=640,height=320", 640, 320, "2/3"),
    ("width=320,height=240", "width=640,height=320,pixel-aspect-ratio=[0/1,1/1]", 640, 320, "2/3"),
    ("width=1920,height=2560,pixel-aspect-ratio=1/1", "width=[1,2048],height=[1,2048],pixel-aspect-ratio=1/1", 1536, 2048, "1/1"),
    ("width=1920,height=2560,pixel-aspect-ratio=1/1", "width=[1,2048],height=[1,2048]", 1920, 2048, "4/5"),
    ("width=1920,height=2560", "width=[1,2048],height=[1,2048]", 1920, 2048, "4/5"),
    ("width=1920,height=2560", "width=1200,height=[1,2048],pixel-aspect-ratio=1/1", 1200, 1600, "1/1"),
    ("width=320,height=240,pixel-aspect-ratio=1/1", "width=200,height=200,pixel-aspect-ratio=1/2", 200, 200, "1/2"),
    ("width=854,height=480", "width=[2,512,2],height=[2,512,2],pixel-aspect-ratio=1/1", 512, 288, "1/1"),
]


@pytest.mark.parametrize("case", NEGOTIATION, ids=lambda c: "%s->%s" % (c[0], c[1]))
def test_fixate_caps_keeps_the_display_aspect_ratio_like_the_reference(gst_env, case):
    """The reference's own negotiation table for videoscale (tests/check/elements/videoscale.c:601-648): what size and pixel aspect
    ratio the element settles on for a given input and downstream restriction (gst_video_convert_scale_fixate_size)."""
    import re
    env, tmp = gst_env
    in_caps, out_caps, w, h, par = case
    r = subprocess.run([env["GSTAMD_LAUNCH_BIN"], "-v", "videotestsrc", "num-buffers=1", "!", "video/x-raw,format=AYUV,framerate=30/1," + in_caps, "!",
                        "videoconvertscale", "!", "video/x-raw,format=AYUV," + out_caps, "!", "fakesink"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [l for l in r.stdout.splitlines() if "fakesink0.GstPad:sink: caps" in l]
    assert lines, r.stdout[-2000:]
    caps = lines[-1]
    assert "width=(int)%d" % w in caps and "height=(int)%d" % h in caps, caps
    m = re.search(r"pixel-aspect-ratio=\(fraction\)(\d+/\d+)", caps)
    assert (m.group(1) if m else "1/1") == par, caps


def test_p010_decoder_output_to_bgra_through_the_element(gst_env, ref):
    """A 10-bit source through the element: P010_10LE frames (videotestsrc's, converted by the stock videoconvert) ->
    videoconvertscale -> BGRA equals the reference's 16-bit chain (unpack to AYUV64, u16 chroma upsampling, matrix16, narrow)."""
    env, tmp = gst_env
    w, h, n = 320, 240, 2
    fin, fout = tmp / "p010_in.raw", tmp / "p010_out.bgra"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=P010_10LE,width=%d,height=%d,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=BGRA ! filesink location=%s"
           % (n, w, h, fin, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("P010_10LE", w, h, "BGRA", w, h, in_colorimetry="bt709", in_chroma_site="mpeg2")
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), i


def _ref_convert_blocks(rc, src_bytes, in_bpf, blk):
    out = []
    for off in range(0, src_bytes.size, blk * in_bpf):
        out.append(rc.samples(src_bytes[off:off + blk * in_bpf]))
    return np.concatenate(out)


def test_audioconvert_element_matches_reference(gst_env, ref):
    """`audioconvert` (plugins/gstamdaudioconvert.c): F32 -> S16 with the element's default triangular dither, then S16 stereo -> S32
    mono, against the reference's converter fed buffer by buffer (the dither generator runs on across buffers)."""
    env, tmp = gst_env
    fin, fout, fmono = tmp / "ac_in.f32", tmp / "ac_out.s16", tmp / "ac_mono.s32"
    launch(env, "audiotestsrc num-buffers=12 wave=white-noise volume=1.0 samplesperbuffer=1024 ! audio/x-raw,format=F32LE,rate=48000,channels=2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdaudioconvert ! audio/x-raw,format=S16LE ! filesink location=%s" % (fin, fout))
    src = np.fromfile(fin, np.uint8)
    out = np.fromfile(fout, np.uint8)
    rc = ref.AudioConverter("F32LE", 48000, 2, "S16LE", 48000, 2, config="GstAudioConverter, GstAudioConverter.dither-method=(GstAudioDitherMethod)tpdf")
    exp = _ref_convert_blocks(rc, src, 8, 1024)
    rc.free()
    assert out.shape == exp.shape and (out == exp).all(), int((out != exp).sum())
    launch(env, "filesrc location=%s blocksize=4096 ! audio/x-raw,format=S16LE,rate=48000,channels=2,layout=interleaved ! amdaudioconvert dithering=none "
                "! audio/x-raw,format=S32LE,channels=1 ! filesink location=%s" % (fout, fmono))
    mono = np.fromfile(fmono, np.uint8)
    rc = ref.AudioConverter("S16LE", 48000, 2, "S32LE", 48000, 1)
    exp = _ref_convert_blocks(rc, out, 4, 1024)
    rc.free()
    assert mono.shape == exp.shape and (mono == exp).all()


def test_audioconvert_element_noise_shaping_downmix_and_resample_chain(gst_env, ref):
    """5.1 (channel-mask 0x3f) F32 -> stereo S16 with noise shaping through the position-based down-mix rules, followed by audioresample."""
    env, tmp = gst_env
    fin, fout = tmp / "ac51_in.f32", tmp / "ac51_out.s16"
    launch(env, "audiotestsrc num-buffers=10 wave=pink-noise samplesperbuffer=1024 ! audio/x-raw,format=F32LE,rate=48000,channels=6,channel-mask=(bitmask)0x3f "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdaudioconvert dithering=rpdf noise-shaping=medium "
                "! audio/x-raw,format=S16LE,channels=2 ! amdaudioresample quality=4 ! audio/x-raw,rate=44100 ! filesink location=%s" % (fin, fout))
    src = np.fromfile(fin, np.uint8)
    out = np.fromfile(fout, np.int16).reshape(-1, 2)
    pos = [0, 1, 2, 3, 4, 5]
    rc = ref.AudioConverter("F32LE", 48000, 6, "S16LE", 48000, 2, in_pos=pos,
                            config="GstAudioConverter, GstAudioConverter.dither-method=(GstAudioDitherMethod)rpdf, "
                                   "GstAudioConverter.noise-shaping-method=(GstAudioNoiseShapingMethod)medium")
    mid = _ref_convert_blocks(rc, src, 24, 1024).view(np.int16).reshape(-1, 2)
    rc.free()
    rr = ref.AudioResampler("S16LE", 2, 48000, 44100, quality=4)
    exp = []
    for off in range(0, len(mid), 1024):
        blk = mid[off:off + 1024]
        exp.append(rr.resample(blk, in_frames=len(blk), out_frames=rr.get_out_frames(len(blk))))
    lat = rr.get_max_latency()
    exp.append(rr.resample(None, in_frames=lat, out_frames=rr.get_out_frames(lat)))
    exp = np.concatenate(exp)
    assert out.shape == exp.shape and (out == exp).all()


def test_upload_honours_the_video_meta_of_the_source(gst_env):
    """amdhipupload with a system-memory frame whose GstVideoMeta has padded strides and a gap between the planes
    (plugins/tests/live_props.c): plane-by-plane pitched copies into the pool frame's default layout"""
    env, tmp = gst_env
    only_on_114(env, "plugins/tests/live_props drives the element through GstHarness (libgstcheck)")
    exe = os.path.join(ROOT, "plugins", "tests", "live_props")
    r = subprocess.run([exe, "upload-meta"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_audioconvert_mix_matrix_changed_while_running(gst_env):
    """`mix-matrix` set between two buffers with unchanged caps (plugins/tests/live_props.c): the converter is re-made by the streaming
    thread at the next transform, as gst_audio_convert_ensure_converter does (gstaudioconvert.c:1700), and the element leaves passthrough"""
    env, tmp = gst_env
    only_on_114(env, "plugins/tests/live_props drives the element through GstHarness (libgstcheck)")
    exe = os.path.join(ROOT, "plugins", "tests", "live_props")
    assert os.path.exists(exe), "plugins/build.py builds it"
    r = subprocess.run([exe, "mix-matrix"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_audioconvert_element_mix_matrix_property(gst_env, ref):
    env, tmp = gst_env
    fin, fout = tmp / "acm_in.s16", tmp / "acm_out.s16"
    r = subprocess.run([env["GSTAMD_LAUNCH_BIN"], "-q"] + ("audiotestsrc num-buffers=6 wave=white-noise samplesperbuffer=1024 ! audio/x-raw,format=S16LE,rate=44100,channels=2 "
                                      "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdaudioconvert" % fin).split() +
                       ["mix-matrix=<<(float)0.25,(float)0.75>,<(float)1.0,(float)0.0>,<(float)0.0,(float)-1.0>>", "!"] +
                       ("audio/x-raw,format=S16LE,channels=3,channel-mask=(bitmask)0x0 ! filesink location=%s" % fout).split(),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    src = np.fromfile(fin, np.uint8)
    out = np.fromfile(fout, np.uint8)
    rc = ref.AudioConverter("S16LE", 44100, 2, "S16LE", 44100, 3, mix=[[0.25, 0.75], [1.0, 0.0], [0.0, -1.0]])
    exp = _ref_convert_blocks(rc, src, 4, 1024)
    rc.free()
    assert out.shape == exp.shape and (out == exp).all()


def test_videoconvertscale_gamma_and_primaries_modes(gst_env, ref):
    """gamma-mode=remap with a downscale, and primaries-mode=fast into bt2020 primaries, through the element (the colorimetry of the
    caps reaches the converter: transfer function and primaries included)."""
    env, tmp = gst_env
    w, h, n = 640, 360, 2
    fin, fout, fprim = tmp / "gm_in.nv12", tmp / "gm_out.bgra", tmp / "gm_prim.bgra"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale gamma-mode=remap method=lanczos "
                "! video/x-raw,format=BGRA,width=320,height=180,colorimetry=sRGB ! filesink location=%s" % (n, w, h, fin, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", 320, 180, in_colorimetry="bt709", in_chroma_site="mpeg2", out_colorimetry="sRGB",
                            config=cases.ref_config_string(ref, dict(cases.LAN, gamma_mode="remap")))
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), (i, int((rc.frame(src[i]) != out[i]).sum()))
    launch(env, "filesrc location=%s blocksize=%d ! video/x-raw,format=NV12,width=%d,height=%d,framerate=30/1,colorimetry=bt709,chroma-site=mpeg2 "
                "! videoconvertscale primaries-mode=fast ! video/x-raw,format=BGRA,colorimetry=1:1:7:7 ! filesink location=%s"
           % (fin, src.shape[1], w, h, fprim))
    out = np.fromfile(fprim, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt709", in_chroma_site="mpeg2", out_colorimetry="1:1:7:7",
                            config=cases.ref_config_string(ref, dict(cases.LIN, primaries_mode="fast")))
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), (i, int((rc.frame(src[i]) != out[i]).sum()))


def test_videoconvertscale_p010_output(gst_env, ref):
    """NV12 -> P010_10LE (the encoder-facing direction) through the element: widening, the default bayer dither of a 10-bit destination."""
    env, tmp = gst_env
    w, h, n = 640, 360, 2
    fin, fout = tmp / "po_in.nv12", tmp / "po_out.p010"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale "
                "! video/x-raw,format=P010_10LE,colorimetry=bt709,chroma-site=mpeg2 ! filesink location=%s" % (n, w, h, fin, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("NV12", w, h, "P010_10LE", w, h, in_colorimetry="bt709", in_chroma_site="mpeg2", out_colorimetry="bt709",
                            out_chroma_site="mpeg2", config=cases.ref_config_string(ref, cases.LIN))
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), (i, int((rc.frame(src[i]) != out[i]).sum()))


def test_audioresample_element_discont_drains_and_resets(gst_env, ref):
    """Two streams through `concat`: the second one starts with a DISCONT buffer, so the element drains the filter's history with the old
    stream's timing, resets, and starts afresh (gstaudioresample.c:898-940) - the output is two independently resampled streams."""
    env, tmp = gst_env
    fin, fout = tmp / "ard_in.f32", tmp / "ard_out.f32"
    launch(env, "concat name=c ! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdaudioresample quality=4 ! audio/x-raw,rate=44100 "
                "! filesink location=%s audiotestsrc num-buffers=10 wave=white-noise samplesperbuffer=1024 ! audio/x-raw,format=F32LE,rate=48000,channels=2 ! c. "
                "audiotestsrc num-buffers=7 wave=pink-noise samplesperbuffer=1024 ! audio/x-raw,format=F32LE,rate=48000,channels=2 ! c." % (fin, fout))
    src = np.fromfile(fin, np.float32).reshape(-1, 2)
    out = np.fromfile(fout, np.float32).reshape(-1, 2)
    assert len(src) == 17 * 1024
    exp = []
    for part in (src[:10 * 1024], src[10 * 1024:]):
        rr = ref.AudioResampler("F32LE", 2, 48000, 44100, quality=4)
        for off in range(0, len(part), 1024):
            blk = part[off:off + 1024]
            exp.append(rr.resample(blk, in_frames=len(blk), out_frames=rr.get_out_frames(len(blk))))
        lat = rr.get_max_latency()
        exp.append(rr.resample(None, in_frames=lat, out_frames=rr.get_out_frames(lat)))
    exp = np.concatenate(exp)
    assert out.shape == exp.shape, (out.shape, exp.shape)
    assert (out == exp).all()


@pytest.mark.parametrize("batch", [1, 4])
def test_videoconvertscale_element_gbr_planes_in_frame_order(gst_env, ref, batch):
    """GBR through the element with system memory on both sides: the single-buffer path (batch-buffers=1) hands GstVideoFrame's planes
    - G, B, R - to gstamd_video_converter_frame_planes; round 4 rotated the channels there while the batched path was right"""
    env, tmp = gst_env
    w, h, n = 322, 180, 4
    fin, fmid, fout = tmp / ("gbr_in_%d.bgra" % batch), tmp / ("gbr_mid_%d.gbr" % batch), tmp / ("gbr_out_%d.rgba" % batch)
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=BGRA,width=%d,height=%d ! tee name=t "
                "t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale batch-buffers=%d ! video/x-raw,format=GBR ! tee name=u "
                "u. ! queue ! filesink location=%s u. ! queue ! videoconvertscale batch-buffers=%d ! video/x-raw,format=RGBA,width=%d,height=%d ! filesink location=%s"
           % (n, w, h, fin, batch, fmid, batch, w // 2, h // 2, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    mid = np.fromfile(fmid, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    a = ref.VideoConverter("BGRA", w, h, "GBR", w, h)
    b = ref.VideoConverter("GBR", w, h, "RGBA", w // 2, h // 2, config=cases.ref_config_string(ref, cases.LIN))
    for i in range(n):
        assert (a.frame(src[i]) == mid[i]).all(), i
        assert (b.frame(mid[i]) == out[i]).all(), i


def test_compositor_is_a_video_aggregator_on_the_references_own_version(gst_env):
    """`compositor` is a GstVideoAggregator subclass (compositor.c:808-809) with pads derived from GstVideoAggregatorConvertPad
    (gstvideoaggregator.c:656-678) - wherever that base class is public API, which includes the reference's own version.  The conda 1.14
    runtime keeps the GstAggregator form (there GstVideoAggregator lived in gst-plugins-bad's unstable library)."""
    env, tmp = gst_env
    if env.get("GSTAMD_RUNTIME") != "1.29":
        pytest.skip("the 1.14 runtime has no public GstVideoAggregator")
    r = subprocess.run([env["GSTAMD_LAUNCH_BIN"], "--types", "compositor"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("GstAmdCompositor < GstVideoAggregator < GstAggregator < GstElement"), lines
    assert lines[1].startswith("pad: GstAmdCompositorPadObj < GstVideoAggregatorConvertPad < GstVideoAggregatorPad < GstAggregatorPad < GstPad"), lines


def test_videoconvertscale_interlaced_caps_are_converted_field_aware(gst_env, ref):
    """interlace-mode=interleaved on both sides (round 6; the round-5 element refused such caps): 576i NV12 -> BGRA through system memory, and 1080i I420
    scaled to 576i in HBM (the interlaced plane scaler, element default method bilinear) - against the reference converter made for interlaced infos"""
    env, tmp = gst_env
    w, h, n = 720, 576, 3
    fin, fout, fsc = tmp / "i_in.nv12", tmp / "i_out.bgra", tmp / "i_sc.i420"
    launch(env, "videotestsrc num-buffers=%d pattern=smpte ! video/x-raw,format=NV12,width=%d,height=%d,interlace-mode=interleaved,colorimetry=bt601,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw,format=BGRA,interlace-mode=interleaved ! filesink location=%s"
           % (n, w, h, fin, fout))
    src = np.fromfile(fin, np.uint8).reshape(n, -1)
    out = np.fromfile(fout, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt601", in_chroma_site="mpeg2", interlaced=True)
    rp = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt601", in_chroma_site="mpeg2")
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), i
    assert (rp.frame(src[0]) != out[0]).any()          # ... which is not what the progressive converter makes of the same bytes
    W, H, ow, oh = 1920, 1080, 720, 576
    fin2 = tmp / "i_in.i420"
    launch(env, "videotestsrc num-buffers=2 pattern=smpte ! video/x-raw,format=I420,width=%d,height=%d,interlace-mode=interleaved,colorimetry=bt709 "
                "! tee name=t t. ! queue ! filesink location=%s t. ! queue ! amdhipupload ! videoconvertscale ! video/x-raw(memory:AMDHIPMemory),format=I420,width=%d,height=%d "
                "! amdhipdownload ! filesink location=%s" % (W, H, fin2, ow, oh, fsc))
    src = np.fromfile(fin2, np.uint8).reshape(2, -1)
    out = np.fromfile(fsc, np.uint8).reshape(2, -1)
    rs = ref.VideoConverter("I420", W, H, "I420", ow, oh, in_colorimetry="bt709", out_colorimetry="bt709", config=cases.ref_config_string(ref, cases.LIN), interlaced=True)
    for i in range(2):
        assert (rs.frame(src[i]) == out[i]).all(), i


def test_videoconvertscale_refuses_alternate_and_mismatched_interlace_modes(gst_env):
    """fields / alternate are not negotiated, and neither is a change of the mode (gst_video_converter_new: "we won't ever do deinterlace")"""
    env, tmp = gst_env
    for caps_in, caps_out in (("interlace-mode=interleaved", "interlace-mode=progressive"),):
        r = subprocess.run([env["GSTAMD_LAUNCH_BIN"], "-q"] + ("videotestsrc num-buffers=1 ! video/x-raw,format=NV12,width=320,height=240,%s ! videoconvertscale ! "
                                                              "video/x-raw,format=BGRA,%s ! fakesink" % (caps_in, caps_out)).split(),
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert r.returncode != 0, r.stdout[-500:]


def test_videoconvertscale_rescales_navigation_events_and_size_tagged_metas(gst_env):
    """src_event and transform_meta of the element (gstvideoconvertscale.c:2008-2037, 773-829; plugins/tests/live_props.c nav-meta): a navigation event
    travelling upstream through a scaling element names input pixels, a GstVideoCropMeta arrives on the output buffer scaled to the output size"""
    env, tmp = gst_env
    only_on_114(env, "plugins/tests/live_props drives the element through GstHarness (libgstcheck)")
    exe = os.path.join(ROOT, "plugins", "tests", "live_props")
    r = subprocess.run([exe, "nav-meta"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_audioresample_resamples_a_buffer_list_in_one_call_with_the_per_buffer_results(gst_env):
    """a GstBufferList on audioresample's sink pad (plugins/gstamdaudioresample.c amd_ar_chain_list; plugins/tests/live_props.c audio-list): the run of buffers
    goes through ONE resample call; the output buffers - count, bytes, timestamps, offsets - are those of the same stream pushed buffer by buffer (which
    test_audioresample_element_matches_reference pins to the reference)"""
    env, tmp = gst_env
    only_on_114(env, "plugins/tests/live_props drives the element through GstHarness (libgstcheck)")
    exe = os.path.join(ROOT, "plugins", "tests", "live_props")
    r = subprocess.run([exe, "audio-list"], env=dict(env, GSTAMD_ELEMENT_STATS="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    assert r.stderr.count("buffers of a list") == 2 and "6 buffers of a list" in r.stderr, r.stderr[-2000:]


def test_hip_allocator_copies_in_hbm_and_shares_windows(gst_env):
    """GstAllocator::mem_copy and ::mem_share of the HIP allocator (plugins/gstamdhipmemory.c; plugins/tests/live_props.c hip-memory): gst_buffer_copy_deep
    of an HBM buffer gives another HBM allocation with the same bytes, gst_memory_share a window that keeps the allocation alive"""
    env, tmp = gst_env
    only_on_114(env, "plugins/tests/live_props links the 1.14 build of the plugin")
    exe = os.path.join(ROOT, "plugins", "tests", "live_props")
    r = subprocess.run([exe, "hip-memory"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("pattern,fmt", [("smpte", "NV12"), ("ball", "I420"), ("snow", "BGRA"), ("colors", "UYVY"), ("blink", "P010_10LE")])
def test_amdhipvideotestsrc_paints_the_reference_sources_frames_in_hbm(gst_env, pattern, fmt):
    """`amdhipvideotestsrc` (plugins/gstamdvideotestsrc.c): frames painted in HBM, downloaded, against the stock videotestsrc of the same runtime's own
    version where that is the reference's (1.29) - on 1.14 the stock source is an older painter, compared on the patterns that did not change"""
    env, tmp = gst_env
    w, h, n = 320, 240, 4
    fa, fb = tmp / ("vts_%s_ours.raw" % pattern), tmp / ("vts_%s_stock.raw" % pattern)
    launch(env, "amdhipvideotestsrc num-buffers=%d pattern=%s ! video/x-raw(memory:AMDHIPMemory),format=%s,width=%d,height=%d,framerate=30/1 ! amdhipdownload ! filesink location=%s"
           % (n, pattern, fmt, w, h, fa))
    launch(env, "videotestsrc num-buffers=%d pattern=%s ! video/x-raw,format=%s,width=%d,height=%d,framerate=30/1 ! filesink location=%s" % (n, pattern, fmt, w, h, fb))
    a, b = np.fromfile(fa, np.uint8), np.fromfile(fb, np.uint8)
    assert a.size == b.size and a.size > 0
    if env.get("GSTAMD_RUNTIME") == "1.29" or pattern in ("smpte", "colors", "blink"):
        import cases
        from oracle import ref
        ri = ref.video_info(fmt, w, h)
        for k in range(n):
            fa_k, fb_k = a.reshape(n, -1)[k], b.reshape(n, -1)[k]
            va = cases.visible_bytes(fmt, w, h, list(ri["stride"]), list(ri["offset"]), fa_k)
            vb = cases.visible_bytes(fmt, w, h, list(ri["stride"]), list(ri["offset"]), fb_k)
            assert (va == vb).all(), (pattern, fmt, k, int((va != vb).sum()))


def test_amdhipvideotestsrc_feeds_the_converter_without_leaving_hbm(gst_env, ref):
    """BASELINE config 1's shape with nothing on the host: amdhipvideotestsrc ! NV12 1080p ! videoconvertscale ! BGRA in HBM; the last step downloads for the check"""
    env, tmp = gst_env
    w, h, n = 1920, 1080, 3
    fa, fb = tmp / "c1_hbm.bgra", tmp / "c1_src.nv12"
    launch(env, "amdhipvideotestsrc num-buffers=%d pattern=ball ! video/x-raw(memory:AMDHIPMemory),format=NV12,width=%d,height=%d,framerate=30/1,colorimetry=bt709,chroma-site=mpeg2 "
                "! tee name=t t. ! queue ! amdhipdownload ! filesink location=%s t. ! queue ! videoconvertscale ! video/x-raw(memory:AMDHIPMemory),format=BGRA ! amdhipdownload "
                "! filesink location=%s" % (n, w, h, fb, fa))
    src = np.fromfile(fb, np.uint8).reshape(n, -1)
    out = np.fromfile(fa, np.uint8).reshape(n, -1)
    rc = ref.VideoConverter("NV12", w, h, "BGRA", w, h, in_colorimetry="bt709", in_chroma_site="mpeg2")
    for i in range(n):
        assert (rc.frame(src[i]) == out[i]).all(), i
