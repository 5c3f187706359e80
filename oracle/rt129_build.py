#!/usr/bin/env python3
"""A GStreamer 1.29 RUNTIME hand-built from /root/reference (TEST INFRASTRUCTURE ONLY): the reference's own core, libgstbase, libgstvideo
(with gstvideoaggregator.c), libgstaudio, and the stock elements the plugin tests need (coreelements, videotestsrc, audiotestsrc), so that
the elements of plugins/ can be loaded and RUN on the reference's own version - not only syntax-checked against its headers.

Same rules as oracle/ref_build.py (whose generated headers and helpers this uses): the sources are compiled where they lie, nothing is
copied into the repository, ORC C backups (-DDISABLE_ORC), the registry and gst_parse are configured out (no bison / flex here; pipelines
are built by plugins/tests/launch129.c, plugins are loaded by path).

Output (git-ignored, shipped to the GPU box by gpurun): oracle/_ref/rt129/
    lib/libgstreamer-1.0.so.0  lib/libgstbase-1.0.so.0  lib/libgstvideo-1.0.so.0  lib/libgstaudio-1.0.so.0   (+ unversioned links)
    plugins/libgstcoreelements.so  plugins/libgstvideotestsrc.so  plugins/libgstaudiotestsrc.so

Only tests/ (through plugins/tests/launch129) may use it."""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_build as R  # noqa: E402

RT = os.path.join(R.OUT, "rt129")
OBJ = os.path.join(RT, "obj")
LIB = os.path.join(RT, "lib")
PLUG = os.path.join(RT, "plugins")

# libgstaudio: what elements built on GstBaseTransform / GstBaseSrc need (the decoder / encoder / sink base classes want libgsttag and are not used)


def main():
    if not os.path.isdir(R.REF):
        print("reference tree not present (%s): keeping prebuilt oracle/_ref/rt129 as is" % R.REF)
        return 0
    for d in (OBJ, LIB, PLUG):
        os.makedirs(d, exist_ok=True)
    R.gen_core_headers()
    R.gen_enums()
    R.gen_orc()
    R.write_if_changed(os.path.join(R.GEN, "videotestsrc", "gstvideotestsrcorc.h"), '#include "%s"\n' % os.path.join(R.PB, "gst/videotestsrc/gstvideotestsrcorc-dist.h"))
    R.write_if_changed(os.path.join(R.GEN, "videotestsrc", "gstvideotestsrcorc.c"), '#include "%s"\n' % os.path.join(R.PB, "gst/videotestsrc/gstvideotestsrcorc-dist.c"))
    common = ["gcc", "-O2", "-w", "-fPIC", "-D_GNU_SOURCE", "-DHAVE_CONFIG_H", "-DDISABLE_ORC", "-ffp-contract=off", "-I" + R.GEN, "-I" + os.path.join(R.GEN, "gst"),
              "-I" + R.CORE, "-I" + os.path.join(R.CORE, "gst"), "-I" + os.path.join(R.CORE, "libs"), "-I" + os.path.join(R.PB, "gst-libs"),
              "-I" + os.path.join(R.GEN, "gst/video"), "-I" + os.path.join(R.GEN, "gst/audio")] + R.GLIB_INC
    groups = {}                # library -> [(src, obj, flags)]
    core_flags = ["-DGST_EXPORTS", "-DBUILDING_GST", '-DG_LOG_DOMAIN="GStreamer"', "-DGST_DISABLE_DEPRECATED"]
    core = []
    for src in sorted(glob.glob(os.path.join(R.CORE, "gst/*.c"))):
        if os.path.basename(src) not in R.CORE_SKIP:
            core.append((src, "core_" + os.path.basename(src)[:-2] + ".o", core_flags))
    for src in sorted(glob.glob(os.path.join(R.CORE, "gst/printf/*.c"))):
        core.append((src, "printf_" + os.path.basename(src)[:-2] + ".o", core_flags + ["-I" + os.path.join(R.CORE, "gst/printf"), "-DSTATIC=G_GNUC_INTERNAL"]))
    core.append((os.path.join(R.GEN, "gst/gstenumtypes.c"), "core_gstenumtypes.o", core_flags))
    groups["gstreamer-1.0"] = core
    bflags = ["-DBUILDING_GST_BASE", '-DG_LOG_DOMAIN="GStreamer-Base"']
    base_src = ("gstadapter.c gstaggregator.c gstbaseparse.c gstbasesink.c gstbasesrc.c gstbasetransform.c gstbitreader.c gstbitwriter.c gstbytereader.c "
                "gstbytewriter.c gstcollectpads.c gstdataqueue.c gstflowcombiner.c gstpushsrc.c gstqueuearray.c gsttypefindhelper.c").split()       # libs/gst/base/meson.build:1-18
    groups["gstbase-1.0"] = [(os.path.join(R.CORE, "libs/gst/base", f), "base_" + f[:-2] + ".o", bflags) for f in base_src]
    vflags = ["-DBUILDING_GST_VIDEO", '-DG_LOG_DOMAIN="GStreamer-Video"', "-I" + os.path.join(R.PB, "gst-libs/gst/video")]
    video_src = ("colorbalance.c colorbalancechannel.c convertframe.c gstvideoaffinetransformationmeta.c gstvideocodecalphameta.c gstvideodscmeta.c gstvideoaggregator.c "
                 "gstvideodecoder.c gstvideodmabufpool.c gstvideoencoder.c gstvideofilter.c gstvideometa.c gstvideopool.c gstvideosink.c gstvideotimecode.c gstvideoutils.c "
                 "gstvideoutilsprivate.c navigation.c video.c video-anc.c video-blend.c video-chroma.c video-color.c video-converter.c video-dither.c video-event.c "
                 "video-format.c video-frame.c video-hdr.c video-info.c video-info-dma.c video-multiview.c video-resampler.c video-scaler.c video-sei.c video-tile.c "
                 "video-overlay-composition.c videodirection.c videoorientation.c videooverlay.c gsth274.c").split()        # gst-libs/gst/video/meson.build:1-43
    video = [(os.path.join(R.PB, "gst-libs/gst/video", f), "video_" + f[:-2] + ".o", vflags) for f in video_src]
    video.append((os.path.join(R.GEN, "gst/video/video-orc.c"), "video_video-orc.o", vflags))
    video.append((os.path.join(R.GEN, "gst/video/video-enumtypes.c"), "video_enumtypes.o", vflags))
    groups["gstvideo-1.0"] = video
    aflags = ["-DBUILDING_GST_AUDIO", '-DG_LOG_DOMAIN="GStreamer-Audio"', "-I" + os.path.join(R.PB, "gst-libs/gst/audio")]
    audio_src = ("audio.c audio-buffer.c audio-channel-mixer.c audio-channels.c audio-converter.c audio-format.c audio-info.c audio-quantize.c audio-resampler.c "
                 "gstaudiofilter.c gstaudiometa.c gstaudiostreamalign.c").split()           # gst-libs/gst/audio/meson.build:1-29 minus the classes that want libgsttag / ring buffers
    audio = [(os.path.join(R.PB, "gst-libs/gst/audio", f), "audio_" + f[:-2] + ".o", aflags) for f in audio_src]
    audio.append((os.path.join(R.GEN, "gst/audio/gstaudiopack.c"), "audio_gstaudiopack.o", aflags))
    audio.append((os.path.join(R.GEN, "gst/audio/audio-enumtypes.c"), "audio_enumtypes.o", aflags))
    groups["gstaudio-1.0"] = audio
    pflags = ['-DG_LOG_DOMAIN="GStreamer-plugins"', "-DGST_USE_UNSTABLE_API"]
    ce_src = ("gstcapsfilter.c gstclocksync.c gstconcat.c gstdataurisrc.c gstdownloadbuffer.c gstcoreelementsplugin.c gstelements_private.c gstfakesink.c gstfakesrc.c "
              "gstfdsink.c gstfdsrc.c gstfilesrc.c gstfilesink.c gstfunnel.c gstidentity.c gstinputselector.c gstmultiqueue.c gstoutputselector.c gstqueue2.c gstqueue.c "
              "gstsparsefile.c gststreamiddemux.c gsttee.c gsttypefindelement.c gstvalve.c").split()                    # plugins/elements/meson.build:1-27
    groups["plugin:gstcoreelements"] = [(os.path.join(R.CORE, "plugins/elements", f), "ce_" + f[:-2] + ".o", pflags + ["-I" + os.path.join(R.CORE, "plugins/elements")])
                                        for f in ce_src]
    vts = os.path.join(R.PB, "gst/videotestsrc")
    groups["plugin:gstvideotestsrc"] = [(os.path.join(vts, f), "vts_" + f[:-2] + ".o", pflags + ["-I" + vts, "-I" + os.path.join(R.GEN, "videotestsrc")])
                                        for f in ("gstvideotestsrc.c", "videotestsrc.c")]
    groups["plugin:gstvideotestsrc"].append((os.path.join(R.GEN, "videotestsrc/gstvideotestsrcorc.c"), "vts_orc.o", pflags))
    groups["plugin:gstaudiotestsrc"] = [(os.path.join(R.PB, "gst/audiotestsrc/gstaudiotestsrc.c"), "ats_gstaudiotestsrc.o", pflags)]

    def compile_one(job):
        src, obj, extra = job
        objp = os.path.join(OBJ, obj)
        if os.path.exists(objp) and os.path.getmtime(objp) > os.path.getmtime(src) and os.path.getmtime(objp) > os.path.getmtime(__file__):
            return src, 0, ""
        rc, out = R.run(common + extra + ["-c", src, "-o", objp])
        return src, rc, out

    jobs = [j for g in groups.values() for j in g]
    failed = []
    with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for src, rc, out in ex.map(compile_one, jobs):
            if rc != 0:
                failed.append(src)
                print("FAILED", src, "\n", out[-2500:])
    if failed:
        print("%d TUs failed" % len(failed))
        return 1
    glibs = ["-L%s/lib" % R.CONDA, "-Wl,-rpath,%s/lib" % R.CONDA, "-lgobject-2.0", "-lgmodule-2.0", "-lgio-2.0", "-lglib-2.0", "-lm", "-ldl", "-lpthread"]
    deps = {"gstreamer-1.0": [], "gstbase-1.0": ["gstreamer-1.0"], "gstvideo-1.0": ["gstbase-1.0", "gstreamer-1.0"], "gstaudio-1.0": ["gstbase-1.0", "gstreamer-1.0"],
            "plugin:gstcoreelements": ["gstbase-1.0", "gstreamer-1.0"], "plugin:gstvideotestsrc": ["gstvideo-1.0", "gstbase-1.0", "gstreamer-1.0"],
            "plugin:gstaudiotestsrc": ["gstaudio-1.0", "gstbase-1.0", "gstreamer-1.0"]}
    for name in ("gstreamer-1.0", "gstbase-1.0", "gstvideo-1.0", "gstaudio-1.0", "plugin:gstcoreelements", "plugin:gstvideotestsrc", "plugin:gstaudiotestsrc"):
        objs = [os.path.join(OBJ, j[1]) for j in groups[name]]
        if name.startswith("plugin:"):
            so = os.path.join(PLUG, "lib%s.so" % name[7:])
            soname = []
        else:
            so = os.path.join(LIB, "lib%s.so.0" % name)
            soname = ["-Wl,-soname,lib%s.so.0" % name]
        cmd = ["gcc", "-shared", "-o", so] + soname + objs + ["-L" + LIB, "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,$ORIGIN"] + ["-l" + d for d in deps[name]] + glibs + [
            "-Wl,--no-undefined"]
        rc, out = R.run(cmd)
        if rc != 0:
            print("LINK FAILED", name, "\n", out[-4000:])
            return 1
        if not name.startswith("plugin:"):
            link = os.path.join(LIB, "lib%s.so" % name)
            if os.path.lexists(link):
                os.remove(link)
            os.symlink(os.path.basename(so), link)
        print("built", so)
    return 0


if __name__ == "__main__":
    sys.exit(main())
