#!/usr/bin/env python3
"""Hand-build of the REFERENCE's own hot-path code into oracle/_ref/ (TEST INFRASTRUCTURE ONLY).

Nothing here is product code and nothing here is copied into the repo: the reference's sources
are compiled *where they lie* under /root/reference (read-only), with a hand-written config.h and
the generated headers meson would normally produce (gstconfig.h, gstversion.h, *-enumtypes.[ch]),
against the glib 2.69 in /opt/conda.  ORC is un-vendored, so the ORC C backups (`*-dist.c`,
-DDISABLE_ORC) are used - exactly what the reference's meson does when orc is missing
(subprojects/gst-plugins-base/gst-libs/gst/video/meson.build:134-152).

Output (git-ignored, but shipped to the GPU box by gpurun):
  oracle/_ref/libgstref.so   core + libgstvideo subset + libgstaudio subset + compositor blend
                             + oracle/ref_driver.c (plain-C entry points for ctypes)

Only tests/, bench.py's cpu_baseline leg and __graft_entry__ may load that library.
SSE paths are left OFF on purpose (no HAVE_SSE*/HAVE_ORC) so the float resampler uses the C
summation order the parity target is pinned to.
"""
import concurrent.futures as cf
import glob
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GST_REFERENCE", "/root/reference")
CORE = os.path.join(REF, "subprojects/gstreamer")
PB = os.path.join(REF, "subprojects/gst-plugins-base")
OUT = os.path.join(HERE, "_ref")
GEN = os.path.join(OUT, "gen")
OBJ = os.path.join(OUT, "obj")
CONDA = "/opt/conda"
GLIB_INC = ["-I%s/include/glib-2.0" % CONDA, "-I%s/lib/glib-2.0/include" % CONDA]
MKENUMS = os.path.join(CONDA, "bin/glib-mkenums")

CONFIG_H = r'''
/* hand-written config.h for the oracle build of the reference (see oracle/ref_build.py) */
#pragma once
#define VERSION "1.29.2.1"
#define PACKAGE "gstreamer"
#define PACKAGE_NAME "GStreamer"
#define PACKAGE_VERSION "1.29.2.1"
#define GETTEXT_PACKAGE "gstreamer-1.0"
#define GST_API_VERSION "1.0"
#define GST_LICENSE "LGPL"
#define GST_PACKAGE_NAME "GStreamer oracle build"
#define GST_PACKAGE_ORIGIN "oracle"
#define GST_PACKAGE_RELEASE_DATETIME "2026-01-01"
#define LOCALEDIR "/nonexistent/locale"
#define LIBDIR "/nonexistent/lib"
#define GST_DATADIR "/nonexistent/share"
#define PLUGINDIR "/nonexistent/lib/gstreamer-1.0"
#define GST_PLUGIN_SCANNER_INSTALLED "/nonexistent/gst-plugin-scanner"
#define GST_PLUGIN_SCANNER_SUBDIR "libexec"
#define GST_PLUGIN_SUBDIR "lib"
#define GST_EXTRA_MODULE_SUFFIX ""
#define HOST_CPU "x86_64"
#define TARGET_CPU "x86_64"
#define HAVE_CPU_X86_64 1
#define HAVE_UNISTD_H 1
#define HAVE_SYS_TIME_H 1
#define HAVE_SYS_SOCKET_H 1
#define HAVE_SYS_TYPES_H 1
#define HAVE_SYS_STAT_H 1
#define HAVE_SYS_WAIT_H 1
#define HAVE_SYS_RESOURCE_H 1
#define HAVE_SYS_UIO_H 1
#define HAVE_POLL 1
#define HAVE_POLL_H 1
#define HAVE_SYS_POLL_H 1
#define HAVE_PPOLL 1
#define HAVE_PTHREAD_H 1
#define HAVE_PTHREAD_CONDATTR_SETCLOCK 1
#define HAVE_CLOCK_GETTIME 1
#define HAVE_POSIX_TIMERS 1
#define HAVE_MONOTONIC_CLOCK 1
#define HAVE_GMTIME_R 1
#define HAVE_LOCALTIME_R 1
#define HAVE_SIGACTION 1
#define HAVE_DLFCN_H 1
#define HAVE_DLADDR 1
#define HAVE_STDINT_H 1
#define HAVE_INTTYPES_H 1
#define HAVE_INTMAX_T 1
#define HAVE_STDINT_H_WITH_UINTMAX 1
#define HAVE_INTTYPES_H_WITH_UINTMAX 1
#define PACKAGE_BUGREPORT "oracle-build"
#define HAVE_LONG_LONG 1
#define HAVE_PTRDIFF_T 1
#define HAVE_LONG_DOUBLE 1
#define HAVE_MMAP 1
#define HAVE_POSIX_MEMALIGN 1
#define HAVE_GETPAGESIZE 1
#define HAVE_EVENTFD 1
#define HAVE_PIPE2 1
#define HAVE_SYS_PRCTL_H 1
#define HAVE_PTHREAD_SETNAME_NP_WITH_TID 1
#define HAVE_TM_GMTOFF 1
#define HAVE_RINT 1
#define HAVE_LRINTF 1
#define HAVE_LOG2 1
#define HAVE_STRINGS_H 1
#define HAVE_STRING_H 1
#define HAVE_STDLIB_H 1
#define HAVE_MEMORY_H 1
#define HAVE_DECL_LOCALTIME_R 1
#define HAVE_DECL_STRSIGNAL 1
#define HAVE_GETRUSAGE 1
#define HAVE_FSEEKO 1
#define HAVE_FTELLO 1
#define HAVE_UNALIGNED_ACCESS 1
#define SIZEOF_CHAR 1
#define SIZEOF_SHORT 2
#define SIZEOF_INT 4
#define SIZEOF_LONG 8
#define SIZEOF_VOIDP 8
#define SIZEOF_OFF_T 8
#define MEMORY_ALIGNMENT_MALLOC 1
#define GST_DISABLE_GST_TRACER_HOOKS 1
#define DISABLE_ORC 1
#define ENABLE_NLS 0
#undef ENABLE_NLS
/* deliberately NOT defined: HAVE_ORC, HAVE_SSE, HAVE_SSE2, HAVE_SSE41 (keeps the C paths) */
'''

CORE_SKIP = {"gstandroid.c", "gstpluginloader-win32.c"}

VIDEO_SRCS = [
    "video-converter.c", "video-format.c", "video-scaler.c", "video-resampler.c",
    "video-chroma.c", "video-color.c", "video-info.c", "video-frame.c", "video-dither.c",
    "video-tile.c", "video-multiview.c", "gstvideometa.c", "video.c", "video-hdr.c",
    "video-info-dma.c", "gstvideocodecalphameta.c", "gstvideopool.c", "gstvideotimecode.c",
]
AUDIO_SRCS = [
    "audio-resampler.c", "audio-format.c", "audio-info.c", "audio-channels.c", "audio.c",
    "gstaudiometa.c", "audio-converter.c", "audio-quantize.c", "audio-channel-mixer.c",
]
COMPOSITOR_SRCS = ["blend.c"]


def run(cmd, merge=True, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT if merge else subprocess.DEVNULL,
                       text=True, **kw)
    return r.returncode, r.stdout


def write_if_changed(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if os.path.exists(path) and open(path).read() == text:
        return
    open(path, "w").write(text)


def gen_core_headers():
    src = open(os.path.join(CORE, "gst/gstconfig.h.in")).read()
    src = src.replace("@GST_DISABLE_PARSE_DEFINE@", "#define GST_DISABLE_PARSE 1")
    src = src.replace("@GST_DISABLE_REGISTRY_DEFINE@", "#define GST_DISABLE_REGISTRY 1")
    src = src.replace("@GST_DISABLE_CAST_CHECKS_DEFINE@", "0")
    src = src.replace("@GST_DISABLE_GLIB_ASSERTS_DEFINE@", "0")
    src = src.replace("@GST_DISABLE_GLIB_CHECKS_DEFINE@", "0")
    src = re.sub(r"@(\w+)_DEFINE@", r"/* #undef \1 */", src)
    src = re.sub(r"@\w+@", "", src)
    write_if_changed(os.path.join(GEN, "gst/gstconfig.h"), src)
    v = open(os.path.join(CORE, "gst/gstversion.h.in")).read()
    for k, val in (("MAJOR", "1"), ("MINOR", "29"), ("MICRO", "2"), ("NANO", "1")):
        v = v.replace("@GST_VERSION_%s@" % k, val).replace("@PACKAGE_VERSION_%s@" % k, val)
    write_if_changed(os.path.join(GEN, "gst/gstversion.h"), v)
    write_if_changed(os.path.join(GEN, "config.h"), CONFIG_H)


def mkenums(headers, out_base, hdr_prefix, body_prefix, decorator, guard, inc_prefix):
    """Equivalent of meson's gnome.mkenums_simple()."""
    h_cmd = [sys.executable, MKENUMS,
             "--fhead", "#pragma once\n\n#include <glib-object.h>\n%s\n\nG_BEGIN_DECLS\n" % hdr_prefix,
             "--fprod", "\n/* enumerations from \"@basename@\" */\n",
             "--vhead", "%s\nGType @enum_name@_get_type (void);\n#define @ENUMPREFIX@_TYPE_@ENUMSHORT@ (@enum_name@_get_type())\n" % decorator,
             "--ftail", "\nG_END_DECLS\n"] + headers
    rc, out = run(h_cmd, merge=False)
    assert rc == 0, out
    write_if_changed(out_base + ".h", out)
    c_cmd = [sys.executable, MKENUMS,
             "--fhead", "%s\n#include \"%s\"\n\n#define C_ENUM(v) ((gint) v)\n#define C_FLAGS(v) ((guint) v)\n" % (
                 body_prefix, os.path.basename(out_base) + ".h"),
             "--fprod", "\n/* enumerations from \"@basename@\" */\n#include \"%s@basename@\"\n" % inc_prefix,
             "--vhead", "\nGType\n@enum_name@_get_type (void)\n{\n  static gsize gtype_id = 0;\n  static const G@Type@Value values[] = {",
             "--vprod", "    { C_@TYPE@(@VALUENAME@), \"@VALUENAME@\", \"@valuenick@\" },",
             "--vtail", "    { 0, NULL, NULL }\n  };\n  if (g_once_init_enter (&gtype_id)) {\n    GType new_type = g_@type@_register_static (g_intern_static_string (\"@EnumName@\"), values);\n    g_once_init_leave (&gtype_id, new_type);\n  }\n  return (GType) gtype_id;\n}"] + headers
    rc, out = run(c_cmd, merge=False)
    assert rc == 0, out
    write_if_changed(out_base + ".c", out)


def gen_enums():
    core_h = sorted(h for h in glob.glob(os.path.join(CORE, "gst/*.h"))
                    if not re.search(r"private|glib-compat|gstmacos|math-compat|gstenumtypes|gst_private", h))
    mkenums(core_h, os.path.join(GEN, "gst/gstenumtypes"), "#include <gst/gstconfig.h>",
            '#include "gst/gst_private.h"\n#include <gst/gst.h>', "GST_API", "", "")
    # header list = video_mkenum_headers (gst-libs/gst/video/meson.build:98-119)
    vid_h = [os.path.join(PB, "gst-libs/gst/video", h) for h in (
        "video.h video-anc.h video-format.h video-frame.h video-chroma.h video-color.h video-converter.h "
        "video-dither.h video-info.h video-overlay-composition.h video-resampler.h video-scaler.h video-tile.h "
        "gstvideometa.h gstvideotimecode.h gstvideoutils.h gstvideoencoder.h gstvideodecoder.h colorbalance.h "
        "navigation.h").split()]
    mkenums(vid_h, os.path.join(GEN, "gst/video/video-enumtypes"), "#include <gst/video/video-prelude.h>",
            "#ifdef HAVE_CONFIG_H\n#include \"config.h\"\n#endif\n#include <gst/video/video.h>\n#include <gst/video/video-chroma.h>\n#include <gst/video/video-resampler.h>\n#include <gst/video/video-scaler.h>\n#include <gst/video/video-converter.h>\n#include <gst/video/video-dither.h>\n#include <gst/video/video-multiview.h>", "GST_VIDEO_API", "", "")
    aud_h = sorted(h for h in glob.glob(os.path.join(PB, "gst-libs/gst/audio/*.h"))
                   if not re.search(r"private|orc|enumtypes|prelude|macros", h))
    mkenums(aud_h, os.path.join(GEN, "gst/audio/audio-enumtypes"), "#include <gst/audio/audio-prelude.h>",
            "#ifdef HAVE_CONFIG_H\n#include \"config.h\"\n#endif\n#include <gst/audio/audio.h>", "GST_AUDIO_API", "", "")


def gen_orc():
    # the reference's meson copies *-dist.[ch] to *-orc.[ch] when orc is absent; emulate with wrappers
    for sub, base, dist in (("gst/video", "video-orc", os.path.join(PB, "gst-libs/gst/video/video-orc-dist")),
                            ("gst/audio", "gstaudiopack", os.path.join(PB, "gst-libs/gst/audio/gstaudiopack-dist")),
                            ("compositor", "compositororc", os.path.join(PB, "gst/compositor/compositororc-dist"))):
        write_if_changed(os.path.join(GEN, sub, base + ".h"), '#include "%s.h"\n' % dist)
        write_if_changed(os.path.join(GEN, sub, base + ".c"), '#include "%s.c"\n' % dist)


def main():
    if not os.path.isdir(REF):
        print("reference tree not present (%s): keeping prebuilt oracle/_ref as is" % REF)
        return 0
    os.makedirs(OBJ, exist_ok=True)
    gen_core_headers()
    gen_enums()
    gen_orc()

    common = ["gcc", "-O2", "-w", "-fPIC", "-D_GNU_SOURCE", "-DHAVE_CONFIG_H", "-DDISABLE_ORC",
              "-ffp-contract=off", "-I" + GEN, "-I" + os.path.join(GEN, "gst"),
              "-I" + CORE, "-I" + os.path.join(CORE, "gst"), "-I" + os.path.join(CORE, "libs"),
              "-I" + os.path.join(PB, "gst-libs"), "-I" + os.path.join(GEN, "gst/video"),
              "-I" + os.path.join(GEN, "gst/audio"), "-I" + os.path.join(GEN, "compositor")] + GLIB_INC
    jobs = []  # (src, obj, extra flags)
    core_flags = ["-DGST_EXPORTS", "-DBUILDING_GST", '-DG_LOG_DOMAIN="GStreamer"', "-DGST_DISABLE_DEPRECATED"]
    for src in sorted(glob.glob(os.path.join(CORE, "gst/*.c"))):
        if os.path.basename(src) in CORE_SKIP:
            continue
        jobs.append((src, "core_" + os.path.basename(src)[:-2] + ".o", core_flags))
    for src in sorted(glob.glob(os.path.join(CORE, "gst/printf/*.c"))):
        jobs.append((src, "printf_" + os.path.basename(src)[:-2] + ".o",
                     core_flags + ["-I" + os.path.join(CORE, "gst/printf"), "-DSTATIC=G_GNUC_INTERNAL"]))
    jobs.append((os.path.join(GEN, "gst/gstenumtypes.c"), "core_gstenumtypes.o", core_flags))
    for b in ("gstbytewriter.c", "gstbytereader.c", "gstbitreader.c", "gstbitwriter.c"):
        jobs.append((os.path.join(CORE, "libs/gst/base", b), "base_" + b[:-2] + ".o",
                     ["-DBUILDING_GST_BASE", '-DG_LOG_DOMAIN="GStreamer-Base"']))
    vflags = ["-DBUILDING_GST_VIDEO", '-DG_LOG_DOMAIN="GStreamer-Video"', "-I" + os.path.join(PB, "gst-libs/gst/video")]
    for s in VIDEO_SRCS:
        jobs.append((os.path.join(PB, "gst-libs/gst/video", s), "video_" + s[:-2] + ".o", vflags))
    jobs.append((os.path.join(GEN, "gst/video/video-orc.c"), "video_video-orc.o", vflags))
    jobs.append((os.path.join(GEN, "gst/video/video-enumtypes.c"), "video_enumtypes.o", vflags))
    aflags = ["-DBUILDING_GST_AUDIO", '-DG_LOG_DOMAIN="GStreamer-Audio"', "-I" + os.path.join(PB, "gst-libs/gst/audio")]
    for s in AUDIO_SRCS:
        jobs.append((os.path.join(PB, "gst-libs/gst/audio", s), "audio_" + s[:-2] + ".o", aflags))
    jobs.append((os.path.join(GEN, "gst/audio/gstaudiopack.c"), "audio_gstaudiopack.o", aflags))
    jobs.append((os.path.join(GEN, "gst/audio/audio-enumtypes.c"), "audio_enumtypes.o", aflags))
    cflags = ["-I" + os.path.join(PB, "gst/compositor")]
    for s in COMPOSITOR_SRCS:
        jobs.append((os.path.join(PB, "gst/compositor", s), "comp_" + s[:-2] + ".o", cflags))
    jobs.append((os.path.join(GEN, "compositor/compositororc.c"), "comp_orc.o", cflags))
    jobs.append((os.path.join(HERE, "ref_driver.c"), "ref_driver.o", cflags))

    def compile_one(job):
        src, obj, extra = job
        objp = os.path.join(OBJ, obj)
        if os.path.exists(objp) and os.path.getmtime(objp) > os.path.getmtime(src) \
                and os.path.getmtime(objp) > os.path.getmtime(__file__):
            return src, 0, ""
        rc, out = run(common + extra + ["-c", src, "-o", objp])
        return src, rc, out

    failed = []
    with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for src, rc, out in ex.map(compile_one, jobs):
            if rc != 0:
                failed.append(src)
                print("FAILED", src, "\n", out[-3000:])
    if failed:
        print("%d TUs failed" % len(failed))
        return 1
    objs = [os.path.join(OBJ, j[1]) for j in jobs]
    so = os.path.join(OUT, "libgstref.so")
    rc, out = run(["gcc", "-shared", "-o", so] + objs + [
        "-L%s/lib" % CONDA, "-Wl,-rpath,%s/lib" % CONDA, "-lgobject-2.0", "-lgmodule-2.0", "-lglib-2.0",
        "-lm", "-ldl", "-lpthread", "-Wl,--no-undefined"])
    print(out[-6000:])
    if rc != 0:
        return rc
    print("built", so)
    return 0


if __name__ == "__main__":
    sys.exit(main())
