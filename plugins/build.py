"""Builds plugins/libgstamdhipdsp.so (GStreamer elements) against the GStreamer development files in
/opt/conda (1.14) and the product library.  python plugins/build.py"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CONDA = os.environ.get("GST_PREFIX", "/opt/conda")
OUT = os.path.join(HERE, "libgstamdhipdsp.so")


def build():
    srcs = [os.path.join(HERE, f) for f in ("gstamdplugin.c", "gstamdhipmemory.c", "gstamdhipbufferpool.c", "gstamdvideoconvertscale.c", "gstamdaudioresample.c", "gstamdaudioconvert.c", "gstamdcompositor.c", "gstamdhiptransfer.c", "gstamdvideotestsrc.c")]
    inc = ["-I%s/include/gstreamer-1.0" % CONDA, "-I%s/lib/gstreamer-1.0/include" % CONDA, "-I%s/include/glib-2.0" % CONDA,
           "-I%s/lib/glib-2.0/include" % CONDA]
    libdir = os.path.join(ROOT, "gstreamer_amd", "lib")
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-deprecated-declarations", "-o", OUT] + srcs + inc + [
        "-L%s/lib" % CONDA, "-Wl,-rpath,%s/lib" % CONDA, "-L" + libdir, "-Wl,-rpath,$ORIGIN/../gstreamer_amd/lib",
        "-lgstamddsp", "-lgstvideo-1.0", "-lgstaudio-1.0", "-lgstbase-1.0", "-lgstreamer-1.0", "-lgobject-2.0", "-lglib-2.0"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("plugin build failed:\n" + r.stdout)
    if r.stdout.strip():
        print(r.stdout)
    build_tools(inc)
    check_against_reference_headers()
    build129()
    return OUT


RT129 = os.path.join(ROOT, "oracle", "_ref", "rt129")
OUT129 = os.path.join(HERE, "rt129", "libgstamdhipdsp.so")


def build129():
    """Third build target: the same element sources compiled and LINKED against the reference's own version - the 1.29 runtime that
    oracle/rt129_build.py hand-builds from /root/reference (test infrastructure; headers from the reference tree, libraries under
    oracle/_ref/rt129/lib).  plugins/rt129/libgstamdhipdsp.so is what tests/test_plugin_gpu.py loads into that runtime through
    plugins/tests/launch129.  Only where the reference tree and the runtime are present (the build container); returns None elsewhere."""
    ref = "/root/reference/subprojects"
    gen = os.path.join(ROOT, "oracle", "_ref", "gen")
    if not os.path.isdir(ref) or not os.path.exists(os.path.join(RT129, "lib", "libgstvideo-1.0.so.0")):
        return OUT129 if os.path.exists(OUT129) else None
    os.makedirs(os.path.dirname(OUT129), exist_ok=True)
    srcs = [os.path.join(HERE, f) for f in ("gstamdplugin.c", "gstamdhipmemory.c", "gstamdhipbufferpool.c", "gstamdvideoconvertscale.c", "gstamdaudioresample.c", "gstamdaudioconvert.c", "gstamdcompositor.c", "gstamdhiptransfer.c", "gstamdvideotestsrc.c")]
    inc = ["-DHAVE_CONFIG_H", "-I" + gen, "-I%s/gstreamer" % ref, "-I%s/gstreamer/libs" % ref, "-I%s/gst-plugins-base/gst-libs" % ref,
           "-I" + os.path.join(gen, "gst/video"), "-I" + os.path.join(gen, "gst/audio"),
           "-I%s/include/glib-2.0" % CONDA, "-I%s/lib/glib-2.0/include" % CONDA]
    libdir = os.path.join(ROOT, "gstreamer_amd", "lib")
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-deprecated-declarations", "-o", OUT129] + srcs + inc + [
        "-L" + os.path.join(RT129, "lib"), "-Wl,-rpath,$ORIGIN/../../oracle/_ref/rt129/lib", "-L%s/lib" % CONDA, "-Wl,-rpath,%s/lib" % CONDA, "-L" + libdir,
        "-Wl,-rpath,$ORIGIN/../../gstreamer_amd/lib", "-lgstamddsp", "-lgstvideo-1.0", "-lgstaudio-1.0", "-lgstbase-1.0", "-lgstreamer-1.0", "-lgobject-2.0", "-lglib-2.0"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("plugin build against the 1.29 runtime failed:\n" + r.stdout[-4000:])
    tool = os.path.join(HERE, "tests", "launch129.c")
    if os.path.exists(tool):
        cmd = ["gcc", "-O2", "-Wall", "-Wno-deprecated-declarations", "-o", os.path.join(HERE, "tests", "launch129"), tool] + inc + [
            "-L" + os.path.join(RT129, "lib"), "-Wl,-rpath,$ORIGIN/../../oracle/_ref/rt129/lib", "-L%s/lib" % CONDA, "-Wl,-rpath,%s/lib" % CONDA,
            "-lgstvideo-1.0", "-lgstbase-1.0", "-lgstreamer-1.0", "-lgobject-2.0", "-lglib-2.0"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("launch129 build failed:\n" + r.stdout[-4000:])
    return OUT129


def check_against_reference_headers():
    """Second build target: every element source compiled (syntax / type check, -Werror=implicit-function-declaration) against the headers of
    the reference's OWN version (1.29: /root/reference + the generated gstconfig.h / enumtypes headers of oracle/ref_build.py) - the API
    subset the elements use must exist there with the same signatures.  Only where the reference tree is present (the build container)."""
    ref = "/root/reference/subprojects"
    gen = os.path.join(ROOT, "oracle", "_ref", "gen")
    if not os.path.isdir(ref) or not os.path.isdir(os.path.join(gen, "gst")):
        return None
    inc = ["-DHAVE_CONFIG_H", "-I" + gen, "-I%s/gstreamer" % ref, "-I%s/gstreamer/libs" % ref, "-I%s/gst-plugins-base/gst-libs" % ref,
           "-I%s/include/glib-2.0" % CONDA, "-I%s/lib/glib-2.0/include" % CONDA]
    checked = []
    for f in sorted(os.listdir(HERE)):
        if not f.endswith(".c"):
            continue
        r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wno-deprecated-declarations", "-Werror=implicit-function-declaration",
                            "-Werror=incompatible-pointer-types", os.path.join(HERE, f)] + inc, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s does not compile against the reference's 1.29 headers:\n%s" % (f, r.stdout[-3000:]))
        checked.append(f)
    return checked


def build_tools(inc):
    """Measurement / test programs that drive the elements through GstHarness (plugins/tests/)."""
    tdir = os.path.join(HERE, "tests")
    for name in ("bench_element", "live_props"):
        src = os.path.join(tdir, name + ".c")
        if not os.path.exists(src):
            continue
        cmd = ["gcc", "-O2", "-Wall", "-Wno-deprecated-declarations", "-Wl,--allow-shlib-undefined", "-o", os.path.join(tdir, name), src] + inc + [
            "-L%s/lib" % CONDA, "-Wl,-rpath,%s/lib" % CONDA, "-L" + HERE, "-Wl,-rpath,$ORIGIN/..", "-lgstamdhipdsp",
            "-L" + os.path.join(ROOT, "gstreamer_amd", "lib"), "-Wl,-rpath,$ORIGIN/../../gstreamer_amd/lib", "-lgstamddsp",
            "-lgstcheck-1.0", "-lgstvideo-1.0", "-lgstaudio-1.0", "-lgstbase-1.0", "-lgstreamer-1.0", "-lgobject-2.0", "-lglib-2.0"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("tool build failed:\n" + r.stdout)


if __name__ == "__main__":
    print(build())
