"""Builds the in-tree native library (C ABI + gfx950 kernels):  python -m gstreamer_amd.build

Output: gstreamer_amd/lib/libgstamddsp.so (git-ignored; travels to the GPU box with gpurun)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgstamddsp.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hip")))


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True, tuning=False, defines=(), suffix=None):
    """tuning=True adds -DGSTAMD_TUNING: the ablation switches (kernels that skip arithmetic on purpose, for profiling sessions)
    exist only in such a build - never commit / ship one; rebuild without it afterwards."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not tuning and not suffix and up_to_date():
        return LIB
    lib = os.path.join(LIBDIR, "libgstamddsp_tuning.so") if tuning else LIB      # a tuning build never replaces the product library
    objdir = os.path.join(LIBDIR, "tuning") if tuning else LIBDIR
    if suffix:                  # profiling variants with extra -D switches: libgstamddsp_<suffix>.so (select with GSTAMD_LIB_PATH)
        lib = os.path.join(LIBDIR, "libgstamddsp_%s.so" % suffix)
        objdir = os.path.join(LIBDIR, suffix)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + (["-DGSTAMD_TUNING"] if tuning else []) + ["-D" + d for d in defines] + ["-x", "hip", "-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,--no-undefined"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return lib


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    sfx = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--suffix=")]
    print(build(force="--force" in sys.argv, tuning="--tuning" in sys.argv, defines=defs, suffix=sfx[0] if sfx else None))
