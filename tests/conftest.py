import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# kernel arguments in HBM (a switch of the HIP runtime, read when it initialises): the launcher's decision - the product library does not
# touch the environment (gstreamer_amd/csrc/tuning.cpp); inherited by the gst-launch / bench_element child processes of the plugin tests
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# torch brings its own HIP runtime: a process that loads the product library (linked against /opt/rocm's) BEFORE torch ends up with two runtimes and the
# second one finds no device (/dev/kfd opens once per process).  The tests hand torch tensors to the library, so torch goes first in every test process.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """The product library; built here when sources are newer (hipcc cross-compiles without a GPU)."""
    from gstreamer_amd import build
    path = build.build(verbose=False)
    assert os.path.exists(path)
    return path


@pytest.fixture(scope="session")
def emu_lib():
    """Host emulator of the kernel bodies (tests/emu) - test infrastructure, built with g++."""
    import ctypes
    emu_dir = os.path.join(ROOT, "tests", "emu")
    so = os.path.join(emu_dir, "libgstamdemu.so")
    srcs = [os.path.join(emu_dir, f) for f in sorted(os.listdir(emu_dir)) if f.endswith(".cpp")]
    srcs.append(os.path.join(ROOT, "gstreamer_amd", "csrc", "planner.cpp"))
    srcs.append(os.path.join(ROOT, "gstreamer_amd", "csrc", "audio_taps.cpp"))
    deps = srcs + [os.path.join(ROOT, "gstreamer_amd", "csrc", f)
                   for f in os.listdir(os.path.join(ROOT, "gstreamer_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so] + srcs)
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def ref():
    """The reference's own code (oracle/_ref).  Skips when the prebuilt library is absent/unloadable."""
    from oracle import ref as r
    if not r.available():
        pytest.skip("oracle/_ref/libgstref.so not built (needs /root/reference; run oracle/ref_build.py)")
    try:
        r.lib()
    except OSError as e:
        pytest.skip("oracle/_ref not loadable here: %s" % e)
    return r


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    return torch.device("cuda:0")
