"""Per-buffer cost breakdown of the videoconvertscale element on HBM buffers (plugins/tests/bench_element with GSTAMD_ELEMENT_STATS=1):
wait (maps + stream waits), convert (the library call = the launch), mark (ticket / event), and the harness's share.
    python scripts/element_stats.py > gpurun_out/element_stats.log"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "plugins"))
from config1_e2e import env_for     # noqa: E402
import build as plugin_build        # noqa: E402

plugin_build.build()
env = env_for(tempfile.mkdtemp(prefix="elstats_"))
env["GSTAMD_ELEMENT_STATS"] = "1"
exe = os.path.join(ROOT, "plugins", "tests", "bench_element")
for args in (["NV12", 3840, 2160, "BGRA", 3840, 2160, 600, 1], ["NV12", 3840, 2160, "BGRA", 3840, 2160, 600, 3],
             ["NV12", 1920, 1080, "BGRA", 1920, 1080, 1500, 1], ["NV12", 1920, 1080, "BGRA", 1920, 1080, 1500, 3]) + tuple(a.split() for a in sys.argv[1:]):
    r = subprocess.run([exe] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    print(" ".join(str(a) for a in args))
    print(r.stdout.strip()[-700:])
    print(r.stderr.strip()[-900:])
    sys.stdout.flush()
