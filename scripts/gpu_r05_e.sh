#!/bin/bash
# round 5: element per-buffer sweep over hip-streams (device kernel arguments by the library's load-time default)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python - <<'PY'
import subprocess, json, sys, os
sys.path.insert(0, os.getcwd())
import bench
env = bench.element_env()
exe = "plugins/tests/bench_element"
for streams in (1, 2, 3, 4, 6, 8):
    for size in ((3840, 2160, 640), (1920, 1080, 1600)):
        r = subprocess.run([exe, "NV12", str(size[0]), str(size[1]), "BGRA", str(size[0]), str(size[1]), str(size[2]), str(streams), "bilinear", "1", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if js:
            d = json.loads(js[-1]); print("streams", streams, size[:2], d["us_per_frame"], round(d["algorithmic_gb_per_s"]/8000, 3))
        else:
            print("streams", streams, "failed", r.stderr[-300:])
PY
