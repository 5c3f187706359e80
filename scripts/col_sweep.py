"""Sweep of k_scale_col's form / geometry knobs on one bench config (GPU box):  python scripts/col_sweep.py [c3] > gpurun_out/col_sweep.log
Every line: the knobs, ms per frame by HIP events, fraction of the HBM peak."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
SETS = []
for opl in (1, 2):
    for waves in (2, 4, 8):
        SETS.append({"GSTAMD_COL_OPL": opl, "GSTAMD_COL_WAVES": waves})
SETS += [{"GSTAMD_NO_COL": 1}]
BATCHES = [int(b) for b in os.environ.get("COL_SWEEP_BATCHES", "1,4").split(",")]
for batch in BATCHES:
    for kn in SETS:
        env = dict(os.environ)
        env.update({k: str(v) for k, v in kn.items()})
        env["GSTAMD_COL_DEBUG"] = "1"
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--no-cpu-baseline", "--steps", "40", "--warmup", "8", "--batch", str(batch)]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        dbg = [l for l in r.stderr.splitlines() if l.startswith("k_scale_col")]
        if not line:
            print("FAILED", kn, batch, r.stderr[-400:], flush=True)
            continue
        j = json.loads(line[-1])
        fps = j["value"]
        print("batch %d %-70s us/frame %7.2f frac %.3f  %s" % (batch, json.dumps(kn), 1e6 / fps, j["roofline"]["frac"], dbg[0] if dbg else ""), flush=True)
