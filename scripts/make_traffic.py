#!/usr/bin/env python3
"""profiles/traffic_<cfg>.json from the PMC summaries of scripts/gpu_profiles.sh (gpurun_out/prof/pmc_<cfg>.json) and the bench line
of the same run (bench_<cfg>.json): bytes per launch of the dominant kernel, with the FETCH_SIZE correction of
MI355X_MICROARCH.md as calibrated in profiles/r02/fetchcal.json.   usage: make_traffic.py <prof dir> <date note> [cfg ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORRECTION = ("FETCH_SIZE x2: on gfx950 a 128-byte fabric read request is tallied as 64 bytes (MI355X_MICROARCH.md); calibrated here on known "
              "byte counts for 4-, 8- and 16-byte coalesced loads per lane: reported / known = 0.5000 in all three (profiles/r02/fetchcal.json). "
              "WRITE_SIZE x1 (1.000 for 4- and 16-byte, plain and nontemporal stores). Loads that ask for only 64 bytes of a line per request "
              "are tallied at FULL size (fetchcal k_read_c2like: 0.667), so x2 is an UPPER bound on the read side where such requests occur")
prof, note = sys.argv[1], sys.argv[2]
for cfg in sys.argv[3:] or ["c1", "c2", "c3", "c4", "c5"]:
    pmc = json.load(open(os.path.join(prof, "pmc_%s.json" % cfg)))
    bench = None
    for line in open(os.path.join(prof, "bench_%s.json" % cfg)):
        if line.startswith("{"):
            bench = json.loads(line)
    kernels = {k: v for k, v in pmc.get("kernels", {}).items() if v.get("FETCH_SIZE_KB_avg") is not None and v.get("WRITE_SIZE_KB_avg") is not None
               and not k.startswith("__amd") and "at::native" not in k}
    if not kernels or bench is None:
        print(cfg, "no counters / bench line")
        continue
    # dominant kernel(s): everything of ours that ran once per launch of the timed loop (the largest launch count)
    top = max(v["launches"] for v in kernels.values())
    mine = {k: v for k, v in kernels.items() if v["launches"] >= top * 0.5}
    fetch = sum(v["FETCH_SIZE_KB_avg"] * 1024 for v in mine.values())
    write = sum(v["WRITE_SIZE_KB_avg"] * 1024 for v in mine.values())
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    out = {"config": cfg, "kernel": " + ".join(sorted(k.split("(")[0].replace("void ", "") for k in mine)),
           "measured": "%s, rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --config %s --no-cpu-baseline` "
                       "(scripts/gpu_profiles.sh)" % (note, cfg),
           "FETCH_SIZE_bytes_per_launch_raw": int(fetch), "WRITE_SIZE_bytes_per_launch_raw": int(write), "correction": CORRECTION,
           "read_bytes_per_launch": int(2 * fetch), "write_bytes_per_launch": int(write), "hbm_bytes_per_launch": int(2 * fetch + write),
           "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((2 * fetch + write) / alg, 3),
           "frames_per_launch": bench["config"].get("frames_per_launch", 1),
           "note": "FETCH_SIZE counts L2 -> fabric requests: Infinity Cache hits are included, so this is L2-miss traffic, an upper bound on HBM bytes"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_%s.json" % cfg), "w"), indent=1)
    print(cfg, out["kernel"][:60], "traffic/alg", out["traffic_over_algorithmic"])
