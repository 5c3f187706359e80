#!/usr/bin/env python3
"""Adds the cases of tests/cases.py that tests/golden/video_golden.json does not hold yet (same recipe as make_golden.py: the REFERENCE's output,
oracle/_ref, for the seeded input); existing entries stay as they are.   python tests/golden/add_missing.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import ref  # noqa: E402

path = os.path.join(ROOT, "tests", "golden", "video_golden.json")
out = json.load(open(path))
n = 0
for name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern in cases.VIDEO_CASES:
    if name in out:
        continue
    ii = ref.video_info(ifmt, w, h)
    src = cases.frame_bytes(ii["size"], pattern, cases.case_seed(name), w)
    col, ocol = cases.split_colorimetry(col)
    dst = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, out_colorimetry=ocol, config=cases.ref_config_string(ref, cfg)).frame(src)
    out[name] = dict(sha256=cases.video_digest(name, dst), head=[int(x) for x in dst[:64]], in_sha256=cases.sha(src), size=int(dst.size))
    print(name, out[name]["sha256"][:16])
    n += 1
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("added", n)
