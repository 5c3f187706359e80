#!/usr/bin/env python3
"""Generates tests/golden/video_golden.json from the REFERENCE itself (oracle/_ref, built from
/root/reference by oracle/ref_build.py).  Run in the build container:  python tests/golden/make_golden.py

For every case of tests/cases.py it stores the sha256 of the reference's output for the seeded
synthetic input (inputs are regenerated from the seed at test time, so only hashes are stored),
plus the first 64 output bytes for eyeballing."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle import ref  # noqa: E402


def main():
    out = {}
    for i, (name, ifmt, w, h, ofmt, ow, oh, cfg, col, site, pattern) in enumerate(cases.VIDEO_CASES):
        ii = ref.video_info(ifmt, w, h)
        src = cases.frame_bytes(ii["size"], pattern, cases.case_seed(name), w)
        col, ocol = cases.split_colorimetry(col)
        rc = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, out_colorimetry=ocol,
                                config=cases.ref_config_string(ref, cfg))
        dst = rc.frame(src)
        out[name] = dict(sha256=cases.video_digest(name, dst), head=[int(x) for x in dst[:64]], in_sha256=cases.sha(src), size=int(dst.size))
        print(name, out[name]["sha256"][:16])
    with open(os.path.join(ROOT, "tests", "golden", "video_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def audio():
    out = {}
    for case in cases.AUDIO_CASES:
        name, fmt, ch, ir, orr, method, quality, bufs = case
        rr = ref.AudioResampler(fmt, ch, ir, orr, method=method, quality=quality, **cases.audio_filter_kwargs(name))
        chunks, counts = [], []
        for i, n in enumerate(list(bufs) + [None]):
            data = None if n is None else cases.audio_buffer(fmt, ch, n, cases.case_seed(name) + i)
            n_in = rr.get_max_latency() if n is None else n
            no = rr.get_out_frames(n_in)
            chunks.append(rr.resample(data, in_frames=n_in, out_frames=no).reshape(-1))
            counts.append(int(no))
        full = __import__("numpy").concatenate(chunks)
        out[name] = dict(sha256=cases.sha(full), out_frames=counts, head=[float(x) for x in full[:8]])
        print(name, out[name]["sha256"][:16], counts[:4])
    for case in cases.AUDIO_UPDATE_CASES:
        name, fmt, ch, ir, orr, method, quality, script = case
        rr = ref.AudioResampler(fmt, ch, ir, orr, method=method, quality=quality, **cases.audio_filter_kwargs(name))
        counts = []

        def do_update(item):
            raw = item.get("raw", (item["in_rate"], item["out_rate"]))
            assert rr.update(raw[0], raw[1], quality=item.get("quality"), filter_mode=item.get("filter_mode"),
                             filter_interpolation=item.get("filter_interpolation"), q_rates=(item["in_rate"], item["out_rate"]))

        def do_resample(data, n_in):
            no = rr.get_out_frames(n_in)
            counts.append(int(no))
            return rr.resample(data, in_frames=n_in, out_frames=no)

        full = cases.audio_update_stream(case, do_update, do_resample, rr.get_max_latency)
        out[name] = dict(sha256=cases.sha(full), out_frames=counts, head=[float(x) for x in full[:8]])
        print(name, out[name]["sha256"][:16], counts)
    with open(os.path.join(ROOT, "tests", "golden", "audio_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    audio()
    main()
