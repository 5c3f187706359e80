#!/usr/bin/env python3
"""Sums the per-seed lines the device fuzz tests (test_video_fuzz / test_compositor_fuzz / test_audio_fuzz) write when GSTAMD_FUZZ_TALLY names a
file.  Video: draws compared with the one-step reference ("ok", 150 draws a seed), announced draws checked stage by stage / on the bytes the reference's
undefined part cannot reach / compared with nothing, per class.  Compositor: scenes per entry and pad-count bracket.  Audio: streams and rounds.
python scripts/fuzz_tally.py gpurun_out/fuzz_tally.jsonl"""
import json
import sys

tot, classes, seeds = {}, {}, 0
for line in open(sys.argv[1]):
    j = json.loads(line)
    seeds += 1
    for k, v in j.items():
        if k == "classes":
            for c, n in v.items():
                classes[c] = classes.get(c, 0) + n
        elif k != "seed":
            tot[k] = tot.get(k, 0) + v
print("%d seeds: %s" % (seeds, tot))
for c in sorted(classes):
    print("  %6d  %s" % (classes[c], c))
