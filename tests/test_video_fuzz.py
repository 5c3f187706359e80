"""A fixed slice of scripts/fuzz_video.py: random format pairs / sizes / options, kernel bodies on the host emulator against the reference
(oracle/_ref).  Every plan is either refused or byte-exact; the refusals are counted so that the test cannot pass by refusing everything."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import fuzz_video  # noqa: E402


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_random_conversions_match_reference_or_are_refused(emu_lib, ref, seed):
    emu = fuzz_video.load_emu()
    rnd = random.Random(seed)
    count = {"ok": 0, "refused": 0, "bad": 0}
    bad = []
    for it in range(120):
        case = fuzz_video.random_case(rnd)
        verdict, text = fuzz_video.run_case(emu, ref, case, seed * 1000 + it)
        count[verdict] += 1
        if verdict == "bad":
            bad.append((case, text))
    assert not bad, bad[:5]
    assert count["ok"] >= 80, count
