"""Draws FIRST..LAST of a GPU fuzz seed in one process, every draw checked; then the failing ones again on their own: history dependence on the device.
python scripts/gpu_fuzz_seq.py SEED [LAST]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch                              # noqa: E402
import cases                              # noqa: E402
import fuzz_video                         # noqa: E402
from gstreamer_amd import video as V      # noqa: E402
from oracle import ref                    # noqa: E402

seed = int(sys.argv[1])
last = int(sys.argv[2]) if len(sys.argv) > 2 else 149
gpu = torch.device("cuda:0")


def run(case, it, poison=None):
    ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
    ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site)
    oi = V.video_info(ofmt, ow, oh)
    try:
        conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    except V.GstAmdError:
        return None
    if conv.divergence() != "":
        conv.free()
        return None
    src = cases.frame_bytes(int(ii.size), "random", seed * 1000 + it, w)
    if poison is not None:                # recycled device memory with a known pattern in it
        junk = [torch.full((1 << 20,), poison, dtype=torch.uint8, device=gpu) for _ in range(8)]
        del junk
    d_src = torch.from_numpy(src).to(gpu)
    d_dst = torch.zeros(int(oi.size), dtype=torch.uint8, device=gpu)
    if os.environ.get("DEBUG_DRAW") and it == int(os.environ["DEBUG_DRAW"]):
        os.environ["GSTAMD_DEEP_DEBUG"] = "1"
        with V.tuning(GSTAMD_DEEP_DEBUG=1):
            conv.frame(d_src, d_dst)
    else:
        conv.frame(d_src, d_dst)
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    desc = conv.describe()
    conv.free()
    if os.environ.get("FUZZ_HASHES"):
        want = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, config=cases.ref_config_string(ref, cfg)).frame(src)
        desc += " | got %s want %s" % (cases.sha(got)[:10], cases.sha(want)[:10])
        import numpy as np
        d = np.nonzero(got != want)[0]
        if len(d):
            st = int(oi.stride[0])
            desc += " | %d bytes, rows %s cols %s first %s" % (len(d), sorted(set(int(i) // st for i in d))[:20], sorted(set(int(i) % st for i in d))[:50],
                                                             [(int(i) // st, int(i) % st, int(got[i]), int(want[i])) for i in d[:8]])
    return fuzz_video.matches_reference(ref, case, src, got, oi)[0], desc


rnd = random.Random(seed)
rects = random.Random(seed + 77) if seed >= 700 else None
draws = [fuzz_video.random_case(rnd, rects) for _ in range(last + 1)]
bad = []
for it, case in enumerate(draws):
    r = run(case, it)
    if it == last and r is not None:
        print("last draw:", r, flush=True)
    if r is not None and not r[0]:
        bad.append(it)
        print("BAD in sequence:", it, case, r[1], flush=True)
for it in bad:
    for poison in (None, 0, 0xff, 0x5a):
        print("again on its own, poison", poison, run(draws[it], it, poison), flush=True)
