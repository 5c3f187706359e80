"""Random resampler STREAMS on the device against the reference (oracle/_ref), 0 ULP: groups of 1 .. 70 streams through
gstamd_audio_resampler_resample_many (one launch where the streams share a filter; strays, duplicates of a rate pair with another quality and
interpolated-mode streams take the one-by-one path inside the same call), buffers of uneven sizes incl. 1 frame and the drain, and
gst_audio_resampler_update events in mid-stream (new rates with new options, new rates keeping the old filter design, a new quality only) -
audio-resampler.c:1503 (update), :1750 (resample).  Every stream has its own reference resampler fed the same buffers.
GSTAMD_AUDIO_SEEDS="300000-300249" runs 250 seeds (scripts/gpu_fuzz_all.sh: >= 5000 streams)."""
import os
import random

import numpy as np
import pytest

import cases
from gstreamer_amd import audio as A

RATES = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000]


def _seed_list(spec):
    out = []
    for part in spec.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


SEEDS = _seed_list(os.environ.get("GSTAMD_AUDIO_SEEDS", "300000-300007"))


def _group(rnd):
    """the streams of one draw: most share format / channels / rates / method / quality (one launch), a few differ"""
    fmt = rnd.choice(["F32LE", "F32LE", "F64LE", "S16LE", "S32LE"])
    ch = rnd.choice([1, 2, 2, 3, 6])
    ir, orr = rnd.choice(RATES), rnd.choice(RATES)
    method = rnd.choice(["kaiser", "kaiser", "kaiser", "blackman-nuttall", "cubic", "linear", "nearest"])
    quality = rnd.randint(0, 10)
    filt = rnd.choice([None, None, None, ("interpolated", "cubic"), ("interpolated", "linear"), ("full", None)])
    n = rnd.choice([1, 3, 8, 20, 20, 40, 70])
    streams = []
    for i in range(n):
        s = dict(ir=ir, orr=orr, quality=quality, filt=filt)
        if rnd.random() < 0.12:
            s["orr"] = rnd.choice(RATES)          # a stray: another rate pair
        if rnd.random() < 0.08:
            s["quality"] = rnd.randint(0, 10)     # the same rates, another filter
        streams.append(s)
    return fmt, ch, method, streams


def _filt_kw(f):
    if not f:
        return {}
    return {k: v for k, v in (("filter_mode", f[0]), ("filter_interpolation", f[1])) if v}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_hip_random_stream_groups_match_reference(native_lib, gpu, ref, seed):
    import torch
    rnd = random.Random(seed)
    fmt, ch, method, streams = _group(rnd)
    dt = cases.AUDIO_DTYPES[fmt]
    tdt = {"F32LE": torch.float32, "S16LE": torch.int16, "F64LE": torch.float64, "S32LE": torch.int32}[fmt]
    rs, rrs = [], []
    for s in streams:
        rs.append(A.AudioResampler(fmt, ch, s["ir"], s["orr"], method, A.options(method, s["quality"], s["ir"], s["orr"], **_filt_kw(s["filt"]))))
        rrs.append(ref.AudioResampler(fmt, ch, s["ir"], s["orr"], method=method, quality=s["quality"], **_filt_kw(s["filt"])))
    n = len(streams)
    sig = [cases.audio_buffer(fmt, ch, 12000, seed * 100 + i) for i in range(n)]
    d_sig = [torch.from_numpy(x).to(gpu) for x in sig]
    pos = [0] * n
    rounds = rnd.randint(3, 7)
    for rd in range(rounds + 1):
        drain = rd == rounds
        if not drain and rd and rnd.random() < 0.35:
            # gst_audio_resampler_update for a subset: new rates with options, new rates with NULL options, or a new quality at the same rates
            kind = rnd.choice(["rates+options", "rates", "quality"])
            nr = (rnd.choice(RATES), rnd.choice(RATES))
            nq = rnd.randint(0, 10)
            for i in range(n):
                if rnd.random() < 0.6:
                    s = streams[i]
                    if kind == "rates+options":
                        rs[i].update(nr[0], nr[1], A.options(method, nq, nr[0], nr[1], **_filt_kw(s["filt"])))
                        rrs[i].update(nr[0], nr[1], quality=nq, **_filt_kw(s["filt"]))
                        s["ir"], s["orr"], s["quality"] = nr[0], nr[1], nq
                    elif kind == "rates":
                        rs[i].update(nr[0], nr[1], None)
                        rrs[i].update(nr[0], nr[1])
                        s["ir"], s["orr"] = nr
                    else:
                        rs[i].update(0, 0, A.options(method, nq, s["ir"], s["orr"], **_filt_kw(s["filt"])))
                        rrs[i].update(0, 0, quality=nq, q_rates=(s["ir"], s["orr"]), **_filt_kw(s["filt"]))
                        s["quality"] = nq
        same_size = rnd.random() < 0.5
        size0 = rnd.choice([1, 37, 256, 1024, 1024, 2000])
        ins, nin, outs, nout, exps = [], [], [], [], []
        for i in range(n):
            if drain:
                k = rrs[i].get_max_latency()
                assert rs[i].get_max_latency() == k, (seed, i, "max_latency")
                src, d_src = None, None
            else:
                k = size0 if same_size else rnd.choice([1, 37, 256, 1024, 1500])
                k = min(k, 12000 - pos[i])
                src, d_src = sig[i][pos[i]:pos[i] + k], d_sig[i][pos[i]:pos[i] + k]
                pos[i] += k
            no = rrs[i].get_out_frames(k)
            assert rs[i].get_out_frames(k) == no, (seed, rd, i, "out_frames", k)
            exps.append(rrs[i].resample(src, in_frames=k, out_frames=no))
            ins.append(d_src)
            nin.append(k)
            outs.append(torch.zeros((max(no, 1), ch), dtype=tdt, device=gpu))
            nout.append(no)
        if rnd.random() < 0.8:
            A.resample_many(rs, ins, nin, outs, nout)
        else:
            for i in range(n):
                rs[i].resample(ins[i], nin[i], outs[i], nout[i])
        torch.cuda.synchronize()
        for i in range(n):
            got = outs[i][:nout[i]].cpu().numpy().reshape(-1)
            want = np.asarray(exps[i], dtype=dt).reshape(-1)
            assert got.tobytes() == want.tobytes(), (seed, rd, i, fmt, ch, method, streams[i], nin[i], int((got != want).sum()))
    for r in rs:
        r.free()
    if os.environ.get("GSTAMD_FUZZ_TALLY"):
        import json
        with open(os.environ["GSTAMD_FUZZ_TALLY"], "a") as f:
            f.write(json.dumps(dict(seed=seed, streams=n, rounds=rounds + 1)) + "\n")
