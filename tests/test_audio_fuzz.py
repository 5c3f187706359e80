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


class _EmuBackend:
    """the product's host bookkeeping (audio_taps.cpp) + the FIR kernel bodies on the host emulator: one stream at a time (no many-launch here)"""

    def __init__(self, emu):
        import ctypes as C
        self.C, self.E = C, emu
        emu.emu_audio_new.restype = C.c_void_p
        emu.emu_audio_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(A.ResamplerOptions), C.POINTER(C.c_int), C.c_char_p, C.c_int]
        emu.emu_audio_get_out_frames.restype = C.c_size_t
        emu.emu_audio_get_out_frames.argtypes = [C.c_void_p, C.c_size_t]
        emu.emu_audio_resample.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        emu.emu_audio_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(A.ResamplerOptions)]
        emu.emu_audio_free.argtypes = [C.c_void_p]
        emu.emu_audio_stale_ahead.argtypes = [C.c_void_p]

    def new(self, fmt, ch, ir, orr, method, opts):
        st = self.C.c_int(0)
        h = self.E.emu_audio_new(A.METHODS[method], 0, A.FORMATS[fmt], ch, ir, orr, self.C.byref(opts), self.C.byref(st), None, 0)
        assert h, st.value
        return h

    def update(self, h, ir, orr, opts):
        assert self.E.emu_audio_update(h, ir, orr, self.C.byref(opts) if opts is not None else None) == 0

    def get_out_frames(self, h, k):
        return self.E.emu_audio_get_out_frames(h, k)

    def max_latency(self, h):
        return None

    def announced(self, h):
        return self.E.emu_audio_stale_ahead(h) > 0

    def run(self, hs, srcs, nin, nout, ch, dt, many):
        outs = []
        for h, src, k, no in zip(hs, srcs, nin, nout):
            o = np.zeros((max(no, 1), ch), dt)
            self.E.emu_audio_resample(h, src.ctypes.data if src is not None else None, k, o.ctypes.data, no)
            outs.append(o[:no].reshape(-1))
        return outs

    def free(self, h):
        self.E.emu_audio_free(h)


class _HipBackend:
    def __init__(self, gpu):
        self.gpu = gpu

    def new(self, fmt, ch, ir, orr, method, opts):
        return A.AudioResampler(fmt, ch, ir, orr, method, opts)

    def update(self, h, ir, orr, opts):
        h.update(ir, orr, opts)

    def get_out_frames(self, h, k):
        return h.get_out_frames(k)

    def max_latency(self, h):
        return h.get_max_latency()

    def announced(self, h):
        return h.divergence() != ""

    def run(self, hs, srcs, nin, nout, ch, dt, many):
        import torch
        tdt = getattr(torch, np.dtype(dt).name)
        ins = [None if s is None else torch.from_numpy(np.ascontiguousarray(s)).to(self.gpu) for s in srcs]
        outs = [torch.zeros((max(no, 1), ch), dtype=tdt, device=self.gpu) for no in nout]
        if many:
            A.resample_many(hs, ins, nin, outs, nout)
        else:
            for h, i, k, o, no in zip(hs, ins, nin, outs, nout):
                h.resample(i, k, o, no)
        torch.cuda.synchronize()
        return [o[:no].cpu().numpy().reshape(-1) for o, no in zip(outs, nout)]

    def free(self, h):
        h.free()


def _run_group(seed, ref, be):
    """one draw: the group's streams through `be` and through one reference resampler each, every buffer compared bit for bit
    -> (streams, rounds, buffers counted instead of compared)"""
    rnd = random.Random(seed)
    fmt, ch, method, streams = _group(rnd)
    dt = cases.AUDIO_DTYPES[fmt]
    rs, rrs = [], []
    for s in streams:
        rs.append(be.new(fmt, ch, s["ir"], s["orr"], method, A.options(method, s["quality"], s["ir"], s["orr"], **_filt_kw(s["filt"]))))
        rrs.append(ref.AudioResampler(fmt, ch, s["ir"], s["orr"], method=method, quality=s["quality"], **_filt_kw(s["filt"])))
    n = len(streams)
    sig = [cases.audio_buffer(fmt, ch, 12000, seed * 100 + i) for i in range(n)]
    pos = [0] * n
    announced = [0]
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
                        be.update(rs[i], nr[0], nr[1], A.options(method, nq, nr[0], nr[1], **_filt_kw(s["filt"])))
                        rrs[i].update(nr[0], nr[1], quality=nq, **_filt_kw(s["filt"]))
                        s["ir"], s["orr"], s["quality"] = nr[0], nr[1], nq
                    elif kind == "rates":
                        be.update(rs[i], nr[0], nr[1], None)
                        rrs[i].update(nr[0], nr[1])
                        s["ir"], s["orr"] = nr
                    else:
                        be.update(rs[i], 0, 0, A.options(method, nq, s["ir"], s["orr"], **_filt_kw(s["filt"])))
                        rrs[i].update(0, 0, quality=nq, q_rates=(s["ir"], s["orr"]), **_filt_kw(s["filt"]))
                        s["quality"] = nq
        same_size = rnd.random() < 0.5
        size0 = rnd.choice([1, 37, 256, 1024, 1024, 2000])
        srcs, nin, nout, exps = [], [], [], []
        for i in range(n):
            if drain:
                k = rrs[i].get_max_latency()
                ml = be.max_latency(rs[i])
                assert ml is None or ml == k, (seed, i, "max_latency")
                src = None
            else:
                k = size0 if same_size else rnd.choice([1, 37, 256, 1024, 1500])
                k = min(k, 12000 - pos[i])
                src = sig[i][pos[i]:pos[i] + k]
                pos[i] += k
            no = rrs[i].get_out_frames(k)
            assert be.get_out_frames(rs[i], k) == no, (seed, rd, i, "out_frames", k)
            exps.append(rrs[i].resample(src, in_frames=k, out_frames=no))
            srcs.append(src)
            nin.append(k)
            nout.append(no)
        # (an update that enlarged the filter past the history: the reference's next outputs depend on stale contents of its sample buffer - announced by
        # gstamd_audio_resampler_divergence until those frames have left the filter window; such buffers are counted, not compared)
        skip = [be.announced(r) for r in rs]
        gots = be.run(rs, srcs, nin, nout, ch, dt, many=rnd.random() < 0.8)
        for i in range(n):
            if skip[i]:
                announced[0] += 1
                continue
            want = np.asarray(exps[i], dtype=dt).reshape(-1)
            assert gots[i].tobytes() == want.tobytes(), (seed, rd, i, fmt, ch, method, streams[i], nin[i], int((gots[i] != want).sum()))
    for r in rs:
        be.free(r)
    return n, rounds + 1, announced[0]


@pytest.mark.parametrize("seed", range(310000, 310030))
def test_random_stream_groups_on_host_match_reference(emu_lib, ref, seed):
    """the same draws through the product's host bookkeeping and the FIR kernel bodies on the host emulator (stream by stream)"""
    _run_group(seed, ref, _EmuBackend(emu_lib))


def test_share_of_announced_buffers_on_host_stays_small(emu_lib, ref):
    """over 60 groups the buffers that are counted instead of compared (an update enlarged the filter past the history) stay a small share"""
    be = _EmuBackend(emu_lib)
    buffers = skipped = 0
    for seed in range(311000, 311060):
        n, rounds, ann = _run_group(seed, ref, be)
        buffers += n * rounds
        skipped += ann
    assert skipped <= 0.05 * buffers, (skipped, buffers)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_hip_random_stream_groups_match_reference(native_lib, gpu, ref, seed):
    n, rounds, ann = _run_group(seed, ref, _HipBackend(gpu))
    if os.environ.get("GSTAMD_FUZZ_TALLY"):
        import json
        with open(os.environ["GSTAMD_FUZZ_TALLY"], "a") as f:
            f.write(json.dumps(dict(seed=seed, streams=n, buffers=n * rounds, announced_buffers=ann)) + "\n")
