"""GPU parity tests of the polyphase FIR (-m gpu): HIP kernels through the C ABI vs the reference's golden
hashes and, when loadable, vs oracle/_ref sample for sample.  The bar is bit-exact (0 ULP): the kernel keeps
the reference's C summation order and is compiled without FMA contraction."""
import json
import os

import numpy as np
import pytest

import cases
from gstreamer_amd import audio as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "audio_golden.json")))
MAX_ULP = 0     # north_star allows 1 ULP for float resampling; we hold 0


def gpu_stream(gpu, case):
    import torch
    name, fmt, ch, ir, orr, method, quality, bufs = case
    dt = cases.AUDIO_DTYPES[fmt]
    r = A.AudioResampler(fmt, ch, ir, orr, method, A.options(method, quality, ir, orr, **cases.audio_filter_kwargs(name)))
    chunks, counts = [], []
    for i, n in enumerate(list(bufs) + [None]):
        if n is None:
            n, d_in = r.get_max_latency(), None
        else:
            d_in = torch.from_numpy(cases.audio_buffer(fmt, ch, n, cases.case_seed(name) + i)).to(gpu)
        no = r.get_out_frames(n)
        d_out = torch.zeros((max(no, 1), ch), dtype=getattr(torch, np.dtype(dt).name), device=gpu)
        r.resample(d_in, n, d_out, no)
        torch.cuda.synchronize()
        chunks.append(d_out[:no].cpu().numpy().reshape(-1))
        counts.append(int(no))
    r.free()
    return np.concatenate(chunks), counts


@pytest.mark.parametrize("case", cases.AUDIO_CASES, ids=lambda c: c[0])
def test_hip_fir_matches_reference_golden(native_lib, gpu, case):
    out, counts = gpu_stream(gpu, case)
    assert counts == GOLDEN[case[0]]["out_frames"]
    assert cases.sha(out) == GOLDEN[case[0]]["sha256"], (list(out[:8]), GOLDEN[case[0]]["head"])


@pytest.mark.parametrize("case", cases.AUDIO_UPDATE_CASES, ids=lambda c: c[0])
def test_hip_update_streams_match_reference_golden(native_lib, gpu, case):
    """gst_audio_resampler_update in mid-stream (rates and / or options; NULL options keep the old filter design): the device
    table is replaced and the HBM history shifted - output identical to the reference's for the whole stream."""
    import torch
    name, fmt, ch, ir, orr, method, quality, script = case
    dt = cases.AUDIO_DTYPES[fmt]
    r = A.AudioResampler(fmt, ch, ir, orr, method, A.options(method, quality, ir, orr, **cases.audio_filter_kwargs(name)))
    counts = []

    def do_update(item):
        raw = item.get("raw", (item["in_rate"], item["out_rate"]))
        uo = None
        if cases.audio_update_has_options(item):
            kw = {k: item[k] for k in ("filter_mode", "filter_interpolation") if k in item}
            uo = A.options(method, item.get("quality"), item["in_rate"], item["out_rate"], **kw)
        r.update(raw[0], raw[1], uo)

    def do_resample(data, n_in):
        d_in = torch.from_numpy(data).to(gpu) if data is not None else None
        no = r.get_out_frames(n_in)
        counts.append(int(no))
        d_out = torch.zeros((max(no, 1), ch), dtype=getattr(torch, np.dtype(dt).name), device=gpu)
        r.resample(d_in, n_in, d_out, no)
        torch.cuda.synchronize()
        return d_out[:no].cpu().numpy()

    out = cases.audio_update_stream(case, do_update, do_resample, r.get_max_latency)
    r.free()
    assert counts == GOLDEN[name]["out_frames"]
    assert cases.sha(out) == GOLDEN[name]["sha256"], (list(out[:8]), GOLDEN[name]["head"])


@pytest.mark.parametrize("layout", [(True, True), (True, False), (False, True)], ids=["planar_planar", "planar_in", "planar_out"])
@pytest.mark.parametrize("name", ["f32_48k_44k1_q4_stereo", "f32_6ch_cubic", "s16_48k_44k1_q4", "f32_8k_16k_gappy", "s32_interp_cubic_48k_32k"])
def test_hip_fir_non_interleaved_layouts(native_lib, gpu, name, layout):
    """Non-interleaved input / output (GstAudioResamplerFlags): same samples as the interleaved golden, planes [channels][frames];
    the planar-planar leg goes through gstamd_audio_resampler_resample_planes with one pointer per plane."""
    import torch
    case = next(c for c in cases.AUDIO_CASES if c[0] == name)
    _, fmt, ch, ir, orr, method, quality, bufs = case
    in_planar, out_planar = layout
    dt = cases.AUDIO_DTYPES[fmt]
    tdt = getattr(torch, np.dtype(dt).name)
    r = A.AudioResampler(fmt, ch, ir, orr, method, A.options(method, quality, ir, orr, **cases.audio_filter_kwargs(name)),
                         in_planar=in_planar, out_planar=out_planar)
    chunks = []
    for i, n in enumerate(list(bufs) + [None]):
        if n is None:
            n, d_in = r.get_max_latency(), None
        else:
            a = cases.audio_buffer(fmt, ch, n, cases.case_seed(name) + i)
            d_in = torch.from_numpy(np.ascontiguousarray(a.T) if in_planar else a).to(gpu)
        no = r.get_out_frames(n)
        d_out = torch.zeros((ch, max(no, 1)) if out_planar else (max(no, 1), ch), dtype=tdt, device=gpu)
        if in_planar and out_planar and no > 0:
            # planes with a pitch of their own: rows of wider buffers
            wide_in = None
            if d_in is not None:
                wide_in = torch.zeros((ch, n + 7), dtype=tdt, device=gpu)
                wide_in[:, :n] = d_in
            wide_out = torch.zeros((ch, no + 5), dtype=tdt, device=gpu)
            r.resample_planes(None if wide_in is None else [wide_in[c] for c in range(ch)], n, [wide_out[c] for c in range(ch)], no)
            d_out = wide_out[:, :no]
        else:
            if out_planar:
                d_out = torch.zeros((ch, no), dtype=tdt, device=gpu) if no > 0 else d_out
            r.resample(d_in, n, d_out, no)
        torch.cuda.synchronize()
        o = d_out.cpu().numpy()
        chunks.append((o.T if out_planar else o[:no]).reshape(-1) if no > 0 else np.zeros(0, dt))
    r.free()
    assert cases.sha(np.concatenate(chunks)) == GOLDEN[name]["sha256"]


def test_hip_fir_c4_ten_seconds_bitwise(native_lib, gpu, ref):
    """BASELINE config 4's audio leg: 2-ch F32, 10 s of noise + a 997 Hz sine block,
    48000 -> 44100, quality 4, buffers of 1024 frames; compared sample by sample with the reference."""
    import torch
    n_total, ch = 480000, 2
    sig = cases.audio_buffer("F32LE", ch, n_total, 4242)
    t = np.arange(48000, dtype=np.float64) / 48000.0
    sig[96000:144000, :] = (0.5 * np.sin(2 * np.pi * 997.0 * t)).astype(np.float32)[:, None]
    rr = ref.AudioResampler("F32LE", ch, 48000, 44100, quality=4)
    r = A.AudioResampler("F32LE", ch, 48000, 44100, "kaiser", A.options("kaiser", 4, 48000, 44100))
    d_sig = torch.from_numpy(sig).to(gpu)
    exp, got = [], []
    for off in range(0, n_total, 1024):
        n = min(1024, n_total - off)
        no = rr.get_out_frames(n)
        assert r.get_out_frames(n) == no
        exp.append(rr.resample(sig[off:off + n], in_frames=n, out_frames=no))
        d_out = torch.zeros((no, ch), dtype=torch.float32, device=gpu)
        r.resample(d_sig[off:off + n], n, d_out, no)
        got.append(d_out)
    torch.cuda.synchronize()
    exp = np.concatenate(exp)
    got = torch.cat(got).cpu().numpy()
    assert exp.shape == got.shape and exp.shape[0] in (440964, 440965, 440966, 441000 - 35, 441000 - 36, 440968, 440967)
    ai, bi = exp.view(np.int32).astype(np.int64), got.view(np.int32).astype(np.int64)
    assert int(np.abs(ai - bi).max()) <= MAX_ULP and (exp == got).all()
    r.free()


def test_one_big_buffer_equals_many_small(native_lib, gpu):
    """Streaming property at full size (no reference needed): any segmentation of the same input stream
    gives the same output stream - history hand-over between calls is exact."""
    import torch
    n_total, ch = 1 << 20, 2
    sig = torch.from_numpy(cases.audio_buffer("F32LE", ch, n_total, 77)).to(gpu)

    def run(chunk):
        r = A.AudioResampler("F32LE", ch, 48000, 44100, "kaiser", None)
        outs = []
        for off in range(0, n_total, chunk):
            n = min(chunk, n_total - off)
            no = r.get_out_frames(n)
            o = torch.zeros((max(no, 1), ch), dtype=torch.float32, device=gpu)
            r.resample(sig[off:off + n], n, o, no)
            outs.append(o[:no])
        torch.cuda.synchronize()
        r.free()
        return torch.cat(outs)

    a, b, c = run(n_total), run(4096), run(1000)
    assert a.shape == b.shape == c.shape and torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("fmt,ch", [("F32LE", 2), ("S16LE", 1), ("F64LE", 3), ("S32LE", 2)])
def test_hip_many_resamplers_in_one_launch_equal_one_by_one(native_lib, gpu, ref, fmt, ch):
    """gstamd_audio_resampler_resample_many: 70 independent streams (more than one launch's 64) that joined at different times - every stream
    stands somewhere else in its history and phase - against the reference stream by stream, 0 ULP, over buffers of uneven sizes incl. the
    drain; a resampler of another filter in the middle of the set goes its own way and the rest still share launches."""
    import torch
    dt = cases.AUDIO_DTYPES[fmt]
    tdt = {"F32LE": torch.float32, "S16LE": torch.int16, "F64LE": torch.float64, "S32LE": torch.int32}[fmt]
    n_streams = 70
    odd = 33                              # this one resamples to another rate: not part of anybody's launch
    rates = [(48000, 32000) if i == odd else (48000, 44100) for i in range(n_streams)]
    rs = [A.AudioResampler(fmt, ch, a, b, "kaiser", A.options("kaiser", 4, a, b)) for a, b in rates]
    rrs = [ref.AudioResampler(fmt, ch, a, b, quality=4) for a, b in rates]
    sigs = [cases.audio_buffer(fmt, ch, 6000, 9000 + i) for i in range(n_streams)]
    d_sigs = [torch.from_numpy(x).to(gpu) for x in sigs]
    pos = [0] * n_streams
    # a head start of its own for every stream, one by one: different histories and phases
    for i in range(n_streams):
        n = 37 * (i % 9) + 5
        no = rrs[i].get_out_frames(n)
        assert rs[i].get_out_frames(n) == no
        exp = rrs[i].resample(sigs[i][:n], in_frames=n, out_frames=no)
        d_out = torch.zeros((max(no, 1), ch), dtype=tdt, device=gpu)
        rs[i].resample(d_sigs[i][:n], n, d_out, no)
        torch.cuda.synchronize()
        assert (d_out[:no].cpu().numpy().reshape(-1) == np.asarray(exp).reshape(-1)).all()
        pos[i] = n
    for size in (1024, 1, 480, 1024, 0, 777):
        ins, nin, outs, nout, exps = [], [], [], [], []
        for i in range(n_streams):
            n = size if size else None
            if n is None:                  # drain: NULL input of get_max_latency frames
                nn = rrs[i].get_max_latency()
                no = rrs[i].get_out_frames(nn)
                assert rs[i].get_out_frames(nn) == no
                exps.append(rrs[i].resample(None, in_frames=nn, out_frames=no))
                ins.append(None)
                nin.append(nn)
            else:
                no = rrs[i].get_out_frames(n)
                assert rs[i].get_out_frames(n) == no
                exps.append(rrs[i].resample(sigs[i][pos[i]:pos[i] + n], in_frames=n, out_frames=no))
                ins.append(d_sigs[i][pos[i]:pos[i] + n])
                nin.append(n)
                pos[i] += n
            outs.append(torch.zeros((max(no, 1), ch), dtype=tdt, device=gpu))
            nout.append(no)
        A.resample_many(rs, ins, nin, outs, nout)
        torch.cuda.synchronize()
        for i in range(n_streams):
            got = outs[i][:nout[i]].cpu().numpy().reshape(-1)
            assert (got == np.asarray(exps[i]).reshape(-1)).all(), (size, i)
    for r in rs:
        r.free()
