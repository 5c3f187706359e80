"""CPU-side audio resampler tests: host bookkeeping + taps + FIR kernel bodies (host emulator) against the
reference (oracle/_ref) and its golden hashes.  Bit-exact for every format, as the C summation order is kept."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import cases
from gstreamer_amd import audio as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "audio_golden.json")))


def _emu(emu_lib):
    emu_lib.emu_audio_new.restype = C.c_void_p
    emu_lib.emu_audio_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(A.ResamplerOptions),
                                      C.POINTER(C.c_int), C.c_char_p, C.c_int]
    emu_lib.emu_audio_get_out_frames.restype = C.c_size_t
    emu_lib.emu_audio_get_out_frames.argtypes = [C.c_void_p, C.c_size_t]
    emu_lib.emu_audio_resample.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    emu_lib.emu_audio_free.argtypes = [C.c_void_p]
    emu_lib.emu_audio_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(A.ResamplerOptions)]
    emu_lib.emu_audio_state.argtypes = [C.c_void_p, C.c_int]
    return emu_lib


@pytest.mark.parametrize("case", cases.AUDIO_CASES, ids=lambda c: c[0])
def test_fir_bodies_on_host_match_golden(native_lib, emu_lib, case):
    E = _emu(emu_lib)
    name, fmt, ch, ir, orr, method, quality, bufs = case
    o = A.options(method, quality, ir, orr, **cases.audio_filter_kwargs(name))
    st = C.c_int(0)
    h = E.emu_audio_new(A.METHODS[method], 0, A.FORMATS[fmt], ch, ir, orr, C.byref(o), C.byref(st), None, 0)
    assert h, st.value
    dt = cases.AUDIO_DTYPES[fmt]
    chunks = []
    latency = None
    for i, n in enumerate(list(bufs) + [None]):
        data = None if n is None else cases.audio_buffer(fmt, ch, n, cases.case_seed(name) + i)
        if n is None:
            r = A.AudioResampler(fmt, ch, ir, orr, method, o)
            n = r.get_max_latency()
            r.free()
        no = E.emu_audio_get_out_frames(h, n)
        assert no == GOLDEN[name]["out_frames"][i]         # gst_audio_resampler_get_out_frames parity
        got = np.zeros((no, ch), dt)
        E.emu_audio_resample(h, data.ctypes.data if data is not None else None, n, got.ctypes.data, no)
        chunks.append(got.reshape(-1))
    E.emu_audio_free(h)
    assert cases.sha(np.concatenate(chunks)) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("lds", [True, False], ids=["k_fir_lds", "k_fir"])
def test_lds_fir_body_is_the_path_taken_for_full_tables(native_lib, emu_lib, lds, monkeypatch):
    """FULL-table streams go through the LDS-staged kernel body (four lanes per output frame, the reference's four partial sums);
    with it switched off the one-lane-per-sample body gives the same samples.  C4's stream: 48k -> 44.1k F32 stereo."""
    E = _emu(emu_lib)
    E.emu_fir_lds_runs.restype = C.c_int
    if not lds:
        monkeypatch.setenv("GSTAMD_NO_FIR_LDS", "1")
    case = [c for c in cases.AUDIO_CASES if c[1] == "F32LE" and c[3] == 48000 and c[4] == 44100 and c[2] == 2][0]
    before = E.emu_fir_lds_runs()
    test_fir_bodies_on_host_match_golden(native_lib, emu_lib, case)
    assert (E.emu_fir_lds_runs() > before) == lds


@pytest.mark.parametrize("case", cases.AUDIO_CASES[::3], ids=lambda c: c[0])
def test_golden_is_the_references_output(ref, case):
    name, fmt, ch, ir, orr, method, quality, bufs = case
    rr = ref.AudioResampler(fmt, ch, ir, orr, method=method, quality=quality, **cases.audio_filter_kwargs(name))
    chunks = []
    for i, n in enumerate(list(bufs) + [None]):
        data = None if n is None else cases.audio_buffer(fmt, ch, n, cases.case_seed(name) + i)
        n_in = rr.get_max_latency() if n is None else n
        chunks.append(rr.resample(data, in_frames=n_in, out_frames=rr.get_out_frames(n_in)).reshape(-1))
    assert cases.sha(np.concatenate(chunks)) == GOLDEN[name]["sha256"]


LAYOUT_CASES = [c for c in cases.AUDIO_CASES if c[0] in ("f32_48k_44k1_q4_stereo", "f32_6ch_cubic", "s16_48k_44k1_q4", "f32_8k_16k_gappy",
                                                         "s32_interp_cubic_48k_32k", "f64_48k_44k1_q4")]


@pytest.mark.parametrize("layout", [(True, True), (True, False), (False, True)], ids=["planar_planar", "planar_in", "planar_out"])
@pytest.mark.parametrize("case", LAYOUT_CASES, ids=lambda c: c[0])
def test_non_interleaved_layouts_on_host_match_golden(native_lib, emu_lib, case, layout):
    """GST_AUDIO_RESAMPLER_FLAG_NON_INTERLEAVED_IN / _OUT: the samples are those of the interleaved stream (the golden), only
    laid out as [channels][frames] on the flagged side."""
    E = _emu(emu_lib)
    name, fmt, ch, ir, orr, method, quality, bufs = case
    in_planar, out_planar = layout
    o = A.options(method, quality, ir, orr, **cases.audio_filter_kwargs(name))
    st = C.c_int(0)
    h = E.emu_audio_new(A.METHODS[method], (1 if in_planar else 0) | (2 if out_planar else 0), A.FORMATS[fmt], ch, ir, orr, C.byref(o),
                        C.byref(st), None, 0)
    assert h, st.value
    dt = cases.AUDIO_DTYPES[fmt]
    chunks = []
    for i, n in enumerate(list(bufs) + [None]):
        data = None if n is None else cases.audio_buffer(fmt, ch, n, cases.case_seed(name) + i)
        if data is not None and in_planar:
            data = np.ascontiguousarray(data.T)
        if n is None:
            r = A.AudioResampler(fmt, ch, ir, orr, method, o)
            n = r.get_max_latency()
            r.free()
        no = E.emu_audio_get_out_frames(h, n)
        got = np.zeros((ch, no) if out_planar else (no, ch), dt)
        E.emu_audio_resample(h, data.ctypes.data if data is not None else None, n, got.ctypes.data, no)
        chunks.append((got.T if out_planar else got).reshape(-1))
    E.emu_audio_free(h)
    assert cases.sha(np.concatenate(chunks)) == GOLDEN[name]["sha256"]


def test_reference_layouts_carry_the_same_samples(ref):
    """The premise of the layout tests, checked on the reference itself: planar in / out == the interleaved stream transposed."""
    name, fmt, ch, ir, orr, method, quality, bufs = next(c for c in cases.AUDIO_CASES if c[0] == "f32_6ch_cubic")
    rr = ref.AudioResampler(fmt, ch, ir, orr, method=method, quality=quality, in_planar=True, out_planar=True)
    chunks = []
    for i, n in enumerate(list(bufs) + [None]):
        data = None if n is None else np.ascontiguousarray(cases.audio_buffer(fmt, ch, n, cases.case_seed(name) + i).T)
        n_in = rr.get_max_latency() if n is None else n
        chunks.append(rr.resample(data, in_frames=n_in, out_frames=rr.get_out_frames(n_in)).T.reshape(-1))
    assert cases.sha(np.concatenate(chunks)) == GOLDEN[name]["sha256"]


def test_c4_plan_48k_to_44k1(native_lib):
    """160/147 reduction, 72 taps, FULL mode with 147 cached phases, 8x oversampled cubic build."""
    r = A.AudioResampler("F32LE", 2, 48000, 44100, "kaiser", A.options("kaiser", 4, 48000, 44100))
    d = r.debug()
    assert (d["in_rate"], d["n_phases"], d["n_taps"], d["oversample"]) == (160, 147, 72, 8)
    assert d["filter_mode"] == A.FILTER_MODE["full"] and d["filter_interpolation"] == A.FILTER_INTERPOLATION["cubic"]
    assert d["samples_avail"] == 35 and r.get_max_latency() == 36
    taps = r.taps()
    assert taps.shape == (147, 72)
    assert np.allclose(taps.sum(axis=1), 1.0, atol=2e-3)      # every phase is DC-normalised
    # NULL options == Kaiser quality 4 (audio-resampler.c:1414-1419)
    r2 = A.AudioResampler("F32LE", 2, 48000, 44100, "kaiser", None)
    assert (r2.taps() == taps).all()
    # reference test_gap_no_extra_samples (tests/check/elements/audioresample.c:1283): 8k -> 16k sample counts
    r3 = A.AudioResampler("S16LE", 1, 8000, 16000, "kaiser", A.options("kaiser", 4, 8000, 16000))
    assert r3.get_out_frames(255) > 0 and r3.get_in_frames(320) == 160
    for x in (r, r2, r3):
        x.free()


def test_resampler_abi_takes_gst_audio_format_values(native_lib):
    """gst_audio_resampler_new's `format` is a GstAudioFormat (audio-resampler.h:218): the ABI takes those values (S16LE 4, S32LE 12,
    F32LE 28, F64LE 30), still understands the private 0 .. 3 of the first releases, and refuses the formats the resampler does not
    accept (audio-resampler.c:1358-1360)."""
    import ctypes as C
    L = A.lib()
    for gst_value, old_value in ((4, 0), (12, 1), (28, 2), (30, 3)):
        taps = []
        for v in (gst_value, old_value):
            st = C.c_int(0)
            h = L.gstamd_audio_resampler_new(4, 0, v, 2, 48000, 44100, None, C.byref(st))
            assert h and st.value == 0, (v, st.value)
            L.gstamd_audio_resampler_get_out_frames.argtypes = [C.c_void_p, C.c_size_t]
            L.gstamd_audio_resampler_get_out_frames.restype = C.c_size_t
            taps.append(L.gstamd_audio_resampler_get_out_frames(C.c_void_p(h), 1024))
            L.gstamd_audio_resampler_free.argtypes = [C.c_void_p]
            L.gstamd_audio_resampler_free(C.c_void_p(h))
        assert taps[0] == taps[1]
    for bad in (5, 8, 16, 29, 31, 40):           # S16BE, S24_32LE, S24LE, F32BE, F64BE, nothing (2 and 3 - S8, U8 - are the old F32 and F64)
        st = C.c_int(0)
        assert not L.gstamd_audio_resampler_new(4, 0, bad, 2, 48000, 44100, None, C.byref(st)) and st.value != 0, bad


def test_filter_mode_auto_picks_interpolated_for_big_tables(native_lib):
    """audio-resampler.c:1110-1130: 48000 -> 44101 would need 44101 phases x 72 taps x 4 B > the 1 MiB threshold, so
    mode AUTO resolves to INTERPOLATED (cubic, 8x oversampled table of 8 + 4 rows)."""
    from gstreamer_amd import video as V
    r = A.AudioResampler("F32LE", 2, 48000, 44101, "kaiser", A.options("kaiser", 4, 48000, 44101))
    d = r.debug()
    assert d["filter_mode"] == A.FILTER_MODE["interpolated"] and d["filter_interpolation"] == A.FILTER_INTERPOLATION["cubic"]
    assert d["oversample"] == 8
    r.free()


def update_options(case, item):
    """The options structure an update item stands for (None = NULL options), built like the reference driver builds it."""
    if not cases.audio_update_has_options(item):
        return None
    kw = {k: item[k] for k in ("filter_mode", "filter_interpolation") if k in item}
    return A.options(case[5], item.get("quality"), item["in_rate"], item["out_rate"], **kw)


@pytest.mark.parametrize("case", cases.AUDIO_UPDATE_CASES, ids=lambda c: c[0])
def test_update_streams_on_host_match_golden(native_lib, emu_lib, case):
    """gst_audio_resampler_update (audio-resampler.c:1503-1614) in mid-stream: phase rescale, rate reduction, filter
    redesign with history shift, and the NULL-options variant that keeps the old design."""
    E = _emu(emu_lib)
    name, fmt, ch, ir, orr, method, quality, script = case
    o = A.options(method, quality, ir, orr, **cases.audio_filter_kwargs(name))
    st = C.c_int(0)
    h = E.emu_audio_new(A.METHODS[method], 0, A.FORMATS[fmt], ch, ir, orr, C.byref(o), C.byref(st), None, 0)
    assert h, st.value
    dt = cases.AUDIO_DTYPES[fmt]
    counts = []

    def do_update(item):
        raw = item.get("raw", (item["in_rate"], item["out_rate"]))
        uo = update_options(case, item)
        assert E.emu_audio_update(h, raw[0], raw[1], C.byref(uo) if uo is not None else None) == 0

    def do_resample(data, n_in):
        no = E.emu_audio_get_out_frames(h, n_in)
        counts.append(no)
        got = np.zeros((no, ch), dt)
        E.emu_audio_resample(h, data.ctypes.data if data is not None else None, n_in, got.ctypes.data, no)
        return got

    full = cases.audio_update_stream(case, do_update, do_resample, lambda: E.emu_audio_state(h, 0) // 2)
    E.emu_audio_free(h)
    assert counts == GOLDEN[name]["out_frames"]
    assert cases.sha(full) == GOLDEN[name]["sha256"]


@pytest.mark.parametrize("case", cases.AUDIO_UPDATE_CASES[::2], ids=lambda c: c[0])
def test_update_golden_is_the_references_output(ref, case):
    name, fmt, ch, ir, orr, method, quality, script = case
    rr = ref.AudioResampler(fmt, ch, ir, orr, method=method, quality=quality, **cases.audio_filter_kwargs(name))

    def do_update(item):
        raw = item.get("raw", (item["in_rate"], item["out_rate"]))
        assert rr.update(raw[0], raw[1], quality=item.get("quality"), filter_mode=item.get("filter_mode"),
                         filter_interpolation=item.get("filter_interpolation"), q_rates=(item["in_rate"], item["out_rate"]))

    full = cases.audio_update_stream(case, do_update, lambda d, n: rr.resample(d, in_frames=n, out_frames=rr.get_out_frames(n)),
                                     rr.get_max_latency)
    assert cases.sha(full) == GOLDEN[name]["sha256"]


def test_failed_update_leaves_the_resampler_alone(native_lib):
    r = A.AudioResampler("F32LE", 2, 48000, 44100)
    before = r.debug()
    bad = A.options("kaiser", 4, 48000, 32000, n_taps=0)
    # kaiser ignores n-taps; linear takes it: a non-positive count is refused
    r2 = A.AudioResampler("F32LE", 2, 48000, 44100, "linear")
    with pytest.raises(Exception):
        r2.update(48000, 32000, bad)
    assert r2.debug() == A.AudioResampler("F32LE", 2, 48000, 44100, "linear").debug()
    r.update(48000, 32000, A.options("kaiser", 4, 48000, 32000))
    after = r.debug()
    assert (after["n_phases"], after["in_rate"]) == (2, 3) and after["n_taps"] > before["n_taps"]
