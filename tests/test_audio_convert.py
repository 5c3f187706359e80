"""GstAudioConverter on the device (include/gstamd_audio.h gstamd_audio_converter_*): every case is compared, byte for byte, with the
reference's own gst_audio_converter_samples (oracle/_ref) on the same seeded stream - format conversion, channel mixing, the
quantiser's dither (the reference's xorshift generator, reproduced by jumping ahead in it), resampling inside the converter.
-m "not gpu": the kernel bodies on the host emulator; -m gpu: the HIP path through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from gstreamer_amd import audio as A

BYTES = A.AFMT_BYTES


def stream(fmt, channels, frames, seed):
    """interleaved frames of `fmt` as bytes: full-range integers / U(-1.2, 1.2) floats (so that clipping happens) + a sine block"""
    rng = np.random.RandomState(seed)
    n = frames * channels
    if fmt in ("F32LE", "F64LE"):
        x = rng.uniform(-1.2, 1.2, n)
        x[: n // 4] = 0.9 * np.sin(np.arange(n // 4) * 0.05)
        if n >= 32:
            x[n // 2: n // 2 + 8] = [0.0, -0.0, 1.0, -1.0, 1e-40, -1e-40, 0.99999999, -0.99999999]
        return x.astype(np.float32 if fmt == "F32LE" else np.float64).view(np.uint8).copy()
    return rng.randint(0, 256, n * BYTES[fmt]).astype(np.uint8)


L51 = ['front-left', 'front-right', 'front-center', 'lfe1', 'rear-left', 'rear-right']
L71 = L51 + ["side-left", "side-right"]
# (name, in_fmt, in_rate, in_ch, out_fmt, out_rate, out_ch, config kwargs, mix matrix [out][in] or None, buffer sizes in frames)
CASES = [
    ("s16_f32", "S16LE", 48000, 2, "F32LE", 48000, 2, {}, None, (1024, 1024, 37)),
    ("f32_s16_tpdf", "F32LE", 48000, 2, "S16LE", 48000, 2, dict(dither_method="tpdf"), None, (1024, 333, 1024)),
    ("f32_s16_default", "F32LE", 48000, 2, "S16LE", 48000, 2, {}, None, (1024, 333)),
    ("f32_s16_rpdf", "F32LE", 44100, 2, "S16LE", 44100, 2, dict(dither_method="rpdf"), None, (1000, 1000)),
    ("f32_s16_nodither", "F32LE", 48000, 1, "S16LE", 48000, 1, dict(dither_method="none"), None, (1024, 1024)),
    ("f64_s32", "F64LE", 48000, 2, "S32LE", 48000, 2, {}, None, (1024,)),
    ("s32_s16", "S32LE", 48000, 2, "S16LE", 48000, 2, dict(dither_method="tpdf"), None, (1024, 1024)),
    ("s32_s24", "S32LE", 48000, 2, "S24LE", 48000, 2, {}, None, (1024,)),
    ("s24_s32", "S24LE", 48000, 2, "S32LE", 48000, 2, {}, None, (1024,)),
    ("s24_32_f64", "S24_32LE", 96000, 1, "F64LE", 96000, 1, {}, None, (777,)),
    ("u8_s16", "U8", 8000, 1, "S16LE", 8000, 1, dict(dither_method="tpdf"), None, (1024,)),
    ("s16_u8", "S16LE", 8000, 2, "U8", 8000, 2, dict(dither_method="tpdf"), None, (1024, 1024)),
    ("s8_f32", "S8", 22050, 2, "F32LE", 22050, 2, {}, None, (512,)),
    ("f32_s8", "F32LE", 22050, 2, "S8", 22050, 2, dict(dither_method="rpdf"), None, (512, 512)),
    ("f32_f64", "F32LE", 48000, 2, "F64LE", 48000, 2, {}, None, (1024,)),
    ("f64_f32", "F64LE", 48000, 2, "F32LE", 48000, 2, {}, None, (1024,)),
    ("f32_s24_32_above_threshold", "F32LE", 48000, 2, "S24_32LE", 48000, 2, {}, None, (1024,)),
    ("s16_s16_passthrough", "S16LE", 48000, 2, "S16LE", 48000, 2, {}, None, (1024,)),
    ("s16_stereo_to_mono", "S16LE", 48000, 2, "S16LE", 48000, 1, {}, None, (1024, 1024)),
    ("f32_mono_to_stereo", "F32LE", 48000, 1, "F32LE", 48000, 2, {}, None, (1024,)),
    ("s16_mono_to_stereo_f32", "S16LE", 44100, 1, "F32LE", 44100, 2, {}, None, (1024,)),
    ("f32_stereo_to_mono_s16", "F32LE", 48000, 2, "S16LE", 48000, 1, dict(dither_method="tpdf"), None, (1024, 500)),
    ("s32_matrix_3_to_2", "S32LE", 48000, 3, "S32LE", 48000, 2, {}, [[1.0, 0.0, 0.7071], [0.0, 1.0, 0.7071]], (1024,)),
    ("f64_matrix_2_to_4", "F64LE", 48000, 2, "F64LE", 48000, 4, {}, [[1, 0], [0, 1], [0.5, 0.5], [0.25, -0.25]], (1024,)),
    ("f32_s16_matrix_swap", "F32LE", 48000, 2, "S16LE", 48000, 2, {}, [[0, 1], [1, 0]], (1024,)),
    ("f32_resample_48k_44k1", "F32LE", 48000, 2, "F32LE", 44100, 2, {}, None, (1024,) * 6),
    ("s16_resample_44k1_48k", "S16LE", 44100, 2, "S16LE", 48000, 2, {}, None, (1024,) * 4),
    ("s16_f32_resample_mono_stereo", "S16LE", 32000, 1, "F32LE", 48000, 2, {}, None, (1024,) * 4),
    ("f32_s16_resample_dither", "F32LE", 48000, 2, "S16LE", 44100, 2, dict(dither_method="tpdf"), None, (1024,) * 5),
    ("f32_s16_tpdf_hf", "F32LE", 48000, 2, "S16LE", 48000, 2, dict(dither_method="tpdf-hf"), None, (1024, 333, 1)),
    ("s32_u8_tpdf_hf_mono", "S32LE", 8000, 1, "U8", 8000, 1, dict(dither_method="tpdf-hf"), None, (500, 500)),
    ("f32_s16_error_feedback", "F32LE", 48000, 2, "S16LE", 48000, 2, dict(dither_method="tpdf", noise_shaping="error-feedback"), None, (1024, 333, 512)),
    ("f32_s16_ns_simple", "F32LE", 48000, 2, "S16LE", 48000, 2, dict(dither_method="tpdf", noise_shaping="simple"), None, (1024, 333)),
    ("f32_s16_ns_medium_rpdf", "F32LE", 44100, 1, "S16LE", 44100, 1, dict(dither_method="rpdf", noise_shaping="medium"), None, (1024, 1024)),
    ("f32_s16_ns_high", "F32LE", 48000, 2, "S16LE", 48000, 2, dict(dither_method="tpdf", noise_shaping="high"), None, (1024, 1024, 7)),
    ("s32_s16_ns_high_no_dither", "S32LE", 48000, 2, "S16LE", 48000, 2, dict(noise_shaping="high"), None, (1024, 1024)),
    ("f32_s8_ns_high_tpdf_hf", "F32LE", 48000, 2, "S8", 48000, 2, dict(dither_method="tpdf-hf", noise_shaping="high"), None, (512, 512)),
    ("f32_s16_ns_high_low_rate", "F32LE", 16000, 2, "S16LE", 16000, 2, dict(dither_method="tpdf", noise_shaping="high"), None, (512, 512)),   # -> error feedback
    ("f32_s16_ns_medium_reset", "F32LE", 48000, 2, "S16LE", 48000, 2, dict(dither_method="tpdf", noise_shaping="medium", reset_after=0), None, (1024, 1024)),
    ("f32_s16_ns_resample", "F32LE", 48000, 2, "S16LE", 44100, 2, dict(dither_method="tpdf", noise_shaping="medium"), None, (1024,) * 4),
    ("f32_51_to_stereo", "F32LE", 48000, 6, "F32LE", 48000, 2, dict(in_pos=L51), None, (1024, 100)),
    ("s16_51_to_stereo", "S16LE", 48000, 6, "S16LE", 48000, 2, dict(in_pos=L51), None, (1024,)),
    ("s32_stereo_to_51_f32", "S32LE", 48000, 2, "F32LE", 48000, 6, dict(out_pos=L51), None, (1024,)),
    ("f64_71_to_51", "F64LE", 48000, 8, "F64LE", 48000, 6, dict(in_pos=L71, out_pos=L51), None, (512,)),
    ("f32_51_to_mono_s16", "F32LE", 48000, 6, "S16LE", 48000, 1, dict(in_pos=L51, dither_method="tpdf"), None, (512,)),
    ("f32_51_to_quad", "F32LE", 48000, 6, "F32LE", 48000, 4, dict(in_pos=L51, out_pos=["front-left", "front-right", "rear-left", "rear-right"]), None, (512,)),
    ("s16_51_reordered", "S16LE", 48000, 6, "S16LE", 48000, 6, dict(in_pos=L51, out_pos=["front-left", "front-right", "rear-left", "rear-right", "front-center", "lfe1"]), None, (512,)),
    ("f32_3mono_to_stereo", "F32LE", 48000, 3, "F32LE", 48000, 2, dict(in_pos=["mono"] * 3), None, (512,)),
    ("f32_4alternate_to_mono", "F32LE", 48000, 4, "F32LE", 48000, 1, dict(in_pos=["front-left", "front-right"] * 2), None, (512,)),
    ("f32_rear_center_to_stereo", "F32LE", 48000, 3, "F32LE", 48000, 2, dict(in_pos=["front-left", "front-right", "rear-center"]), None, (512,)),
    ("f32_sides_to_51", "F32LE", 48000, 4, "F32LE", 48000, 6, dict(in_pos=["front-left", "front-right", "side-left", "side-right"], out_pos=L51), None, (512,)),
    ("f32_8ch_unpositioned_identity_s16", "F32LE", 48000, 8, "S16LE", 48000, 8, {}, None, (256,)),
    ("s24_s16_resample_down_mix", "S24LE", 48000, 2, "S16LE", 16000, 1, {}, None, (960,) * 4),
]
REFUSED = [
    ("F32LE", 48000, 6, "F32LE", 48000, 2, {}),          # unpositioned channels, different counts, no mix-matrix (audio-converter.c:1370)
]


def config_kw(kw):
    return {k: v for k, v in kw.items() if k not in ("reset_after", "in_pos", "out_pos")}


def pos_values(names):
    return None if names is None else [A.POSITION[n] for n in names]


def infos(case):
    name, ifmt, ir, ic, ofmt, orr, oc, kw, mix, bufs = case
    return A.audio_info(ifmt, ir, ic, positions=pos_values(kw.get("in_pos"))), A.audio_info(ofmt, orr, oc, positions=pos_values(kw.get("out_pos")))


def config_string(kw):
    parts = []
    if "dither_method" in kw:
        parts.append("GstAudioConverter.dither-method=(GstAudioDitherMethod)%s" % kw["dither_method"])
    if "noise_shaping" in kw:
        parts.append("GstAudioConverter.noise-shaping-method=(GstAudioNoiseShapingMethod)%s" % kw["noise_shaping"])
    return ("GstAudioConverter, " + ", ".join(parts)) if parts else None


def reference_stream(ref, case):
    name, ifmt, ir, ic, ofmt, orr, oc, kw, mix, bufs = case
    rc = ref.AudioConverter(ifmt, ir, ic, ofmt, orr, oc, config=config_string(kw), mix=mix, in_pos=pos_values(kw.get("in_pos")),
                            out_pos=pos_values(kw.get("out_pos")))
    srcs, outs = [], []
    for k, n in enumerate(bufs):
        src = stream(ifmt, ic, n, 1000 + 17 * k + len(name))
        srcs.append(src)
        outs.append(rc.samples(src))
        if kw.get("reset_after") == k:
            rc.reset()
    rc.free()
    return srcs, outs


class EmuConverter:
    def __init__(self, emu, ii, oi, cfg):
        emu.emu_aconv_new.restype = C.c_void_p
        emu.emu_aconv_new.argtypes = [C.c_int, C.POINTER(A.AudioInfo), C.POINTER(A.AudioInfo), C.POINTER(A.AudioConverterConfig), C.c_char_p, C.c_int]
        emu.emu_aconv_get_out_frames.restype = C.c_size_t
        emu.emu_aconv_get_out_frames.argtypes = [C.c_void_p, C.c_size_t]
        emu.emu_aconv_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        emu.emu_aconv_free.argtypes = [C.c_void_p]
        emu.emu_aconv_reset.argtypes = [C.c_void_p]
        emu.emu_aconv_is_passthrough.argtypes = [C.c_void_p]
        self.emu = emu
        self.err = C.create_string_buffer(512)
        self.h = emu.emu_aconv_new(0, C.byref(ii), C.byref(oi), C.byref(cfg), self.err, 512)

    def samples(self, src, in_bpf, out_bpf):
        n = src.size // in_bpf
        on = self.emu.emu_aconv_get_out_frames(self.h, n)
        out = np.zeros(on * out_bpf, np.uint8)
        self.emu.emu_aconv_samples(self.h, src.ctypes.data, n, out.ctypes.data, on)
        return out


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_converter_bodies_on_host_match_reference(native_lib, emu_lib, ref, case):
    name, ifmt, ir, ic, ofmt, orr, oc, kw, mix, bufs = case
    srcs, exp = reference_stream(ref, case)
    cv = EmuConverter(emu_lib, *infos(case), A.audio_converter_config(mix_matrix=mix, **config_kw(kw)))
    assert cv.h, cv.err.value
    for k, src in enumerate(srcs):
        got = cv.samples(src, BYTES[ifmt] * ic, BYTES[ofmt] * oc)
        assert got.size == exp[k].size, (k, got.size, exp[k].size)
        assert (got == exp[k]).all(), (name, k, int((got != exp[k]).sum()), got[:16], exp[k][:16])
        if kw.get("reset_after") == k:
            emu_lib.emu_aconv_reset(cv.h)
    emu_lib.emu_aconv_free(cv.h)


@pytest.mark.parametrize("case", REFUSED, ids=lambda c: "%s_%dch_%s_%dch_%s" % (c[0], c[2], c[3], c[5], "_".join(map(str, c[6].values()))))
def test_unsupported_conversions_are_refused_not_approximated(native_lib, emu_lib, case):
    ifmt, ir, ic, ofmt, orr, oc, kw = case
    ii = A.audio_info(ifmt, ir, ic)
    cv = EmuConverter(emu_lib, ii, A.audio_info(ofmt, orr, oc), A.audio_converter_config(**kw))
    assert not cv.h and cv.err.value


def test_default_mix_matrices_are_the_references(native_lib, emu_lib, ref):
    """mono <-> stereo: the matrix the plan holds equals what the reference's mixer does to unit impulses"""
    for ic, oc in ((1, 2), (2, 1), (2, 2)):
        rc = ref.AudioConverter("F32LE", 48000, ic, "F32LE", 48000, oc)
        imp = np.zeros((ic, ic), np.float32)
        np.fill_diagonal(imp, 1.0)
        got = rc.samples(imp.view(np.uint8).reshape(-1).copy()).view(np.float32).reshape(ic, oc)
        rc.free()
        cv = EmuConverter(emu_lib, A.audio_info("F32LE", 48000, ic), A.audio_info("F32LE", 48000, oc), A.audio_converter_config())
        mine = cv.samples(imp.view(np.uint8).reshape(-1).copy(), 4 * ic, 4 * oc).view(np.float32).reshape(ic, oc)
        assert (got == mine).all(), (ic, oc, got, mine)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_hip_converter_matches_reference(native_lib, gpu, ref, case):
    import torch
    name, ifmt, ir, ic, ofmt, orr, oc, kw, mix, bufs = case
    srcs, exp = reference_stream(ref, case)
    cv = A.AudioConverter(*infos(case), A.audio_converter_config(mix_matrix=mix, **config_kw(kw)))
    for k, src in enumerate(srcs):
        n = src.size // (BYTES[ifmt] * ic)
        on = cv.get_out_frames(n)
        assert on * BYTES[ofmt] * oc == exp[k].size
        d_in = torch.from_numpy(src).to(gpu)
        d_out = torch.zeros(max(1, exp[k].size), dtype=torch.uint8, device=gpu)
        cv.samples(d_in, n, d_out, on)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()[: exp[k].size]
        assert (got == exp[k]).all(), (name, k, int((got != exp[k]).sum()))
        if kw.get("reset_after") == k:
            cv.reset()
    cv.free()


@pytest.mark.gpu
def test_hip_converter_refuses_what_it_cannot_do(native_lib, gpu):
    from gstreamer_amd import video as V
    with pytest.raises(V.GstAmdError):
        A.AudioConverter(A.audio_info("F32LE", 48000, 6), A.audio_info("F32LE", 48000, 2), A.audio_converter_config())
