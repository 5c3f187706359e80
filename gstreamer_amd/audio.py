"""ctypes binding of include/gstamd_audio.h - mirrors gst_audio_resampler_* (audio-resampler.h:218-256).

Sample buffers live in HBM (device pointers or torch CUDA tensors).  No CPU implementation."""
import ctypes as C

from . import video as _v

METHODS = {"nearest": 0, "linear": 1, "cubic": 2, "blackman-nuttall": 3, "kaiser": 4}
FORMATS = {"S16LE": 4, "S32LE": 12, "F32LE": 28, "F64LE": 30}      # GstAudioFormat values, as gst_audio_resampler_new takes them
FILTER_MODE = {"interpolated": 0, "full": 1, "auto": 2}
FILTER_INTERPOLATION = {"none": 0, "linear": 1, "cubic": 2}


class ResamplerOptions(C.Structure):
    _fields_ = [("cutoff", C.c_double), ("stop_attenuation", C.c_double), ("transition_bandwidth", C.c_double),
                ("cubic_b", C.c_double), ("cubic_c", C.c_double), ("max_phase_error", C.c_double),
                ("n_taps", C.c_int32), ("filter_mode", C.c_int32), ("filter_mode_threshold", C.c_int32),
                ("filter_interpolation", C.c_int32), ("filter_oversample", C.c_int32), ("reserved", C.c_int32 * 7)]


_ready = False


def lib():
    global _ready
    L = _v.lib()
    if not _ready:
        L.gstamd_audio_resampler_options_init.argtypes = [C.POINTER(ResamplerOptions)]
        L.gstamd_audio_resampler_options_set_quality.argtypes = [C.c_int, C.c_uint, C.c_int, C.c_int, C.POINTER(ResamplerOptions)]
        L.gstamd_audio_resampler_new.restype = C.c_void_p
        L.gstamd_audio_resampler_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.POINTER(ResamplerOptions), C.POINTER(C.c_int)]
        L.gstamd_audio_resampler_free.argtypes = [C.c_void_p]
        L.gstamd_audio_resampler_reset.argtypes = [C.c_void_p]
        L.gstamd_audio_resampler_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(ResamplerOptions)]
        for n in ("get_out_frames", "get_in_frames"):
            f = getattr(L, "gstamd_audio_resampler_" + n)
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_size_t]
        L.gstamd_audio_resampler_get_max_latency.restype = C.c_size_t
        L.gstamd_audio_resampler_get_max_latency.argtypes = [C.c_void_p]
        L.gstamd_audio_resampler_resample.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gstamd_audio_resampler_resample_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gstamd_audio_resampler_resample_many.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                                           C.POINTER(C.c_size_t), C.c_void_p]
        L.gstamd_audio_resampler_debug_get.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        L.gstamd_audio_resampler_debug_taps.restype = C.c_long
        L.gstamd_audio_resampler_debug_taps.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_long]
        _ready = True
    return L


def options(method="kaiser", quality=None, in_rate=0, out_rate=0, **kw):
    """Empty options structure; quality=N applies gst_audio_resampler_options_set_quality; kw overrides fields."""
    o = ResamplerOptions()
    lib().gstamd_audio_resampler_options_init(C.byref(o))
    if quality is not None:
        lib().gstamd_audio_resampler_options_set_quality(METHODS[method], quality, in_rate, out_rate, C.byref(o))
    enums = {"filter_mode": FILTER_MODE, "filter_interpolation": FILTER_INTERPOLATION}
    for k, v in kw.items():
        if k in enums and isinstance(v, str):
            v = enums[k][v]
        setattr(o, k, v)
    return o


_ptr = _v._ptr


class ManyBuffers:
    """The argument arrays of one gstamd_audio_resampler_resample_many call, built once (a caller that cycles through a ring of buffers keeps
    one of these per slot: building five ctypes arrays of 64 entries costs more than the launch)."""

    def __init__(self, resamplers, srcs, in_frames, dsts, out_frames):
        n = self.n = len(resamplers)
        assert n == len(srcs) == len(in_frames) == len(dsts) == len(out_frames)
        self.keep = (list(resamplers), list(srcs), list(dsts))
        self.rs = (C.c_void_p * n)(*[r._h for r in resamplers])
        self.ip = (C.c_void_p * n)(*[None if x is None else _ptr(x) for x in srcs])
        self.op = (C.c_void_p * n)(*[_ptr(x) for x in dsts])
        self.inf = (C.c_size_t * n)(*in_frames)
        self.outf = (C.c_size_t * n)(*out_frames)

    def run(self, stream=None):
        r = lib().gstamd_audio_resampler_resample_many(self.n, self.rs, self.ip, self.inf, self.op, self.outf, stream)
        if r != 0:
            raise RuntimeError("gstamd_audio_resampler_resample_many: %d" % r)


def resample_many(resamplers, srcs, in_frames, dsts, out_frames, stream=None):
    """gstamd_audio_resampler_resample_many: one buffer per resampler, one launch where the resamplers share a filter"""
    ManyBuffers(resamplers, srcs, in_frames, dsts, out_frames).run(stream)


class AudioResampler:
    """gst_audio_resampler_new (method, flags, format, channels, in_rate, out_rate, options)."""

    def __init__(self, fmt, channels, in_rate, out_rate, method="kaiser", opts=None, in_planar=False, out_planar=False):
        status = C.c_int(0)
        self.channels, self.fmt = channels, fmt
        flags = (1 if in_planar else 0) | (2 if out_planar else 0)        # GstAudioResamplerFlags
        self._h = lib().gstamd_audio_resampler_new(METHODS[method], flags, FORMATS[fmt], channels, in_rate, out_rate,
                                                   C.byref(opts) if opts is not None else None, C.byref(status))
        if not self._h:
            raise _v.GstAmdError(status.value, "audio resampler plan refused")

    def get_out_frames(self, in_frames):
        return lib().gstamd_audio_resampler_get_out_frames(self._h, in_frames)

    def get_in_frames(self, out_frames):
        return lib().gstamd_audio_resampler_get_in_frames(self._h, out_frames)

    def get_max_latency(self):
        return lib().gstamd_audio_resampler_get_max_latency(self._h)

    def resample(self, src, in_frames, dst, out_frames, stream=None):
        _v._check(lib().gstamd_audio_resampler_resample(self._h, _v._ptr(src), in_frames, _v._ptr(dst), out_frames, stream))

    def resample_planes(self, src_planes, in_frames, dst_planes, out_frames, stream=None):
        """gst_audio_resampler_resample's argument shape: lists of device pointers (one per plane; one for an interleaved side)."""
        ia = (C.c_void_p * len(src_planes))(*[_v._ptr(p) for p in src_planes]) if src_planes is not None else None
        oa = (C.c_void_p * len(dst_planes))(*[_v._ptr(p) for p in dst_planes])
        _v._check(lib().gstamd_audio_resampler_resample_planes(self._h, ia, in_frames, oa, out_frames, stream))

    def update(self, in_rate=0, out_rate=0, opts=None):
        """gst_audio_resampler_update (resampler, in_rate, out_rate, options); opts None keeps the previous filter design."""
        _v._check(lib().gstamd_audio_resampler_update(self._h, in_rate, out_rate, C.byref(opts) if opts is not None else None))

    def divergence(self):
        """"" while the output is the reference's bit for bit (gstamd_audio_resampler_divergence)"""
        f = lib().gstamd_audio_resampler_divergence
        f.restype, f.argtypes = C.c_char_p, [C.c_void_p]
        return f(self._h).decode()

    def reset(self):
        lib().gstamd_audio_resampler_reset(self._h)

    def debug(self):
        buf = (C.c_int32 * 16)()
        n = lib().gstamd_audio_resampler_debug_get(self._h, buf, 16)
        keys = ["n_taps", "n_phases", "in_rate", "oversample", "filter_mode", "filter_interpolation", "taps_stride",
                "samples_avail", "samp_phase", "skip"]
        return dict(zip(keys, list(buf[:n])))

    def taps(self):
        import numpy as np
        n = lib().gstamd_audio_resampler_debug_taps(self._h, None, 0)
        a = np.zeros(n, np.float64)
        lib().gstamd_audio_resampler_debug_taps(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), n)
        d = self.debug()
        return a.reshape(d["n_phases"], d["n_taps"])

    def free(self):
        if self._h:
            lib().gstamd_audio_resampler_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- GstAudioConverter (include/gstamd_audio.h) -------------------------------------------------------------------------------
AFMT = {"S8": 2, "U8": 3, "S16LE": 4, "S24_32LE": 8, "S32LE": 12, "S24LE": 16, "F32LE": 28, "F64LE": 30}
AFMT_BYTES = {"S8": 1, "U8": 1, "S16LE": 2, "S24LE": 3, "S24_32LE": 4, "S32LE": 4, "F32LE": 4, "F64LE": 8}
DITHER = {"none": 0, "rpdf": 1, "tpdf": 2, "tpdf-hf": 3}
NOISE_SHAPING = {"none": 0, "error-feedback": 1, "simple": 2, "medium": 3, "high": 4}
MAX_CHANNELS = 8
# default positions of gst_audio_info_set_format (audio-info.c: gst_audio_channel_positions ... default_channel_order) for 1 / 2 channels
DEFAULT_POSITIONS = {1: [-2], 2: [0, 1]}
# GstAudioChannelPosition (audio-channels.h:101-133)
POSITION = {"none": -3, "mono": -2, "invalid": -1, "front-left": 0, "front-right": 1, "front-center": 2, "lfe1": 3, "rear-left": 4, "rear-right": 5,
            "front-left-of-center": 6, "front-right-of-center": 7, "rear-center": 8, "lfe2": 9, "side-left": 10, "side-right": 11}


class AudioInfo(C.Structure):
    _fields_ = [("format", C.c_int32), ("rate", C.c_int32), ("channels", C.c_int32), ("layout", C.c_int32), ("unpositioned", C.c_int32),
                ("position", C.c_int32 * MAX_CHANNELS)]


class AudioConverterConfig(C.Structure):
    _fields_ = [("dither_method", C.c_int32), ("noise_shaping", C.c_int32), ("dither_threshold", C.c_uint32), ("resampler_method", C.c_int32),
                ("has_resampler_options", C.c_int32), ("resampler_options", ResamplerOptions), ("has_mix_matrix", C.c_int32),
                ("mix_matrix", (C.c_float * MAX_CHANNELS) * MAX_CHANNELS)]


def audio_info(fmt, rate, channels, positions=None, unpositioned=False):
    ai = AudioInfo()
    ai.format, ai.rate, ai.channels, ai.layout, ai.unpositioned = AFMT[fmt], rate, channels, 0, int(unpositioned)
    pos = positions if positions is not None else DEFAULT_POSITIONS.get(channels)
    if pos is None:                             # gst_audio_info_set_format without positions beyond stereo: unpositioned, NONE everywhere
        pos = [POSITION["none"]] * channels
        ai.unpositioned = 1
    for i, v in enumerate(pos):
        ai.position[i] = POSITION[v] if isinstance(v, str) else v
    return ai


_conv_ready = False


def _conv_lib():
    global _conv_ready
    L = lib()
    if not _conv_ready:
        L.gstamd_audio_converter_config_init.argtypes = [C.POINTER(AudioConverterConfig)]
        L.gstamd_audio_converter_new.restype = C.c_void_p
        L.gstamd_audio_converter_new.argtypes = [C.c_int, C.POINTER(AudioInfo), C.POINTER(AudioInfo), C.POINTER(AudioConverterConfig), C.POINTER(C.c_int)]
        L.gstamd_audio_converter_free.argtypes = [C.c_void_p]
        L.gstamd_audio_converter_reset.argtypes = [C.c_void_p]
        for n in ("get_out_frames", "get_in_frames"):
            f = getattr(L, "gstamd_audio_converter_" + n)
            f.restype, f.argtypes = C.c_size_t, [C.c_void_p, C.c_size_t]
        L.gstamd_audio_converter_get_max_latency.restype = C.c_size_t
        L.gstamd_audio_converter_get_max_latency.argtypes = [C.c_void_p]
        L.gstamd_audio_converter_is_passthrough.argtypes = [C.c_void_p]
        L.gstamd_audio_converter_samples.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gstamd_audio_converter_get_mix_matrix.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        _conv_ready = True
    return L


def audio_converter_config(dither_method=None, noise_shaping=None, dither_threshold=None, mix_matrix=None, resampler_method=None):
    cfg = AudioConverterConfig()
    _conv_lib().gstamd_audio_converter_config_init(C.byref(cfg))
    if dither_method is not None:
        cfg.dither_method = DITHER[dither_method] if isinstance(dither_method, str) else dither_method
    if noise_shaping is not None:
        cfg.noise_shaping = NOISE_SHAPING[noise_shaping] if isinstance(noise_shaping, str) else noise_shaping
    if dither_threshold is not None:
        cfg.dither_threshold = dither_threshold
    if resampler_method is not None:
        cfg.resampler_method = METHODS[resampler_method] if isinstance(resampler_method, str) else resampler_method
    if mix_matrix is not None:                  # [out][in], as the GstAudioConverter.mix-matrix option
        cfg.has_mix_matrix = 1
        for j, row in enumerate(mix_matrix):
            for i, v in enumerate(row):
                cfg.mix_matrix[j][i] = v
    return cfg


class AudioConverter:
    """gst_audio_converter_new (flags, in_info, out_info, config) -> .samples (in, in_frames, out, out_frames) on device buffers."""

    def __init__(self, in_info, out_info, config=None, flags=0):
        st = C.c_int(0)
        self._h = _conv_lib().gstamd_audio_converter_new(flags, C.byref(in_info), C.byref(out_info),
                                                         C.byref(config) if config is not None else None, C.byref(st))
        if not self._h:
            raise _v.GstAmdError(st.value, _v.last_error())

    def get_out_frames(self, in_frames):
        return _conv_lib().gstamd_audio_converter_get_out_frames(self._h, in_frames)

    def is_passthrough(self):
        return bool(_conv_lib().gstamd_audio_converter_is_passthrough(self._h))

    def samples(self, src, in_frames, dst, out_frames, stream=None):
        _v._check(_conv_lib().gstamd_audio_converter_samples(self._h, 0, _v._ptr(src), in_frames, _v._ptr(dst), out_frames, stream))

    def reset(self):
        _conv_lib().gstamd_audio_converter_reset(self._h)

    def mix_matrix(self, in_ch, out_ch):
        buf = (C.c_float * (in_ch * out_ch))()
        _conv_lib().gstamd_audio_converter_get_mix_matrix(self._h, buf, in_ch * out_ch)
        return [[buf[i * out_ch + j] for j in range(out_ch)] for i in range(in_ch)]

    def free(self):
        if self._h:
            _conv_lib().gstamd_audio_converter_free(self._h)
            self._h = None
