"""ctypes binding of include/gstamd_audio.h - mirrors gst_audio_resampler_* (audio-resampler.h:218-256).

Sample buffers live in HBM (device pointers or torch CUDA tensors).  No CPU implementation."""
import ctypes as C

from . import video as _v

METHODS = {"nearest": 0, "linear": 1, "cubic": 2, "blackman-nuttall": 3, "kaiser": 4}
FORMATS = {"S16LE": 0, "S32LE": 1, "F32LE": 2, "F64LE": 3}
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
