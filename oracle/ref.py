"""ctypes binding of oracle/_ref/libgstref.so - the REFERENCE's own code (built by oracle/ref_build.py).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke().
The product package (gstreamer_amd/) never imports this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libgstref.so")

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.ref_init.restype = C.c_int
        L.ref_video_info.restype = C.c_int
        L.ref_video_info.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_size_t), C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.ref_video_converter_new.restype = C.c_void_p
        L.ref_video_converter_new.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p,
                                              C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p,
                                              C.c_char_p]
        L.ref_video_converter_frame.restype = C.c_int
        L.ref_video_converter_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ref_video_converter_bench.restype = C.c_double
        L.ref_video_converter_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.ref_video_converter_free.argtypes = [C.c_void_p]
        L.ref_compositor_blend.restype = C.c_int
        L.ref_compositor_blend.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int]
        L.ref_compositor_fill.restype = C.c_int
        L.ref_compositor_fill.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_audio_converter_new.restype = C.c_void_p
        L.ref_audio_converter_new.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_void_p]
        L.ref_audio_converter_new_positions.restype = C.c_void_p
        L.ref_audio_converter_new_positions.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p,
                                                        C.c_char_p, C.c_void_p]
        L.ref_audio_converter_get_out_frames.restype = C.c_size_t
        L.ref_audio_converter_get_out_frames.argtypes = [C.c_void_p, C.c_size_t]
        L.ref_audio_converter_is_passthrough.argtypes = [C.c_void_p]
        L.ref_audio_converter_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ref_audio_converter_free.argtypes = [C.c_void_p]
        L.ref_audio_converter_reset.argtypes = [C.c_void_p]
        L.ref_audio_resampler_new.restype = C.c_void_p
        L.ref_audio_resampler_new.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
        for n in ("get_out_frames", "get_in_frames"):
            f = getattr(L, "ref_audio_resampler_" + n)
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_size_t]
        L.ref_audio_resampler_get_max_latency.restype = C.c_size_t
        L.ref_audio_resampler_get_max_latency.argtypes = [C.c_void_p]
        L.ref_audio_resampler_resample.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ref_audio_resampler_resample_planar.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_audio_resampler_reset.argtypes = [C.c_void_p]
        L.ref_audio_resampler_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
        L.ref_audio_resampler_free.argtypes = [C.c_void_p]
        L.ref_init()
        _lib = L
    return _lib


def _b(s):
    return s.encode() if isinstance(s, str) and s else None


def video_info(fmt, w, h):
    """Default layout + caps defaults as the reference negotiates them."""
    stride = (C.c_int * 4)()
    off = (C.c_size_t * 4)()
    size = C.c_size_t()
    col = C.create_string_buffer(64)
    chroma = C.create_string_buffer(64)
    n = lib().ref_video_info(fmt.encode(), w, h, stride, off, C.byref(size), col, 64, chroma, 64)
    if n < 0:
        raise ValueError("reference rejects %s %dx%d" % (fmt, w, h))
    return dict(n_planes=n, stride=list(stride), offset=list(off), size=size.value,
                colorimetry=col.value.decode(), chroma_site=chroma.value.decode())


_props = {}


def format_props(fmt):
    """the reference's format table entry: dict(flags, yuv, rgb, gray, alpha, unpack, bits, w_sub, h_sub)"""
    if fmt not in _props:
        L = lib()
        flags, bits, ws, hs = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        name = C.create_string_buffer(32)
        L.ref_video_format_props.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if L.ref_video_format_props(fmt.encode(), C.byref(flags), name, 32, C.byref(bits), C.byref(ws), C.byref(hs)) != 0:
            raise ValueError(fmt)
        f = flags.value          # GstVideoFormatFlags (video-format.h): YUV 1, RGB 2, GRAY 4, ALPHA 8
        _props[fmt] = dict(flags=f, yuv=bool(f & 1), rgb=bool(f & 2), gray=bool(f & 4), alpha=bool(f & 8), unpack=name.value.decode(), bits=bits.value,
                           w_sub=ws.value, h_sub=hs.value)
    return _props[fmt]


def scaler_windows(config, in_size, out_size):
    """(first source line of every output line, taps) of the scaler gst_video_converter_new would make under `config` (a config_string () or None)"""
    L = lib()
    L.ref_video_scaler_windows.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    offs = (C.c_int * out_size)()
    n = C.c_int()
    L.ref_video_scaler_windows(_b(config), in_size, out_size, offs, C.byref(n))
    return list(offs), n.value


def config_string(**opts):
    """Serialise converter options the way gst_structure_from_string wants them.

    Keys use '_' for '-' and '__' for '.': e.g. GstVideoConverter__resampler_method='lanczos'."""
    types = {
        "GstVideoConverter.resampler-method": "GstVideoResamplerMethod",
        "GstVideoConverter.chroma-resampler-method": "GstVideoResamplerMethod",
        "GstVideoConverter.dither-method": "GstVideoDitherMethod",
        "GstVideoConverter.alpha-mode": "GstVideoAlphaMode",
        "GstVideoConverter.chroma-mode": "GstVideoChromaMode",
        "GstVideoConverter.matrix-mode": "GstVideoMatrixMode",
        "GstVideoConverter.gamma-mode": "GstVideoGammaMode",
        "GstVideoConverter.primaries-mode": "GstVideoPrimariesMode",
    }
    parts = ["GstVideoConverter"]
    for k, v in opts.items():
        key = k.replace("__", ".").replace("_", "-")
        if key in types:
            parts.append("%s=(%s)%s" % (key, types[key], v))
        elif isinstance(v, bool):
            parts.append("%s=(boolean)%s" % (key, "true" if v else "false"))
        elif isinstance(v, float):
            parts.append("%s=(double)%r" % (key, v))
        elif key.endswith("threads") or key.endswith("resampler-taps") or key.endswith("quantization") or key.endswith("border-argb"):
            parts.append("%s=(uint)%d" % (key, v))
        else:
            parts.append("%s=(int)%d" % (key, v))
    return ", ".join(parts)


class VideoConverter:
    """gst_video_converter_new/frame of the reference on host numpy buffers."""

    def __init__(self, in_fmt, in_w, in_h, out_fmt, out_w, out_h, in_colorimetry=None, in_chroma_site=None,
                 out_colorimetry=None, out_chroma_site=None, config=None, interlaced=False):
        """interlaced: both infos carry interlace-mode=interleaved, every frame is mapped with GST_VIDEO_FRAME_FLAG_INTERLACED"""
        L = lib()
        L.ref_video_converter_new_interlaced.restype = C.c_void_p
        L.ref_video_converter_new_interlaced.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p,
                                                         C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        self.h = L.ref_video_converter_new_interlaced(_b(in_fmt), in_w, in_h, _b(in_colorimetry), _b(in_chroma_site), None, None,
                                                      _b(out_fmt), out_w, out_h, _b(out_colorimetry), _b(out_chroma_site), None,
                                                      None, _b(config), 1 if interlaced else 0)
        if not self.h:
            raise ValueError("reference could not create converter")
        self.in_info = video_info(in_fmt, in_w, in_h)
        self.out_info = video_info(out_fmt, out_w, out_h)

    def frame(self, src):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        assert src.size == self.in_info["size"], (src.size, self.in_info["size"])
        # slack behind the frame: some of the reference's fastpaths store whole pixel pairs (convert_UYVY_AYUV on an odd width writes one
        # pixel past every line, the last one past the frame) - without it that store lands in the allocator's bookkeeping
        size = self.out_info["size"]
        buf = np.zeros(size + 256, dtype=np.uint8)
        dst = buf[:size]
        r = lib().ref_video_converter_frame(self.h, src.ctypes.data, src.size, dst.ctypes.data, size)
        assert r == 0
        return dst.copy()

    def bench(self, src, n_frames):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.zeros(self.out_info["size"], dtype=np.uint8)
        return lib().ref_video_converter_bench(self.h, src.ctypes.data, src.size, dst.ctypes.data, dst.size, n_frames)

    def close(self):
        if self.h:
            lib().ref_video_converter_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def compositor_blend(func, fmt, src, sw, sh, xpos, ypos, alpha, dst, dw, dh, y0, y1, mode):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    assert dst.flags["C_CONTIGUOUS"] and dst.dtype == np.uint8
    r = lib().ref_compositor_blend(func.encode(), fmt.encode(), src.ctypes.data, src.size, sw, sh, xpos, ypos, float(alpha),
                                   dst.ctypes.data, dst.size, dw, dh, y0, y1, mode)
    assert r == 0, r
    return dst


def compositor_fill(kind, fmt_func, fmt, dst, dw, dh, y0, y1, c1=0, c2=0, c3=0):
    r = lib().ref_compositor_fill(kind, fmt_func.encode(), fmt.encode(), dst.ctypes.data, dst.size, dw, dh, y0, y1, c1, c2, c3)
    assert r == 0, r
    return dst


class AudioResampler:
    """gst_audio_resampler_* of the reference (interleaved in/out)."""
    METHODS = {"nearest": 0, "linear": 1, "cubic": 2, "blackman-nuttall": 3, "kaiser": 4}

    FILTER_MODE = {"interpolated": 0, "full": 1, "auto": 2}
    FILTER_INTERPOLATION = {"none": 0, "linear": 1, "cubic": 2}

    def __init__(self, fmt, channels, in_rate, out_rate, method="kaiser", quality=4, options=None, filter_mode=None,
                 filter_interpolation=None, in_planar=False, out_planar=False):
        self.channels = channels
        self.in_planar, self.out_planar = in_planar, out_planar
        self.fmt = fmt
        self.method = self.METHODS[method]
        self.dtype = {"F32LE": np.float32, "F64LE": np.float64, "S16LE": np.int16, "S32LE": np.int32}[fmt]
        lib().ref_audio_resampler_set_filter(-1 if filter_mode is None else self.FILTER_MODE[filter_mode],
                                              -1 if filter_interpolation is None else self.FILTER_INTERPOLATION[filter_interpolation])
        self.h = lib().ref_audio_resampler_new(self.METHODS[method], (1 if in_planar else 0) | (2 if out_planar else 0), fmt.encode(), channels, in_rate, out_rate, quality,
                                               _b(options))
        lib().ref_audio_resampler_set_filter(-1, -1)
        if not self.h:
            raise ValueError("reference could not create resampler")

    def get_out_frames(self, in_frames):
        return lib().ref_audio_resampler_get_out_frames(self.h, in_frames)

    def get_in_frames(self, out_frames):
        return lib().ref_audio_resampler_get_in_frames(self.h, out_frames)

    def get_max_latency(self):
        return lib().ref_audio_resampler_get_max_latency(self.h)

    def resample(self, data, in_frames=None, out_frames=None):
        """data: interleaved [frames, channels] array or None (silence / drain)."""
        if data is not None:
            data = np.ascontiguousarray(data, dtype=self.dtype)
            in_frames = data.size // self.channels
        if out_frames is None:
            out_frames = self.get_out_frames(in_frames)
        if self.in_planar or self.out_planar:
            # a non-interleaved side is a [channels, frames] array
            out = np.zeros((self.channels, out_frames) if self.out_planar else (out_frames, self.channels), dtype=self.dtype)
            lib().ref_audio_resampler_resample_planar(self.h, data.ctypes.data if data is not None else None, in_frames,
                                                      out.ctypes.data, out_frames, self.channels, out.itemsize,
                                                      int(self.in_planar), int(self.out_planar))
            return out
        out = np.zeros((out_frames, self.channels), dtype=self.dtype)
        lib().ref_audio_resampler_resample(self.h, data.ctypes.data if data is not None else None, in_frames,
                                           out.ctypes.data, out_frames)
        return out

    def update(self, in_rate=0, out_rate=0, quality=None, filter_mode=None, filter_interpolation=None, with_options=None,
               q_rates=None):
        """gst_audio_resampler_update: quality / filter_* given -> new options (set_quality for q_rates), else NULL options."""
        if with_options is None:
            with_options = quality is not None or filter_mode is not None or filter_interpolation is not None
        qi, qo = q_rates if q_rates else (in_rate, out_rate)
        lib().ref_audio_resampler_set_filter(-1 if filter_mode is None else self.FILTER_MODE[filter_mode],
                                              -1 if filter_interpolation is None else self.FILTER_INTERPOLATION[filter_interpolation])
        ok = lib().ref_audio_resampler_update(self.h, in_rate, out_rate, int(bool(with_options)), self.method,
                                              -1 if quality is None else quality, qi, qo, None)
        lib().ref_audio_resampler_set_filter(-1, -1)
        return bool(ok)

    def reset(self):
        lib().ref_audio_resampler_reset(self.h)

    def close(self):
        if self.h:
            lib().ref_audio_resampler_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


AUDIO_BYTES = {"S8": 1, "U8": 1, "S16LE": 2, "S24LE": 3, "S24_32LE": 4, "S32LE": 4, "F32LE": 4, "F64LE": 8}


class AudioConverter:
    """gst_audio_converter_new / _samples of the reference on host numpy byte buffers (interleaved frames)."""

    def __init__(self, in_fmt, in_rate, in_ch, out_fmt, out_rate, out_ch, config=None, mix=None, flags=0, in_pos=None, out_pos=None):
        import numpy as np
        m = None
        if mix is not None:
            m = np.ascontiguousarray(np.asarray(mix, np.float32))
            assert m.shape == (out_ch, in_ch)
        ip = np.asarray(in_pos, np.int32) if in_pos is not None else None
        op = np.asarray(out_pos, np.int32) if out_pos is not None else None
        self.h = lib().ref_audio_converter_new_positions(flags, in_fmt.encode(), in_rate, in_ch, ip.ctypes.data if ip is not None else None,
                                                         out_fmt.encode(), out_rate, out_ch, op.ctypes.data if op is not None else None, _b(config),
                                                         m.ctypes.data if m is not None else None)
        if not self.h:
            raise ValueError("reference could not create the audio converter")
        self.in_bpf, self.out_bpf = AUDIO_BYTES[in_fmt] * in_ch, AUDIO_BYTES[out_fmt] * out_ch

    def get_out_frames(self, in_frames):
        return lib().ref_audio_converter_get_out_frames(self.h, in_frames)

    def is_passthrough(self):
        return bool(lib().ref_audio_converter_is_passthrough(self.h))

    def samples(self, src, in_frames=None):
        """src: uint8 array of whole frames, or None with in_frames (silence into the resampler)."""
        import numpy as np
        n = in_frames if src is None else src.size // self.in_bpf
        out_frames = self.get_out_frames(n)
        out = np.zeros(out_frames * self.out_bpf, np.uint8)
        ok = lib().ref_audio_converter_samples(self.h, src.ctypes.data if src is not None else None, n, out.ctypes.data, out_frames)
        assert ok
        return out

    def reset(self):
        lib().ref_audio_converter_reset(self.h)

    def free(self):
        if self.h:
            lib().ref_audio_converter_free(self.h)
            self.h = None
