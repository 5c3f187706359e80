"""ctypes binding of include/gstamd_video.h (libgstamddsp.so).

Mirrors the reference's library API for this path - GstVideoInfo / gst_video_converter_new /
gst_video_converter_frame (gst-libs/gst/video/video-converter.h:291-316) - with the same argument
meaning.  Frames live in HBM: `frame()` takes device pointers (ints) or torch CUDA tensors.
There is no CPU implementation behind this module; without the native library or without a GPU it
raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libgstamddsp.so")
if os.environ.get("GSTAMD_LIB_PATH"):               # profiling sessions only: a variant built with python -m gstreamer_amd.build -DX=n --suffix=name
    LIB_PATH = os.environ["GSTAMD_LIB_PATH"]
elif os.environ.get("GSTAMD_TUNING_LIB") == "1":         # profiling sessions only: the -DGSTAMD_TUNING build (python -m gstreamer_amd.build --tuning)
    LIB_PATH = os.path.join(HERE, "lib", "libgstamddsp_tuning.so")

FORMATS = {"I420": 2, "YV12": 3, "AYUV": 6, "RGBx": 7, "BGRx": 8, "xRGB": 9, "xBGR": 10, "RGBA": 11, "BGRA": 12,
           "ARGB": 13, "ABGR": 14, "Y42B": 18, "Y444": 20, "NV12": 23, "NV21": 24,
           "YUY2": 4, "UYVY": 5, "RGB": 15, "BGR": 16, "YVYU": 19, "NV16": 51, "NV24": 52, "NV61": 60, "VYUY": 64,
           "GBR": 48, "I420_10LE": 43, "P010_10LE": 62, "ARGB64": 39, "AYUV64": 40, "v308": 28, "IYU2": 63, "VUYA": 84, "I422_10LE": 45, "Y444_10LE": 47,
           "I420_12LE": 73, "I422_12LE": 75, "Y444_12LE": 77, "Y444_16LE": 88, "P016_LE": 90, "P012_LE": 92, "GRAY8": 25, "v210": 21, "Y210": 82, "Y410": 83, "Y212_LE": 94,
           "BGR10A2_LE": 85, "RGB10A2_LE": 86, "BGR10x2_LE": 140, "RGB10x2_LE": 141, "GRAY16_BE": 26, "GRAY16_LE": 27, "ARGB64_LE": 102, "ARGB64_BE": 103,
           "RGBA64_LE": 104, "RGBA64_BE": 105, "BGRA64_LE": 106, "BGRA64_BE": 107, "ABGR64_LE": 108, "ABGR64_BE": 109,
           "RGB16": 29, "BGR16": 30, "RGB15": 31, "BGR15": 32, "A420": 34,
           "GBR_10LE": 50, "GBRA": 65, "GBR_12LE": 69, "Y412_LE": 96, "RGBP": 99, "BGRP": 100, "A422": 117, "A444": 118, "GBR_16LE": 131, "RBGA": 133,
           "Y216_LE": 134, "Y416_LE": 136,
           "A420_10LE": 55, "A422_10LE": 57, "A444_10LE": 59, "GBRA_10LE": 67, "GBRA_12LE": 71, "A444_12LE": 119, "A422_12LE": 121, "A420_12LE": 123,
           "A444_16LE": 125, "A422_16LE": 127, "A420_16LE": 129, "v216": 22, "NV12_10LE40_4L4": 113, "NV12_64Z32": 53, "NV12_4L4": 97, "NV12_32L32": 98, "NV12_16L32S": 110, "NV12_8L128": 111, "RGBA_F16LE": 143, "RGBA_F16BE": 144, "UYVP": 33, "NV12_10LE40": 81, "NV16_10LE40": 139, "GRAY10_LE32": 78, "NV12_10LE32": 79, "NV16_10LE32": 80, "IYU1": 38, "r210": 41, "GRAY10_LE16": 138, "AV12": 101, "Y41B": 17,
           "I420_10BE": 42, "I422_10BE": 44, "Y444_10BE": 46, "GBR_10BE": 49, "A420_10BE": 54, "A422_10BE": 56, "A444_10BE": 58, "P010_10BE": 61, "GBRA_10BE": 66, "GBR_12BE": 68, "GBRA_12BE": 70, "I420_12BE": 72, "I422_12BE": 74, "Y444_12BE": 76, "Y444_16BE": 87, "P016_BE": 89, "P012_BE": 91, "Y212_BE": 93, "Y412_BE": 95, "A444_12BE": 120, "A422_12BE": 122, "A420_12BE": 124, "A444_16BE": 126, "A422_16BE": 128, "A420_16BE": 130, "GBR_16BE": 132, "Y216_BE": 135, "Y416_BE": 137}
COLOR_RANGE = {"unknown": 0, "0-255": 1, "16-235": 2, "0-1": 3}
COLOR_MATRIX = {"unknown": 0, "rgb": 1, "fcc": 2, "bt709": 3, "bt601": 4, "smpte240m": 5, "bt2020": 6}
CHROMA_SITE = {"unknown": 0, "none": 1, "jpeg": 1, "h-cosited": 2, "mpeg2": 2, "v-cosited": 4, "cosited": 6,
               "alt-line": 8, "dv": 14}
RESAMPLER_METHOD = {"nearest": 0, "linear": 1, "cubic": 2, "sinc": 3, "lanczos": 4}
ALPHA_MODE = {"copy": 0, "set": 1, "mult": 2}
CHROMA_MODE = {"full": 0, "upsample-only": 1, "downsample-only": 2, "none": 3}
MATRIX_MODE = {"full": 0, "input-only": 1, "output-only": 2, "none": 3}
# GstVideoTransferFunction / GstVideoColorPrimaries (video-color.h:132-148, 197-209)
TRANSFER = {"unknown": 0, "gamma10": 1, "gamma18": 2, "gamma20": 3, "gamma22": 4, "bt709": 5, "smpte240m": 6, "srgb": 7, "gamma28": 8, "log100": 9,
            "log316": 10, "bt2020-12": 11, "adobergb": 12, "bt2020-10": 13, "smpte2084": 14, "arib-std-b67": 15, "bt601": 16}
PRIMARIES = {"unknown": 0, "bt709": 1, "bt470m": 2, "bt470bg": 3, "smpte170m": 4, "smpte240m": 5, "film": 6, "bt2020": 7, "adobergb": 8,
             "smptest428": 9, "smpterp431": 10, "smpteeg432": 11, "ebu3213": 12}
GAMMA_MODE = {"none": 0, "remap": 1}
PRIMARIES_MODE = {"none": 0, "merge-only": 1, "fast": 2}
# colorimetry strings of video-color.c:72-86 -> (range, matrix, transfer, primaries); "r:m:t:p" with the numbers of the enums also works
COLORIMETRY = {"bt601": ("16-235", "bt601", "bt601", "smpte170m"), "bt709": ("16-235", "bt709", "bt709", "bt709"),
               "smpte240m": ("16-235", "smpte240m", "smpte240m", "smpte240m"), "sRGB": ("0-255", "rgb", "srgb", "bt709"),
               "bt2020": ("16-235", "bt2020", "bt2020-12", "bt2020"), "bt2020-10": ("16-235", "bt2020", "bt2020-10", "bt2020"),
               "bt2100-pq": ("16-235", "bt2020", "smpte2084", "bt2020"), "bt2100-hlg": ("16-235", "bt2020", "arib-std-b67", "bt2020")}

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP = 0, -1, -2, -3


class VideoInfo(C.Structure):
    _fields_ = [("format", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("n_planes", C.c_int32),
                ("stride", C.c_int32 * 4), ("offset", C.c_uint64 * 4), ("size", C.c_uint64),
                ("color_range", C.c_int32), ("color_matrix", C.c_int32), ("chroma_site", C.c_int32),
                ("color_transfer", C.c_int32), ("color_primaries", C.c_int32), ("interlace_mode", C.c_int32), ("frame_height", C.c_int32),
                ("reserved", C.c_int32 * 1)]


class ConverterConfig(C.Structure):
    _fields_ = [("resampler_method", C.c_int32), ("resampler_taps", C.c_uint32), ("max_taps", C.c_int32),
                ("envelope", C.c_double), ("sharpness", C.c_double), ("sharpen", C.c_double),
                ("cubic_b", C.c_double), ("cubic_c", C.c_double), ("alpha_mode", C.c_int32),
                ("alpha_value", C.c_double), ("chroma_mode", C.c_int32), ("matrix_mode", C.c_int32),
                ("dither_quantization", C.c_uint32), ("chroma_resampler_method", C.c_int32), ("dither_method", C.c_int32),
                ("gamma_mode", C.c_int32), ("primaries_mode", C.c_int32), ("internal_flags", C.c_int32), ("reserved", C.c_int32 * 3),
                ("src_x", C.c_int32), ("src_y", C.c_int32), ("src_width", C.c_int32), ("src_height", C.c_int32),
                ("dest_x", C.c_int32), ("dest_y", C.c_int32), ("dest_width", C.c_int32), ("dest_height", C.c_int32),
                ("fill_border", C.c_int32), ("border_argb", C.c_uint32)]


class CompositorPad(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
                ("xpos", C.c_int32), ("ypos", C.c_int32), ("alpha", C.c_double), ("blend_mode", C.c_int32),
                ("reserved", C.c_int32)]


class CompositorPadOpacity(C.Structure):
    """GstAmdCompositorPadOpacity: what is known about a pad's pixel alpha (gstamd_compositor_aggregate_opaque)"""
    _fields_ = [("map", C.c_void_p), ("all_opaque", C.c_int32), ("reserved", C.c_int32)]


class CompositorScaledPad(C.Structure):
    """GstAmdCompositorScaledPad: a pad's frame as it arrives + the converter that scales it (None: blended as it is)"""
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
                ("xpos", C.c_int32), ("ypos", C.c_int32), ("alpha", C.c_double), ("blend_mode", C.c_int32),
                ("reserved", C.c_int32), ("scaler", C.c_void_p)]


class CompositorFramePad(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("stride", C.c_int32 * 3), ("width", C.c_int32), ("height", C.c_int32),
                ("xpos", C.c_int32), ("ypos", C.c_int32), ("alpha", C.c_double), ("blend_mode", C.c_int32), ("reserved", C.c_int32)]


class NativeLibraryMissing(RuntimeError):
    pass


_lib = None


def lib():
    """Load the native library; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing("%s not built - run `python -m gstreamer_amd.build`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.gstamd_last_error.restype = C.c_char_p
        L.gstamd_video_info_set_format.argtypes = [C.POINTER(VideoInfo), C.c_int, C.c_int, C.c_int]
        L.gstamd_video_converter_config_init.argtypes = [C.POINTER(ConverterConfig)]
        L.gstamd_video_converter_new.restype = C.c_void_p
        L.gstamd_video_converter_new.argtypes = [C.POINTER(VideoInfo), C.POINTER(VideoInfo), C.POINTER(ConverterConfig),
                                                 C.POINTER(C.c_int)]
        L.gstamd_video_converter_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gstamd_video_converter_frame_planes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                                          C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_void_p]
        L.gstamd_video_converter_frames.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                    C.c_void_p]
        L.gstamd_video_converter_list_launches.argtypes = [C.c_void_p]
        L.gstamd_video_converter_free.argtypes = [C.c_void_p]
        L.gstamd_video_converter_describe.restype = C.c_char_p
        L.gstamd_video_converter_describe.argtypes = [C.c_void_p]
        L.gstamd_video_converter_divergence.restype = C.c_char_p
        L.gstamd_video_converter_divergence.argtypes = [C.c_void_p]
        L.gstamd_video_converter_set_config.argtypes = [C.c_void_p, C.POINTER(ConverterConfig)]
        L.gstamd_video_converter_get_config.argtypes = [C.c_void_p, C.POINTER(ConverterConfig)]
        L.gstamd_video_converter_algorithmic_bytes.restype = C.c_uint64
        L.gstamd_video_converter_algorithmic_bytes.argtypes = [C.c_void_p]
        L.gstamd_video_converter_debug_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int]
        L.gstamd_device_alloc.restype = C.c_void_p
        L.gstamd_device_alloc.argtypes = [C.c_size_t]
        L.gstamd_device_free.argtypes = [C.c_void_p]
        L.gstamd_device_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gstamd_device_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gstamd_stream_synchronize.argtypes = [C.c_void_p]
        if hasattr(L, "gstamd_compositor_blend"):
            L.gstamd_compositor_blend.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_void_p]
            L.gstamd_compositor_fill_checker.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                         C.c_int, C.c_void_p]
            L.gstamd_compositor_fill_color.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                       C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.gstamd_compositor_aggregate.argtypes = [C.c_int, C.c_int, C.POINTER(CompositorPad), C.c_int, C.c_void_p,
                                                      C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.gstamd_compositor_aggregate_opaque.argtypes = [C.c_int, C.c_int, C.POINTER(CompositorPad), C.POINTER(CompositorPadOpacity), C.c_int,
                                                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.gstamd_compositor_pad_opacity_map.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.gstamd_compositor_aggregate_scaled.argtypes = [C.c_int, C.c_int, C.POINTER(CompositorScaledPad), C.c_int, C.c_void_p,
                                                             C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.gstamd_compositor_pad_scaler_usable.argtypes = [C.c_void_p]
            L.gstamd_compositor_aggregate_frame.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                            C.POINTER(CompositorFramePad), C.c_int, C.POINTER(C.c_void_p),
                                                            C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class tuning:
    """with V.tuning(GSTAMD_NO_H420_REG=1): ...   - sets knobs of the library's tuning table (gstamd_tuning_set) for the block"""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        L = lib()
        L.gstamd_tuning_set.argtypes = [C.c_char_p, C.c_int]
        L.gstamd_tuning_get.argtypes = [C.c_char_p]
        self.old = {k: L.gstamd_tuning_get(k.encode()) for k in self.knobs}
        for k, v in self.knobs.items():
            assert L.gstamd_tuning_set(k.encode(), int(v)) == 0, k
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            lib().gstamd_tuning_set(k.encode(), v)
        return False


def last_error():
    return lib().gstamd_last_error().decode()


class GstAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (status %d)" % (msg, code))
        self.code = code


def _check(code):
    if code != OK:
        raise GstAmdError(code, last_error())


def video_info(fmt, width, height, colorimetry=None, chroma_site=None, stride=None, offset=None):
    """GstVideoInfo as the elements negotiate it (defaults by height), optional overrides."""
    info = VideoInfo()
    _check(lib().gstamd_video_info_set_format(C.byref(info), FORMATS[fmt], width, height))
    if colorimetry:
        if colorimetry in COLORIMETRY:
            rng, mtx, trc, prim = COLORIMETRY[colorimetry]
            info.color_range, info.color_matrix = COLOR_RANGE[rng], COLOR_MATRIX[mtx]
            info.color_transfer, info.color_primaries = TRANSFER[trc], PRIMARIES[prim]
        else:                                   # "range:matrix:transfer:primaries" (gst_video_colorimetry_from_string's numeric form)
            info.color_range, info.color_matrix, info.color_transfer, info.color_primaries = [int(v) for v in colorimetry.split(":")]
        # gst_video_info_from_caps (video-info.c:580-586): "force the 0_1 range for float formats, other ranges are reserved"
        if fmt.startswith("RGBA_F") and info.color_range not in (0, COLOR_RANGE["0-1"]):
            info.color_range = COLOR_RANGE["0-1"]
    if chroma_site:
        info.chroma_site = CHROMA_SITE[chroma_site]
    if stride is not None:
        for i, s in enumerate(stride):
            info.stride[i] = s
    if offset is not None:
        for i, o in enumerate(offset):
            info.offset[i] = o
    return info


DITHER_METHOD = {"none": 0, "verterr": 1, "floyd-steinberg": 2, "sierra-lite": 3, "bayer": 4}


def converter_config(**kw):
    """Library defaults of video-converter.c:778-796; keyword overrides use the field names."""
    cfg = ConverterConfig()
    lib().gstamd_video_converter_config_init(C.byref(cfg))
    enums = {"resampler_method": RESAMPLER_METHOD, "alpha_mode": ALPHA_MODE, "chroma_mode": CHROMA_MODE,
             "matrix_mode": MATRIX_MODE, "dither_method": DITHER_METHOD, "gamma_mode": GAMMA_MODE, "primaries_mode": PRIMARIES_MODE}
    for k, v in kw.items():
        if k in enums and isinstance(v, str):
            v = enums[k][v]
        setattr(cfg, k, v)
    return cfg


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):        # torch tensor in HBM
        if not x.is_cuda:
            raise ValueError("frames must be device (HBM) tensors")
        return x.data_ptr()
    raise TypeError("expected device pointer or CUDA tensor")


TEST_PATTERNS = {"smpte": 0, "snow": 1, "black": 2, "white": 3, "red": 4, "green": 5, "blue": 6, "checkers-1": 7, "checkers-2": 8, "checkers-4": 9,
                 "checkers-8": 10, "circular": 11, "blink": 12, "smpte75": 13, "zone-plate": 14, "gamut": 15, "chroma-zone-plate": 16, "solid-color": 17,
                 "ball": 18, "smpte100": 19, "bar": 20, "pinwheel": 21, "spokes": 22, "gradient": 23, "colors": 24, "smpte-rp-219": 25}


class VideoTestPattern:
    """gstamd_video_test_pattern_*: GstVideoTestSrc's frames painted in HBM (include/gstamd_video.h)"""

    def __init__(self, info, pattern, foreground=0xffffffff, background=0xff000000):
        L = lib()
        L.gstamd_video_test_pattern_new.restype = C.c_void_p
        L.gstamd_video_test_pattern_new.argtypes = [C.POINTER(VideoInfo), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]
        L.gstamd_video_test_pattern_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.gstamd_video_test_pattern_free.argtypes = [C.c_void_p]
        L.gstamd_video_test_pattern_free.restype = None
        L.gstamd_video_test_pattern_describe.argtypes = [C.c_void_p]
        L.gstamd_video_test_pattern_describe.restype = C.c_char_p
        status = C.c_int(0)
        self._h = L.gstamd_video_test_pattern_new(C.byref(info), TEST_PATTERNS[pattern] if isinstance(pattern, str) else pattern, foreground, background, C.byref(status))
        if not self._h:
            raise GstAmdError(status.value, last_error())

    def frame(self, n_frames, dest, stream=None):
        _check(lib().gstamd_video_test_pattern_frame(self._h, n_frames, _ptr(dest), stream))

    def describe(self):
        return lib().gstamd_video_test_pattern_describe(self._h).decode()

    def free(self):
        if self._h:
            lib().gstamd_video_test_pattern_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class VideoConverter:
    """gst_video_converter_new(in_info, out_info, config) -> .frame(src, dest)."""

    def __init__(self, in_info, out_info, config=None):
        status = C.c_int(0)
        self.in_info, self.out_info = in_info, out_info
        self._h = lib().gstamd_video_converter_new(C.byref(in_info), C.byref(out_info),
                                                   C.byref(config) if config is not None else None, C.byref(status))
        if not self._h:
            raise GstAmdError(status.value, last_error())

    def frame(self, src, dest, stream=None):
        _check(lib().gstamd_video_converter_frame(self._h, _ptr(src), _ptr(dest), stream))

    def frame_planes(self, src_planes, src_strides, dest_planes, dest_strides, stream=None):
        """One device pointer and pitch per plane, in the FRAME's plane order (GstVideoFrame.data[]: GBR's planes are G, B, R)."""
        def arr(t, xs):
            xs = list(xs) + [0] * (4 - len(xs))
            return (t * 4)(*xs)
        sp = arr(C.c_void_p, [_ptr(x) for x in src_planes])
        dp = arr(C.c_void_p, [_ptr(x) for x in dest_planes])
        ss = arr(C.c_int32, src_strides) if src_strides is not None else None
        ds = arr(C.c_int32, dest_strides) if dest_strides is not None else None
        _check(lib().gstamd_video_converter_frame_planes(self._h, sp, ss, dp, ds, stream))

    def frames(self, srcs, dests, stream=None):
        """Convert a list of frames (GstBufferList analogue) - one launch when the plan allows."""
        n = len(srcs)
        assert n == len(dests)
        sp = (C.c_void_p * n)(*[_ptr(x) for x in srcs])
        dp = (C.c_void_p * n)(*[_ptr(x) for x in dests])
        _check(lib().gstamd_video_converter_frames(self._h, n, sp, dp, stream))

    def list_launches(self):
        """launches of the last frames() call that each served a whole list (0: the frames went one by one)"""
        return int(lib().gstamd_video_converter_list_launches(self._h))

    def describe(self):
        return lib().gstamd_video_converter_describe(self._h).decode()

    def divergence(self):
        """"" when the plan reproduces the reference bit for bit; else why not (the reference's own output is undefined there)"""
        return lib().gstamd_video_converter_divergence(self._h).decode()

    def set_config(self, config):
        """gst_video_converter_set_config: re-plan with new options; a refused config leaves the converter as it was"""
        _check(lib().gstamd_video_converter_set_config(self._h, C.byref(config)))

    def algorithmic_bytes(self):
        return lib().gstamd_video_converter_algorithmic_bytes(self._h)

    def debug_get(self, what):
        n = lib().gstamd_video_converter_debug_get(self._h, what, None, 0)
        if n < 0:
            return None
        buf = (C.c_int32 * max(n, 1))()
        lib().gstamd_video_converter_debug_get(self._h, what, buf, n)
        return list(buf[:n])

    def free(self):
        if self._h:
            lib().gstamd_video_converter_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
