/* gstamd_video.h - C ABI of the MI355X-native GstVideoConverter / compositor-blend replacement.
 *
 * This is the drop-in boundary for the video half of the hot path (SURVEY.md section 8b,
 * "Library-level C ABI").  Every entry point names the reference interface it replaces
 * (paths relative to /root/reference/subprojects/gst-plugins-base):
 *
 *   gstamd_video_converter_new      <- gst_video_converter_new        gst-libs/gst/video/video-converter.h:291
 *   gstamd_video_converter_frame    <- gst_video_converter_frame      gst-libs/gst/video/video-converter.h:313
 *   gstamd_video_converter_free     <- gst_video_converter_free       gst-libs/gst/video/video-converter.h:304
 *   gstamd_video_info_set_format    <- gst_video_info_set_format +    gst-libs/gst/video/video-info.c:890-1100
 *                                      the caps defaults of           gst-libs/gst/video/video-info.c:155-225, 540-600
 *   gstamd_compositor_*             <- BlendFunction / FillChecker /  gst/compositor/blend.h:50-52,
 *                                      blend_pads / _draw_background  gst/compositor/compositor.c:1619-1697
 *
 * Plain C types only: pointers, sizes, ints, doubles.  All frame pointers are DEVICE pointers
 * (HBM, e.g. from gstamd_device_alloc, a GstAmdHipMemory or a torch tensor's data_ptr());
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue work on
 * `stream` and return without synchronising, exactly like a HIP kernel launch.
 *
 * Integer enum values are numerically identical to the reference's public enums
 * (GstVideoFormat video-format.h:195-, GstVideoColorRange/Matrix/... video-color.h,
 * GstVideoChromaSite video-chroma.h:43-52, GstVideoResamplerMethod video-resampler.h:45-49,
 * GstVideoAlphaMode/ChromaMode/MatrixMode video-converter.h) so a binding can pass them through.
 */
#ifndef GSTAMD_VIDEO_H
#define GSTAMD_VIDEO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSTAMD_VIDEO_MAX_PLANES 4

/* GstVideoFormat subset (same numeric values as video-format.h:195-) */
enum {
  GSTAMD_VIDEO_FORMAT_UNKNOWN = 0,
  GSTAMD_VIDEO_FORMAT_I420 = 2,
  GSTAMD_VIDEO_FORMAT_YV12 = 3,
  GSTAMD_VIDEO_FORMAT_YUY2 = 4,
  GSTAMD_VIDEO_FORMAT_UYVY = 5,
  GSTAMD_VIDEO_FORMAT_AYUV = 6,
  GSTAMD_VIDEO_FORMAT_RGBx = 7,
  GSTAMD_VIDEO_FORMAT_BGRx = 8,
  GSTAMD_VIDEO_FORMAT_xRGB = 9,
  GSTAMD_VIDEO_FORMAT_xBGR = 10,
  GSTAMD_VIDEO_FORMAT_RGBA = 11,
  GSTAMD_VIDEO_FORMAT_BGRA = 12,
  GSTAMD_VIDEO_FORMAT_ARGB = 13,
  GSTAMD_VIDEO_FORMAT_ABGR = 14,
  GSTAMD_VIDEO_FORMAT_RGB = 15,
  GSTAMD_VIDEO_FORMAT_BGR = 16,
  GSTAMD_VIDEO_FORMAT_Y41B = 17,       /* planar 4:1:1 */
  GSTAMD_VIDEO_FORMAT_Y42B = 18,
  GSTAMD_VIDEO_FORMAT_YVYU = 19,
  GSTAMD_VIDEO_FORMAT_Y444 = 20,
  GSTAMD_VIDEO_FORMAT_NV12 = 23,
  GSTAMD_VIDEO_FORMAT_GRAY8 = 25,       /* one plane of luma; unpacks to A = 0xff, Y, U = V = 0x80 (video-format.c:1207-1229) */
  GSTAMD_VIDEO_FORMAT_GRAY16_BE = 26,   /* one plane of 16-bit luma, big / little endian; unpacks to AYUV64 with U = V = 0x8000 */
  GSTAMD_VIDEO_FORMAT_GRAY16_LE = 27,
  GSTAMD_VIDEO_FORMAT_v308 = 28,
  GSTAMD_VIDEO_FORMAT_RGB16 = 29,       /* one little-endian 16-bit word per pixel: R 5, G 6, B 5 from the high bits; BGR16 the other way round */
  GSTAMD_VIDEO_FORMAT_BGR16 = 30,
  GSTAMD_VIDEO_FORMAT_RGB15 = 31,       /* x 1, R 5, G 5, B 5; BGR15 the other way round */
  GSTAMD_VIDEO_FORMAT_BGR15 = 32,        /* packed 4:4:4, 3 bytes per pixel: Y U V */
  GSTAMD_VIDEO_FORMAT_IYU2 = 63,        /* the same in the order U Y V */
  GSTAMD_VIDEO_FORMAT_VUYA = 84,        /* packed 4:4:4:4, 4 bytes per pixel: V U Y A */
  GSTAMD_VIDEO_FORMAT_NV21 = 24,
  GSTAMD_VIDEO_FORMAT_GBR = 48,         /* planar 8-bit RGB, planes in the order G, B, R */
  GSTAMD_VIDEO_FORMAT_GBR_10LE = 50,    /* GBR with 10 / 12 / 16 bits in little-endian 16-bit words (GBR_12LE = 69, GBR_16LE = 131) */
  GSTAMD_VIDEO_FORMAT_NV16 = 51,
  GSTAMD_VIDEO_FORMAT_NV24 = 52,
  GSTAMD_VIDEO_FORMAT_A420 = 34,        /* I420 with a fourth, full-size plane of alpha */
  GSTAMD_VIDEO_FORMAT_IYU1 = 38,        /* packed 4:1:1: six bytes U Y0 Y1 V Y2 Y3 per group of four pixels */
  GSTAMD_VIDEO_FORMAT_GRAY10_LE32 = 78, /* one plane of luma, three 10-bit samples per little-endian 32-bit word (2 bits of padding on top) */
  GSTAMD_VIDEO_FORMAT_NV12_10LE32 = 79, /* NV12 with both planes packed that way (the UV plane: U V U | V U V) */
  GSTAMD_VIDEO_FORMAT_NV16_10LE32 = 80, /* NV16 likewise */
  GSTAMD_VIDEO_FORMAT_NV12_10LE40 = 81, /* NV12 with fully packed 10-bit samples: a little-endian bit stream, four samples in five bytes */
  GSTAMD_VIDEO_FORMAT_NV16_10LE40 = 139,/* NV16 likewise */
  GSTAMD_VIDEO_FORMAT_UYVP = 33,        /* packed 4:2:2, 10 bits: U Y0 V Y1 as a big-endian bit stream, five bytes per two pixels */
  GSTAMD_VIDEO_FORMAT_RGBA_F16LE = 143, /* four IEEE half floats per pixel, R G B A, little endian (0.0 .. 1.0 = the 16-bit chain's 0 .. 65535) */
  GSTAMD_VIDEO_FORMAT_RGBA_F16BE = 144, /* the same big endian */
  GSTAMD_VIDEO_FORMAT_NV12_64Z32 = 53,  /* NV12 in 64 x 32 tiles, zigzag order (GST_VIDEO_TILE_MODE_ZFLIPZ_2X2); stride[] holds the tile counts (y tiles << 16 | x tiles) */
  GSTAMD_VIDEO_FORMAT_NV12_4L4 = 97,    /* NV12 in 4 x 4 tiles, linear order */
  GSTAMD_VIDEO_FORMAT_NV12_32L32 = 98,  /* 32 x 32 */
  GSTAMD_VIDEO_FORMAT_NV12_16L32S = 110,/* 16 x 32, the UV plane in 16 x 16 sub-tiles */
  GSTAMD_VIDEO_FORMAT_NV12_8L128 = 111, /* 8 x 128 */
  GSTAMD_VIDEO_FORMAT_NV12_10LE40_4L4 = 113, /* NV12_10LE40's bit stream in 4 x 4 tiles (five bytes a tile row), linear order */
  GSTAMD_VIDEO_FORMAT_v216 = 22,        /* packed 4:2:2, little-endian 16-bit words U Y0 V Y1 */
  GSTAMD_VIDEO_FORMAT_r210 = 41,        /* one big-endian 32-bit word per pixel: x 2, R 10, G 10, B 10 */
  GSTAMD_VIDEO_FORMAT_GRAY10_LE16 = 138,/* one plane of luma, 10 bits in the low bits of little-endian words */
  GSTAMD_VIDEO_FORMAT_ARGB64 = 39,      /* 16 bits per component, native endianness (little endian here), memory order A R G B */
  GSTAMD_VIDEO_FORMAT_AYUV64 = 40,      /* the same with A Y U V */
  GSTAMD_VIDEO_FORMAT_I420_10LE = 43,   /* 10 bits in the low bits of little-endian 16-bit words; see DESIGN.md 3.6 / 3.7 for the combinations */
  GSTAMD_VIDEO_FORMAT_I422_10LE = 45,   /* planar 4:2:2 and 4:4:4 with the same 10-bit samples */
  GSTAMD_VIDEO_FORMAT_Y444_10LE = 47,
  GSTAMD_VIDEO_FORMAT_I420_12LE = 73,   /* 12 bits in the low bits */
  GSTAMD_VIDEO_FORMAT_I422_12LE = 75,
  GSTAMD_VIDEO_FORMAT_Y444_12LE = 77,
  GSTAMD_VIDEO_FORMAT_Y444_16LE = 88,   /* all 16 bits */
  GSTAMD_VIDEO_FORMAT_P016_LE = 90,
  GSTAMD_VIDEO_FORMAT_P012_LE = 92,     /* 12 bits in the high bits */
  GSTAMD_VIDEO_FORMAT_v210 = 21,      /* packed 4:2:2, 10 bits: six pixels in four little-endian 32-bit words (three samples a word) */
  GSTAMD_VIDEO_FORMAT_Y210 = 82,      /* packed 4:2:2, 16-bit little-endian words Y0 U Y1 V, 10 bits in the high bits */
  GSTAMD_VIDEO_FORMAT_Y410 = 83,      /* packed 4:4:4 in one little-endian 32-bit word: U 10, Y 10, V 10, A 2 (from the low bits) */
  GSTAMD_VIDEO_FORMAT_BGR10A2_LE = 85,  /* one little-endian 32-bit word per pixel: B 10, G 10, R 10, A 2 (from the low bits) */
  GSTAMD_VIDEO_FORMAT_RGB10A2_LE = 86,  /* the same with R 10, G 10, B 10, A 2 */
  GSTAMD_VIDEO_FORMAT_BGR10x2_LE = 140, /* BGR10A2_LE's word declared without an alpha component (video-format.c:8504: the same pack / unpack functions) */
  GSTAMD_VIDEO_FORMAT_RGB10x2_LE = 141, /* RGB10A2_LE's word, likewise */
  GSTAMD_VIDEO_FORMAT_Y212_LE = 94,
  GSTAMD_VIDEO_FORMAT_A420_10LE = 55,   /* A420 / A422 / A444 with 10, 12 (A444_12LE = 119, A422_12LE = 121, A420_12LE = 123) or 16 bits (125, 127, 129) in LE words */
  GSTAMD_VIDEO_FORMAT_A422_10LE = 57,
  GSTAMD_VIDEO_FORMAT_A444_10LE = 59,
  GSTAMD_VIDEO_FORMAT_GBRA = 65,
  GSTAMD_VIDEO_FORMAT_GBRA_10LE = 67,
  GSTAMD_VIDEO_FORMAT_GBRA_12LE = 71,
  GSTAMD_VIDEO_FORMAT_A444_12LE = 119,
  GSTAMD_VIDEO_FORMAT_A422_12LE = 121,
  GSTAMD_VIDEO_FORMAT_A420_12LE = 123,
  GSTAMD_VIDEO_FORMAT_A444_16LE = 125,
  GSTAMD_VIDEO_FORMAT_A422_16LE = 127,
  GSTAMD_VIDEO_FORMAT_A420_16LE = 129,        /* planar 8-bit RGB with alpha, planes G, B, R, A */
  GSTAMD_VIDEO_FORMAT_GBR_12LE = 69,
  GSTAMD_VIDEO_FORMAT_Y412_LE = 96,     /* packed 4:4:4:4, four little-endian 16-bit words U Y V A, 12 bits in the high bits */
  GSTAMD_VIDEO_FORMAT_RGBP = 99,        /* planar 8-bit RGB, planes R, G, B / B, G, R */
  GSTAMD_VIDEO_FORMAT_BGRP = 100,
  GSTAMD_VIDEO_FORMAT_A422 = 117,       /* Y42B / Y444 with a fourth, full-size plane of alpha */
  GSTAMD_VIDEO_FORMAT_A444 = 118,
  GSTAMD_VIDEO_FORMAT_GBR_16LE = 131,
  GSTAMD_VIDEO_FORMAT_RBGA = 133,       /* packed 4-byte RGB with alpha, bytes R, B, G, A */
  GSTAMD_VIDEO_FORMAT_Y216_LE = 134,    /* Y210's layout with all 16 bits */
  GSTAMD_VIDEO_FORMAT_Y416_LE = 136,    /* Y412_LE's layout with all 16 bits */
  GSTAMD_VIDEO_FORMAT_AV12 = 101,       /* NV12 with a third, full-size plane of alpha */
  GSTAMD_VIDEO_FORMAT_ARGB64_LE = 102,  /* 16 bits per component in the named memory order and endianness (ARGB64 is ARGB64_LE on this host) */
  GSTAMD_VIDEO_FORMAT_ARGB64_BE = 103,
  GSTAMD_VIDEO_FORMAT_RGBA64_LE = 104,
  GSTAMD_VIDEO_FORMAT_RGBA64_BE = 105,
  GSTAMD_VIDEO_FORMAT_BGRA64_LE = 106,
  GSTAMD_VIDEO_FORMAT_BGRA64_BE = 107,
  GSTAMD_VIDEO_FORMAT_ABGR64_LE = 108,
  GSTAMD_VIDEO_FORMAT_ABGR64_BE = 109,   /* Y210's layout with 12 bits */
  /* big-endian forms of the 10 / 12 / 16-bit word formats (round 5) */
  GSTAMD_VIDEO_FORMAT_I420_10BE = 42,
  GSTAMD_VIDEO_FORMAT_I422_10BE = 44,
  GSTAMD_VIDEO_FORMAT_Y444_10BE = 46,
  GSTAMD_VIDEO_FORMAT_GBR_10BE = 49,
  GSTAMD_VIDEO_FORMAT_A420_10BE = 54,
  GSTAMD_VIDEO_FORMAT_A422_10BE = 56,
  GSTAMD_VIDEO_FORMAT_A444_10BE = 58,
  GSTAMD_VIDEO_FORMAT_P010_10BE = 61,
  GSTAMD_VIDEO_FORMAT_GBRA_10BE = 66,
  GSTAMD_VIDEO_FORMAT_GBR_12BE = 68,
  GSTAMD_VIDEO_FORMAT_GBRA_12BE = 70,
  GSTAMD_VIDEO_FORMAT_I420_12BE = 72,
  GSTAMD_VIDEO_FORMAT_I422_12BE = 74,
  GSTAMD_VIDEO_FORMAT_Y444_12BE = 76,
  GSTAMD_VIDEO_FORMAT_Y444_16BE = 87,
  GSTAMD_VIDEO_FORMAT_P016_BE = 89,
  GSTAMD_VIDEO_FORMAT_P012_BE = 91,
  GSTAMD_VIDEO_FORMAT_Y212_BE = 93,
  GSTAMD_VIDEO_FORMAT_Y412_BE = 95,
  GSTAMD_VIDEO_FORMAT_A444_12BE = 120,
  GSTAMD_VIDEO_FORMAT_A422_12BE = 122,
  GSTAMD_VIDEO_FORMAT_A420_12BE = 124,
  GSTAMD_VIDEO_FORMAT_A444_16BE = 126,
  GSTAMD_VIDEO_FORMAT_A422_16BE = 128,
  GSTAMD_VIDEO_FORMAT_A420_16BE = 130,
  GSTAMD_VIDEO_FORMAT_GBR_16BE = 132,
  GSTAMD_VIDEO_FORMAT_Y216_BE = 135,
  GSTAMD_VIDEO_FORMAT_Y416_BE = 137,
  GSTAMD_VIDEO_FORMAT_NV61 = 60,
  GSTAMD_VIDEO_FORMAT_P010_10LE = 62,   /* 10 bits in the high bits of little-endian 16-bit words */
  GSTAMD_VIDEO_FORMAT_VYUY = 64
};

enum { GSTAMD_COLOR_RANGE_UNKNOWN = 0, GSTAMD_COLOR_RANGE_0_255 = 1, GSTAMD_COLOR_RANGE_16_235 = 2, GSTAMD_COLOR_RANGE_0_1 = 3 /* the float formats': full-range code values */ };
enum {
  GSTAMD_COLOR_MATRIX_UNKNOWN = 0, GSTAMD_COLOR_MATRIX_RGB = 1, GSTAMD_COLOR_MATRIX_FCC = 2,
  GSTAMD_COLOR_MATRIX_BT709 = 3, GSTAMD_COLOR_MATRIX_BT601 = 4, GSTAMD_COLOR_MATRIX_SMPTE240M = 5,
  GSTAMD_COLOR_MATRIX_BT2020 = 6
};
enum {
  GSTAMD_CHROMA_SITE_UNKNOWN = 0, GSTAMD_CHROMA_SITE_NONE = 1, GSTAMD_CHROMA_SITE_H_COSITED = 2,
  GSTAMD_CHROMA_SITE_V_COSITED = 4, GSTAMD_CHROMA_SITE_ALT_LINE = 8
};
enum {
  GSTAMD_RESAMPLER_METHOD_NEAREST = 0, GSTAMD_RESAMPLER_METHOD_LINEAR = 1,
  GSTAMD_RESAMPLER_METHOD_CUBIC = 2, GSTAMD_RESAMPLER_METHOD_SINC = 3,
  GSTAMD_RESAMPLER_METHOD_LANCZOS = 4
};
enum { GSTAMD_ALPHA_MODE_COPY = 0, GSTAMD_ALPHA_MODE_SET = 1, GSTAMD_ALPHA_MODE_MULT = 2 };
enum {
  GSTAMD_CHROMA_MODE_FULL = 0, GSTAMD_CHROMA_MODE_UPSAMPLE_ONLY = 1,
  GSTAMD_CHROMA_MODE_DOWNSAMPLE_ONLY = 2, GSTAMD_CHROMA_MODE_NONE = 3
};
enum {
  GSTAMD_MATRIX_MODE_FULL = 0, GSTAMD_MATRIX_MODE_INPUT_ONLY = 1,
  GSTAMD_MATRIX_MODE_OUTPUT_ONLY = 2, GSTAMD_MATRIX_MODE_NONE = 3
};

enum { GSTAMD_GAMMA_MODE_NONE = 0, GSTAMD_GAMMA_MODE_REMAP = 1 };
enum { GSTAMD_PRIMARIES_MODE_NONE = 0, GSTAMD_PRIMARIES_MODE_MERGE_ONLY = 1, GSTAMD_PRIMARIES_MODE_FAST = 2 };
/* GstVideoTransferFunction / GstVideoColorPrimaries values used by the defaults (the fields take any value of the reference's enums) */
enum { GSTAMD_TRANSFER_UNKNOWN = 0, GSTAMD_TRANSFER_BT709 = 5, GSTAMD_TRANSFER_SRGB = 7, GSTAMD_TRANSFER_BT601 = 16 };
enum { GSTAMD_PRIMARIES_UNKNOWN = 0, GSTAMD_PRIMARIES_BT709 = 1, GSTAMD_PRIMARIES_SMPTE170M = 4 };

/* status codes (0 = ok).  The library never falls back to a CPU path: a conversion it does not
 * implement on the GPU is refused with GSTAMD_ERR_UNSUPPORTED, mirroring the reference returning
 * NULL from gst_video_converter_new for an impossible conversion (video-converter.c:2543-2562). */
enum {
  GSTAMD_OK = 0,
  GSTAMD_ERR_INVALID = -1,
  GSTAMD_ERR_UNSUPPORTED = -2,
  GSTAMD_ERR_HIP = -3
};

/* Mirror of the GstVideoInfo fields the converter reads (video-info.h:400-440). */
typedef struct GstAmdVideoInfo {
  int32_t format;                               /* GSTAMD_VIDEO_FORMAT_* */
  int32_t width, height;
  int32_t n_planes;
  int32_t stride[GSTAMD_VIDEO_MAX_PLANES];      /* pitch in bytes */
  uint64_t offset[GSTAMD_VIDEO_MAX_PLANES];     /* plane offsets from the frame base */
  uint64_t size;                                /* total frame bytes for the default layout */
  int32_t color_range;                          /* GstVideoColorRange */
  int32_t color_matrix;                         /* GstVideoColorMatrix */
  int32_t chroma_site;                          /* GstVideoChromaSite flags */
  int32_t color_transfer;                       /* GstVideoTransferFunction (video-color.h:132-148); read with gamma-mode = remap */
  int32_t color_primaries;                      /* GstVideoColorPrimaries (video-color.h:197-209); read with primaries-mode != none */
  int32_t interlace_mode;                       /* GSTAMD_INTERLACE_MODE_* (GstVideoInterlaceMode, video-info.h:38-58); 0 = progressive.  Both infos of a
                                                 * converter carry the same mode (gst_video_converter_new, video-converter.c:2435).  With _INTERLEAVED every
                                                 * frame is converted as two fields (GST_VIDEO_FRAME_IS_INTERLACED: field-aware 4:2:0 lines, chroma
                                                 * resampling and vertical scaling, video-converter.c:3303, 3383, 1651, 7977) */
  int32_t frame_height;                         /* 0 for callers (the library's own field conversions: height of the frame a field belongs to) */
  int32_t reserved[1];
} GstAmdVideoInfo;

/* GstVideoInterlaceMode.  _MIXED streams flag each buffer (GST_VIDEO_BUFFER_FLAG_INTERLACED): the caller converts flagged frames with a converter
 * made for _INTERLEAVED infos and the others with one made for progressive infos (what gst_video_frame_map + video_converter_generic :3303 do per
 * frame); _FIELDS and _ALTERNATE are refused (GSTAMD_ERR_UNSUPPORTED). */
#define GSTAMD_INTERLACE_MODE_PROGRESSIVE 0
#define GSTAMD_INTERLACE_MODE_INTERLEAVED 1
#define GSTAMD_INTERLACE_MODE_MIXED 2
#define GSTAMD_INTERLACE_MODE_FIELDS 3
#define GSTAMD_INTERLACE_MODE_ALTERNATE 4
/* (internal: the two field conversions an interleaved frame is split into) */
#define GSTAMD_INTERLACE_FIELD_TOP 16
#define GSTAMD_INTERLACE_FIELD_BOTTOM 17

/* Mirror of the GstVideoConverter option keys (video-converter.h:34-286) that this
 * implementation honours; gstamd_video_converter_config_init() sets the library defaults of
 * video-converter.c:778-796 (NB: resampler-method CUBIC; the videoconvertscale element itself
 * defaults to LINEAR with max-taps 2, gstvideoconvertscale.c:1000-1005). */
/* GstVideoDitherMethod (video-dither.h:44-50) */
#define GSTAMD_DITHER_NONE 0
#define GSTAMD_DITHER_VERTERR 1
#define GSTAMD_DITHER_FLOYD_STEINBERG 2
#define GSTAMD_DITHER_SIERRA_LITE 3
#define GSTAMD_DITHER_BAYER 4

typedef struct GstAmdVideoConverterConfig {
  int32_t resampler_method;      /* GstVideoConverter.resampler-method */
  uint32_t resampler_taps;       /* GstVideoConverter.resampler-taps (0 = auto) */
  int32_t max_taps;              /* GstVideoResampler.max-taps (default 128) */
  double envelope;               /* GstVideoResampler.envelope (2.0) */
  double sharpness;              /* GstVideoResampler.sharpness (1.0) */
  double sharpen;                /* GstVideoResampler.sharpen (0.0) */
  double cubic_b, cubic_c;       /* GstVideoResampler.cubic-b/-c (1/3, 1/3) */
  int32_t alpha_mode;            /* GstVideoConverter.alpha-mode */
  double alpha_value;            /* GstVideoConverter.alpha-value (1.0) */
  int32_t chroma_mode;           /* GstVideoConverter.chroma-mode */
  int32_t matrix_mode;           /* GstVideoConverter.matrix-mode */
  uint32_t dither_quantization;  /* GstVideoConverter.dither-quantization (1) */
  int32_t chroma_resampler_method; /* GstVideoConverter.chroma-resampler-method (LINEAR): chroma planes of the plane scaler */
  int32_t dither_method;         /* GstVideoConverter.dither-method (BAYER); only matters with dither-quantization > 1 on this path */
  int32_t gamma_mode;            /* GstVideoConverter.gamma-mode: GSTAMD_GAMMA_MODE_NONE (default) / _REMAP */
  int32_t primaries_mode;        /* GstVideoConverter.primaries-mode: GSTAMD_PRIMARIES_MODE_NONE (default) / _MERGE_ONLY / _FAST */
  int32_t internal_flags;        /* 0 for callers; the library's own sub-conversions of the gamma chain set bit 0 (generic chain only) */
  int32_t reserved[3];
  /* source crop and destination rectangle (GstVideoConverter.src-x/-y/-width/-height, dest-x/-y/-width/-height,
   * video-converter.h:64-131); width / height 0 = "to the frame's edge" (the option absent) */
  int32_t src_x, src_y, src_width, src_height;
  int32_t dest_x, dest_y, dest_width, dest_height;
  int32_t fill_border;           /* GstVideoConverter.fill-border (TRUE) */
  uint32_t border_argb;          /* GstVideoConverter.border-argb (0xff000000) */
} GstAmdVideoConverterConfig;

typedef struct GstAmdVideoConverter GstAmdVideoConverter;

/* Fill `info` like the elements do: gst_video_info_set_format's default pitch-linear layout plus
 * the caps defaults (colorimetry by height, chroma-site by height).  Returns GSTAMD_OK or an error. */
int gstamd_video_info_set_format (GstAmdVideoInfo *info, int format, int width, int height);

void gstamd_video_converter_config_init (GstAmdVideoConverterConfig *config);

/* Plan a conversion.  `config` may be NULL (library defaults).  On failure returns NULL and, when
 * `status` is non-NULL, stores the reason there. */
GstAmdVideoConverter *gstamd_video_converter_new (const GstAmdVideoInfo *in_info,
    const GstAmdVideoInfo *out_info, const GstAmdVideoConverterConfig *config, int *status);

/* Convert one frame.  `src`/`dest` are device pointers to the frame base (planes are found
 * through info.offset[] / info.stride[]).  Enqueues on `stream`; does not synchronise. */
int gstamd_video_converter_frame (GstAmdVideoConverter *convert, const void *src, void *dest,
    void *stream);

/* Same with one device pointer per plane (what a GstVideoFrame carries in data[] after a
 * device map) and explicit pitches overriding the infos'. */
int gstamd_video_converter_frame_planes (GstAmdVideoConverter *convert,
    const void *const src_planes[GSTAMD_VIDEO_MAX_PLANES], const int32_t src_stride[GSTAMD_VIDEO_MAX_PLANES],
    void *const dest_planes[GSTAMD_VIDEO_MAX_PLANES], const int32_t dest_stride[GSTAMD_VIDEO_MAX_PLANES],
    void *stream);

/* A list of independent frames (what a GstBufferList hands to chain_list, gstpad.h) converted by ONE
 * kernel launch where the plan allows it; results are identical to n calls of _frame.  Amortises the
 * per-launch ramp-up/drain, which is ~25 % of a 4K frame's kernel time on MI355X. */
int gstamd_video_converter_frames (GstAmdVideoConverter *convert, int n_frames, const void *const *src,
    void *const *dest, void *stream);

/* ---- GstVideoTestSrc's frames, painted in HBM --------------------------------------------------------------------------------------------------------
 * gst/videotestsrc/videotestsrc.c: the element's painters (gst_video_test_src_smpte :381 ... _colors :1869) fill a GstVideoFrame line by line on the CPU.
 * _new takes the caps' info, the `pattern` value (GstVideoTestSrcPattern, gstvideotestsrc.h:84-112) and the foreground-color / background-color
 * properties (0xAARRGGBB; the element's defaults 0xffffffff / 0xff000000); _frame paints frame number `n_frames` (the element's running count: blink, the
 * ball's position and the random generator of snow / smpte follow it) into the HBM frame `dest`.  Byte for byte the reference's frame.  Built: smpte,
 * snow, black, white, red, green, blue, checkers-1 / -2 / -4 / -8, blink, smpte75, smpte100, solid-color, bar, gradient, colors, ball (motion wavy,
 * animation-mode frames, no horizontal-speed); the other patterns: GSTAMD_ERR_UNSUPPORTED. */
typedef struct GstAmdVideoTestPattern GstAmdVideoTestPattern;
GstAmdVideoTestPattern *gstamd_video_test_pattern_new (const GstAmdVideoInfo *info, int pattern, uint32_t foreground_argb, uint32_t background_argb, int *status);
int gstamd_video_test_pattern_frame (GstAmdVideoTestPattern *pattern, uint64_t n_frames, void *dest, void *stream);
const char *gstamd_video_test_pattern_describe (const GstAmdVideoTestPattern *pattern);
void gstamd_video_test_pattern_free (GstAmdVideoTestPattern *pattern);

/* How the last _frames call on this converter ran: the number of kernel launches that each served a whole list (or a
 * chunk of up to 32 / 16 frames of it) - 0 when the plan's kernels took the frames one by one.  What tests and the
 * element's statistics read; no reference counterpart (the reference converts buffer lists buffer by buffer,
 * gstbasetransform.c default chain_list). */
int gstamd_video_converter_list_launches (GstAmdVideoConverter *convert);

void gstamd_video_converter_free (GstAmdVideoConverter *convert);

/* Introspection used by tests / bench: name of the kernel plan chosen ("fused_convert",
 * "front+hscale+vscale_back", ...) and algorithmic bytes per frame (source planes read once +
 * destination written once). */
const char *gstamd_video_converter_describe (const GstAmdVideoConverter *convert);
/* "" when the plan reproduces gst_video_converter_frame bit for bit.  Otherwise: why it does not - the conversions for which the
 * reference's own output is undefined (it reads lines it has not converted, or converts a repeated line twice: video-converter.c
 * do_convert_lines :3112, video_scale_v_near + the in-place stages, the unpack ring of setup_allocators :2115-2187, unpack_VYUY's
 * fallback loop).  There this library computes what the chain's stages mean - the reference's result when the same stages are run
 * as separate conversions - instead of refusing the caps. */
const char *gstamd_video_converter_divergence (const GstAmdVideoConverter *convert);
uint64_t gstamd_video_converter_algorithmic_bytes (const GstAmdVideoConverter *convert);

/* ---- compositor ----------------------------------------------------------------------- */

enum { GSTAMD_COMPOSITOR_BLEND_MODE_SOURCE = 0, GSTAMD_COMPOSITOR_BLEND_MODE_OVER = 1,
  GSTAMD_COMPOSITOR_BLEND_MODE_ADD = 2 };       /* GstCompositorBlendMode, blend.h:35-40 */
enum { GSTAMD_COMPOSITOR_BACKGROUND_CHECKER = 0, GSTAMD_COMPOSITOR_BACKGROUND_BLACK = 1,
  GSTAMD_COMPOSITOR_BACKGROUND_WHITE = 2, GSTAMD_COMPOSITOR_BACKGROUND_TRANSPARENT = 3 };
                                                /* GstCompositorBackground, compositor.h */

/* One BlendFunction call (blend.h:50): composite `src` (sw x sh, pitch sstride) at (xpos,ypos) with
 * pad alpha `src_alpha` onto `dest` rows [dst_y_start, dst_y_end).  format: BGRA/RGBA ("bgra"
 * family, alpha in byte 3) or ARGB/ABGR/AYUV (alpha in byte 0).  `overlay` selects the
 * gst_compositor_overlay_* variant used on a transparent background (compositor.c:850-854). */
int gstamd_compositor_blend (int format, int overlay, const void *src, int sw, int sh, int sstride,
    int xpos, int ypos, double src_alpha, void *dest, int dw, int dh, int dstride,
    int dst_y_start, int dst_y_end, int mode, void *stream);

/* FillCheckerFunction / FillColorFunction (blend.h:51-52) on rows [y_start, y_end). */
int gstamd_compositor_fill_checker (int format, void *dest, int dw, int dh, int dstride,
    int y_start, int y_end, void *stream);
int gstamd_compositor_fill_color (int format, void *dest, int dw, int dh, int dstride,
    int y_start, int y_end, int c1, int c2, int c3, void *stream);

typedef struct GstAmdCompositorPad {
  const void *data;             /* device pointer, same format as the output */
  int32_t width, height, stride;
  int32_t xpos, ypos;
  double alpha;                 /* pad alpha 0..1 */
  int32_t blend_mode;           /* per-pad operator -> GstCompositorBlendMode */
  int32_t reserved;
} GstAmdCompositorPad;

/* Whole aggregate step of one output frame = _draw_background + blend_pads loop
 * (compositor.c:1619-1697, 1739-1870) in ONE fused pass over the canvas: every output pixel is
 * written once, pads are applied in array order (= zorder) in registers.  Result is identical
 * to calling fill + gstamd_compositor_blend per pad. */
int gstamd_compositor_aggregate (int format, int background, const GstAmdCompositorPad *pads,
    int n_pads, void *dest, int dw, int dh, int dstride, void *stream);

/* gstamd_compositor_aggregate that reads fewer bytes than blend_pads (compositor.c:1678-1697) does: where a pad with alpha 1.0 covers a whole
 * 256-pixel strip of a canvas row with pixels of alpha 255, OVER / ADD leave the pad's own pixel ((s * 255 + d * 0) / 255, BLEND_A32 blend.c:96-132)
 * and nothing under that pad is read for the strip.  The output is the same byte for byte.  opacity[i] says what is known about pad i (NULL array:
 * nothing, the plain aggregate): all_opaque for a pad whose frame was converted from a format without alpha, or a map made once per pad frame by
 * gstamd_compositor_pad_opacity_map (a still image, a logo, a frame composited more than once).  Launches without the direct form (a SOURCE pad, a
 * transparent background, pads narrower than 4 pixels) and 64-bit canvases ignore the hints. */
typedef struct GstAmdCompositorPadOpacity {
  const uint64_t *map;          /* device pointer: pad-height words, bit b of word r = pixels [64 b, 64 b + 64) of pad row r all have alpha 255; or NULL */
  int32_t all_opaque;           /* 1: every pixel of the pad has alpha 255 */
  int32_t reserved;
} GstAmdCompositorPadOpacity;
int gstamd_compositor_aggregate_opaque (int format, int background, const GstAmdCompositorPad *pads,
    const GstAmdCompositorPadOpacity *opacity, int n_pads, void *dest, int dw, int dh, int dstride, void *stream);
/* the map of one pad frame (width <= 4096): `map` is a device buffer of `height` words, written on `stream` */
int gstamd_compositor_pad_opacity_map (int format, const void *data, int width, int height, int stride, uint64_t *map, void *stream);

/* Pads that are SCALED into the canvas (BASELINE C4 variant A): the reference gives each such pad a converter
 * (GstVideoAggregatorConvertPad, gstvideoaggregator.c:479-513) and blends the converted frame.  Here the pad hands over its frame as
 * it arrived plus its converter, and the scaled pixels are evaluated inside the blend pass - no scaled frame in HBM, one launch per
 * output frame.  `scaler` must be a converter for which gstamd_compositor_pad_scaler_usable () returns 1: the same 4-byte 8-bit
 * format as the canvas on both sides, whole frames, any resampler method / taps (the result is the converter's own output, bit for
 * bit); the pad then covers the converter's OUTPUT size on the canvas and data / stride describe a frame of its INPUT size.
 * scaler == NULL: the frame is blended as it is (width x height).  Other pads (format conversions, crops, borders) are converted by
 * the caller first, as before.  Formats: BGRA, RGBA, ARGB, ABGR, AYUV. */
typedef struct GstAmdCompositorScaledPad {
  const void *data;
  int32_t width, height, stride;        /* the frame in `data` */
  int32_t xpos, ypos;
  double alpha;
  int32_t blend_mode;
  int32_t reserved;
  GstAmdVideoConverter *scaler;
} GstAmdCompositorScaledPad;
int gstamd_compositor_pad_scaler_usable (GstAmdVideoConverter *convert);
int gstamd_compositor_aggregate_scaled (int format, int background, const GstAmdCompositorScaledPad *pads,
    int n_pads, void *dest, int dw, int dh, int dstride, void *stream);

/* Outputs WITHOUT per-pixel alpha - I420, YV12, Y42B, Y444, NV12, NV21, RGB, BGR: the reference converts every pad to the
 * output format and blends plane by plane with the pad alpha only (blend.c PLANAR_YUV_BLEND / NV_YUV_BLEND / RGB_BLEND,
 * fill_checker_* / fill_color_*; compositor.c:1619-1697).  One pass per destination plane, pads applied in array order.
 * black / white: the element's black_color / white_color in component order (compositor.c:1131-1149); NULL = 16,128,128 /
 * 235,128,128 for YUV and 0 / 255 for RGB. */
typedef struct GstAmdCompositorFramePad {
  const void *data[3];          /* device pointers of the pad's planes, same format as the output */
  int32_t stride[3];
  int32_t width, height;
  int32_t xpos, ypos;
  double alpha;
  int32_t blend_mode;
  int32_t reserved;
} GstAmdCompositorFramePad;
int gstamd_compositor_aggregate_frame (int format, int background, const int32_t black[3], const int32_t white[3],
    const GstAmdCompositorFramePad *pads, int n_pads, void *const dest[3], const int32_t dstride[3], int dw, int dh,
    void *stream);

/* ---- device memory helpers (thin wrappers so non-HIP hosts can manage HBM) -------------- */
void *gstamd_device_alloc (size_t size);
void gstamd_device_free (void *ptr);
int gstamd_device_upload (void *dst_device, const void *src_host, size_t size, void *stream);
int gstamd_device_download (void *dst_host, const void *src_device, size_t size, void *stream);
int gstamd_stream_synchronize (void *stream);
int gstamd_device_count (void);
int gstamd_set_device (int device);
int gstamd_get_device (void);
const char *gstamd_last_error (void);

/* ---- streams and events: what an element instance needs to run without device-wide synchronisation ----
 * (SURVEY 8b Threading: "one HIP stream per element instance, synchronise before a CPU gst_buffer_map (READ)").  A stream belongs to
 * the device that is current when it is created (gstamd_set_device).  Events order work ACROSS streams: the producer of a
 * frame records an event after its kernel, a consumer on another stream makes its stream wait on it - no host round trip. */
void *gstamd_stream_new (void);                  /* non-blocking stream; NULL on failure */
void gstamd_stream_free (void *stream);
void *gstamd_event_new (void);                   /* timing disabled */
void gstamd_event_free (void *event);
int gstamd_event_record (void *event, void *stream);
int gstamd_stream_wait_event (void *stream, void *event);
int gstamd_event_synchronize (void *event);      /* host waits for the event */
int gstamd_event_query (void *event);            /* 1: reached, 0: not yet, < 0: error */
/* page-locked host memory for system-memory pads (pageable memory makes hipMemcpyAsync synchronous and slow) */
void *gstamd_host_alloc (size_t size);
void gstamd_host_free (void *ptr);
/* 1 when `ptr` lies in page-locked host memory known to HIP (hipHostMalloc / hipHostRegister) - a hipMemcpyAsync from it is only QUEUED
 * when the call returns; 0 for pageable memory, where the call returns once the source has been staged */
int gstamd_host_is_pinned (const void *ptr);
int gstamd_device_copy (void *dst_device, const void *src_device, size_t size, void *stream);
int gstamd_device_upload_async (void *dst_device, const void *src_host, size_t size, void *stream);
int gstamd_device_download_async (void *dst_host, const void *src_device, size_t size, void *stream);
/* one plane whose host rows and device rows have different pitches (a GstVideoMeta with padded strides, gstvideometa.h:55-77) */
int gstamd_device_upload_2d_async (void *dst_device, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t row_bytes, size_t rows, void *stream);
int gstamd_device_download_2d_async (void *dst_host, size_t dst_pitch, const void *src_device, size_t src_pitch, size_t row_bytes, size_t rows, void *stream);

/* 1: frames of this converter may be in flight on several streams at once.  Since round 3 that is every plan: the intermediate images
 * of multi-kernel plans (two-pass and 16-bit scalers, planar packers, plane scalers, gamma remap) exist once per stream a frame has
 * been sent on.  Calls into one converter remain the caller's to serialise (one streaming thread per element). */
int gstamd_video_converter_is_reentrant (GstAmdVideoConverter *convert);

/* Development / support knobs (gstreamer_amd/csrc/tuning.h holds the table): the library reads them from the environment ONCE, at its
 * first look at any of them; afterwards only this call changes one.  value < 0 unsets it (the library's own choice).  None of them
 * changes a result: they choose between kernels that are held to the same byte-exact tests.  -1: unknown name. */
int gstamd_tuning_set (const char *name, int value);
int gstamd_tuning_get (const char *name);

/* gst_video_converter_get_config / _set_config (video-converter.c:2736-2790) for the options this library reads: the config the
 * converter was planned with; set_config re-plans with the new options applied on top (unknown conversions: GSTAMD_ERR_UNSUPPORTED,
 * the converter keeps its old plan). */
int gstamd_video_converter_get_config (const GstAmdVideoConverter *convert, GstAmdVideoConverterConfig *config);
int gstamd_video_converter_set_config (GstAmdVideoConverter *convert, const GstAmdVideoConverterConfig *config);
/* gst_video_converter_frame_finish (video-converter.c:2827): waits for the conversions enqueued on `stream` (the async-tasks
 * analogue: _frame enqueues, _frame_finish joins). */
int gstamd_video_converter_frame_finish (GstAmdVideoConverter *convert, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSTAMD_VIDEO_H */
