/* gstamd_audio.h - C ABI of the MI355X-native GstAudioResampler replacement (polyphase FIR).
 *
 * Drop-in boundary for the audio half of the hot path.  Reference interface being
 * replaced (subprojects/gst-plugins-base/gst-libs/gst/audio/audio-resampler.h):
 *
 *   gstamd_audio_resampler_options_set_quality <- gst_audio_resampler_options_set_quality  :218
 *   gstamd_audio_resampler_new                 <- gst_audio_resampler_new                  :224
 *   gstamd_audio_resampler_free / _reset       <- gst_audio_resampler_free / _reset        :231, :234
 *   gstamd_audio_resampler_get_out_frames      <- gst_audio_resampler_get_out_frames       :242
 *   gstamd_audio_resampler_get_in_frames       <- gst_audio_resampler_get_in_frames        :246
 *   gstamd_audio_resampler_get_max_latency     <- gst_audio_resampler_get_max_latency      :250
 *   gstamd_audio_resampler_resample(_planes)   <- gst_audio_resampler_resample             :253
 *
 * Sample buffers are DEVICE pointers (interleaved frames); the FIR runs as HIP kernels on `stream`
 * (hipStream_t as void*, NULL = default stream) and the call returns without synchronising.  All
 * bookkeeping (history length, sample index/phase, skip) is host state that follows the reference
 * function for function, so get_out_frames() etc. give the reference's numbers.  The taps tables are
 * computed on the host with the reference's double-precision recipe and uploaded once.
 * Float summation order is the reference's C order (4 interleaved partial sums, ((r0+r1)+r2)+r3,
 * audio-resampler.c:693-707), without FMA contraction: results are bit-identical to the C path.
 */
#ifndef GSTAMD_AUDIO_H
#define GSTAMD_AUDIO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GstAudioResamplerMethod (audio-resampler.h:181-187) */
enum {
  GSTAMD_AUDIO_RESAMPLER_METHOD_NEAREST = 0, GSTAMD_AUDIO_RESAMPLER_METHOD_LINEAR = 1,
  GSTAMD_AUDIO_RESAMPLER_METHOD_CUBIC = 2, GSTAMD_AUDIO_RESAMPLER_METHOD_BLACKMAN_NUTTALL = 3,
  GSTAMD_AUDIO_RESAMPLER_METHOD_KAISER = 4
};
/* GstAudioResamplerFilterMode / FilterInterpolation (audio-resampler.h:103-107, 137-141) */
enum { GSTAMD_AUDIO_FILTER_MODE_INTERPOLATED = 0, GSTAMD_AUDIO_FILTER_MODE_FULL = 1, GSTAMD_AUDIO_FILTER_MODE_AUTO = 2 };
enum { GSTAMD_AUDIO_FILTER_INTERPOLATION_NONE = 0, GSTAMD_AUDIO_FILTER_INTERPOLATION_LINEAR = 1,
  GSTAMD_AUDIO_FILTER_INTERPOLATION_CUBIC = 2 };
/* gstamd_audio_resampler_new's `format` is a GstAudioFormat like gst_audio_resampler_new's (audio-resampler.h:218) - GSTAMD_AFMT_S16LE,
 * _S32LE, _F32LE, _F64LE below, whose values are GstAudioFormat's: a binding passes GST_AUDIO_INFO_FORMAT through.  The private
 * values 0 .. 3 of the first two releases stay understood (they collide with no GstAudioFormat the resampler accepts). */
enum { GSTAMD_AUDIO_FORMAT_S16 = 0, GSTAMD_AUDIO_FORMAT_S32 = 1, GSTAMD_AUDIO_FORMAT_F32 = 2, GSTAMD_AUDIO_FORMAT_F64 = 3 };

/* Mirror of the GstAudioResampler.* option keys (audio-resampler.h:42-165).  A field left at its
 * "unset" value (NaN for doubles, -1 for ints) means the key is absent from the options structure. */
typedef struct GstAmdAudioResamplerOptions {
  double cutoff;                 /* GstAudioResampler.cutoff */
  double stop_attenuation;       /* GstAudioResampler.stop-attenutation */
  double transition_bandwidth;   /* GstAudioResampler.transition-bandwidth */
  double cubic_b, cubic_c;       /* GstAudioResampler.cubic-b / cubic-c */
  double max_phase_error;        /* GstAudioResampler.max-phase-error */
  int32_t n_taps;                /* GstAudioResampler.n-taps */
  int32_t filter_mode;           /* GstAudioResampler.filter-mode */
  int32_t filter_mode_threshold; /* GstAudioResampler.filter-mode-threshold */
  int32_t filter_interpolation;  /* GstAudioResampler.filter-interpolation */
  int32_t filter_oversample;     /* GstAudioResampler.filter-oversample */
  int32_t reserved[7];
} GstAmdAudioResamplerOptions;

typedef struct GstAmdAudioResampler GstAmdAudioResampler;

/* all keys unset (an empty options structure) */
void gstamd_audio_resampler_options_init (GstAmdAudioResamplerOptions *options);
void gstamd_audio_resampler_options_set_quality (int method, unsigned quality, int in_rate, int out_rate,
    GstAmdAudioResamplerOptions *options);

/* options == NULL: Kaiser quality 4 like the reference (audio-resampler.c:1414-1419).  flags: GstAudioResamplerFlags
 * (audio-resampler.h:177-182) - 1 non-interleaved input, 2 non-interleaved output, 4 variable rate.  Returns NULL and sets *status (GSTAMD_ERR_* of gstamd_video.h) on failure.  Both filter
 * modes are implemented: FULL (one row of taps per phase) and INTERPOLATED (the taps of every output sample are blended
 * on the device from the oversampled table, linear or cubic, audio-resampler.c:567-757). */
GstAmdAudioResampler *gstamd_audio_resampler_new (int method, int flags, int format, int channels, int in_rate,
    int out_rate, const GstAmdAudioResamplerOptions *options, int *status);
void gstamd_audio_resampler_free (GstAmdAudioResampler *resampler);
void gstamd_audio_resampler_reset (GstAmdAudioResampler *resampler);
/* gst_audio_resampler_update (audio-resampler.h:231, audio-resampler.c:1503-1614): new rates (<= 0: unchanged) and / or new
 * options on a running stream; the phase is rescaled, and when the tap count changes the history moves by half the difference.
 * options == NULL keeps the previous filter design, as the reference does.  Returns GSTAMD_OK (the reference's TRUE) or an
 * error with the resampler left as it was. */
int gstamd_audio_resampler_update (GstAmdAudioResampler *resampler, int in_rate, int out_rate,
    const GstAmdAudioResamplerOptions *options);
/* Where the reference's own result is not a function of the stream: gst_audio_resampler_update with options that ENLARGE the filter by more than
 * twice the history it holds leaves the head of the new history as it finds it in its sample buffer - input of earlier calls past the valid
 * samples ("FIXME, probably do something better like mirror or fill with zeroes", audio-resampler.c:1587-1590).  This library fills those
 * frames with silence and says so here: a text while such frames are inside the filter window, "" otherwise (and always for streams that
 * never enlarge their filter that far).  Same contract as gstamd_video_converter_divergence. */
const char *gstamd_audio_resampler_divergence (GstAmdAudioResampler *resampler);
size_t gstamd_audio_resampler_get_out_frames (GstAmdAudioResampler *resampler, size_t in_frames);
size_t gstamd_audio_resampler_get_in_frames (GstAmdAudioResampler *resampler, size_t out_frames);
size_t gstamd_audio_resampler_get_max_latency (GstAmdAudioResampler *resampler);

/* in: device pointer to in_frames interleaved frames, or NULL for silence (drain); out: device pointer
 * with room for out_frames frames.  A non-interleaved side holds its channels one after the other, in_frames
 * (out_frames) samples apart. */
int gstamd_audio_resampler_resample (GstAmdAudioResampler *resampler, const void *in, size_t in_frames, void *out,
    size_t out_frames, void *stream);

/* gst_audio_resampler_resample's own argument shape (audio-resampler.h:253): in[] / out[] hold ONE pointer for an
 * interleaved side and `channels` pointers for a non-interleaved one (equally spaced planes in ascending order; anything
 * else is GSTAMD_ERR_UNSUPPORTED); in == NULL feeds silence. */
int gstamd_audio_resampler_resample_planes (GstAmdAudioResampler *resampler, const void *const in[], size_t in_frames,
    void *const out[], size_t out_frames, void *stream);

/* introspection for tests: n_taps, n_phases (reduced out_rate), reduced in_rate, oversample, filter mode */
/* `n` independent resamplers, one buffer each (in[i] / out[i] interleaved, or non-interleaved planes following each other as in _resample),
 * in ONE kernel launch where they share a filter - resamplers made with the same arguments, full filter mode; up to 64 per launch, longer
 * or mixed sets run as several launches / one by one.  Outputs and resampler states are exactly those of n _resample calls.  For what
 * carries many streams at once: the channels of a non-interleaved capture, the inputs of a mixer, a buffer list.  No reference counterpart:
 * gst_audio_resampler_resample (audio-resampler.h:253) takes one stream and the reference has no cross-stream batching. */
int gstamd_audio_resampler_resample_many (int n, GstAmdAudioResampler *const *resamplers, const void *const *in, const size_t *in_frames,
    void *const *out, const size_t *out_frames, void *stream);

int gstamd_audio_resampler_debug_get (GstAmdAudioResampler *resampler, int32_t *out, int max_out);
/* copies the [n_phases][n_taps] taps table (as doubles) out; returns number of values or < 0 */
long gstamd_audio_resampler_debug_taps (GstAmdAudioResampler *resampler, double *out, long max_out);

/* ---- GstAudioConverter (gst-libs/gst/audio/audio-converter.h:85-140, audio-converter.c) ------------------------------------------
 * The stages of gst_audio_converter_new's chain (audio-converter.c:708-1090) on the device: unpack -> S32 / F64, S32 -> F64
 * (convert_in), channel mix, resample, F64 -> S32 (convert_out), quantize (dither), pack.  Formats are GstAudioFormat values
 * (audio-format.h:80-130, little-endian ones): */
enum {
  GSTAMD_AFMT_S8 = 2, GSTAMD_AFMT_U8 = 3, GSTAMD_AFMT_S16LE = 4, GSTAMD_AFMT_S24_32LE = 8, GSTAMD_AFMT_S32LE = 12, GSTAMD_AFMT_S24LE = 16,
  GSTAMD_AFMT_F32LE = 28, GSTAMD_AFMT_F64LE = 30
};
/* GstAudioDitherMethod / GstAudioNoiseShapingMethod (audio-quantize.h:45-72) */
enum { GSTAMD_AUDIO_DITHER_NONE = 0, GSTAMD_AUDIO_DITHER_RPDF = 1, GSTAMD_AUDIO_DITHER_TPDF = 2, GSTAMD_AUDIO_DITHER_TPDF_HF = 3 };
#define GSTAMD_AUDIO_MAX_CHANNELS 8

typedef struct GstAmdAudioInfo {
  int32_t format;               /* GSTAMD_AFMT_* */
  int32_t rate, channels;
  int32_t layout;               /* 0 interleaved (the only layout of the converter so far) */
  int32_t unpositioned;         /* GST_AUDIO_FLAG_UNPOSITIONED */
  int32_t position[GSTAMD_AUDIO_MAX_CHANNELS];  /* GstAudioChannelPosition values (audio-channels.h:101-133): NONE -3, MONO -2, FRONT_LEFT 0, FRONT_RIGHT 1, FRONT_CENTER 2, LFE1 3, REAR_LEFT 4, ... */
} GstAmdAudioInfo;

typedef struct GstAmdAudioConverterConfig {
  int32_t dither_method;        /* GstAudioConverter.dither-method (library default: none) */
  int32_t noise_shaping;        /* GstAudioConverter.noise-shaping-method (NONE; the error-feedback methods are sequential per channel) */
  uint32_t dither_threshold;    /* GstAudioConverter.dither-threshold (20) */
  int32_t resampler_method;     /* GstAudioConverter.resampler-method (BLACKMAN_NUTTALL = 3) */
  int32_t has_resampler_options;
  GstAmdAudioResamplerOptions resampler_options;
  int32_t has_mix_matrix;       /* GstAudioConverter.mix-matrix: mix_matrix[out][in] as in the option (audio-converter.c:797-845) */
  float mix_matrix[GSTAMD_AUDIO_MAX_CHANNELS][GSTAMD_AUDIO_MAX_CHANNELS];
} GstAmdAudioConverterConfig;

typedef struct GstAmdAudioConverter GstAmdAudioConverter;

void gstamd_audio_converter_config_init (GstAmdAudioConverterConfig *config);
/* flags: GstAudioConverterFlags (audio-converter.h:99-103; 2 = VARIABLE_RATE).  NULL + *status on failure
 * (GSTAMD_ERR_UNSUPPORTED: the reference converts this, the GPU path does not yet - never a CPU fallback). */
GstAmdAudioConverter *gstamd_audio_converter_new (int flags, const GstAmdAudioInfo *in_info, const GstAmdAudioInfo *out_info,
    const GstAmdAudioConverterConfig *config, int *status);
void gstamd_audio_converter_free (GstAmdAudioConverter *convert);
void gstamd_audio_converter_reset (GstAmdAudioConverter *convert);
size_t gstamd_audio_converter_get_out_frames (GstAmdAudioConverter *convert, size_t in_frames);
size_t gstamd_audio_converter_get_in_frames (GstAmdAudioConverter *convert, size_t out_frames);
size_t gstamd_audio_converter_get_max_latency (GstAmdAudioConverter *convert);
int gstamd_audio_converter_is_passthrough (GstAmdAudioConverter *convert);
/* gst_audio_converter_samples (audio-converter.c:1545): in / out are device pointers to interleaved frames; in == NULL feeds
 * silence into the resampler (drain) */
int gstamd_audio_converter_samples (GstAmdAudioConverter *convert, int flags, const void *in, size_t in_frames, void *out, size_t out_frames,
    void *stream);
/* the mix matrix the converter uses, matrix[in][out] as GstAudioChannelMixer holds it; returns in_channels * out_channels */
int gstamd_audio_converter_get_mix_matrix (GstAmdAudioConverter *convert, float *matrix, int max);

#ifdef __cplusplus
}
#endif
#endif /* GSTAMD_AUDIO_H */
