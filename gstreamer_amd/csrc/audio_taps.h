// audio_taps.h - host-side plan of the polyphase FIR resampler (see audio_taps.cpp)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gstamd_audio.h"
#include "../../include/gstamd_video.h"   // status codes

namespace gstamd {

struct AudioPlan {
  int method, format, channels, bps;
  bool variable_rate;
  bool in_planar, out_planar;     // GST_AUDIO_RESAMPLER_FLAG_NON_INTERLEAVED_IN / _OUT
  int in_rate, out_rate;          // gcd-reduced
  int samp_inc, samp_frac;
  int n_taps, oversample;
  int filter_mode, filter_interpolation;
  double cutoff, kaiser_beta, b, c;
  double max_phase_error;         // GstAudioResampler.max-phase-error of the stored options (read by the next update)
  int n_phases;                   // == out_rate in FULL mode
  int taps_stride;                // elements per table row (n_taps rounded up to 4, zero padded)
  std::vector<uint8_t> table;     // [n_phases][taps_stride] in the sample format
};

// host bookkeeping of one stream, field for field as struct _GstAudioResampler (audio-resampler-private.h:99-106)
struct AudioState {
  long long samp_index = 0, samp_phase = 0, skip = 0;
  size_t samples_avail = 0;
  long long stale_ahead = 0;     // > 0: the first stale_ahead frames of the history still hold frames the reference fills otherwise (audio_update); counts down as input is consumed
};

// what one gst_audio_resampler_resample() call has to launch
struct AudioStep {
  bool skipped_all;         // all input swallowed by `skip`: nothing to launch, history untouched
  bool run_fir;
  long long n_out;
  // FirParams fields (audio_device.h)
  long long samp_index0;
  int samp_phase0;
  long long hist_frames, total_frames;
  // history rebuild: new_hist[i] = i < moved ? logical[src_start + i] : logical[i], for i < keep
  long long src_start, moved, keep;
};

// history move of gst_audio_resampler_update when the tap count changes (audio-resampler.c:1572-1596), in frames
struct AudioHistoryShift {
  bool changed;
  long long src_off, dst_off, frames;
  long long stale;              // frames of the new history that the reference takes from beyond its valid samples (see audio_update)
};

void audio_state_reset (const AudioPlan &plan, AudioState *st);
int audio_update (AudioPlan *plan, AudioState *st, int in_rate, int out_rate, const GstAmdAudioResamplerOptions *options,
    AudioHistoryShift *shift, std::string *error);
void audio_history_shift (const AudioHistoryShift &shift, size_t frame_bytes, std::vector<uint8_t> *hist);
AudioStep audio_step (const AudioPlan &plan, AudioState *st, size_t in_frames, size_t out_frames);
size_t audio_get_out_frames (const AudioPlan &plan, const AudioState &st, size_t in_frames);
size_t audio_get_in_frames (const AudioPlan &plan, const AudioState &st, size_t out_frames);

void audio_options_init (GstAmdAudioResamplerOptions *o);
void audio_options_set_quality (int method, unsigned quality, int in_rate, int out_rate, GstAmdAudioResamplerOptions *o);
int plan_audio_resampler (int method, int flags, int format, int channels, int in_rate, int out_rate,
    const GstAmdAudioResamplerOptions *options, AudioPlan *plan, std::string *error);

}  // namespace gstamd
