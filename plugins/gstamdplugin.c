/* gstamdplugin.c - plugin entry: registers the MI355X elements.  `videoconvertscale` takes the reference's
 * factory name (gstvideoconvertscaleplugin.c:33-53; rank SECONDARY there, one above here so autopluggers prefer
 * it when both plugins are visible; with a dedicated GST_PLUGIN_PATH it simply replaces the stock element).
 * The other elements carry an `amd` prefix for now so they can sit next to the stock videoconvert / videoscale
 * / audioresample in one registry; INTEGRATION.md shows the one-line change to claim those names. */
#include <gst/gst.h>

GType gst_amd_video_convert_scale_get_type (void);
GType gst_amd_audio_resample_get_type (void);
GType gst_amd_compositor_get_type (void);

static gboolean
plugin_init (GstPlugin * plugin)
{
  gboolean ok = TRUE;
  ok &= gst_element_register (plugin, "videoconvertscale", GST_RANK_SECONDARY + 1, gst_amd_video_convert_scale_get_type ());
  ok &= gst_element_register (plugin, "amdvideoconvert", GST_RANK_MARGINAL + 1, gst_amd_video_convert_scale_get_type ());
  ok &= gst_element_register (plugin, "amdvideoscale", GST_RANK_MARGINAL + 1, gst_amd_video_convert_scale_get_type ());
  ok &= gst_element_register (plugin, "amdaudioresample", GST_RANK_PRIMARY + 1, gst_amd_audio_resample_get_type ());
  /* the reference registers `compositor` with GST_RANK_PRIMARY + 1 (compositor.c, GST_ELEMENT_REGISTER_DEFINE) */
  ok &= gst_element_register (plugin, "compositor", GST_RANK_PRIMARY + 1, gst_amd_compositor_get_type ());
  return ok;
}

#ifndef PACKAGE
#define PACKAGE "gstreamer_amd"
#endif
GST_PLUGIN_DEFINE (GST_VERSION_MAJOR, GST_VERSION_MINOR, amdhipdsp, "MI355X-native raw video/audio DSP elements",
    plugin_init, "0.1", "LGPL", "gstreamer_amd", "https://example.invalid/gstreamer_amd")
