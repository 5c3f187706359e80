/* gstamdplugin.c - plugin entry: registers the MI355X elements under the REFERENCE's factory names and ranks
 * (subprojects/gst-plugins-base/gst/videoconvertscale/gstvideoconvertscaleplugin.c:33-53: `videoscale`, `videoconvert` rank
 * MARGINAL, `videoconvertscale` rank SECONDARY - gstvideoconvert.c:46, gstvideoscale.c:58, gstvideoconvertscale.c:121;
 * gst/audioresample/gstaudioresample.c:139 `audioresample` rank PRIMARY; gst/audioconvert/gstaudioconvert.c `audioconvert` PRIMARY; gst/compositor/compositor.c `compositor` PRIMARY + 1),
 * one rank step above the stock elements so that autopluggers prefer them when both plugins are visible.  A registry keeps ONE
 * feature per name: with this plugin scanned after gst-plugins-base (or in a GST_PLUGIN_PATH of its own) the names resolve to these
 * elements.  `amd`-prefixed aliases are always there, so both implementations can be addressed side by side in one registry;
 * GSTAMD_PLUGIN_REFERENCE_NAMES=0 registers the aliases only.  Plus `amdhipupload` / `amdhipdownload` for a pipeline's edges and
 * `amdhipvideotestsrc`, GstVideoTestSrc's frames painted in HBM. */
#include <gst/gst.h>
#include <string.h>

GType gst_amd_video_convert_scale_get_type (void);
GType gst_amd_video_convert_element_get_type (void);
GType gst_amd_video_scale_element_get_type (void);
GType gst_amd_audio_resample_get_type (void);
GType gst_amd_audio_convert_get_type (void);
GType gst_amd_compositor_get_type (void);
GType gst_amd_hip_upload_element_get_type (void);
GType gst_amd_hip_download_element_get_type (void);
GType gst_amd_video_test_src_get_type (void);

/* A registry holds one feature per name, and GStreamer does not define what happens when two plugins claim the same one (1.14 ends up
 * with the details of one factory and the type of the other).  A reference name is therefore claimed only while no OTHER plugin of the
 * registry holds it: in a registry of its own (GST_PLUGIN_PATH without gst-plugins-base's videoconvert / videoscale /
 * audioresample plugins, or a deployment that ships this plugin instead of them) the names are ours; next to the stock plugins
 * the stock elements keep their names and ours answer to the amd-prefixed aliases. */
static gboolean
name_is_free (const gchar * name)
{
  GstPluginFeature *f = gst_registry_lookup_feature (gst_registry_get (), name);
  gboolean free_name = TRUE;

  if (f) {
    free_name = g_strcmp0 (gst_plugin_feature_get_plugin_name (f), "amdhipdsp") == 0;
    gst_object_unref (f);
  }
  return free_name;
}

static gboolean
claim (GstPlugin * plugin, const gchar * name, guint rank, GType type)
{
  if (!name_is_free (name)) {
    GST_INFO ("factory name %s is held by another plugin of this registry: not claimed (the amd-prefixed alias is registered)", name);
    return TRUE;
  }
  return gst_element_register (plugin, name, rank, type);
}

static gboolean
plugin_init (GstPlugin * plugin)
{
  const gchar *e = g_getenv ("GSTAMD_PLUGIN_REFERENCE_NAMES");
  const gboolean ref_names = !(e && strcmp (e, "0") == 0);
  gboolean ok = TRUE;

  ok &= gst_element_register (plugin, "videoconvertscale", GST_RANK_SECONDARY + 1, gst_amd_video_convert_scale_get_type ());
  /* the reference registers `compositor` with GST_RANK_PRIMARY + 1 (compositor.c, GST_ELEMENT_REGISTER_DEFINE) */
  ok &= claim (plugin, "compositor", GST_RANK_PRIMARY + 1, gst_amd_compositor_get_type ());
  if (ref_names) {
    ok &= claim (plugin, "videoconvert", GST_RANK_MARGINAL + 1, gst_amd_video_convert_element_get_type ());
    ok &= claim (plugin, "videoscale", GST_RANK_MARGINAL + 1, gst_amd_video_scale_element_get_type ());
    ok &= claim (plugin, "audioresample", GST_RANK_PRIMARY + 1, gst_amd_audio_resample_get_type ());
    ok &= claim (plugin, "audioconvert", GST_RANK_PRIMARY + 1, gst_amd_audio_convert_get_type ());
  }
  ok &= gst_element_register (plugin, "amdvideoconvertscale", GST_RANK_NONE, gst_amd_video_convert_scale_get_type ());
  ok &= gst_element_register (plugin, "amdvideoconvert", GST_RANK_NONE, gst_amd_video_convert_element_get_type ());
  ok &= gst_element_register (plugin, "amdvideoscale", GST_RANK_NONE, gst_amd_video_scale_element_get_type ());
  ok &= gst_element_register (plugin, "amdaudioresample", GST_RANK_NONE, gst_amd_audio_resample_get_type ());
  ok &= gst_element_register (plugin, "amdaudioconvert", GST_RANK_NONE, gst_amd_audio_convert_get_type ());
  ok &= gst_element_register (plugin, "amdcompositor", GST_RANK_NONE, gst_amd_compositor_get_type ());
  ok &= gst_element_register (plugin, "amdhipupload", GST_RANK_NONE, gst_amd_hip_upload_element_get_type ());
  ok &= gst_element_register (plugin, "amdhipdownload", GST_RANK_NONE, gst_amd_hip_download_element_get_type ());
  /* (videotestsrc keeps its name: the stock source feeds system-memory pipelines and the tests' references) */
  ok &= gst_element_register (plugin, "amdhipvideotestsrc", GST_RANK_NONE, gst_amd_video_test_src_get_type ());
  return ok;
}

#ifndef PACKAGE
#define PACKAGE "gstreamer_amd"
#endif
GST_PLUGIN_DEFINE (GST_VERSION_MAJOR, GST_VERSION_MINOR, amdhipdsp, "MI355X-native raw video/audio DSP elements",
    plugin_init, "0.2", "LGPL", "gstreamer_amd", "https://example.invalid/gstreamer_amd")
