/* gstamdcompositor.c - `compositor` element backed by the MI355X aggregate kernel (C ABI of include/gstamd_video.h).
 *
 * Mirrors the reference element's contract for the blend path
 * (subprojects/gst-plugins-base/gst/compositor/compositor.c):
 *   - factory name `compositor`, request pads `sink_%u`                        compositor.c:112, 2180
 *   - pad properties xpos / ypos / width / height / alpha / operator (+ zorder, which the reference inherits from
 *     GstVideoAggregatorPad) with the reference names, ranges and defaults      compositor.c:678-714
 *   - element property background (checker / black / white / transparent)       compositor.c:2103-2120
 *   - output size = bounding box of the pads (xpos + width, ypos + height)       compositor.c:1060-1160 (_fixate_caps)
 *   - one output frame = background fill + pads blended in zorder               compositor.c:1619-1697, 1739-1870
 * The reference subclasses GstVideoAggregator (compositor.c:808-809, aggregate_frames :1739 / :2098, pads derived from
 * GstVideoAggregatorConvertPad gstvideoaggregator.c:656-678).  So does this element wherever that base class is public API - libgstvideo
 * from 1.16 on, which includes the reference's own version (AMD_COMP_VAGG below): GstVideoAggregator does the frame selection, the QoS and
 * latency bookkeeping, zorder / repeat-after-eos / max-last-buffer-repeat / converter-config and the child proxy; the element supplies
 * create_output_buffer (HBM frames of its own pool) and aggregate_frames, the pads a prepare_frame that maps and converts NOTHING on the CPU
 * (the frames stay where they are, per-pad conversions are GstAmdVideoConverters on the GPU).  On the 1.14 runtime of this image, where
 * GstVideoAggregator still lived in gst-plugins-bad's unstable library, the same file compiles onto GstAggregator (public in libgstbase since
 * 1.14) and does the part of GstVideoAggregator the blend path needs itself (described below).  Either way all pads of one frame go to the GPU
 * in ONE fused kernel launch (gstamd_compositor_aggregate): every canvas pixel is written once, where the reference read-modify-writes the
 * canvas once per pad.
 *
 * What of GstVideoAggregator is here: frame selection by running time (a pad shows the queued frame that overlaps the output
 * frame's interval, older ones are dropped, a slower pad's frame is repeated; gstvideoaggregator.c:1753-2000), repeat-after-eos,
 * the pads' output size with sizing-policy and zero-size-is-unscaled and the pixel-aspect-ratio rules of
 * _mixer_pad_get_output_size (compositor.c:290-412), and the visibility rules of prepare_frame_start (compositor.c:464-601):
 * frames with alpha 0, off the canvas or fully under an opaque pad never reach the GPU.  Pads of another format or size are
 * converted / scaled by a per-pad GstAmdVideoConverter with the library's default config (GstVideoAggregatorConvertPad).
 * Not there (1.14 form only): QoS and the latency bookkeeping of the base class.  max-threads is accepted and means nothing (the blend is one
 * kernel launch per output frame), ignore-inactive-pads needs the 1.20 aggregator (gst_aggregator_set_ignore_inactive_pads) and is accepted
 * without effect on the 1.14 runtime this image has.
 */
#include <gst/base/gstaggregator.h>
#include <gst/gst.h>
#include <gst/video/video.h>
#if GST_CHECK_VERSION (1, 16, 0)
#define AMD_COMP_VAGG 1
#include <gst/video/gstvideoaggregator.h>
#else
#define AMD_COMP_VAGG 0
#endif
#include <gst/video/gstvideosink.h>
#include <string.h>

#include "../include/gstamd_video.h"
#include "gstamdhipbufferpool.h"
#include "gstamdhipmemory.h"

GST_DEBUG_CATEGORY_STATIC (amd_comp_debug);
#define GST_CAT_DEFAULT amd_comp_debug

/* Y444_16LE, P012_LE and P016_LE joined the format enum in 1.18: there when the headers this is compiled against have them */
#if GST_CHECK_VERSION (1, 18, 0)
#define AMD_COMP_NEWER_CANVAS ", VUYA, Y444_16LE"
#define AMD_COMP_NEWER_PADS ", VUYA, Y444_16LE, P012_LE, P016_LE"
#elif GST_CHECK_VERSION (1, 16, 0)
#define AMD_COMP_NEWER_CANVAS ", VUYA"
#define AMD_COMP_NEWER_PADS ", VUYA"
#else
#define AMD_COMP_NEWER_CANVAS ""
#define AMD_COMP_NEWER_PADS ""
#endif
#if GST_CHECK_VERSION (1, 20, 0)
#define AMD_COMP_NEWEST_PADS ", AV12"
#else
#define AMD_COMP_NEWEST_PADS ""
#endif
#define AMD_COMP_FORMATS "{ BGRA, RGBA, ARGB, ABGR, AYUV, ARGB64, AYUV64, I420, YV12, Y42B, Y444, NV12, NV21, RGB, BGR, RGBx, BGRx, xRGB, xBGR, YUY2, UYVY, YVYU, " \
    "I420_10LE, I420_12LE, I422_10LE, I422_12LE, Y444_10LE, Y444_12LE" AMD_COMP_NEWER_CANVAS " }"
/* what a pad may carry: anything the converter takes; it is brought to the output format / the pad's width x height by a
 * per-pad GstAmdVideoConverter (the reference's GstVideoAggregatorConvertPad, gstvideoaggregator.c:479-513) */
#define AMD_COMP_PAD_FORMATS "{ BGRA, RGBA, ARGB, ABGR, AYUV, ARGB64, AYUV64, RGBx, BGRx, xRGB, xBGR, RGB, BGR, NV12, NV21, NV16, NV61, NV24, I420, YV12, Y42B, Y444, YUY2, UYVY, YVYU, VYUY, GRAY8, GBR, " \
    "I420_10LE, I420_12LE, I422_10LE, I422_12LE, Y444_10LE, Y444_12LE, P010_10LE, " \
    /* round 5: the converter's newer formats (those every supported runtime's headers know) - A420 first, the alpha-plane format WebM / VP8 alpha decodes to */ \
    "A420, A420_10LE, A422_10LE, A444_10LE, GBRA, GBR_10LE, GBR_12LE, GBRA_10LE, GBRA_12LE, GRAY16_LE, GRAY16_BE, RGB16, BGR16, RGB15, BGR15, v210, v216, r210, v308, IYU2, Y41B" AMD_COMP_NEWER_PADS AMD_COMP_NEWEST_PADS " }"
#define AMD_COMP_MAX_PADS 64

static GstStaticPadTemplate comp_sink_tmpl = GST_STATIC_PAD_TEMPLATE ("sink_%u", GST_PAD_SINK, GST_PAD_REQUEST,
    GST_STATIC_CAPS (GST_VIDEO_CAPS_MAKE_WITH_FEATURES (GST_CAPS_FEATURE_MEMORY_AMD_HIP, AMD_COMP_PAD_FORMATS) ";"
        GST_VIDEO_CAPS_MAKE (AMD_COMP_PAD_FORMATS)));
static GstStaticPadTemplate comp_src_tmpl = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VIDEO_CAPS_MAKE_WITH_FEATURES (GST_CAPS_FEATURE_MEMORY_AMD_HIP, AMD_COMP_FORMATS) ";"
        GST_VIDEO_CAPS_MAKE (AMD_COMP_FORMATS)));

/* ---- pad ------------------------------------------------------------------------------------------------ */
typedef struct {
#if AMD_COMP_VAGG
  GstVideoAggregatorConvertPad parent;
#else
  GstAggregatorPad parent;
#endif
  gint xpos, ypos, width, height;
  gdouble alpha;
  gint op;                     /* GstCompositorOperator: 0 source, 1 over, 2 add */
  guint zorder;
  gint sizing_policy;          /* GstCompositorSizingPolicy: 0 none, 1 keep-aspect-ratio (compositor.c:207-232) */
  gboolean repeat_after_eos;   /* GstVideoAggregatorPad: keep showing the last frame after EOS (gstvideoaggregator.c:166-176) */
  guint64 max_last_buffer_repeat;   /* GstVideoAggregatorPad (gstvideoaggregator.c:282-303): ns past its end a frame is still shown without a successor */
  GstStructure *converter_config;     /* GstVideoAggregatorConvertPad::converter-config (gstvideoaggregator.c:444-488), object lock */
  gboolean converter_config_changed;
  GstVideoInfo info;            /* layout of `current` (and of the buffers taken since the last promoted CAPS event) */
  gboolean have_info;
  GstVideoInfo pending_info;    /* caps that arrived while a frame of the old caps is still shown (object lock) */
  gboolean have_pending;
  /* frame selection by running time (gst_video_aggregator_fill_queues, gstvideoaggregator.c:1753-2000) */
  GstBuffer *current;          /* the frame this pad shows now */
  GstClockTime cur_start, cur_end;     /* its running-time interval (end NONE: until replaced) */
  gpointer staging;            /* device copy of a system-memory input frame */
  gsize staging_size;
  /* per-pad conversion (format and / or size): converter, its key, its device output frame */
  GstAmdVideoConverter *conv;
  gboolean conv_inline;         /* conv only scales the canvas format: it is sampled inside the blend kernel (gstamd_compositor_aggregate_scaled) */
  gint conv_key[6];            /* in format, in w, in h, out format, out w, out h */
  gpointer conv_buf;
  gsize conv_buf_size;
  GstAmdVideoInfo conv_out;      /* layout of the converted frame in conv_buf */
} GstAmdCompositorPadObj;

#if AMD_COMP_VAGG
typedef struct { GstVideoAggregatorConvertPadClass parent_class; } GstAmdCompositorPadObjClass;
#define AMD_COMP_PAD_PARENT_TYPE GST_TYPE_VIDEO_AGGREGATOR_CONVERT_PAD
#else
typedef struct { GstAggregatorPadClass parent_class; } GstAmdCompositorPadObjClass;
#define AMD_COMP_PAD_PARENT_TYPE GST_TYPE_AGGREGATOR_PAD
#endif

enum { PAD_PROP_0, PAD_PROP_XPOS, PAD_PROP_YPOS, PAD_PROP_WIDTH, PAD_PROP_HEIGHT, PAD_PROP_ALPHA, PAD_PROP_OPERATOR, PAD_PROP_ZORDER,
  PAD_PROP_SIZING_POLICY, PAD_PROP_REPEAT_AFTER_EOS, PAD_PROP_MAX_LAST_BUFFER_REPEAT, PAD_PROP_CONVERTER_CONFIG };

G_DEFINE_TYPE (GstAmdCompositorPadObj, gst_amd_compositor_pad, AMD_COMP_PAD_PARENT_TYPE);
#define AMD_COMP_PAD(o) ((GstAmdCompositorPadObj *) (o))

static GType
amd_comp_operator_get_type (void)
{
  static GType t = 0;
  static const GEnumValue v[] = { {0, "Source", "source"}, {1, "Over", "over"}, {2, "Add", "add"}, {0, NULL, NULL} };
  if (!t)
    t = g_enum_register_static ("GstAmdCompositorOperator", v);
  return t;
}

static GType
amd_comp_background_get_type (void)
{
  static GType t = 0;
  static const GEnumValue v[] = { {0, "Checker pattern", "checker"}, {1, "Black", "black"}, {2, "White", "white"},
    {3, "Transparent Background to enable further compositing", "transparent"}, {0, NULL, NULL} };
  if (!t)
    t = g_enum_register_static ("GstAmdCompositorBackground", v);
  return t;
}

static void
amd_comp_pad_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstAmdCompositorPadObj *p = AMD_COMP_PAD (object);
  GST_OBJECT_LOCK (p);
  switch (id) {
    case PAD_PROP_XPOS: p->xpos = g_value_get_int (value); break;
    case PAD_PROP_YPOS: p->ypos = g_value_get_int (value); break;
    case PAD_PROP_WIDTH: p->width = g_value_get_int (value); break;
    case PAD_PROP_HEIGHT: p->height = g_value_get_int (value); break;
    case PAD_PROP_ALPHA: p->alpha = g_value_get_double (value); break;
    case PAD_PROP_OPERATOR: p->op = g_value_get_enum (value); break;
    case PAD_PROP_ZORDER: p->zorder = g_value_get_uint (value); break;
    case PAD_PROP_SIZING_POLICY: p->sizing_policy = g_value_get_enum (value); break;
    case PAD_PROP_REPEAT_AFTER_EOS: p->repeat_after_eos = g_value_get_boolean (value); break;
    case PAD_PROP_MAX_LAST_BUFFER_REPEAT: p->max_last_buffer_repeat = g_value_get_uint64 (value); break;
    case PAD_PROP_CONVERTER_CONFIG:
      if (p->converter_config)
        gst_structure_free (p->converter_config);
      p->converter_config = g_value_dup_boxed (value);
      p->converter_config_changed = TRUE;       /* the aggregate thread re-makes the pad's converter (gstvideoaggregator.c:468-476) */
      break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (p);
}

static void
amd_comp_pad_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstAmdCompositorPadObj *p = AMD_COMP_PAD (object);
  GST_OBJECT_LOCK (p);
  switch (id) {
    case PAD_PROP_XPOS: g_value_set_int (value, p->xpos); break;
    case PAD_PROP_YPOS: g_value_set_int (value, p->ypos); break;
    case PAD_PROP_WIDTH: g_value_set_int (value, p->width); break;
    case PAD_PROP_HEIGHT: g_value_set_int (value, p->height); break;
    case PAD_PROP_ALPHA: g_value_set_double (value, p->alpha); break;
    case PAD_PROP_OPERATOR: g_value_set_enum (value, p->op); break;
    case PAD_PROP_ZORDER: g_value_set_uint (value, p->zorder); break;
    case PAD_PROP_SIZING_POLICY: g_value_set_enum (value, p->sizing_policy); break;
    case PAD_PROP_REPEAT_AFTER_EOS: g_value_set_boolean (value, p->repeat_after_eos); break;
    case PAD_PROP_MAX_LAST_BUFFER_REPEAT: g_value_set_uint64 (value, p->max_last_buffer_repeat); break;
    case PAD_PROP_CONVERTER_CONFIG: g_value_set_boxed (value, p->converter_config); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (p);
}

static void
amd_comp_pad_finalize (GObject * object)
{
  GstAmdCompositorPadObj *p = AMD_COMP_PAD (object);
  gstamd_device_free (p->staging);
  gstamd_device_free (p->conv_buf);
  if (p->conv)
    gstamd_video_converter_free (p->conv);
  gst_buffer_replace (&p->current, NULL);
  if (p->converter_config)
    gst_structure_free (p->converter_config);
  G_OBJECT_CLASS (gst_amd_compositor_pad_parent_class)->finalize (object);
}

#if !AMD_COMP_VAGG
static GstFlowReturn
amd_comp_pad_flush (GstAggregatorPad * pad, GstAggregator * agg)
{
  (void) agg;
  gst_buffer_replace (&AMD_COMP_PAD (pad)->current, NULL);
  return GST_FLOW_OK;
}
#else
/* GstVideoAggregatorPadClass::prepare_frame / clean_frame: the base classes would map the buffer on the CPU (an HBM frame: a download) and
 * GstVideoAggregatorConvertPad would run a CPU GstVideoConverter on it (gstvideoaggregator.c:479-569).  Nothing of that: aggregate_frames
 * takes the pad's current buffer (gst_video_aggregator_pad_get_current_buffer) as it is */
static gboolean
amd_comp_pad_prepare_frame (GstVideoAggregatorPad * pad, GstVideoAggregator * vagg, GstBuffer * buffer, GstVideoFrame * prepared_frame)
{
  (void) pad;
  (void) vagg;
  (void) buffer;
  (void) prepared_frame;
  return TRUE;
}

static void
amd_comp_pad_clean_frame (GstVideoAggregatorPad * pad, GstVideoAggregator * vagg, GstVideoFrame * prepared_frame)
{
  (void) pad;
  (void) vagg;
  (void) prepared_frame;
}
#endif

static void
gst_amd_compositor_pad_class_init (GstAmdCompositorPadObjClass * klass)
{
  GObjectClass *oc = (GObjectClass *) klass;
#if AMD_COMP_VAGG
  ((GstVideoAggregatorPadClass *) klass)->prepare_frame = amd_comp_pad_prepare_frame;
  ((GstVideoAggregatorPadClass *) klass)->clean_frame = amd_comp_pad_clean_frame;
#if GST_CHECK_VERSION (1, 20, 0)          /* members since 1.20 (gstvideoaggregator.h) */
  ((GstVideoAggregatorPadClass *) klass)->prepare_frame_start = NULL;
  ((GstVideoAggregatorPadClass *) klass)->prepare_frame_finish = NULL;
#endif
#else
  ((GstAggregatorPadClass *) klass)->flush = amd_comp_pad_flush;
#endif
  const GParamFlags f = G_PARAM_READWRITE | GST_PARAM_CONTROLLABLE | G_PARAM_STATIC_STRINGS;
  oc->set_property = amd_comp_pad_set_property;
  oc->get_property = amd_comp_pad_get_property;
  oc->finalize = amd_comp_pad_finalize;
  /* names, ranges and defaults of compositor.c:678-714 (+ zorder of gstvideoaggregator.c:150-160) */
  g_object_class_install_property (oc, PAD_PROP_XPOS, g_param_spec_int ("xpos", "X Position", "X Position of the picture", G_MININT, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PAD_PROP_YPOS, g_param_spec_int ("ypos", "Y Position", "Y Position of the picture", G_MININT, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PAD_PROP_WIDTH, g_param_spec_int ("width", "Width", "Width of the picture", G_MININT, G_MAXINT, -1, f));
  g_object_class_install_property (oc, PAD_PROP_HEIGHT, g_param_spec_int ("height", "Height", "Height of the picture", G_MININT, G_MAXINT, -1, f));
  g_object_class_install_property (oc, PAD_PROP_ALPHA, g_param_spec_double ("alpha", "Alpha", "Alpha of the picture", 0.0, 1.0, 1.0, f));
  g_object_class_install_property (oc, PAD_PROP_OPERATOR, g_param_spec_enum ("operator", "Operator",
          "Blending operator to use for blending this pad over the previous ones", amd_comp_operator_get_type (), 1, f));
#if !AMD_COMP_VAGG              /* (GstVideoAggregatorPad's own, gstvideoaggregator.c:150-176, 282-303) */
  g_object_class_install_property (oc, PAD_PROP_ZORDER, g_param_spec_uint ("zorder", "Z-Order", "Z Order of the picture", 0, G_MAXUINT, 0, f));
#endif
  {
    static const GEnumValue sp[] = { {0, "None: Image is scaled to fill configured destination rectangle without padding or keeping the aspect ratio", "none"},
      {1, "Keep Aspect Ratio: Image is scaled to fit destination rectangle specified by GstCompositorPad:{xpos, ypos, width, height} "
            "with preserved aspect ratio", "keep-aspect-ratio"}, {0, NULL, NULL} };
    GType t = g_type_from_name ("GstAmdCompositorSizingPolicy");
    if (!t)
      t = g_enum_register_static ("GstAmdCompositorSizingPolicy", sp);
    g_object_class_install_property (oc, PAD_PROP_SIZING_POLICY, g_param_spec_enum ("sizing-policy", "Sizing policy",
            "Sizing policy to use for image scaling", t, 0, f));
  }
#if !AMD_COMP_VAGG
  g_object_class_install_property (oc, PAD_PROP_REPEAT_AFTER_EOS, g_param_spec_boolean ("repeat-after-eos", "Repeat After EOS",
          "Repeat the last frame after EOS until all pads are EOS", FALSE, f));
  g_object_class_install_property (oc, PAD_PROP_MAX_LAST_BUFFER_REPEAT, g_param_spec_uint64 ("max-last-buffer-repeat", "Max Last Buffer Repeat",
          "Repeat last buffer for time (in ns, -1=until EOS), behaviour on EOS is not affected", 0, G_MAXUINT64, G_MAXUINT64, f));
  /* GstVideoAggregatorConvertPad::converter-config (gstvideoaggregator.c:560-570) */
  g_object_class_install_property (oc, PAD_PROP_CONVERTER_CONFIG, g_param_spec_boxed ("converter-config", "Converter configuration",
          "A GstStructure describing the configuration that should be used when scaling and converting this pad's video frames", GST_TYPE_STRUCTURE,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
#endif
}

static void
gst_amd_compositor_pad_init (GstAmdCompositorPadObj * p)
{
  p->xpos = p->ypos = 0;
  p->width = p->height = -1;
  p->alpha = 1.0;
  p->op = 1;
  p->zorder = 0;
  p->sizing_policy = 0;
  p->repeat_after_eos = FALSE;
  p->max_last_buffer_repeat = G_MAXUINT64;
  p->have_info = FALSE;
  p->current = NULL;
  p->cur_start = p->cur_end = GST_CLOCK_TIME_NONE;
}

/* ---- element -------------------------------------------------------------------------------------------- */
typedef struct {
#if AMD_COMP_VAGG
  GstVideoAggregator parent;
#else
  GstAggregator parent;
#endif
  gint background;
  GstVideoInfo out_info;
  gboolean have_out, out_hip;
  GstBufferPool *out_pool;     /* HBM output frames (downstream negotiated memory:AMDHIPMemory) */
  gpointer d_out;              /* device canvas when downstream wants system memory */
  gsize d_out_size;
  guint64 n_frames;
  gpointer stream;             /* this instance's HIP stream (SURVEY 8b Threading); buffers are ordered by the tickets of gstamdhipmemory.h */
  gint device_id;              /* device-id property: -1 = the process's current device */
  guint next_pad;
  gboolean zero_size_is_unscaled;      /* compositor.c:2131: width / height 0 mean "unscaled" (TRUE) or "invisible" (FALSE) */
  gboolean ignore_inactive_pads;       /* accepted (GstAggregator's inactive-pad tracking does not exist in this runtime) */
  guint max_threads;                   /* accepted; the GPU grid replaces the blend thread pool */
  guint64 n_inline_scaled;             /* pad frames scaled inside the blend kernel instead of by a converter launch (GSTAMD_ELEMENT_STATS) */
  guint64 n_culled;                    /* pad frames left out because nothing of them can be seen (culled-frames, read-only) */
} GstAmdCompositor;

#if AMD_COMP_VAGG
typedef struct { GstVideoAggregatorClass parent_class; } GstAmdCompositorClass;
#else
typedef struct { GstAggregatorClass parent_class; } GstAmdCompositorClass;
#endif

enum { PROP_0, PROP_BACKGROUND, PROP_DEVICE_ID, PROP_ZERO_SIZE_IS_UNSCALED, PROP_MAX_THREADS, PROP_IGNORE_INACTIVE_PADS, PROP_CULLED_FRAMES };

/* GstChildProxy (as compositor.c:2185-2220 does): lets `sink_1::xpos=..` address pad properties from gst-launch */
static GObject *
amd_comp_child_by_index (GstChildProxy * proxy, guint index)
{
  GObject *obj;
  GST_OBJECT_LOCK (proxy);
  obj = g_list_nth_data (GST_ELEMENT_CAST (proxy)->sinkpads, index);
  if (obj)
    gst_object_ref (obj);
  GST_OBJECT_UNLOCK (proxy);
  return obj;
}

static guint
amd_comp_children_count (GstChildProxy * proxy)
{
  guint n;
  GST_OBJECT_LOCK (proxy);
  n = GST_ELEMENT_CAST (proxy)->numsinkpads;
  GST_OBJECT_UNLOCK (proxy);
  return n;
}

static void
amd_comp_child_proxy_init (gpointer g_iface, gpointer iface_data)
{
  GstChildProxyInterface *iface = g_iface;
  iface->get_child_by_index = amd_comp_child_by_index;
  iface->get_children_count = amd_comp_children_count;
}

#if AMD_COMP_VAGG
G_DEFINE_TYPE_WITH_CODE (GstAmdCompositor, gst_amd_compositor, GST_TYPE_VIDEO_AGGREGATOR,          /* compositor.c:808-810 */
    G_IMPLEMENT_INTERFACE (GST_TYPE_CHILD_PROXY, amd_comp_child_proxy_init));
#else
G_DEFINE_TYPE_WITH_CODE (GstAmdCompositor, gst_amd_compositor, GST_TYPE_AGGREGATOR,
    G_IMPLEMENT_INTERFACE (GST_TYPE_CHILD_PROXY, amd_comp_child_proxy_init));
#endif
#define AMD_COMP(o) ((GstAmdCompositor *) (o))

static GstPad *
amd_comp_request_new_pad (GstElement * element, GstPadTemplate * templ, const gchar * req_name, const GstCaps * caps)
{
  GstPad *pad = GST_ELEMENT_CLASS (g_type_class_peek_parent (G_OBJECT_GET_CLASS (element)))->request_new_pad (element, templ, req_name, caps);
  if (pad)
    gst_child_proxy_child_added (GST_CHILD_PROXY (element), G_OBJECT (pad), GST_OBJECT_NAME (pad));
  return pad;
}

static void
amd_comp_release_pad (GstElement * element, GstPad * pad)
{
  gst_child_proxy_child_removed (GST_CHILD_PROXY (element), G_OBJECT (pad), GST_OBJECT_NAME (pad));
  GST_ELEMENT_CLASS (g_type_class_peek_parent (G_OBJECT_GET_CLASS (element)))->release_pad (element, pad);
}

static int
amd_format_of (GstVideoFormat f)
{
  switch (f) {
    case GST_VIDEO_FORMAT_BGRA: return GSTAMD_VIDEO_FORMAT_BGRA;
    case GST_VIDEO_FORMAT_RGBA: return GSTAMD_VIDEO_FORMAT_RGBA;
    case GST_VIDEO_FORMAT_ARGB: return GSTAMD_VIDEO_FORMAT_ARGB;
    case GST_VIDEO_FORMAT_ABGR: return GSTAMD_VIDEO_FORMAT_ABGR;
    case GST_VIDEO_FORMAT_AYUV: return GSTAMD_VIDEO_FORMAT_AYUV;
    /* 16 bits per component: blend_argb64 / overlay_argb64 (compositor.c:1048-1059) */
    case GST_VIDEO_FORMAT_ARGB64: return GSTAMD_VIDEO_FORMAT_ARGB64;
    case GST_VIDEO_FORMAT_AYUV64: return GSTAMD_VIDEO_FORMAT_AYUV64;
    /* outputs without per-pixel alpha: pads are blended plane by plane (gstamd_compositor_aggregate_frame) */
    case GST_VIDEO_FORMAT_I420: return GSTAMD_VIDEO_FORMAT_I420;
    case GST_VIDEO_FORMAT_YV12: return GSTAMD_VIDEO_FORMAT_YV12;
    case GST_VIDEO_FORMAT_Y42B: return GSTAMD_VIDEO_FORMAT_Y42B;
    case GST_VIDEO_FORMAT_Y444: return GSTAMD_VIDEO_FORMAT_Y444;
    case GST_VIDEO_FORMAT_NV12: return GSTAMD_VIDEO_FORMAT_NV12;
    case GST_VIDEO_FORMAT_NV21: return GSTAMD_VIDEO_FORMAT_NV21;
    case GST_VIDEO_FORMAT_RGB: return GSTAMD_VIDEO_FORMAT_RGB;
    case GST_VIDEO_FORMAT_BGR: return GSTAMD_VIDEO_FORMAT_BGR;
    /* RGB_BLEND with four bytes per pixel and PACKED_422_BLEND (blend.c:1768-1925) */
    case GST_VIDEO_FORMAT_RGBx: return GSTAMD_VIDEO_FORMAT_RGBx;
    case GST_VIDEO_FORMAT_BGRx: return GSTAMD_VIDEO_FORMAT_BGRx;
    case GST_VIDEO_FORMAT_xRGB: return GSTAMD_VIDEO_FORMAT_xRGB;
    case GST_VIDEO_FORMAT_xBGR: return GSTAMD_VIDEO_FORMAT_xBGR;
    case GST_VIDEO_FORMAT_YUY2: return GSTAMD_VIDEO_FORMAT_YUY2;
    case GST_VIDEO_FORMAT_UYVY: return GSTAMD_VIDEO_FORMAT_UYVY;
    case GST_VIDEO_FORMAT_YVYU: return GSTAMD_VIDEO_FORMAT_YVYU;
#if GST_CHECK_VERSION (1, 16, 0)
    case GST_VIDEO_FORMAT_VUYA: return GSTAMD_VIDEO_FORMAT_VUYA;          /* BGRA's blend / overlay, its own fills (blend.h:58-65) */
#endif
    /* ... and the planar canvases of 10 / 12 / 16 bits (blend.c:609-697: compositor_orc_blend_u10 / u12 / u16) */
    case GST_VIDEO_FORMAT_I420_10LE: return GSTAMD_VIDEO_FORMAT_I420_10LE;
    case GST_VIDEO_FORMAT_I420_12LE: return GSTAMD_VIDEO_FORMAT_I420_12LE;
    case GST_VIDEO_FORMAT_I422_10LE: return GSTAMD_VIDEO_FORMAT_I422_10LE;
    case GST_VIDEO_FORMAT_I422_12LE: return GSTAMD_VIDEO_FORMAT_I422_12LE;
    case GST_VIDEO_FORMAT_Y444_10LE: return GSTAMD_VIDEO_FORMAT_Y444_10LE;
    case GST_VIDEO_FORMAT_Y444_12LE: return GSTAMD_VIDEO_FORMAT_Y444_12LE;
#if GST_CHECK_VERSION (1, 18, 0)
    case GST_VIDEO_FORMAT_Y444_16LE: return GSTAMD_VIDEO_FORMAT_Y444_16LE;
#endif
    default: return 0;
  }
}

static int
amd_pad_format_of (GstVideoFormat f)
{
  switch (f) {
    case GST_VIDEO_FORMAT_NV16: return GSTAMD_VIDEO_FORMAT_NV16;
    case GST_VIDEO_FORMAT_NV61: return GSTAMD_VIDEO_FORMAT_NV61;
    case GST_VIDEO_FORMAT_NV24: return GSTAMD_VIDEO_FORMAT_NV24;
    case GST_VIDEO_FORMAT_VYUY: return GSTAMD_VIDEO_FORMAT_VYUY;
    case GST_VIDEO_FORMAT_GRAY8: return GSTAMD_VIDEO_FORMAT_GRAY8;
    case GST_VIDEO_FORMAT_GBR: return GSTAMD_VIDEO_FORMAT_GBR;
    case GST_VIDEO_FORMAT_P010_10LE: return GSTAMD_VIDEO_FORMAT_P010_10LE;
#if GST_CHECK_VERSION (1, 18, 0)
    case GST_VIDEO_FORMAT_P012_LE: return GSTAMD_VIDEO_FORMAT_P012_LE;
    case GST_VIDEO_FORMAT_P016_LE: return GSTAMD_VIDEO_FORMAT_P016_LE;
#endif
    default:
      break;
  }
  if (amd_format_of (f))
    return amd_format_of (f);
  /* every other format the converter knows: the ABI's format numbers ARE GstVideoFormat's (include/gstamd_video.h), the library says whether it has this one */
  {
    GstAmdVideoInfo probe;
    return gstamd_video_info_set_format (&probe, (int) f, 16, 16) == GSTAMD_OK ? (int) f : 0;
  }
}

/* GstVideoInfo -> GstAmdVideoInfo with the caps' colorimetry / chroma site (as the videoconvertscale element does) */
static gboolean
amd_fill_info (const GstVideoInfo * vi, int w, int h, int fmt, GstAmdVideoInfo * ai)
{
  if (!fmt || gstamd_video_info_set_format (ai, fmt, w, h) != GSTAMD_OK)
    return FALSE;
  if (vi) {
    ai->color_range = vi->colorimetry.range;
    ai->color_matrix = vi->colorimetry.matrix;
    ai->chroma_site = vi->chroma_site;
  }
  return TRUE;
}

/* Size and in-rectangle offset of the pad's picture on the canvas: _mixer_pad_get_output_size (compositor.c:290-412).  The width /
 * height properties (or the frame size), corrected for the pixel aspect ratios of pad and canvas; sizing-policy none stretches the
 * picture over that rectangle (preferring to keep the height), keep-aspect-ratio fits it inside, centred. */
static void
amd_comp_pad_output_size (gboolean zero_is_unscaled, GstAmdCompositorPadObj * p, gint out_par_n, gint out_par_d, gint * width, gint * height,
    gint * x_off, gint * y_off)
{
  const gint fw = GST_VIDEO_INFO_WIDTH (&p->info), fh = GST_VIDEO_INFO_HEIGHT (&p->info);
  const gint pn = GST_VIDEO_INFO_PAR_N (&p->info), pd = GST_VIDEO_INFO_PAR_D (&p->info);
  gint pw, ph;
  guint dar_n, dar_d;

  *width = *height = *x_off = *y_off = 0;
  if (!p->have_info)
    return;
  pw = zero_is_unscaled ? (p->width <= 0 ? fw : p->width) : (p->width < 0 ? fw : p->width);
  ph = zero_is_unscaled ? (p->height <= 0 ? fh : p->height) : (p->height < 0 ? fh : p->height);
  if (pw == 0 || ph == 0)
    return;
  if (!gst_video_calculate_display_ratio (&dar_n, &dar_d, pw, ph, pn, pd, out_par_n, out_par_d))
    return;
  if (p->sizing_policy == 0) {
    if (ph % dar_n == 0)
      pw = gst_util_uint64_scale_int (ph, dar_n, dar_d);
    else if (pw % dar_d == 0)
      ph = gst_util_uint64_scale_int (pw, dar_d, dar_n);
    else
      pw = gst_util_uint64_scale_int (ph, dar_n, dar_d);
  } else {
    gint from_n, from_d, to_n, to_d, num, den;
    if (!gst_util_fraction_multiply (fw, fh, pn, pd, &from_n, &from_d))
      from_n = from_d = -1;
    if (!gst_util_fraction_multiply (pw, ph, out_par_n, out_par_d, &to_n, &to_d))
      to_n = to_d = -1;
    if (from_n != to_n || from_d != to_d) {
      if (from_n != -1 && from_d != -1 && gst_util_fraction_multiply (from_n, from_d, out_par_d, out_par_n, &num, &den)) {
        GstVideoRectangle src, dst, res;
        src.x = src.y = 0;
        src.w = pw;
        src.h = gst_util_uint64_scale_int (pw, den, num);
        if (src.h == 0)
          return;
        dst.x = dst.y = 0;
        dst.w = pw;
        dst.h = ph;
        gst_video_sink_center_rect (src, dst, &res, TRUE);
        *x_off = res.x;
        *y_off = res.y;
        pw = res.w;
        ph = res.h;
      } else {
        return;
      }
    }
  }
  *width = pw;
  *height = ph;
}

static void
amd_comp_pad_target_size (GstAmdCompositorPadObj * p, gint * w, gint * h)
{
  gint xo, yo;
  GstAmdCompositor *c = GST_OBJECT_PARENT (p) ? (GstAmdCompositor *) GST_OBJECT_PARENT (p) : NULL;
  amd_comp_pad_output_size (c ? c->zero_size_is_unscaled : TRUE, p, c && c->have_out ? GST_VIDEO_INFO_PAR_N (&c->out_info) : 1,
      c && c->have_out ? GST_VIDEO_INFO_PAR_D (&c->out_info) : 1, w, h, &xo, &yo);
}

static void
amd_comp_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  if (id == PROP_BACKGROUND)
    AMD_COMP (object)->background = g_value_get_enum (value);
  else if (id == PROP_DEVICE_ID)
    AMD_COMP (object)->device_id = g_value_get_int (value);
  else if (id == PROP_ZERO_SIZE_IS_UNSCALED)
    AMD_COMP (object)->zero_size_is_unscaled = g_value_get_boolean (value);
  else if (id == PROP_MAX_THREADS)
    AMD_COMP (object)->max_threads = g_value_get_uint (value);
  else if (id == PROP_IGNORE_INACTIVE_PADS) {
    AMD_COMP (object)->ignore_inactive_pads = g_value_get_boolean (value);
#if GST_CHECK_VERSION (1, 20, 0)
    gst_aggregator_set_ignore_inactive_pads (GST_AGGREGATOR (object), AMD_COMP (object)->ignore_inactive_pads);        /* compositor.c:2057 */
#endif
  } else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec);
}

static void
amd_comp_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  if (id == PROP_BACKGROUND)
    g_value_set_enum (value, AMD_COMP (object)->background);
  else if (id == PROP_DEVICE_ID)
    g_value_set_int (value, AMD_COMP (object)->device_id);
  else if (id == PROP_ZERO_SIZE_IS_UNSCALED)
    g_value_set_boolean (value, AMD_COMP (object)->zero_size_is_unscaled);
  else if (id == PROP_MAX_THREADS)
    g_value_set_uint (value, AMD_COMP (object)->max_threads);
  else if (id == PROP_IGNORE_INACTIVE_PADS)
    g_value_set_boolean (value, AMD_COMP (object)->ignore_inactive_pads);
  else if (id == PROP_CULLED_FRAMES)
    g_value_set_uint64 (value, AMD_COMP (object)->n_culled);
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec);
}

#if !AMD_COMP_VAGG              /* GstVideoAggregator makes the pads, takes the CAPS events and answers the caps queries itself */
static GstAggregatorPad *
amd_comp_create_new_pad (GstAggregator * agg, GstPadTemplate * templ, const gchar * req_name, const GstCaps * caps)
{
  GstAmdCompositor *c = AMD_COMP (agg);
  guint serial = 0;
  gchar *name;
  GstAmdCompositorPadObj *pad;

  if (templ->direction != GST_PAD_SINK || !g_str_has_prefix (templ->name_template, "sink_"))
    return NULL;
  GST_OBJECT_LOCK (agg);
  if (req_name && strlen (req_name) > 5 && g_str_has_prefix (req_name, "sink_")) {
    serial = (guint) g_ascii_strtoull (req_name + 5, NULL, 10);
    if (serial >= c->next_pad)
      c->next_pad = serial + 1;
  } else {
    serial = c->next_pad++;
  }
  GST_OBJECT_UNLOCK (agg);
  name = g_strdup_printf ("sink_%u", serial);
  pad = g_object_new (gst_amd_compositor_pad_get_type (), "name", name, "direction", GST_PAD_SINK, "template", templ, NULL);
  g_free (name);
  pad->zorder = serial;        /* new pads go on top, as in gst_video_aggregator_request_new_pad */
  return GST_AGGREGATOR_PAD (pad);
}

static gboolean
amd_comp_sink_event (GstAggregator * agg, GstAggregatorPad * apad, GstEvent * event)
{
  if (GST_EVENT_TYPE (event) == GST_EVENT_CAPS) {
    GstAmdCompositorPadObj *p = AMD_COMP_PAD (apad);
    GstCaps *caps;
    gst_event_parse_caps (event, &caps);
    GstVideoInfo vi;
    /* progressive frames only: neither the blend nor the per-pad converter has a field-aware path (an interlaced pad would be
     * composited as if progressive), so such caps are refused like any other unsupported format */
    const gboolean ok = gst_video_info_from_caps (&vi, caps) && amd_pad_format_of (GST_VIDEO_INFO_FORMAT (&vi)) != 0 &&
        !GST_VIDEO_INFO_IS_INTERLACED (&vi);
    if (!ok) {
      GST_ERROR_OBJECT (apad, "unsupported caps %" GST_PTR_FORMAT, caps);
      gst_event_unref (event);
      return FALSE;
    }
    /* The pad may still show a frame of the OLD caps (p->current, repeated until a frame of the new caps starts): the new layout is
     * kept pending and becomes p->info at the moment a buffer queued after this event becomes the pad's current frame
     * (amd_comp_pad_select) - gst_video_aggregator keeps pending_vinfo / pending_caps the same way (gstvideoaggregator.c:1825-1831,
     * 1931-1937).  Describing the old buffer with the new width / height / stride made the blend kernels read past it. */
    GST_OBJECT_LOCK (p);
    p->pending_info = vi;
    p->have_pending = TRUE;
    if (!p->have_info) {          /* first caps: nothing is shown yet */
      p->info = vi;
      p->have_info = TRUE;
      p->have_pending = FALSE;
    }
    GST_OBJECT_UNLOCK (p);
    gst_pad_mark_reconfigure (agg->srcpad);
  }
  return GST_AGGREGATOR_CLASS (gst_amd_compositor_parent_class)->sink_event (agg, apad, event);
}

#endif

/* (also on GstVideoAggregator: its own answer ties a sink pad to the memory type downstream negotiated and - when the output format has no
 * alpha - to formats without alpha, gstvideoaggregator.c:1597-1665; here system-memory and HBM pads mix freely and every pad is converted
 * to the canvas format on the GPU, so the pads accept what their template says) */
static gboolean
amd_comp_sink_query (GstAggregator * agg, GstAggregatorPad * apad, GstQuery * query)
{
  if (GST_QUERY_TYPE (query) == GST_QUERY_CAPS) {
    /* any supported format, size and framerate: pads are converted / scaled individually */
    GstCaps *filter, *tmpl = gst_pad_get_pad_template_caps (GST_PAD (apad)), *res;
    gst_query_parse_caps (query, &filter);
    if (filter) {
      res = gst_caps_intersect_full (filter, tmpl, GST_CAPS_INTERSECT_FIRST);
      gst_caps_unref (tmpl);
    } else {
      res = tmpl;
    }
    gst_query_set_caps_result (query, res);
    gst_caps_unref (res);
    return TRUE;
  }
  if (GST_QUERY_TYPE (query) == GST_QUERY_ACCEPT_CAPS) {
    GstCaps *caps, *tmpl = gst_pad_get_pad_template_caps (GST_PAD (apad));
    gst_query_parse_accept_caps (query, &caps);
    gst_query_set_accept_caps_result (query, gst_caps_can_intersect (caps, tmpl));
    gst_caps_unref (tmpl);
    return TRUE;
  }
  return GST_AGGREGATOR_CLASS (gst_amd_compositor_parent_class)->sink_query (agg, apad, query);
}

/* bounding box of the pads, best framerate, the pads' common format (compositor.c _fixate_caps / _update_caps) */
static GstFlowReturn
amd_comp_update_src_caps (GstAggregator * agg, GstCaps * caps, GstCaps ** ret)
{
  GList *l;
  gint best_w = 0, best_h = 0, fps_n = 0, fps_d = 1;
  gdouble best_fps = -1.0;
  GstVideoFormat fmt = GST_VIDEO_FORMAT_UNKNOWN;
  gboolean all = TRUE, any = FALSE;
  GstCaps *want, *want_hip;

  GST_OBJECT_LOCK (agg);
  for (l = GST_ELEMENT (agg)->sinkpads; l; l = l->next) {
    GstAmdCompositorPadObj *p = AMD_COMP_PAD (l->data);
    gint w, h;
    gdouble fps;
#if AMD_COMP_VAGG
    p->info = GST_VIDEO_AGGREGATOR_PAD (p)->info;         /* the base class keeps the pad's negotiated layout */
    p->have_info = p->info.finfo != NULL && GST_VIDEO_INFO_FORMAT (&p->info) != GST_VIDEO_FORMAT_UNKNOWN;
#endif
    if (!p->have_info) {
      all = FALSE;
      continue;
    }
    any = TRUE;
    amd_comp_pad_target_size (p, &w, &h);
    w += p->xpos;
    h += p->ypos;
    best_w = MAX (best_w, w);
    best_h = MAX (best_h, h);
    if (fmt == GST_VIDEO_FORMAT_UNKNOWN && amd_format_of (GST_VIDEO_INFO_FORMAT (&p->info)))
      fmt = GST_VIDEO_INFO_FORMAT (&p->info);
    fps = GST_VIDEO_INFO_FPS_D (&p->info) ? (gdouble) GST_VIDEO_INFO_FPS_N (&p->info) / GST_VIDEO_INFO_FPS_D (&p->info) : 0.0;
    if (fps > best_fps) {
      best_fps = fps;
      fps_n = GST_VIDEO_INFO_FPS_N (&p->info);
      fps_d = GST_VIDEO_INFO_FPS_D (&p->info);
    }
  }
  GST_OBJECT_UNLOCK (agg);
  if (!any || !all)
    return GST_AGGREGATOR_FLOW_NEED_DATA;
  if (fps_d == 0 || fps_n == 0) {
    fps_n = 25;
    fps_d = 1;
  }
  if (fmt == GST_VIDEO_FORMAT_UNKNOWN)
    fmt = GST_VIDEO_FORMAT_BGRA;        /* no pad carries a blendable format: composite in BGRA */
  /* preferences: pads' format; downstream may still pick another size (the canvas is simply that big) */
  want = gst_caps_new_simple ("video/x-raw", "format", G_TYPE_STRING, gst_video_format_to_string (fmt),
      "framerate", GST_TYPE_FRACTION, fps_n, fps_d, "pixel-aspect-ratio", GST_TYPE_FRACTION, 1, 1, NULL);
  want_hip = gst_caps_copy (want);
  gst_caps_set_features (want_hip, 0, gst_caps_features_from_string (GST_CAPS_FEATURE_MEMORY_AMD_HIP));
  gst_caps_append (want_hip, want);
  *ret = gst_caps_intersect_full (caps, want_hip, GST_CAPS_INTERSECT_FIRST);
  gst_caps_unref (want_hip);
  if (gst_caps_is_empty (*ret)) {
    /* downstream insists on another blendable format: composite in that one, every pad gets converted */
    GstCaps *any = gst_caps_new_simple ("video/x-raw", "framerate", GST_TYPE_FRACTION, fps_n, fps_d, "pixel-aspect-ratio",
        GST_TYPE_FRACTION, 1, 1, NULL);
    GstCaps *any_hip = gst_caps_copy (any);
    gst_caps_set_features (any_hip, 0, gst_caps_features_from_string (GST_CAPS_FEATURE_MEMORY_AMD_HIP));
    gst_caps_append (any_hip, any);
    gst_caps_unref (*ret);
    *ret = gst_caps_intersect_full (caps, any_hip, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (any_hip);
  }
  g_object_set_data (G_OBJECT (agg), "amd-best-w", GINT_TO_POINTER (best_w));
  g_object_set_data (G_OBJECT (agg), "amd-best-h", GINT_TO_POINTER (best_h));
  return GST_FLOW_OK;
}

static GstCaps *
amd_comp_fixate_src_caps (GstAggregator * agg, GstCaps * caps)
{
  const gint best_w = GPOINTER_TO_INT (g_object_get_data (G_OBJECT (agg), "amd-best-w"));
  const gint best_h = GPOINTER_TO_INT (g_object_get_data (G_OBJECT (agg), "amd-best-h"));
  GstStructure *s;
  caps = gst_caps_truncate (gst_caps_make_writable (caps));
  s = gst_caps_get_structure (caps, 0);
  gst_structure_fixate_field_nearest_int (s, "width", best_w > 0 ? best_w : 320);
  gst_structure_fixate_field_nearest_int (s, "height", best_h > 0 ? best_h : 240);
  return gst_caps_fixate (caps);
}

static gboolean
amd_comp_negotiated_src_caps (GstAggregator * agg, GstCaps * caps)
{
  GstAmdCompositor *c = AMD_COMP (agg);
  GstCapsFeatures *f = gst_caps_get_features (caps, 0);
  if (!gst_video_info_from_caps (&c->out_info, caps) || !amd_format_of (GST_VIDEO_INFO_FORMAT (&c->out_info)))
    return FALSE;
  c->out_hip = f && gst_caps_features_contains (f, GST_CAPS_FEATURE_MEMORY_AMD_HIP);
  if (c->out_pool) {
    gst_buffer_pool_set_active (c->out_pool, FALSE);
    gst_object_unref (c->out_pool);
    c->out_pool = NULL;
  }
  if (c->out_hip && !(c->out_pool = gst_amd_hip_buffer_pool_new_for_caps (caps, 2)))
    return FALSE;
  c->have_out = TRUE;
  GST_INFO_OBJECT (c, "output %" GST_PTR_FORMAT, caps);
#if AMD_COMP_VAGG
  return GST_AGGREGATOR_CLASS (gst_amd_compositor_parent_class)->negotiated_src_caps (agg, caps);          /* GstVideoAggregator::info, the pads' conversion info, latency */
#else
  return TRUE;
#endif
}

#if !AMD_COMP_VAGG
static gint
pad_zorder_cmp (gconstpointer a, gconstpointer b)
{
  const GstAmdCompositorPadObj *pa = *(GstAmdCompositorPadObj * const *) a, *pb = *(GstAmdCompositorPadObj * const *) b;
  return pa->zorder < pb->zorder ? -1 : (pa->zorder > pb->zorder ? 1 : 0);
}

#endif

static gboolean
ensure_device (gpointer * p, gsize * have, gsize need)
{
  if (*have >= need)
    return TRUE;
  gstamd_device_free (*p);
  *p = gstamd_device_alloc (need);
  *have = *p ? need : 0;
  return *p != NULL;
}

#if !AMD_COMP_VAGG
/* Which queued frame does the pad show during the output frame [out_start, out_end) (running time)?  The rule of
 * gst_video_aggregator_fill_queues (gstvideoaggregator.c:1753-2000): frames that ended before the output frame starts are dropped,
 * a frame that starts at or after its end stays queued (the pad keeps showing what it shows), anything else becomes the pad's
 * current frame and stays that until replaced - a 15 fps pad under a 30 fps output is shown twice.  After EOS the last frame stays
 * only with repeat-after-eos.  FALSE: the pad has nothing queued, is not EOS and its current frame does not reach out_end -
 * the aggregator has to wait for data. */
/* a buffer queued after the pad's last CAPS event is about to become its current frame: the pending layout is now the frame's */
static void
amd_comp_pad_promote_caps (GstAmdCompositorPadObj * p)
{
  gboolean promoted = FALSE;
  GST_OBJECT_LOCK (p);
  if (p->have_pending) {
    p->info = p->pending_info;
    p->have_pending = FALSE;
    promoted = TRUE;
  }
  GST_OBJECT_UNLOCK (p);
  if (promoted) {               /* the canvas size / format follow the pads' layouts: renegotiate before the next output frame */
    GstObject *agg = gst_object_get_parent (GST_OBJECT (p));
    if (agg) {
      gst_pad_mark_reconfigure (GST_AGGREGATOR (agg)->srcpad);
      gst_object_unref (agg);
    }
  }
}

static gboolean
amd_comp_pad_select (GstAmdCompositorPadObj * p, GstClockTime out_start, GstClockTime out_end, gboolean * is_eos)
{
  GstAggregatorPad *ap = GST_AGGREGATOR_PAD (p);

  *is_eos = FALSE;
  for (;;) {
    GstBuffer *b = gst_aggregator_pad_peek_buffer (ap);
    GstClockTime start, end = GST_CLOCK_TIME_NONE;

    if (!b) {
      if (gst_aggregator_pad_is_eos (ap)) {
        /* the last frame is shown until it ends; after that only with repeat-after-eos (which never ends the stream by itself) */
        if (p->current && !p->repeat_after_eos && GST_CLOCK_TIME_IS_VALID (p->cur_end) && p->cur_end > out_start)
          return TRUE;
        *is_eos = TRUE;
        if (!p->repeat_after_eos)
          gst_buffer_replace (&p->current, NULL);
        return TRUE;
      }
      /* max-last-buffer-repeat (gstvideoaggregator.c:1984-2027): a frame that is over and has no successor yet goes on being shown -
       * unless the output frame starts more than that many ns after its end (its start, for a frame without a duration): then the pad
       * shows nothing until data arrives (a stalled network source falls back to the pads below it) */
      if (p->current && GST_CLOCK_TIME_IS_VALID (p->max_last_buffer_repeat)) {
        const GstClockTime ref_t = GST_CLOCK_TIME_IS_VALID (p->cur_end) ? p->cur_end : p->cur_start;
        if (GST_CLOCK_TIME_IS_VALID (ref_t) && ref_t <= out_start && out_start - ref_t > p->max_last_buffer_repeat)
          gst_buffer_replace (&p->current, NULL);
      }
      /* an untimed frame is used once; a frame without a duration ends where the next one starts, which is not known yet */
      return p->current && GST_CLOCK_TIME_IS_VALID (p->cur_end) && p->cur_end >= out_end;
    }
    if (!GST_BUFFER_PTS_IS_VALID (b)) {         /* untimed: shown as it comes */
      amd_comp_pad_promote_caps (p);
      gst_buffer_replace (&p->current, b);
      p->cur_start = p->cur_end = GST_CLOCK_TIME_NONE;
      gst_buffer_unref (b);
      gst_aggregator_pad_drop_buffer (ap);
      return TRUE;
    }
    start = gst_segment_to_running_time (&ap->segment, GST_FORMAT_TIME, GST_BUFFER_PTS (b));
    if (GST_BUFFER_DURATION_IS_VALID (b))
      end = gst_segment_to_running_time (&ap->segment, GST_FORMAT_TIME, GST_BUFFER_PTS (b) + GST_BUFFER_DURATION (b));
    if (!GST_CLOCK_TIME_IS_VALID (start)) {     /* outside the segment */
      gst_buffer_unref (b);
      gst_aggregator_pad_drop_buffer (ap);
      continue;
    }
    if (GST_CLOCK_TIME_IS_VALID (end) && end <= out_start) {    /* over before this output frame begins */
      gst_buffer_unref (b);
      gst_aggregator_pad_drop_buffer (ap);
      continue;
    }
    if (start >= out_end) {                     /* belongs to a later output frame */
      gst_buffer_unref (b);
      return p->current != NULL || start >= out_end;
    }
    amd_comp_pad_promote_caps (p);
    gst_buffer_replace (&p->current, b);
    p->cur_start = start;
    p->cur_end = end;
    gst_buffer_unref (b);
    gst_aggregator_pad_drop_buffer (ap);
    return TRUE;
  }
}

#endif

/* rectangle (x, y, w, h) clamped to the canvas (clamp_rectangle, compositor.c:441-459) */
static GstVideoRectangle
amd_comp_clamp (gint x, gint y, gint w, gint h, gint cw, gint ch)
{
  GstVideoRectangle r;
  r.x = CLAMP (x, 0, cw);
  r.y = CLAMP (y, 0, ch);
  r.w = CLAMP (x + w, 0, cw) - r.x;
  r.h = CLAMP (y + h, 0, ch) - r.y;
  return r;
}

/* One output frame: the pads' current frames (bufs[i], NULL: the pad shows nothing; pads in z order) composited into the canvas.
 * outbuf_in: the frame to fill (GstVideoAggregator's create_output_buffer made it), NULL: made here.  Consumes the references in pads[] / bufs[]. */
static GstFlowReturn
amd_comp_compose (GstAmdCompositor * c, GstAmdCompositorPadObj ** pads, GstBuffer ** bufs, guint n, GstBuffer * outbuf_in, GstBuffer ** outbuf_out)
{
  GstMapInfo maps[AMD_COMP_MAX_PADS];
  gboolean mapped_dev[AMD_COMP_MAX_PADS];
  GstAmdCompositorPad desc[AMD_COMP_MAX_PADS];
  GstBuffer *outbuf = outbuf_in;
  GstMapInfo omap;
  GstMemory *omem = NULL;
  gpointer canvas;
  GstFlowReturn flow = GST_FLOW_OK;
  guint i, n_desc = 0;
  const int fmt = amd_format_of (GST_VIDEO_INFO_FORMAT (&c->out_info));
  const gboolean by_planes = !GST_VIDEO_INFO_HAS_ALPHA (&c->out_info);
  GstAmdCompositorFramePad fdesc[AMD_COMP_MAX_PADS];
  GstAmdCompositorPadOpacity odesc[AMD_COMP_MAX_PADS];
  GstAmdVideoConverter *inline_conv[AMD_COMP_MAX_PADS];
  gboolean any_inline = FALSE;
  gint pad_w[AMD_COMP_MAX_PADS], pad_h[AMD_COMP_MAX_PADS], pad_x[AMD_COMP_MAX_PADS], pad_y[AMD_COMP_MAX_PADS];
  guint n_culled = 0;
  int r;

  memset (mapped_dev, 0, sizeof (mapped_dev));
  *outbuf_out = NULL;
  gst_amd_hip_select_device (c->device_id);
  if (!c->stream && !(c->stream = gstamd_stream_new ())) {
    GST_ELEMENT_ERROR (c, LIBRARY, INIT, ("no HIP stream"), ("%s", gstamd_last_error ()));
    flow = GST_FLOW_ERROR;
    goto done_inputs;
  }

  /* where each pad lands, and which pads cannot be seen at all (_should_draw_background / prepare_frame_start,
   * compositor.c:464-601): alpha 0, nothing left after clamping to the canvas, or fully under an opaque pad above */
  {
    const gint cw = GST_VIDEO_INFO_WIDTH (&c->out_info), ch = GST_VIDEO_INFO_HEIGHT (&c->out_info);
    const gint opn = GST_VIDEO_INFO_PAR_N (&c->out_info), opd = GST_VIDEO_INFO_PAR_D (&c->out_info);
    for (i = 0; i < n; i++) {
      gint xo = 0, yo = 0;
      pad_w[i] = pad_h[i] = pad_x[i] = pad_y[i] = 0;
      if (!bufs[i] || !pads[i]->have_info)
        continue;
      amd_comp_pad_output_size (c->zero_size_is_unscaled, pads[i], opn, opd, &pad_w[i], &pad_h[i], &xo, &yo);
      pad_x[i] = pads[i]->xpos + xo;
      pad_y[i] = pads[i]->ypos + yo;
    }
    for (i = 0; i < n; i++) {
      GstVideoRectangle fr;
      guint j;
      if (!bufs[i] || !pads[i]->have_info)
        continue;
      fr = amd_comp_clamp (pad_x[i], pad_y[i], pad_w[i], pad_h[i], cw, ch);
      if (pads[i]->alpha == 0.0 || pad_w[i] <= 0 || pad_h[i] <= 0 || fr.w <= 0 || fr.h <= 0) {
        gst_buffer_replace (&bufs[i], NULL);
        n_culled++;
        continue;
      }
      for (j = i + 1; j < n; j++) {
        if (!bufs[j] || !pads[j]->have_info || pads[j]->alpha != 1.0 || GST_VIDEO_INFO_HAS_ALPHA (&pads[j]->info))
          continue;
        if (pad_x[j] <= fr.x && pad_y[j] <= fr.y && pad_x[j] + pad_w[j] >= fr.x + fr.w && pad_y[j] + pad_h[j] >= fr.y + fr.h) {
          gst_buffer_replace (&bufs[i], NULL);
          n_culled++;
          break;
        }
      }
    }
    c->n_culled += n_culled;
  }

  /* pads -> device pointers */
  for (i = 0; i < n; i++) {
    GstAmdCompositorPadObj *p = pads[i];
    GstMemory *mem;
    GstVideoMeta *vmeta;
    const guint8 *base;
    if (!bufs[i] || !p->have_info)
      continue;
    mem = gst_buffer_peek_memory (bufs[i], 0);
    if (gst_buffer_n_memory (bufs[i]) == 1 && gst_is_amd_hip_memory (mem)) {
      if (!gst_memory_map (mem, &maps[i], GST_MAP_READ | GST_MAP_AMDHIP)) {
        flow = GST_FLOW_ERROR;
        goto done_inputs;
      }
      mapped_dev[i] = TRUE;
      gst_amd_hip_memory_wait_written (mem, c->stream);       /* produced on another element's stream */
      base = maps[i].data;
    } else {
      GstMapInfo m;
      if (!gst_buffer_map (bufs[i], &m, GST_MAP_READ)) {
        flow = GST_FLOW_ERROR;
        goto done_inputs;
      }
      if (!ensure_device (&p->staging, &p->staging_size, m.size) || gstamd_device_upload_async (p->staging, m.data, m.size, c->stream) != GSTAMD_OK) {
        gst_buffer_unmap (bufs[i], &m);
        flow = GST_FLOW_ERROR;
        goto done_inputs;
      }
      gst_buffer_unmap (bufs[i], &m);
      base = p->staging;
    }
    {
      gint tw, th, k;
      const int ifmt = amd_pad_format_of (GST_VIDEO_INFO_FORMAT (&p->info));
      tw = pad_w[i];
      th = pad_h[i];
      GstAmdVideoConverterConfig pcfg;
      gboolean have_cfg, cfg_changed;
      odesc[n_desc].all_opaque = 0;
#if AMD_COMP_VAGG
      {
        /* GstVideoAggregatorConvertPad keeps converter-config privately: read it through the property, notice changes by comparing */
        GstStructure *cc = NULL;
        g_object_get (p, "converter-config", &cc, NULL);
        GST_OBJECT_LOCK (p);
        if ((cc == NULL) != (p->converter_config == NULL) || (cc && !gst_structure_is_equal (cc, p->converter_config))) {
          if (p->converter_config)
            gst_structure_free (p->converter_config);
          p->converter_config = cc;
          p->converter_config_changed = TRUE;
        } else if (cc) {
          gst_structure_free (cc);
        }
        GST_OBJECT_UNLOCK (p);
      }
#endif
      GST_OBJECT_LOCK (p);
      have_cfg = p->converter_config != NULL;
      cfg_changed = p->converter_config_changed;
      p->converter_config_changed = FALSE;
      if (have_cfg) {
        gstamd_video_converter_config_init (&pcfg);
        gst_amd_converter_config_from_structure (p->converter_config, &pcfg);
      }
      GST_OBJECT_UNLOCK (p);
      /* GstVideoAggregatorConvertPad (gstvideoaggregator.c:479-513): a converter exists when the pad's frames differ from what the canvas
       * wants - or when the pad has a converter-config, whose options (resampler method, alpha / chroma / matrix modes, dither ...) are the
       * converter's; without one the library defaults apply (cubic) */
      if (ifmt != fmt || tw != GST_VIDEO_INFO_WIDTH (&p->info) || th != GST_VIDEO_INFO_HEIGHT (&p->info) || have_cfg) {
        const gint key[6] = { ifmt, GST_VIDEO_INFO_WIDTH (&p->info), GST_VIDEO_INFO_HEIGHT (&p->info), fmt, tw, th };
        if (!p->conv || cfg_changed || memcmp (key, p->conv_key, sizeof (key)) != 0) {
          GstAmdVideoInfo ai, ao;
          int status = 0;
          if (p->conv)
            gstamd_video_converter_free (p->conv);
          p->conv = NULL;
          if (amd_fill_info (&p->info, key[1], key[2], ifmt, &ai) && amd_fill_info (NULL, tw, th, fmt, &ao)) {
            if (GST_VIDEO_INFO_IS_YUV (&c->out_info)) {
              ao.color_range = c->out_info.colorimetry.range;
              ao.color_matrix = c->out_info.colorimetry.matrix;
            }
            p->conv = gstamd_video_converter_new (&ai, &ao, have_cfg ? &pcfg : NULL, &status);
            p->conv_out = ao;
          }
          if (!p->conv) {
            GST_ELEMENT_ERROR (c, STREAM, FORMAT, ("no HIP conversion for pad %s", GST_OBJECT_NAME (p)), ("%s", gstamd_last_error ()));
            flow = GST_FLOW_NOT_NEGOTIATED;
            goto done_inputs;
          }
          memcpy (p->conv_key, key, sizeof (key));
          p->conv_inline = !by_planes && ifmt == fmt && GST_VIDEO_INFO_COMP_DEPTH (&c->out_info, 0) == 8 &&
              g_getenv ("GSTAMD_COMPOSITOR_NO_INLINE_SCALE") == NULL && gstamd_compositor_pad_scaler_usable (p->conv) == 1;
        }
        inline_conv[n_desc] = NULL;
        if (p->conv_inline) {
          /* the frame as it arrived + its scaler: no scaled frame in HBM, no extra launch */
          vmeta = gst_buffer_get_video_meta (bufs[i]);
          desc[n_desc].data = base + (vmeta ? vmeta->offset[0] : GST_VIDEO_INFO_PLANE_OFFSET (&p->info, 0));
          desc[n_desc].width = GST_VIDEO_INFO_WIDTH (&p->info);
          desc[n_desc].height = GST_VIDEO_INFO_HEIGHT (&p->info);
          desc[n_desc].stride = vmeta ? vmeta->stride[0] : GST_VIDEO_INFO_PLANE_STRIDE (&p->info, 0);
          memset (&fdesc[n_desc], 0, sizeof (fdesc[n_desc]));
          inline_conv[n_desc] = p->conv;
          any_inline = TRUE;
          c->n_inline_scaled++;
        } else {
        if (!ensure_device (&p->conv_buf, &p->conv_buf_size, (gsize) p->conv_out.size) ||
            gstamd_video_converter_frame (p->conv, base, p->conv_buf, c->stream) != GSTAMD_OK) {
          flow = GST_FLOW_ERROR;
          goto done_inputs;
        }
        desc[n_desc].data = p->conv_buf;
        odesc[n_desc].all_opaque = !have_cfg && !GST_VIDEO_INFO_HAS_ALPHA (&p->info);
        desc[n_desc].width = tw;
        desc[n_desc].height = th;
        desc[n_desc].stride = p->conv_out.stride[0];
        memset (&fdesc[n_desc], 0, sizeof (fdesc[n_desc]));
        for (k = 0; k < p->conv_out.n_planes && k < 3; k++) {
          fdesc[n_desc].data[k] = (const guint8 *) p->conv_buf + p->conv_out.offset[k];
          fdesc[n_desc].stride[k] = p->conv_out.stride[k];
        }
        }
      } else {
        inline_conv[n_desc] = NULL;
        vmeta = gst_buffer_get_video_meta (bufs[i]);
        desc[n_desc].data = base + (vmeta ? vmeta->offset[0] : GST_VIDEO_INFO_PLANE_OFFSET (&p->info, 0));
        desc[n_desc].width = GST_VIDEO_INFO_WIDTH (&p->info);
        desc[n_desc].height = GST_VIDEO_INFO_HEIGHT (&p->info);
        desc[n_desc].stride = vmeta ? vmeta->stride[0] : GST_VIDEO_INFO_PLANE_STRIDE (&p->info, 0);
        memset (&fdesc[n_desc], 0, sizeof (fdesc[n_desc]));
        for (k = 0; k < (gint) GST_VIDEO_INFO_N_PLANES (&p->info) && k < 3; k++) {
          fdesc[n_desc].data[k] = base + (vmeta ? vmeta->offset[k] : GST_VIDEO_INFO_PLANE_OFFSET (&p->info, k));
          fdesc[n_desc].stride[k] = vmeta ? vmeta->stride[k] : GST_VIDEO_INFO_PLANE_STRIDE (&p->info, k);
        }
      }
    }
    /* a frame a default converter made from a format without alpha has alpha 255 everywhere (the unpackers' 0xff, alpha-mode copy): strips of the
     * canvas it covers at pad alpha 1.0 need nothing from the pads under it (gstamd_compositor_aggregate_opaque - blend_pads' bytes all the same) */
    odesc[n_desc].map = NULL;
    odesc[n_desc].reserved = 0;
    desc[n_desc].xpos = pad_x[i];
    desc[n_desc].ypos = pad_y[i];
    desc[n_desc].alpha = p->alpha;
    desc[n_desc].blend_mode = p->op;
    desc[n_desc].reserved = 0;
    fdesc[n_desc].width = desc[n_desc].width;
    fdesc[n_desc].height = desc[n_desc].height;
    fdesc[n_desc].xpos = pad_x[i];
    fdesc[n_desc].ypos = pad_y[i];
    fdesc[n_desc].alpha = p->alpha;
    fdesc[n_desc].blend_mode = p->op;
    n_desc++;
  }

  /* canvas */
  if (c->out_hip) {
    GstFlowReturn pool_flow = outbuf ? GST_FLOW_OK : (c->out_pool ? gst_buffer_pool_acquire_buffer (c->out_pool, &outbuf, NULL) : GST_FLOW_ERROR);
    if (pool_flow != GST_FLOW_OK)
      outbuf = NULL;
    omem = outbuf ? gst_buffer_peek_memory (outbuf, 0) : NULL;
    if (!omem || !gst_memory_map (omem, &omap, GST_MAP_WRITE | GST_MAP_AMDHIP)) {
      /* a flushing pool (seek, state change) is passed on as it is; only a real failure posts an error below */
      flow = pool_flow != GST_FLOW_OK ? pool_flow : GST_FLOW_ERROR;
      goto done_inputs;
    }
    gst_amd_hip_memory_wait_idle (omem, c->stream);           /* a recycled canvas may still be read downstream */
    canvas = omap.data;
  } else {
    if (!outbuf)
      outbuf = gst_buffer_new_allocate (NULL, GST_VIDEO_INFO_SIZE (&c->out_info), NULL);
    if (!outbuf || !ensure_device (&c->d_out, &c->d_out_size, GST_VIDEO_INFO_SIZE (&c->out_info))) {
      flow = GST_FLOW_ERROR;
      goto done_inputs;
    }
    canvas = c->d_out;
  }
  if (by_planes) {
    /* black_color / white_color of the element (compositor.c:1131-1149): the range's offset and offset + scale */
    const gboolean yuv = GST_VIDEO_INFO_IS_YUV (&c->out_info);
    gint offset[GST_VIDEO_MAX_COMPONENTS], scale[GST_VIDEO_MAX_COMPONENTS];
    int32_t black[3], white[3];
    gst_video_color_range_offsets (c->out_info.colorimetry.range, c->out_info.finfo, offset, scale);          /* at the format's own depth */
    black[0] = offset[0], white[0] = scale[0] + offset[0];
    black[1] = offset[1], white[1] = yuv ? offset[1] : scale[1] + offset[1];
    black[2] = offset[2], white[2] = yuv ? offset[2] : scale[2] + offset[2];
    void *dplanes[3] = { NULL, NULL, NULL };
    int32_t dstrides[3] = { 0, 0, 0 };
    gint k;
    for (k = 0; k < (gint) GST_VIDEO_INFO_N_PLANES (&c->out_info) && k < 3; k++) {
      dplanes[k] = (guint8 *) canvas + GST_VIDEO_INFO_PLANE_OFFSET (&c->out_info, k);
      dstrides[k] = GST_VIDEO_INFO_PLANE_STRIDE (&c->out_info, k);
    }
    r = gstamd_compositor_aggregate_frame (fmt, c->background, black, white, fdesc, (int) n_desc, dplanes, dstrides,
        GST_VIDEO_INFO_WIDTH (&c->out_info), GST_VIDEO_INFO_HEIGHT (&c->out_info), c->stream);
  } else if (any_inline) {
    GstAmdCompositorScaledPad sdesc[AMD_COMP_MAX_PADS];
    guint k;
    for (k = 0; k < n_desc; k++) {
      sdesc[k].data = desc[k].data;
      sdesc[k].width = desc[k].width;
      sdesc[k].height = desc[k].height;
      sdesc[k].stride = desc[k].stride;
      sdesc[k].xpos = desc[k].xpos;
      sdesc[k].ypos = desc[k].ypos;
      sdesc[k].alpha = desc[k].alpha;
      sdesc[k].blend_mode = desc[k].blend_mode;
      sdesc[k].reserved = 0;
      sdesc[k].scaler = inline_conv[k];
    }
    r = gstamd_compositor_aggregate_scaled (fmt, c->background, sdesc, (int) n_desc, canvas, GST_VIDEO_INFO_WIDTH (&c->out_info),
        GST_VIDEO_INFO_HEIGHT (&c->out_info), GST_VIDEO_INFO_PLANE_STRIDE (&c->out_info, 0), c->stream);
  } else {
    r = gstamd_compositor_aggregate_opaque (fmt, c->background, desc, odesc, (int) n_desc, canvas, GST_VIDEO_INFO_WIDTH (&c->out_info),
        GST_VIDEO_INFO_HEIGHT (&c->out_info), GST_VIDEO_INFO_PLANE_STRIDE (&c->out_info, 0), c->stream);
  }
  if (r == GSTAMD_OK) {
    /* one ticket for the whole frame: `written` on the canvas, `read` on every HBM pad frame (staged inputs and per-pad
     * conversions are reused by the next frame on the same stream, which orders them) */
    GstAmdHipTicket *t = gst_amd_hip_ticket_new (c->stream);
    if (c->out_hip)
      gst_amd_hip_memory_set_written (omem, t);
    for (i = 0; i < n; i++)
      if (bufs[i] && mapped_dev[i])
        gst_amd_hip_memory_set_read (gst_buffer_peek_memory (bufs[i], 0), t);
    gst_amd_hip_ticket_unref (t);
  }
  if (c->out_hip) {
    gst_memory_unmap (omem, &omap);
  } else if (r == GSTAMD_OK) {
    GstMapInfo m;
    if (gst_buffer_map (outbuf, &m, GST_MAP_WRITE)) {
      r = gstamd_device_download_async (m.data, c->d_out, m.size, c->stream);
      if (r == GSTAMD_OK)
        r = gstamd_stream_synchronize (c->stream);            /* the CPU is about to look at the frame */
      gst_buffer_unmap (outbuf, &m);
    } else {
      r = GSTAMD_ERR_INVALID;
    }
  }
  if (r != GSTAMD_OK) {
    flow = GST_FLOW_ERROR;
    goto done_inputs;
  }
  c->n_frames++;

done_inputs:
  for (i = 0; i < n; i++) {
    if (bufs[i]) {
      if (mapped_dev[i])
        gst_memory_unmap (gst_buffer_peek_memory (bufs[i], 0), &maps[i]);
      gst_buffer_unref (bufs[i]);
    }
    gst_object_unref (pads[i]);
  }
  if (flow != GST_FLOW_OK) {
    if (outbuf && !outbuf_in)
      gst_buffer_unref (outbuf);
    if (flow == GST_FLOW_ERROR)
      GST_ELEMENT_ERROR (c, LIBRARY, FAILED, ("HIP compositing failed"), ("%s", gstamd_last_error ()));
    return flow;
  }
  *outbuf_out = outbuf;
  return GST_FLOW_OK;
}

#if AMD_COMP_VAGG
/* GstVideoAggregatorClass::create_output_buffer: an HBM frame of this element's pool when downstream negotiated memory:AMDHIPMemory */
static GstFlowReturn
amd_comp_create_output_buffer (GstVideoAggregator * vagg, GstBuffer ** outbuf)
{
  GstAmdCompositor *c = AMD_COMP (vagg);
  if (!c->have_out)
    return GST_FLOW_NOT_NEGOTIATED;
  if (c->out_hip) {
    gst_amd_hip_select_device (c->device_id);
    return c->out_pool ? gst_buffer_pool_acquire_buffer (c->out_pool, outbuf, NULL) : GST_FLOW_ERROR;
  }
  *outbuf = gst_buffer_new_allocate (NULL, GST_VIDEO_INFO_SIZE (&c->out_info), NULL);
  return *outbuf ? GST_FLOW_OK : GST_FLOW_ERROR;
}

/* GstVideoAggregatorClass::aggregate_frames (compositor.c:1739): the base class has chosen every pad's frame for this output frame */
static GstFlowReturn
amd_comp_aggregate_frames (GstVideoAggregator * vagg, GstBuffer * outbuf)
{
  GstAmdCompositor *c = AMD_COMP (vagg);
  GstAmdCompositorPadObj *pads[AMD_COMP_MAX_PADS];
  GstBuffer *bufs[AMD_COMP_MAX_PADS], *out = NULL;
  guint n = 0, i;
  GList *l;

  if (!c->have_out)
    return GST_FLOW_NOT_NEGOTIATED;
  GST_OBJECT_LOCK (vagg);
  for (l = GST_ELEMENT (vagg)->sinkpads; l && n < AMD_COMP_MAX_PADS; l = l->next)          /* kept in z order by the base class */
    pads[n++] = gst_object_ref (l->data);
  GST_OBJECT_UNLOCK (vagg);
  for (i = 0; i < n; i++) {
    GstVideoAggregatorPad *vp = GST_VIDEO_AGGREGATOR_PAD (pads[i]);
    GstBuffer *b = gst_video_aggregator_pad_get_current_buffer (vp);
    pads[i]->info = vp->info;
    pads[i]->have_info = vp->info.finfo != NULL && GST_VIDEO_INFO_FORMAT (&vp->info) != GST_VIDEO_FORMAT_UNKNOWN;
    bufs[i] = b ? gst_buffer_ref (b) : NULL;
  }
  return amd_comp_compose (c, pads, bufs, n, outbuf, &out);
}
#else
static GstFlowReturn
amd_comp_aggregate (GstAggregator * agg, gboolean timeout)
{
  GstAmdCompositor *c = AMD_COMP (agg);
  GstAmdCompositorPadObj *pads[AMD_COMP_MAX_PADS];
  GstBuffer *bufs[AMD_COMP_MAX_PADS], *outbuf = NULL;
  GstFlowReturn flow;
  GList *l;
  guint n = 0, i;
  gboolean all_eos = TRUE;

  if (!c->have_out)
    return GST_FLOW_NOT_NEGOTIATED;
  GST_OBJECT_LOCK (agg);
  for (l = GST_ELEMENT (agg)->sinkpads; l && n < AMD_COMP_MAX_PADS; l = l->next)
    pads[n++] = gst_object_ref (l->data);
  GST_OBJECT_UNLOCK (agg);
  qsort (pads, n, sizeof (pads[0]), pad_zorder_cmp);
  {
    /* the output frame's running-time interval: frame counter at the output framerate, from running time 0 */
    const guint64 fn = GST_VIDEO_INFO_FPS_N (&c->out_info) > 0 ? GST_VIDEO_INFO_FPS_N (&c->out_info) : 25;
    const guint64 fd = GST_VIDEO_INFO_FPS_N (&c->out_info) > 0 ? GST_VIDEO_INFO_FPS_D (&c->out_info) : 1;
    const GstClockTime out_start = gst_util_uint64_scale (c->n_frames, fd * GST_SECOND, fn);
    const GstClockTime out_end = gst_util_uint64_scale (c->n_frames + 1, fd * GST_SECOND, fn);
    gboolean need_data = FALSE;
    for (i = 0; i < n; i++) {
      gboolean eos = FALSE;
      if (!amd_comp_pad_select (pads[i], out_start, out_end, &eos) && !timeout)
        need_data = TRUE;
      if (!eos)
        all_eos = FALSE;
    }
    if (need_data && !all_eos) {
      for (i = 0; i < n; i++)
        gst_object_unref (pads[i]);
      return GST_AGGREGATOR_FLOW_NEED_DATA;
    }
    for (i = 0; i < n; i++)
      bufs[i] = pads[i]->current ? gst_buffer_ref (pads[i]->current) : NULL;
  }
  GST_LOG_OBJECT (c, "aggregate: %u pads, all_eos %d, timeout %d", n, all_eos, timeout);
  if (all_eos) {
    for (i = 0; i < n; i++) {
      gst_buffer_replace (&bufs[i], NULL);
      gst_object_unref (pads[i]);
    }
    return GST_FLOW_EOS;
  }
  for (i = 0; i < n; i++)
    if (pads[i]->current && !GST_CLOCK_TIME_IS_VALID (pads[i]->cur_start))
      gst_buffer_replace (&pads[i]->current, NULL);             /* untimed frames are shown once (bufs[] holds this frame's reference) */
  {
    const guint64 frame_no = c->n_frames;
    flow = amd_comp_compose (c, pads, bufs, n, NULL, &outbuf);
    if (flow != GST_FLOW_OK)
      return flow;
    /* timestamps: the running frame count at the output framerate (gst_video_aggregator_do_aggregate) */
    if (GST_VIDEO_INFO_FPS_N (&c->out_info) > 0) {
      const guint64 fn = GST_VIDEO_INFO_FPS_N (&c->out_info), fd = GST_VIDEO_INFO_FPS_D (&c->out_info);
      GST_BUFFER_PTS (outbuf) = gst_util_uint64_scale (frame_no, fd * GST_SECOND, fn);
      GST_BUFFER_DURATION (outbuf) = gst_util_uint64_scale (frame_no + 1, fd * GST_SECOND, fn) - GST_BUFFER_PTS (outbuf);
    }
  }
  return gst_aggregator_finish_buffer (agg, outbuf);
}
#endif

static gboolean
amd_comp_stop (GstAggregator * agg)
{
  GstAmdCompositor *c = AMD_COMP (agg);
  if (c->out_pool) {
    gst_buffer_pool_set_active (c->out_pool, FALSE);
    gst_object_unref (c->out_pool);
    c->out_pool = NULL;
  }
  gst_amd_hip_select_device (c->device_id);
  if (c->stream) {
    gstamd_stream_synchronize (c->stream);
    gstamd_stream_free (c->stream);
    c->stream = NULL;
  }
  gstamd_device_free (c->d_out);
  c->d_out = NULL;
  c->d_out_size = 0;
  c->have_out = FALSE;
  if (g_getenv ("GSTAMD_ELEMENT_STATS"))
    g_printerr ("amdcompositor %s: frames %" G_GUINT64_FORMAT " culled-frames %" G_GUINT64_FORMAT " inline-scaled %" G_GUINT64_FORMAT "\n", GST_OBJECT_NAME (c),
        c->n_frames, c->n_culled, c->n_inline_scaled);
  c->n_frames = 0;
  {
    GList *l;
    GST_OBJECT_LOCK (agg);
    for (l = GST_ELEMENT (agg)->sinkpads; l; l = l->next)
      gst_buffer_replace (&AMD_COMP_PAD (l->data)->current, NULL);
    GST_OBJECT_UNLOCK (agg);
  }
#if AMD_COMP_VAGG
  return GST_AGGREGATOR_CLASS (gst_amd_compositor_parent_class)->stop (agg);
#else
  return TRUE;
#endif
}

static void
gst_amd_compositor_class_init (GstAmdCompositorClass * klass)
{
  GObjectClass *oc = (GObjectClass *) klass;
  GstElementClass *ec = (GstElementClass *) klass;
  GstAggregatorClass *ac = (GstAggregatorClass *) klass;

  GST_DEBUG_CATEGORY_INIT (amd_comp_debug, "amdcompositor", 0, "MI355X compositor");
  gst_amd_converter_config_register_types ();
  oc->set_property = amd_comp_set_property;
  oc->get_property = amd_comp_get_property;
  g_object_class_install_property (oc, PROP_BACKGROUND, g_param_spec_enum ("background", "Background", "Background type",
          amd_comp_background_get_type (), 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "HIP device this instance runs on (-1 = the process's current device)", -1, G_MAXINT, -1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  /* names and defaults of compositor.c:2103-2162 */
  g_object_class_install_property (oc, PROP_ZERO_SIZE_IS_UNSCALED, g_param_spec_boolean ("zero-size-is-unscaled", "Zero size is unscaled",
          "If TRUE, then input video is unscaled in that dimension if width or height is 0 (for backwards compatibility)", TRUE,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_MAX_THREADS, g_param_spec_uint ("max-threads", "Max Threads",
          "Accepted for compatibility (the GPU grid replaces the blend threads)", 0, G_MAXINT, 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_IGNORE_INACTIVE_PADS, g_param_spec_boolean ("ignore-inactive-pads", "Ignore inactive pads",
          "Avoid timing out waiting for inactive pads", FALSE, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_CULLED_FRAMES, g_param_spec_uint64 ("culled-frames", "Culled frames",
          "Pad frames left out so far because nothing of them was visible (alpha 0, off canvas, or under an opaque pad)", 0, G_MAXUINT64, 0,
          G_PARAM_READABLE | G_PARAM_STATIC_STRINGS));
  gst_element_class_add_static_pad_template_with_gtype (ec, &comp_sink_tmpl, gst_amd_compositor_pad_get_type ());
  /* the base class keeps its output segment in the src pad: it has to be a GstAggregatorPad */
  gst_element_class_add_static_pad_template_with_gtype (ec, &comp_src_tmpl, GST_TYPE_AGGREGATOR_PAD);
  gst_element_class_set_static_metadata (ec, "Compositor (MI355X/HIP)", "Filter/Editor/Video/Compositor",
      "Composite multiple video streams in one fused GPU pass", "gstreamer_amd");
#if AMD_COMP_VAGG
  /* GstVideoAggregator: pads, events, queries, frame selection, timestamps, QoS are the base class's; the canvas size / format rules and the
   * output memory are this element's, and so is the frame (compositor.c:2098 aggregate_frames, :2096 fixate_src_caps) */
  ((GstVideoAggregatorClass *) klass)->aggregate_frames = amd_comp_aggregate_frames;
  ((GstVideoAggregatorClass *) klass)->create_output_buffer = amd_comp_create_output_buffer;
#else
  ac->create_new_pad = amd_comp_create_new_pad;
  ac->sink_event = amd_comp_sink_event;
  ac->aggregate = amd_comp_aggregate;
#endif
  ac->sink_query = amd_comp_sink_query;
  ec->request_new_pad = amd_comp_request_new_pad;           /* child-added / child-removed around the base class's (compositor.c:1887-1930) */
  ec->release_pad = amd_comp_release_pad;
  ac->update_src_caps = amd_comp_update_src_caps;
  ac->fixate_src_caps = amd_comp_fixate_src_caps;
  ac->negotiated_src_caps = amd_comp_negotiated_src_caps;
  ac->stop = amd_comp_stop;
}

static void
gst_amd_compositor_init (GstAmdCompositor * c)
{
  c->background = 0;
  c->have_out = FALSE;
  c->n_frames = 0;
  c->next_pad = 0;
  c->device_id = -1;
  c->stream = NULL;
  c->zero_size_is_unscaled = TRUE;
  c->ignore_inactive_pads = FALSE;
  c->max_threads = 0;
}
