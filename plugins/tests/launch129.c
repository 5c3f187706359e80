/* launch129.c - a small gst-launch for the hand-built 1.29 runtime (oracle/rt129_build.py), which has no gst_parse (no bison / flex in this
 * image) and no registry: test / measurement tool, not part of the plugin.
 *
 *   GSTAMD_LAUNCH_PLUGINS=/path/libgstcoreelements.so:/path/libgstamdhipdsp.so  launch129 [-q] <pipeline description, one token per argument>
 *
 * The description is the subset of the gst-launch syntax tests/test_plugin_gpu.py uses (so that the same pipelines run on the conda 1.14
 * runtime through gst-launch-1.0 and on the reference's own version through this):
 *   factory [prop=value ...]            an element; values go through gst_util_set_object_arg (enums by nick, structures, caps ...)
 *   pad::prop=value                     a property of a (request) pad - GstChildProxy, applied once every link has been made
 *   video/x-raw,... | audio/x-raw,...   a caps filter
 *   !                                   link
 *   name. | name.pad                    the element called `name` (name=...), optionally one of its pads
 * Runs the pipeline to EOS; an error message on the bus is printed and the exit code is 1. */
#include <gst/gst.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
  GstElement *element;
  gchar *pad;                   /* NULL: any */
} End;

typedef struct
{
  GstElement *element;
  gchar *name, *value;
} Deferred;

/* -v: the caps a pad settles on, in gst-launch's own words (what tests grep for) */
static void
on_deep_notify (GstObject * obj, GstObject * orig, GParamSpec * pspec, gpointer user_data)
{
  if (GST_IS_PAD (orig) && !strcmp (pspec->name, "caps")) {
    GstCaps *caps = gst_pad_get_current_caps (GST_PAD (orig));
    if (caps) {
      gchar *path = gst_object_get_path_string (orig), *str = gst_caps_to_string (caps);
      /* gst-launch prints /GstPipeline:pipeline0/GstFakeSink:fakesink0.GstPad:sink: caps = ... */
      gchar *dot = strrchr (path, '/');
      if (dot)
        *dot = '.';
      printf ("%s: caps = %s\n", path, str);
      g_free (path);
      g_free (str);
      gst_caps_unref (caps);
    }
  }
}

static gboolean
is_caps (const gchar * t)
{
  return g_str_has_prefix (t, "video/") || g_str_has_prefix (t, "audio/");
}

int
main (int argc, char **argv)
{
  GstElement *pipeline, *cur = NULL;
  End prev = { NULL, NULL };
  gboolean link_pending = FALSE;
  GArray *deferred = g_array_new (FALSE, TRUE, sizeof (Deferred));
  g_setenv ("HIP_FORCE_DEV_KERNARG", "1", FALSE);     /* kernel arguments in HBM; the launcher's decision, single-threaded here (tuning.cpp) */
  const gchar *plugins = g_getenv ("GSTAMD_LAUNCH_PLUGINS");
  int i, rc = 0;

  gst_init (NULL, NULL);
  if (plugins) {
    gchar **paths = g_strsplit (plugins, ":", -1);
    for (i = 0; paths[i]; i++) {
      GError *err = NULL;
      GstPlugin *p;
      if (!*paths[i])
        continue;
      p = gst_plugin_load_file (paths[i], &err);
      if (!p) {
        fprintf (stderr, "launch129: cannot load %s: %s\n", paths[i], err ? err->message : "?");
        return 2;
      }
      gst_object_unref (p);
    }
    g_strfreev (paths);
  }
  if (argc == 3 && !strcmp (argv[1], "--types")) {
    /* the type chain of an element and of its request sink pads: what `gst-inspect-1.0` shows under "GObject" */
    GstElement *e = gst_element_factory_make (argv[2], NULL);
    GType t;
    GstPad *pad;
    if (!e) {
      fprintf (stderr, "launch129: no element '%s'\n", argv[2]);
      return 2;
    }
    for (t = G_OBJECT_TYPE (e); t; t = g_type_parent (t))
      printf ("%s%s", g_type_name (t), g_type_parent (t) ? " < " : "\n");
    pad = gst_element_request_pad_simple (e, "sink_%u");
    if (pad) {
      printf ("pad: ");
      for (t = G_OBJECT_TYPE (pad); t; t = g_type_parent (t))
        printf ("%s%s", g_type_name (t), g_type_parent (t) ? " < " : "\n");
      gst_element_release_request_pad (e, pad);
      gst_object_unref (pad);
    }
    gst_object_unref (e);
    return 0;
  }
  pipeline = gst_pipeline_new ("pipeline");
  for (i = 1; i < argc; i++) {
    const gchar *t = argv[i];
    End here = { NULL, NULL };

    if (!strcmp (t, "-q") || !*t)
      continue;
    if (!strcmp (t, "-v")) {
      g_signal_connect (pipeline, "deep-notify", G_CALLBACK (on_deep_notify), NULL);
      continue;
    }
    if (!strcmp (t, "!")) {
      if (!prev.element) {
        fprintf (stderr, "launch129: '!' with nothing before it\n");
        return 2;
      }
      link_pending = TRUE;
      cur = NULL;
      continue;
    }
    if (is_caps (t)) {
      GstCaps *caps = gst_caps_from_string (t);
      if (!caps) {
        fprintf (stderr, "launch129: bad caps '%s'\n", t);
        return 2;
      }
      here.element = gst_element_factory_make ("capsfilter", NULL);
      if (!here.element) {
        fprintf (stderr, "launch129: no capsfilter element\n");
        return 2;
      }
      g_object_set (here.element, "caps", caps, NULL);
      gst_caps_unref (caps);
      gst_bin_add (GST_BIN (pipeline), here.element);
      cur = NULL;
    } else if (strchr (t, '=') && cur) {
      /* a property of the current element (or of one of its pads) */
      gchar *name = g_strndup (t, strchr (t, '=') - t);
      const gchar *value = strchr (t, '=') + 1;
      if (strstr (name, "::")) {
        Deferred d = { cur, name, g_strdup (value) };
        g_array_append_val (deferred, d);
      } else if (!strcmp (name, "name")) {
        g_free (name);          /* given to gst_element_factory_make below: an object takes a new name only while it has no parent */
      } else {
        if (!g_object_class_find_property (G_OBJECT_GET_CLASS (cur), name)) {
          fprintf (stderr, "launch129: no property '%s' in element '%s'\n", name, GST_OBJECT_NAME (cur));
          return 2;
        }
        gst_util_set_object_arg (G_OBJECT (cur), name, value);
        g_free (name);
      }
      continue;
    } else if (strchr (t, '.') && !strchr (t, '=')) {
      /* a reference: name. or name.pad */
      gchar *name = g_strndup (t, strchr (t, '.') - t);
      const gchar *pad = strchr (t, '.') + 1;
      here.element = gst_bin_get_by_name (GST_BIN (pipeline), name);
      if (!here.element) {
        fprintf (stderr, "launch129: no element named '%s'\n", name);
        return 2;
      }
      gst_object_unref (here.element);          /* the bin keeps it */
      here.pad = *pad ? g_strdup (pad) : NULL;
      g_free (name);
      cur = NULL;
    } else {
      const gchar *ename = NULL;
      int k;
      for (k = i + 1; k < argc && strchr (argv[k], '=') && strcmp (argv[k], "!"); k++)
        if (g_str_has_prefix (argv[k], "name="))
          ename = argv[k] + 5;
      here.element = gst_element_factory_make (t, ename);
      if (!here.element) {
        fprintf (stderr, "launch129: no element '%s'\n", t);
        return 2;
      }
      gst_bin_add (GST_BIN (pipeline), here.element);
      cur = here.element;
    }
    if (link_pending) {
      if (!gst_element_link_pads (prev.element, prev.pad, here.element, here.pad)) {
        fprintf (stderr, "launch129: could not link %s.%s to %s.%s\n", GST_OBJECT_NAME (prev.element), prev.pad ? prev.pad : "", GST_OBJECT_NAME (here.element),
            here.pad ? here.pad : "");
        return 2;
      }
      link_pending = FALSE;
      /* a chain that ended in a reference: the next element starts a new chain */
    }
    g_free (prev.pad);
    prev = here;
    if (here.pad && !cur) {
      /* `name.pad` as the END of a chain names a sink pad: what follows is a new chain; as the START of one it names a source pad - both are
       * covered by keeping it as prev until the next '!' or element */
    }
  }
  for (i = 0; i < (int) deferred->len; i++) {
    Deferred *d = &g_array_index (deferred, Deferred, i);
    GObject *target = NULL;
    GParamSpec *pspec = NULL;
    if (!GST_IS_CHILD_PROXY (d->element) || !gst_child_proxy_lookup (GST_CHILD_PROXY (d->element), d->name, &target, &pspec)) {
      fprintf (stderr, "launch129: no child property '%s' in element '%s'\n", d->name, GST_OBJECT_NAME (d->element));
      return 2;
    }
    gst_util_set_object_arg (target, pspec->name, d->value);
    g_object_unref (target);
  }
  {
    GstBus *bus = gst_element_get_bus (pipeline);
    GstMessage *msg;
    if (gst_element_set_state (pipeline, GST_STATE_PLAYING) == GST_STATE_CHANGE_FAILURE) {
      fprintf (stderr, "launch129: the pipeline does not go to PLAYING\n");
      rc = 1;
    }
    msg = gst_bus_timed_pop_filtered (bus, rc ? 0 : 120 * GST_SECOND, GST_MESSAGE_EOS | GST_MESSAGE_ERROR);
    if (!msg && !rc) {
      fprintf (stderr, "launch129: no EOS within 120 s\n");
      rc = 1;
    } else if (msg && GST_MESSAGE_TYPE (msg) == GST_MESSAGE_ERROR) {
      GError *err = NULL;
      gchar *dbg = NULL;
      gst_message_parse_error (msg, &err, &dbg);
      fprintf (stderr, "ERROR: from element %s: %s\n%s\n", GST_OBJECT_NAME (GST_MESSAGE_SRC (msg)), err->message, dbg ? dbg : "");
      g_clear_error (&err);
      g_free (dbg);
      rc = 1;
    }
    if (msg)
      gst_message_unref (msg);
    gst_element_set_state (pipeline, GST_STATE_NULL);
    gst_object_unref (bus);
  }
  gst_object_unref (pipeline);
  return rc;
}
