// tuning.cpp - see tuning.h
#include "tuning.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace {

struct Knob {
  const char *name;
  bool text;
  int value;            /* -1: unset */
  std::string str;
  bool has_str;
};

Knob g_knobs[] = {
  {"GSTAMD_NO_FAST_PRE", false, -1, "", false}, {"GSTAMD_STORE_POLICY", false, -1, "", false}, {"GSTAMD_STORE_WT_BELOW", false, -1, "", false},
  {"GSTAMD_NO_FUSED420", false, -1, "", false}, {"GSTAMD_NO_COL", false, -1, "", false}, {"GSTAMD_COL_OPL", false, -1, "", false}, {"GSTAMD_COL_SHARE", false, -1, "", false},
  {"GSTAMD_COL_WAVES", false, -1, "", false}, {"GSTAMD_COL_CHUNKS", false, -1, "", false}, {"GSTAMD_COL_SOLO", false, -1, "", false}, {"GSTAMD_COL_NO_REGWIN", false, -1, "", false}, {"GSTAMD_COL_DEBUG", false, -1, "", false}, {"GSTAMD_LIST_DEBUG", false, -1, "", false}, {"GSTAMD_DEEP_DEBUG", false, -1, "", false}, {"GSTAMD_NO_DEEP_PLANES16", false, -1, "", false}, {"GSTAMD_NO_ENCODE16", false, -1, "", false}, {"GSTAMD_ENCODE16_NARROW", false, -1, "", false}, {"GSTAMD_ENCODE16_WIDE", false, -1, "", false}, {"GSTAMD_NO_PLANE_QUAD", false, -1, "", false}, {"GSTAMD_PLANE_QUAD_ROWS", false, -1, "", false}, {"GSTAMD_PLANE_QUAD_MODE", false, -1, "", false}, {"GSTAMD_PLANE_QUAD_NT", false, -1, "", false}, {"GSTAMD_PLANE_QUAD_ONLY", false, -1, "", false}, {"GSTAMD_PLANE_QUAD_NO_DSTEP", false, -1, "", false}, {"GSTAMD_NO_H420_REG", false, -1, "", false}, {"GSTAMD_H420_ROWS", false, -1, "", false},
  {"GSTAMD_VSCALE_ROWS", false, -1, "", false}, {"GSTAMD_NO_FAST420P", false, -1, "", false}, {"GSTAMD_NO_FAST422", false, -1, "", false}, {"GSTAMD_NO_CONVERT_PACK", false, -1, "", false}, {"GSTAMD_NO_CONVERT_PACK_WIDE", false, -1, "", false}, {"GSTAMD_NO_CONVERT_PACK_422UP", false, -1, "", false}, {"GSTAMD_NO_PLANE_FRAME", false, -1, "", false}, {"GSTAMD_NO_RELAYOUT", false, -1, "", false}, {"GSTAMD_NO_SWIZZLE34", false, -1, "", false}, {"GSTAMD_NO_GAMMA_COMP", false, -1, "", false}, {"GSTAMD_NO_BILINEAR4", false, -1, "", false}, {"GSTAMD_NO_BILINEAR4_UP", false, -1, "", false}, {"GSTAMD_BIL4_UP_ROWS", false, -1, "", false}, {"GSTAMD_NO_CONVERT16_FAST", false, -1, "", false}, {"GSTAMD_NO_DEEP_SCALE_PACK", false, -1, "", false}, {"GSTAMD_NO_BILINEAR_AYUV", false, -1, "", false}, {"GSTAMD_DEEP_PACK_NARROW", false, -1, "", false},
  {"GSTAMD_NO_BILINEAR420", false, -1, "", false}, {"GSTAMD_NO_BILINEAR_ROWS", false, -1, "", false}, {"GSTAMD_NO_BILINEAR_HALF", false, -1, "", false}, {"GSTAMD_BIL_HALF_ROWS", false, -1, "", false}, {"GSTAMD_BIL_HALF_STORE", false, -1, "", false}, {"GSTAMD_BIL_HALF_SMALL", false, -1, "", false}, {"GSTAMD_BIL_TILE", false, -1, "", false},
  {"GSTAMD_BIL_TABLE", false, -1, "", false}, {"GSTAMD_BIL_ROWS", false, -1, "", false}, {"GSTAMD_BIL_ROWS_TILE", false, -1, "", false},
  {"GSTAMD_BIL_SLOTS", false, -1, "", false}, {"GSTAMD_BIL_WG", false, -1, "", false}, {"GSTAMD_BIL_VERBOSE", false, -1, "", false},
  {"GSTAMD_FUSED_WAVES", false, -1, "", false}, {"GSTAMD_FUSED_ROWS", false, -1, "", false}, {"GSTAMD_FUSED_SCHED", false, -1, "", false},
  {"GSTAMD_FUSED_FIRST", false, -1, "", false}, {"GSTAMD_FUSED_DEBUG", false, -1, "", false},
  {"GSTAMD_NO_FIR_LDS", false, -1, "", false}, {"GSTAMD_NO_FIR_MANY", false, -1, "", false},
  {"GSTAMD_SCALED_TILE_ROWS", false, -1, "", false}, {"GSTAMD_AGG_BX", false, -1, "", false},
  {"GSTAMD_NO_AGG_WALK", false, -1, "", false}, {"GSTAMD_WALK_ROWS", false, -1, "", false}, {"GSTAMD_WALK_XCD", false, -1, "", false},
  /* tuning builds */
  {"GSTAMD_ABLATE", false, -1, "", false}, {"GSTAMD_AGG_ABLATE", false, -1, "", false}, {"GSTAMD_AGG_ROWS", false, -1, "", false}, {"GSTAMD_AGG_CULL_ROWS", false, -1, "", false},
  {"GSTAMD_AGG_DEPTH", false, -1, "", false}, {"GSTAMD_AGG_STRIP_ROWS", false, -1, "", false}, {"GSTAMD_AGG_STRIP_PX", false, -1, "", false},
  {"GSTAMD_AGG_NT", false, -1, "", false}, {"GSTAMD_FAST_VARIANT", true, -1, "", false}, {"GSTAMD_FUSED_TRACE", true, -1, "", false},
};

std::once_flag g_once;
std::mutex g_lock;

void read_environment ()
{
  for (Knob &k : g_knobs) {
    const char *e = getenv (k.name);
    if (!e)
      continue;
    if (k.text) {
      k.str = e;
      k.has_str = true;
    } else {
      const int v = atoi (e);
      k.value = v < 0 ? 0 : v;          /* "set" with a value that is no number counts as 0 */
    }
  }
}

Knob *find (const char *name)
{
  std::call_once (g_once, read_environment);
  for (Knob &k : g_knobs)
    if (strcmp (k.name, name) == 0)
      return &k;
  return nullptr;
}

}  // namespace

// Kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1): the command processor fetches a launch's arguments from HBM instead of over PCIe
// from host memory.  On MI355X that takes ~3.7 us off every launch (4K NV12 -> BGRA, one frame per launch: 15.2 -> 11.5 us; lists of 32:
// 245 -> 232 us; profiles/r04/launch_kernarg.log) - a frame a launch is what a live pipeline runs.  It is a switch of the HIP runtime, read once
// when the runtime initialises, and it belongs to the PROCESS: this library does not touch the environment (a setenv from a dlopen'ed plugin
// races with getenv in the host's other threads and changes the runtime for every other HIP user).  The launcher sets it - bench.py,
// plugins/tests/bench_element.c, plugins/tests/launch129.c and tests/conftest.py do; INTEGRATION.md says so for applications.

namespace gstamd {

int tuning_int (const char *name, int unset)
{
  Knob *k = find (name);
  if (!k)
    return unset;
  std::lock_guard<std::mutex> g (g_lock);
  return k->value < 0 ? unset : k->value;
}

const char *tuning_text (const char *name)
{
  Knob *k = find (name);
  return k && k->has_str ? k->str.c_str () : nullptr;
}

}  // namespace gstamd

extern "C" {

int gstamd_tuning_set (const char *name, int value)
{
  Knob *k = name ? find (name) : nullptr;
  if (!k || k->text)
    return -1;
  std::lock_guard<std::mutex> g (g_lock);
  k->value = value < 0 ? -1 : value;
  return 0;
}

int gstamd_tuning_get (const char *name)
{
  return name ? gstamd::tuning_int (name, -1) : -1;
}

}
