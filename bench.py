#!/usr/bin/env python3
"""bench.py - headline benchmark: 4K frames/s of videoconvertscale's NV12->BGRA path per GPU (BASELINE C2),
plus the other BASELINE configs behind --config.

  python bench.py [--config c2|c3|c4|c4audio|c5] [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

`--gpus N` without a torchrun environment re-executes itself under torch.distributed.run with N ranks
(127.0.0.1 rendezvous); a rank count that differs from --gpus is an error, never a silent 1-GPU run.
Independent streams shard one per GPU (no collective on the data path): the only torch.distributed use
is the timing barrier and the MAX / SUM of the per-rank numbers.  Rank 0 prints ONE JSON line.

Every config: frames resident in HBM, pools of distinct frames larger than the 256 MiB Infinity Cache on
both sides, output pool >= frames per step (no two frames of one step share a buffer), 60 ms untimed
pre-heat, W warm-up steps, K timed steps between barrier + synchronize, HIP events on the launch stream.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# kernel arguments in device memory (a HIP runtime switch read when the runtime initialises - PyTorch does that long before the library is
# loaded, so the library's own load-time default comes too late here; gstreamer_amd/csrc/tuning.cpp says what it buys)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HEADLINE = "4K frames/s (videoconvertscale NV12->BGRA) per GPU; % HBM roofline"

# name -> (in format, w, h, out format, ow, oh, converter options, frames per step, frames per launch, metric, dominant kernel)
VIDEO_CONFIGS = {
    "c2": ("NV12", 3840, 2160, "BGRA", 3840, 2160, {}, 32, 32, HEADLINE, "k_convert_strip<CHROMA_H_H2_CS, layout BGRA>"),
    "c1": ("NV12", 1920, 1080, "BGRA", 1920, 1080, {}, 32, 32,
           "1080p frames/s (videoconvertscale NV12->BGRA) per GPU; % HBM roofline", "k_convert_strip<CHROMA_H_H2_CS, layout BGRA>"),
    "c3": ("I420", 7680, 4320, "RGBA", 1920, 1080, {"resampler_method": "lanczos"}, 16, 16,
           "8K->1080p frames/s (videoconvertscale I420->RGBA, Lanczos) per GPU; % HBM roofline", "k_scale_col (--batch 1: one frame per launch)"),
    "c5": ("NV12", 7680, 4320, "BGRA", 3840, 2160, {"resampler_method": "linear", "max_taps": 2}, 16, 16,
           "8K->4K frames/s (videoconvertscale NV12->BGRA, bilinear) per GPU; % HBM roofline",
           "k_bilinear420_half (an exact halving; other ratios: k_bilinear420_rows)"),
    # SURVEY 8(f) rows, measured the same way (composite plans: several launches per frame, the whole frame is what is timed)
    "f2gamma": ("NV12", 3840, 2160, "BGRA", 3840, 2160, {"gamma_mode": "remap"}, 8, 8,
                "4K frames/s (videoconvertscale NV12->BGRA, gamma-mode=remap) per GPU; % HBM roofline", "k_convert_strip<.., GSTAMD_FAST_LUT> (the direct conversion made with the gamma chain's to_RGB matrix, the composed decode/encode table ahead of the store)"),
    "f2p010out": ("NV12", 3840, 2160, "P010_10LE", 3840, 2160, {}, 8, 8,
                  "4K frames/s (videoconvertscale NV12->P010_10LE) per GPU; % HBM roofline", "sub-conversion + k_gamma_stage + k_pack16"),
    "f2p010in": ("P010_10LE", 3840, 2160, "NV12", 3840, 2160, {}, 8, 8,
                 "4K frames/s (videoconvertscale P010_10LE->NV12) per GPU; % HBM roofline", "k_front16 + k_gamma_stage + sub-conversion"),
    # SURVEY 8(f) generic paths: the plane scaler behind `videoscale` on NV12, a planar 4:2:0 pack, a byte swizzle
    "f8scale": ("NV12", 3840, 2160, "NV12", 1920, 1080, {"resampler_method": "linear", "max_taps": 2}, 8, 8,
                "4K->1080p frames/s (videoscale NV12->NV12, bilinear) per GPU; % HBM roofline", "k_plane_quad (every plane of the list in one grid: eight output bytes per lane from one 16-byte window per source row, passes as v_dot2; k_plane_tiles serves longer filters)"),
    "f8pack": ("YUY2", 3840, 2160, "I420", 3840, 2160, {}, 8, 8,
               "4K frames/s (videoconvert YUY2->I420) per GPU; % HBM roofline", "k_convert_pack_422 (the planar packer fed by the unscaled chain, no AYUV image)"),
    "f5encode16": ("BGRA", 3840, 2160, "P010_10LE", 3840, 2160, {}, 8, 8,
                   "4K frames/s (videoconvertscale BGRA->P010_10LE) per GPU; % HBM roofline", "k_encode16 (widen, matrix16, chroma down, ordered dither, pack in one kernel)"),
    "f8swizzle": ("BGRA", 3840, 2160, "RGBA", 3840, 2160, {}, 8, 8,
                  "4K frames/s (videoconvert BGRA->RGBA) per GPU; % HBM roofline", "k_swizzle4"),
    # the scaled pairs of real pipelines - a 10-bit decoder's frames at half the size into an encoder's / a display's format
    "f6p010nv12": ("P010_10LE", 3840, 2160, "NV12", 1920, 1080, {"resampler_method": "linear", "max_taps": 2}, 8, 8,
                   "4K->1080p frames/s (videoconvertscale P010_10LE->NV12, bilinear) per GPU; % HBM roofline",
                   "k_deep_scale_pack<semi-planar, CHROMA_H_H2_CS> (16-bit front, both u16 passes, narrowing, chroma downsampler and pack in one kernel)"),
    "f6p010p010": ("P010_10LE", 3840, 2160, "P010_10LE", 1920, 1080, {"resampler_method": "linear", "max_taps": 2}, 8, 8,
                   "4K->1080p frames/s (videoconvertscale P010_10LE->P010_10LE, bilinear) per GPU; % HBM roofline",
                   "k_deep_scale_pack16<semi-planar, CHROMA_H_H2_CS> (16-bit front, both u16 passes, u16 chroma downsamplers, ordered dither and pack in one kernel)"),
    "f6nv12i420": ("NV12", 3840, 2160, "I420", 1920, 1080, {"resampler_method": "linear", "max_taps": 2}, 8, 8,
                   "4K->1080p frames/s (videoconvertscale NV12->I420, bilinear) per GPU; % HBM roofline",
                   "k_bilinear420_half<.., GSTAMD_LAYOUT_AYUV> into the pack image + k_pack_planar (two launches per list)"),
    "f6p010bgra": ("P010_10LE", 3840, 2160, "BGRA", 1920, 1080, {"resampler_method": "linear", "max_taps": 2}, 8, 8,
                   "4K->1080p frames/s (videoconvertscale P010_10LE->BGRA, bilinear) per GPU; % HBM roofline",
                   "k_deep_scale4<semi-planar, CHROMA_H_H2_CS> (16-bit front, both u16 passes, matrix16, narrowing and pack in one kernel)"),
}
CONFIG_TEXT = {
    "f8scale": "SURVEY 8(f): 3840x2160 NV12 -> 1920x1080 NV12, bilinear (the elements' default method: linear, max-taps 2), plane by plane (convert_scale_planes)",
    "f8pack": "SURVEY 8(f): 3840x2160 YUY2 -> I420 (unpack, chroma downsample, planar pack)",
    "f8swizzle": "SURVEY 8(f): 3840x2160 BGRA -> RGBA (a byte permutation)",
    "f5encode16": "SURVEY 8(f)2 / VERDICT r03 item 5: 3840x2160 BGRA -> P010_10LE (widen, matrix16, cosited chroma down, ordered dither, pack)",
    "f6p010nv12": "VERDICT r05 item 4: 3840x2160 P010_10LE -> 1920x1080 NV12, bilinear (the elements' default method) - an HDR decoder's frames into an encoder's format",
    "f6p010p010": "VERDICT r05 item 4: 3840x2160 P010_10LE -> 1920x1080 P010_10LE, bilinear - an HDR transcode that keeps ten bits",
    "f6nv12i420": "VERDICT r05 item 4: 3840x2160 NV12 -> 1920x1080 I420, bilinear - a decoder's frames into a software encoder's format",
    "f6p010bgra": "VERDICT r05 item 4: 3840x2160 P010_10LE -> 1920x1080 BGRA, bilinear (the elements' default method) - an HDR decoder's frames for display",
    "c2": "C2: 3840x2160 NV12 (bt709 limited, chroma-site mpeg2) -> BGRA, fused unpack + chroma upsample + matrix + pack",
    "c1": "C1 size on the GPU: 1920x1080 NV12 -> BGRA (the reference's CPU-runnable case)",
    "c3": "C3: 7680x4320 I420 -> 1920x1080 RGBA, Lanczos (16 x 16 taps), horizontal then vertical like chain_scale",
    "c5": "C5 (per GPU): 7680x4320 NV12 -> 3840x2160 BGRA, bilinear (element default method); frame lists "
          "(gstamd_video_converter_frames, what the element's chain_list calls) - `--batch 1` for one launch per frame",
    "f2gamma": "SURVEY 8(f)2: 3840x2160 NV12 bt709 -> BGRA sRGB with gamma-mode=remap (decode table, linear ARGB64, encode table)",
    "f2p010out": "SURVEY 8(f)2: 3840x2160 NV12 -> P010_10LE (widen, ordered dither, pack)",
    "f2p010in": "SURVEY 8(f)2: 3840x2160 P010_10LE -> NV12 (16-bit front, narrowing, pack)",
    "c4": "C4: compositor, 16 x 1920x1080 BGRA pads (xpos 640*(i%4), ypos 360*(i/4), alpha 0.25+0.05i, random pixel alpha, "
          "operator over) on a checker background -> 3840x2160 BGRA, one fused launch per output frame",
    "c4audio": "C4 audio: audioresample 48000 -> 44100 Hz, F32 stereo interleaved, Kaiser quality 4 (72 taps x 147 phases)",
}


# ------------------------------------------------------------------------------------------------
# workloads: setup() allocates the HBM pools, step(s) enqueues one step on `stream`
# ------------------------------------------------------------------------------------------------
class VideoWorkload:
    def __init__(self, name, batch=None, size=None):
        self.name = name
        (self.ifmt, self.w, self.h, self.ofmt, self.ow, self.oh, self.cfg, self.frames_per_step, self.batch, self.metric,
         self.kernel) = VIDEO_CONFIGS[name]
        if size:
            self.w, self.h = size
            if name in ("c2", "c1"):
                self.ow, self.oh = size
        if batch:
            self.batch = max(1, min(batch, self.frames_per_step))
        assert self.frames_per_step % self.batch == 0
        self.unit = "frames/s"
        self.dtype = "u8"

    def setup(self, dev, rank):
        import torch

        import cases
        from gstreamer_amd import video as V
        self.V = V
        ii, oi = V.video_info(self.ifmt, self.w, self.h), V.video_info(self.ofmt, self.ow, self.oh)
        self.conv = V.VideoConverter(ii, oi, V.converter_config(**self.cfg))
        self.alg_bytes = self.conv.algorithmic_bytes()
        self.pool_in = max(self.frames_per_step, int(400e6 // int(ii.size)) + 1)
        self.pool_out = max(self.frames_per_step, int(530e6 // int(oi.size)) + 1)       # >= the frames of one step
        pin = torch.empty((self.pool_in, int(ii.size)), dtype=torch.uint8, device=dev)
        base = torch.from_numpy(cases.frame_bytes(int(ii.size), "random", 2000 + rank)).to(dev)
        for i in range(self.pool_in):
            pin[i] = torch.roll(base, shifts=i * 4099)
        pout = torch.zeros((self.pool_out, int(oi.size)), dtype=torch.uint8, device=dev)
        self.keep = (pin, pout)
        self.in_ptrs = [pin[i].data_ptr() for i in range(self.pool_in)]
        self.out_ptrs = [pout[i].data_ptr() for i in range(self.pool_out)]
        self.stream = torch.cuda.current_stream().cuda_stream
        self.launches_per_step = self.frames_per_step // self.batch
        self.units_per_step = self.frames_per_step
        self.alg_bytes_per_launch = self.alg_bytes * self.batch

    def step(self, s):
        B = self.batch
        for f in range(0, self.frames_per_step, B):
            n = s * self.frames_per_step + f
            if B == 1:
                self.conv.frame(self.in_ptrs[n % self.pool_in], self.out_ptrs[n % self.pool_out], self.stream)
            else:
                self.conv.frames([self.in_ptrs[(n + i) % self.pool_in] for i in range(B)],
                                 [self.out_ptrs[(n + i) % self.pool_out] for i in range(B)], self.stream)

    def config(self, world):
        return {"workload": "%s, %d frames/step, pools %d in / %d out resident in HBM, 1 stream per GPU, %d frame(s) per kernel launch"
                            % (CONFIG_TEXT[self.name], self.frames_per_step, self.pool_in, self.pool_out, self.batch),
                "plan": self.conv.describe(), "frames_per_step": self.frames_per_step, "frames_per_launch": self.batch,
                "parallelism": "stream-per-gpu x%d" % world}

    def cpu_baseline(self):
        """The reference's own gst_video_converter_frame (oracle/_ref, C-backup ORC, -O2) on this host, element-default
        n-threads=1, bounded sample of the same workload."""
        import cases
        from oracle import ref
        if not ref.available():
            return None
        # at least 100 frames or 5 s of the reference's work on one core (C3 / C5 run 4-8 frames/s: the 5 s rule)
        n = {"c2": 100, "c1": 400, "c3": 24, "c5": 40}.get(self.name, 10)
        src = cases.frame_bytes(ref.video_info(self.ifmt, self.w, self.h)["size"], "random", 1)

        def conv(threads):
            cfg = dict(self.cfg, threads=threads)
            return ref.VideoConverter(self.ifmt, self.w, self.h, self.ofmt, self.ow, self.oh, config=cases.ref_config_string(ref, cfg))
        rc = conv(1)
        rc.bench(src, 2)
        secs = rc.bench(src, n)
        out = {"value": round(n / secs, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": "%d frames of %dx%d %s -> %dx%d %s, gst_video_converter_frame of the reference built from /root/reference "
                         "with -O2 -DDISABLE_ORC (ORC C backups, no JIT SIMD), n-threads=1 (element default)"
                         % (n, self.w, self.h, self.ifmt, self.ow, self.oh, self.ofmt)}
        ncpu = os.cpu_count() or 1
        if ncpu > 1:
            rc2 = conv(ncpu)
            rc2.bench(src, 2)
            s2 = rc2.bench(src, n)
            out["all_cores"] = {"value": round(n / s2, 3), "cores": ncpu,
                                "note": "n-threads=%d; NB the reference's output for 4:2:0 input changes with n-threads "
                                        "(tests/test_video_host.py)" % ncpu}
        return out


class CompositorWorkload:
    """BASELINE C4, video half: SURVEY.md 8d's primary layout."""
    name, unit, dtype = "c4", "frames/s", "u8"
    metric = "4K output frames/s (compositor, 16 x 1080p BGRA pads alpha-blended) per GPU; % HBM roofline"
    kernel = "k_aggregate_direct"
    DW, DH, PW, PH, N = 3840, 2160, 1920, 1080, 16
    frames_per_step = 8
    SETS = 5                # 5 x 16 pad frames of 8.3 MB = 663 MB of distinct pad pixels

    def pad_geometry(self, i):
        return (i % 4) * 640, (i // 4) * 360, 0.25 + 0.05 * i

    def setup(self, dev, rank):
        import torch

        import cases
        from gstreamer_amd import video as V
        self.V = V
        n, pw, ph = self.N, self.PW, self.PH
        base = torch.from_numpy(cases.frame_bytes(pw * ph * 4, "random", 31 + rank)).to(dev)
        self.pads = [torch.roll(base, shifts=i * 4099) for i in range(n * self.SETS)]
        self.arrs = []
        for s_ in range(self.SETS):
            arr = (V.CompositorPad * n)()
            for i in range(n):
                x, y, a = self.pad_geometry(i)
                arr[i].data, arr[i].width, arr[i].height, arr[i].stride = self.pads[s_ * n + i].data_ptr(), pw, ph, pw * 4
                arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = x, y, a, 1
            self.arrs.append(arr)
        self.pool_out = 16
        self.outs = [torch.zeros(self.DW * self.DH * 4, dtype=torch.uint8, device=dev) for _ in range(self.pool_out)]
        self.stream = torch.cuda.current_stream().cuda_stream
        self.L = V.lib()
        self.alg_bytes = n * pw * ph * 4 + self.DW * self.DH * 4        # every pad read once + one canvas write
        self.alg_bytes_per_launch = self.alg_bytes
        self.launches_per_step = self.frames_per_step
        self.units_per_step = self.frames_per_step

    def step(self, s):
        for f in range(self.frames_per_step):
            k = s * self.frames_per_step + f
            r = self.L.gstamd_compositor_aggregate(self.V.FORMATS["BGRA"], 0, self.arrs[k % self.SETS], self.N,
                                                   self.outs[k % self.pool_out].data_ptr(), self.DW, self.DH, self.DW * 4, self.stream)
            assert r == 0, self.V.last_error()

    def config(self, world):
        return {"workload": "%s, %d output frames/step, %d pad frames / %d canvases resident in HBM" %
                            (CONFIG_TEXT["c4"], self.frames_per_step, len(self.pads), self.pool_out),
                "frames_per_step": self.frames_per_step, "frames_per_launch": 1, "parallelism": "stream-per-gpu x%d" % world}

    def cpu_baseline(self):
        import numpy as np

        import cases
        from oracle import ref
        if not ref.available():
            return None
        pad_np = cases.frame_bytes(self.PW * self.PH * 4, "random", 31)
        canvas = np.zeros(self.DW * self.DH * 4, np.uint8)
        n_frames = 40             # 7.4 frames/s on one core: >= 5 s of the reference
        t0 = time.perf_counter()
        for _ in range(n_frames):
            ref.compositor_fill(0, "bgra", "BGRA", canvas, self.DW, self.DH, 0, self.DH)
            for i in range(self.N):
                x, y, a = self.pad_geometry(i)
                ref.compositor_blend("blend_bgra", "BGRA", pad_np, self.PW, self.PH, x, y, a, canvas, self.DW, self.DH, 0, self.DH, 1)
        secs = time.perf_counter() - t0
        return {"value": round(n_frames / secs, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                "sample": "%d output frames: fill_checker + 16 x blend_bgra of the reference (compositor/blend.c, ORC C backups), "
                          "one thread (max-threads default)" % n_frames}


class CompositorOpaqueWorkload(CompositorWorkload):
    """C4's layout with opaque pads (pad alpha 1.0, pixel alpha 255) through gstamd_compositor_aggregate_opaque: blend_pads' canvas byte for byte, without
    the reads under a strip that an opaque pad covers.  --opaque-hint map: per-pad opacity maps made once before the timed region (still pads / frames
    composited more than once); all: the pads flagged all_opaque (frames converted from a format without alpha); none: no hints (the plain kernel)."""
    name = "c4opaque"
    metric = "4K output frames/s (compositor, 16 x 1080p opaque BGRA pads, C4's overlapping layout, strips under an opaque pad not read) per GPU"
    kernel = "k_aggregate_direct_cull"
    hint = "map"

    def pad_geometry(self, i):
        return (i % 4) * 640, (i // 4) * 360, 1.0

    def setup(self, dev, rank):
        import torch
        super().setup(dev, rank)
        V = self.V
        for t in self.pads:
            t.view(-1, 4)[:, 3] = 255
        self.opas, self.maps = [], []
        for s_ in range(self.SETS):
            opa = (V.CompositorPadOpacity * self.N)()
            for i in range(self.N):
                if self.hint == "all":
                    opa[i].all_opaque = 1
                elif self.hint == "map":
                    m = torch.zeros(self.PH, dtype=torch.int64, device=dev)
                    r = self.L.gstamd_compositor_pad_opacity_map(V.FORMATS["BGRA"], self.pads[s_ * self.N + i].data_ptr(), self.PW, self.PH, self.PW * 4,
                                                                 m.data_ptr(), self.stream)
                    assert r == 0, V.last_error()
                    self.maps.append(m)
                    opa[i].map = m.data_ptr()
            self.opas.append(opa)
        torch.cuda.synchronize()
        if self.hint == "none":
            self.kernel = "k_aggregate_direct"
        # what the culled pass has to move: the topmost pad's pixels of every canvas pixel + one canvas write (the 4 x 4 grid at 640 x 360 steps covers the canvas)
        self.alg_bytes_ref = self.alg_bytes
        self.alg_bytes = self.alg_bytes_per_launch = 2 * self.DW * self.DH * 4

    def step(self, s):
        for f in range(self.frames_per_step):
            k = s * self.frames_per_step + f
            r = self.L.gstamd_compositor_aggregate_opaque(self.V.FORMATS["BGRA"], 0, self.arrs[k % self.SETS], self.opas[k % self.SETS] if self.hint != "none" else None,
                                                          self.N, self.outs[k % self.pool_out].data_ptr(), self.DW, self.DH, self.DW * 4, self.stream)
            assert r == 0, self.V.last_error()

    def config(self, world):
        c = super().config(world)
        c["workload"] = ("C4 opaque: compositor, 16 x 1920x1080 BGRA pads at xpos 640*(i%%4), ypos 360*(i/4), pad alpha 1.0, pixel alpha 255, checker background, "
                         "opacity hint '%s'; bytes counted: the visible pad pixels + the canvas (%d MB; blend_pads reads %d MB), %d output frames/step"
                         % (self.hint, self.alg_bytes // 1000000, self.alg_bytes_ref // 1000000, self.frames_per_step))
        return c

    def cpu_baseline(self):
        return None


class CompositorScaledWorkload(CompositorWorkload):
    """SURVEY 8d's C4 variant A ("next"): a 4 x 4 grid of non-overlapping pads, each 1080p BGRA frame scaled to 960x540 by a per-pad
    converter with the library defaults (cubic) - what GstVideoAggregatorConvertPad does - then one aggregate launch."""
    name = "c4a"
    metric = "4K output frames/s (compositor, 16 x 1080p BGRA pads each scaled to 960x540, 4x4 grid) per GPU; % HBM roofline"
    kernel = "k_aggregate_walk (exact halvings with 8-tap passes, scaled pads side by side; other pad sets: k_aggregate_scaled)"
    SW, SH = 960, 540
    FUSED = os.environ.get("GSTAMD_BENCH_C4A_CONVERTERS") is None      # set: the round-2 form, 16 x per-pad converter launches + k_aggregate

    def pad_geometry(self, i):
        return (i % 4) * 960, (i // 4) * 540, 1.0

    def setup(self, dev, rank):
        import torch
        CompositorWorkload.setup(self, dev, rank)
        V = self.V
        self.convs = [V.VideoConverter(V.video_info("BGRA", self.PW, self.PH), V.video_info("BGRA", self.SW, self.SH)) for _ in range(self.N)]
        self.scaled = [torch.zeros(self.SW * self.SH * 4, dtype=torch.uint8, device=dev) for _ in range(self.N)]
        self.sarr = (V.CompositorPad * self.N)()
        for i in range(self.N):
            x, y, a = self.pad_geometry(i)
            self.sarr[i].data, self.sarr[i].width, self.sarr[i].height, self.sarr[i].stride = self.scaled[i].data_ptr(), self.SW, self.SH, self.SW * 4
            self.sarr[i].xpos, self.sarr[i].ypos, self.sarr[i].alpha, self.sarr[i].blend_mode = x, y, a, 1
        self.farr = [(V.CompositorScaledPad * self.N)() for _ in range(self.SETS)]
        for st in range(self.SETS):
            for i in range(self.N):
                x, y, a = self.pad_geometry(i)
                e = self.farr[st][i]
                e.data, e.width, e.height, e.stride = self.pads[st * self.N + i].data_ptr(), self.PW, self.PH, self.PW * 4
                e.xpos, e.ypos, e.alpha, e.blend_mode, e.scaler = x, y, a, 1, self.convs[i]._h
        if self.FUSED:
            assert self.L.gstamd_compositor_pad_scaler_usable(self.convs[0]._h) == 1
        else:
            self.kernel = "16 x per-pad converter (plane scaler) + k_aggregate"
        self.launches_per_step = self.frames_per_step
        self.plan = self.convs[0].describe()

    def step(self, s):
        for f in range(self.frames_per_step):
            k = s * self.frames_per_step + f
            if self.FUSED:
                r = self.L.gstamd_compositor_aggregate_scaled(self.V.FORMATS["BGRA"], 0, self.farr[k % self.SETS], self.N,
                                                              self.outs[k % self.pool_out].data_ptr(), self.DW, self.DH, self.DW * 4, self.stream)
                assert r == 0, self.V.last_error()
                continue
            for i in range(self.N):
                self.convs[i].frame(self.pads[(k % self.SETS) * self.N + i].data_ptr(), self.scaled[i].data_ptr(), self.stream)
            r = self.L.gstamd_compositor_aggregate(self.V.FORMATS["BGRA"], 0, self.sarr, self.N,
                                                   self.outs[k % self.pool_out].data_ptr(), self.DW, self.DH, self.DW * 4, self.stream)
            assert r == 0, self.V.last_error()

    def config(self, world):
        c = CompositorWorkload.config(self, world)
        c["workload"] = ("C4 variant A: compositor, 16 x 1920x1080 BGRA pads each scaled to 960x540 (%s, cubic: %s), 4 x 4 grid, operator over, "
                         "-> 3840x2160 BGRA; %d output frames/step" % ("scaled inside the blend kernel" if self.FUSED else "per-pad converter launches", self.plan,
                                                                      self.frames_per_step))
        return c

    def cpu_baseline(self):
        import numpy as np

        import cases
        from oracle import ref
        if not ref.available():
            return None
        pad_np = cases.frame_bytes(self.PW * self.PH * 4, "random", 31)
        canvas = np.zeros(self.DW * self.DH * 4, np.uint8)
        rc = ref.VideoConverter("BGRA", self.PW, self.PH, "BGRA", self.SW, self.SH)
        n_frames = 18            # 3.4 frames/s on one core: >= 5 s
        t0 = time.perf_counter()
        for _ in range(n_frames):
            ref.compositor_fill(0, "bgra", "BGRA", canvas, self.DW, self.DH, 0, self.DH)
            for i in range(self.N):
                x, y, a = self.pad_geometry(i)
                small = rc.frame(pad_np)
                ref.compositor_blend("blend_bgra", "BGRA", small, self.SW, self.SH, x, y, a, canvas, self.DW, self.DH, 0, self.DH, 1)
        secs = time.perf_counter() - t0
        return {"value": round(n_frames / secs, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                "sample": "%d output frames: fill_checker + 16 x (gst_video_converter_frame 1080p -> 540p cubic + blend_bgra) of the reference, one thread" % n_frames}


class AudioWorkload:
    """BASELINE C4, audio half.  A step = 10 s of stereo F32 handed over in `block`-frame buffers (1024 = what the element sees)."""
    name, unit, dtype = "c4audio", "input frames/s", "f32"
    metric = "audio input frames/s (audioresample 48k->44.1k F32 stereo, polyphase FIR) per GPU"
    kernel = "k_fir<float>"
    CH, N = 2, 48000 * 10

    def __init__(self, block=1024):
        self.block = block

    def setup(self, dev, rank):
        import torch

        import cases
        from gstreamer_amd import audio as A
        self.sig = torch.from_numpy(cases.audio_buffer("F32LE", self.CH, self.N, 4242 + rank)).to(dev)
        self.r = A.AudioResampler("F32LE", self.CH, 48000, 44100, "kaiser", None)
        self.out = torch.zeros((self.N + 4096, self.CH), dtype=torch.float32, device=dev)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.blocks = [(o, min(self.block, self.N - o)) for o in range(0, self.N, self.block)]
        self.launches_per_step = len(self.blocks)
        self.units_per_step = self.N
        no = self.N * 147 // 160
        self.alg_bytes = self.N * self.CH * 4 + no * self.CH * 4
        self.alg_bytes_per_launch = self.alg_bytes / len(self.blocks)
        self.flops_per_step = 2.0 * 72 * no * self.CH

    def step(self, s):
        po = 0
        fsz = self.CH * 4
        for o, n in self.blocks:
            m = self.r.get_out_frames(n)
            self.r.resample(self.sig.data_ptr() + o * fsz, n, self.out.data_ptr() + po * fsz, m, self.stream)
            po += m

    def config(self, world):
        return {"workload": "%s, a step = 10 s of audio in buffers of %d frames (%d launches)" % (CONFIG_TEXT["c4audio"], self.block,
                                                                                                len(self.blocks)),
                "block_frames": self.block, "parallelism": "stream-per-gpu x%d" % world}

    def cpu_baseline(self):
        import cases
        from oracle import ref
        if not ref.available():
            return None
        rr = ref.AudioResampler("F32LE", self.CH, 48000, 44100, quality=4)
        n = 48000 * 60
        data = cases.audio_buffer("F32LE", self.CH, n, 4242)
        t0 = time.perf_counter()
        rr.resample(data, in_frames=n, out_frames=rr.get_out_frames(n))
        secs = time.perf_counter() - t0
        return {"value": round(n / secs, 1), "unit": "input frames/s", "cores": 1, "kind": "reference",
                "sample": "60 s of stereo F32 48k->44.1k in one gst_audio_resampler_resample call (C inner product, no SSE: the summation "
                          "order parity is defined on)"}


class AudioManyWorkload(AudioWorkload):
    """The same buffers for STREAMS independent stereo streams at once (a mixer's inputs, the sessions of a transcoding farm): one
    gstamd_audio_resampler_resample_many call - one launch - per round of 1024-frame buffers.  A step = 10 s of every stream."""
    name = "c4audiomany"
    metric = "audio input frames/s (audioresample 48k->44.1k F32 stereo, 64 independent streams per launch) per GPU"
    kernel = "k_fir_lds_many<float> (blockIdx.y = stream)"
    STREAMS = 64

    def setup(self, dev, rank):
        import torch

        import cases
        from gstreamer_amd import audio as A
        self.A = A
        self.sigs = [torch.from_numpy(cases.audio_buffer("F32LE", self.CH, self.N, 4242 + rank + 7 * i)).to(dev) for i in range(self.STREAMS)]
        self.rs = [A.AudioResampler("F32LE", self.CH, 48000, 44100, "kaiser", None) for _ in range(self.STREAMS)]
        self.outs = [torch.zeros((self.N + 4096, self.CH), dtype=torch.float32, device=dev) for _ in range(self.STREAMS)]
        self.stream = torch.cuda.current_stream().cuda_stream
        self.blocks = [(o, min(self.block, self.N - o)) for o in range(0, self.N, self.block)]
        self.launches_per_step = len(self.blocks)
        self.units_per_step = self.N * self.STREAMS
        no = self.N * 147 // 160
        self.alg_bytes = (self.N * self.CH * 4 + no * self.CH * 4) * self.STREAMS
        self.alg_bytes_per_launch = self.alg_bytes / len(self.blocks)
        self.flops_per_step = 2.0 * 72 * no * self.CH * self.STREAMS

        # the argument arrays of every round, built once: the output frame counts of a stream follow from its input sizes alone
        # (gst_audio_resampler_get_out_frames is a function of the phase, which a throw-away plan walks through here)
        walker = A.AudioResampler("F32LE", self.CH, 48000, 44100, "kaiser", None)
        tmp = torch.zeros((self.block + 64, self.CH), dtype=torch.float32, device=dev)
        self.calls, po, fsz = [], 0, self.CH * 4
        for o, n in self.blocks:
            m = walker.get_out_frames(n)
            walker.resample(self.sigs[0].data_ptr() + o * fsz, n, tmp, m, self.stream)
            self.calls.append(A.ManyBuffers(self.rs, [x.data_ptr() + o * fsz for x in self.sigs], [n] * self.STREAMS,
                                            [x.data_ptr() + po * fsz for x in self.outs], [m] * self.STREAMS))
            po += m
        torch.cuda.synchronize()

    def step(self, s):
        for r in self.rs:
            r.reset()
        for c in self.calls:
            c.run(self.stream)

    def config(self, world):
        return {"workload": "%s, %d independent streams, a step = 10 s of each in buffers of %d frames (%d launches of %d streams)"
                            % (CONFIG_TEXT["c4audio"], self.STREAMS, self.block, len(self.blocks), self.STREAMS),
                "block_frames": self.block, "streams": self.STREAMS, "parallelism": "stream-per-gpu x%d" % world}


class AudioConvertWorkload:
    """audioconvert ! audioresample on the device.  A step = 10 s of stereo F32 48 kHz -> S16 44.1 kHz with the element's default
    triangular dither, in `block`-frame buffers: unpack / resample / quantize + dither / pack per buffer (three launches)."""
    name, unit, dtype = "f4audioconv", "input frames/s", "f64"
    metric = "audio input frames/s (audioconvert F32->S16 tpdf + audioresample 48k->44.1k, stereo) per GPU"
    kernel = "k_aconv_pre + k_fir<double> + k_aconv_post"
    CH, N = 2, 48000 * 10

    def __init__(self, block=1024):
        self.block = block

    def setup(self, dev, rank):
        import torch

        import cases
        from gstreamer_amd import audio as A
        self.A = A
        self.sig = torch.from_numpy(cases.audio_buffer("F32LE", self.CH, self.N, 4242 + rank)).to(dev)
        self.c = A.AudioConverter(A.audio_info("F32LE", 48000, self.CH), A.audio_info("S16LE", 44100, self.CH),
                                  A.audio_converter_config(dither_method="tpdf", resampler_method="kaiser"))
        self.out = torch.zeros((self.N + 4096, self.CH), dtype=torch.int16, device=dev)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.blocks = [(o, min(self.block, self.N - o)) for o in range(0, self.N, self.block)]
        self.launches_per_step = len(self.blocks)
        self.units_per_step = self.N
        no = self.N * 147 // 160
        self.alg_bytes = self.N * self.CH * 4 + no * self.CH * 2
        self.alg_bytes_per_launch = self.alg_bytes / len(self.blocks)

    def step(self, s):
        po = 0
        for o, n in self.blocks:
            m = self.c.get_out_frames(n)
            self.c.samples(self.sig.data_ptr() + o * self.CH * 4, n, self.out.data_ptr() + po * self.CH * 2, m, self.stream)
            po += m

    def config(self, world):
        return {"workload": "SURVEY 8(f)4: audioconvert F32LE 48000 Hz -> S16LE 44100 Hz stereo (tpdf dither, Kaiser resampler inside the converter), "
                            "a step = 10 s of audio in buffers of %d frames (%d converter calls)" % (self.block, len(self.blocks)),
                "block_frames": self.block, "parallelism": "stream-per-gpu x%d" % world}

    def cpu_baseline(self):
        import cases
        from oracle import ref
        if not ref.available():
            return None
        rc = ref.AudioConverter("F32LE", 48000, self.CH, "S16LE", 44100, self.CH,
                                config="GstAudioConverter, GstAudioConverter.dither-method=(GstAudioDitherMethod)tpdf, "
                                       "GstAudioConverter.resampler-method=(GstAudioResamplerMethod)kaiser")
        n = 48000 * 30
        data = cases.audio_buffer("F32LE", self.CH, n, 4242).view("uint8").reshape(-1)
        t0 = time.perf_counter()
        rc.samples(data)
        secs = time.perf_counter() - t0
        rc.free()
        return {"value": round(n / secs, 1), "unit": "input frames/s", "cores": 1, "kind": "reference",
                "sample": "30 s of stereo F32 48k -> S16 44.1k in one gst_audio_converter_samples call (ORC C backups, C inner product)"}


class StubWorkload:
    """CPU stand-in used by tests/test_bench_dist.py to run THIS file's N>1 control flow under gloo: no converter, a step just takes
    a rank-dependent time.  Never part of a measurement."""
    name, unit, dtype, metric, kernel = "stub", "frames/s", "none", "stub", "none"
    frames_per_step = 4

    def setup(self, dev, rank):
        self.rank = rank
        self.launches_per_step = 1
        self.units_per_step = self.frames_per_step
        self.alg_bytes = self.alg_bytes_per_launch = 1000

    def step(self, s):
        time.sleep(0.001 * (1 + self.rank))

    def config(self, world):
        return {"workload": "stub", "parallelism": "stream-per-gpu x%d" % world}

    def cpu_baseline(self):
        return None


def make_workload(args):
    size = tuple(int(v) for v in args.size.split("x")) if args.size else None
    if args.config in VIDEO_CONFIGS:
        return VideoWorkload(args.config, args.batch, size)
    if args.config == "c4":
        return CompositorWorkload()
    if args.config == "c4opaque":
        w = CompositorOpaqueWorkload()
        w.hint = args.opaque_hint
        return w
    if args.config == "c4a":
        return CompositorScaledWorkload()
    if args.config == "c4audio":
        return AudioWorkload(args.audio_block)
    if args.config == "c4audiomany":
        return AudioManyWorkload(args.audio_block)
    if args.config == "f4audioconv":
        return AudioConvertWorkload(args.audio_block)
    if args.config == "stub":
        return StubWorkload()
    raise SystemExit("unknown --config %r" % args.config)


# ------------------------------------------------------------------------------------------------
def reduce_job(wall_s, units_this_rank, device, distributed):
    """Whole-job numbers from per-rank measurements: ranks convert independent streams (no data-path
    collective), so the job time is the MAX over ranks and the job's frames are the SUM over ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([wall_s], dtype=torch.float64, device=device)
    n = torch.tensor([float(units_this_rank)], dtype=torch.float64, device=device)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(float(n.item())))


def gather_ranks(values, device, distributed, world):
    """[values of rank 0, values of rank 1, ...] (a short list of floats per rank) on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if not distributed:
        return [list(map(float, t.tolist()))]
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [list(map(float, o.tolist())) for o in out]


def stored_traffic(name, frames_per_launch=None):
    """HBM bytes per launch from the PMC passes kept under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    runs of this same command; scripts/gpu_profiles.sh).  A stored figure, not a per-run measurement: the source says so.  The
    passes were taken at some number of frames per launch: the figure is scaled to this run's (traffic per frame does not
    depend on the list length for these kernels - every frame of a list is read and written once)."""
    for path in (os.path.join(ROOT, "profiles", "traffic_%s.json" % name),):
        if os.path.exists(path):
            try:
                d = json.load(open(path))
                b, at = int(d["hbm_bytes_per_launch"]), int(d.get("frames_per_launch", 0) or 0)
                note = ""
                if frames_per_launch and at and at != frames_per_launch:
                    b = int(round(b * frames_per_launch / at))
                    note = "; taken at %d frame(s) per launch, scaled to %d" % (at, frames_per_launch)
                return b, "stored PMC figure: profiles/%s (%s%s)" % (os.path.basename(path), d.get("measured", "round 1"), note)
            except Exception:
                pass
    return None, None


def element_env():
    """Environment in which plugins/tests/bench_element finds the elements of plugins/ (the GStreamer runtime of this image: conda 1.14)."""
    env = dict(os.environ)
    env.update(GST_PLUGIN_PATH=os.path.join(ROOT, "plugins") + ":/opt/conda/lib/gstreamer-1.0", GST_PLUGIN_SYSTEM_PATH="/nonexistent",
               GST_REGISTRY="/tmp/gstamd_bench_registry.bin", GST_REGISTRY_FORK="no", GSTAMD_ELEMENT_STATS="0",
               LD_LIBRARY_PATH=os.path.join(ROOT, "gstreamer_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", ""))
    if os.path.exists("/usr/lib/x86_64-linux-gnu/libstdc++.so.6"):
        env["LD_PRELOAD"] = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    return env


def secondary_c2(wl, sync):
    """What the headline does NOT say: the same C2 conversion with ONE frame per launch at the C ABI, and through the
    `videoconvertscale` element (GstHarness, HBM buffers; plugins/tests/bench_element.c) per buffer, in buffer lists of 4 and with
    batch-buffers=8.  Same accounting (45,619,200 algorithmic bytes per 4K frame over the 8 TB/s peak), measured after the headline's timed
    region, never part of `value`."""
    import torch
    out = {"note": "same workload and accounting as the headline; measured after it, not part of `value`"}
    # one launch per frame at the C ABI
    n = 1600
    for i in range(200):
        wl.conv.frame(wl.in_ptrs[i % wl.pool_in], wl.out_ptrs[i % wl.pool_out], wl.stream)
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(n):
        wl.conv.frame(wl.in_ptrs[i % wl.pool_in], wl.out_ptrs[i % wl.pool_out], wl.stream)
    ev1.record()
    sync()
    us = ev0.elapsed_time(ev1) * 1e3 / n
    out["c_abi_one_frame_per_launch"] = {"us_per_frame": round(us, 3), "frames_per_s": round(1e6 / us, 1),
                                         "frac": round(wl.alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "frames": n}
    exe = os.path.join(ROOT, "plugins", "tests", "bench_element")
    if not os.path.exists(exe):
        out["element"] = "unavailable: plugins/tests/bench_element not built"
        return out
    rows = []
    # (label, hip-streams, buffers per list, batch-buffers)
    for label, streams, list_n, batch in (("per buffer, 1 stream", 1, 1, 1), ("per buffer, hip-streams=3 (the element default)", 3, 1, 1),
                                          ("buffer lists of 4", 1, 4, 1), ("batch-buffers=8", 1, 1, 8)):
        try:
            argv = [exe, wl.ifmt, str(wl.w), str(wl.h), wl.ofmt, str(wl.ow), str(wl.oh), "640", str(streams), "bilinear", str(list_n), str(batch)]
            r = subprocess.run(argv, env=element_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            js = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not js:
                rows.append({"case": label, "error": (r.stderr or r.stdout)[-200:]})
                continue
            d = json.loads(js[-1])
            rows.append({"case": label, "hip_streams": d.get("hip_streams"), "buffers_per_list": list_n, "batch_buffers": batch,
                         "us_per_frame": d["us_per_frame"], "frames_per_s": d["frames_per_s"],
                         "frac": round(d["algorithmic_gb_per_s"] / HBM_PEAK_GBS, 4), "frames": d["frames"]})
        except Exception as e:      # a report, never a reason to fail the bench
            rows.append({"case": label, "error": repr(e)[:200]})
    out["element"] = rows
    try:
        out["round6_pairs"] = secondary_pairs(sync)
    except Exception as e:      # a report, never a reason to fail the bench
        out["round6_pairs"] = {"error": repr(e)[:200]}
    return out


def secondary_pairs(sync):
    """The transcoding pairs round 6 gave kernels of their own (k_deep_scale_pack / _scale4 / _pack16, the A Y U V layout of the bilinear
    4:2:0 kernels, k_deep_planes16 across layouts and deep -> deep), so that the driver's own run records them: microseconds per 4K source frame with one
    frame per call and in lists of 8 (HIP events), fraction of the 8 TB/s peak on the plan's algorithmic bytes.  Not part of `value`."""
    import torch
    from gstreamer_amd import video as V
    bil = {"resampler_method": "linear", "max_taps": 2}
    pairs = [("P010_10LE", "NV12", 1920, 1080, bil), ("P010_10LE", "BGRA", 1920, 1080, bil), ("P010_10LE", "P010_10LE", 1920, 1080, bil), ("NV12", "I420", 1920, 1080, bil),
             ("P010_10LE", "I420", 3840, 2160, {}), ("P010_10LE", "I420_10LE", 3840, 2160, {}), ("NV12", "I420_10LE", 3840, 2160, {})]
    dev = torch.device("cuda", torch.cuda.current_device())
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for ifmt, ofmt, ow, oh, cfg in pairs:
        ii, oi = V.video_info(ifmt, 3840, 2160), V.video_info(ofmt, ow, oh)
        conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
        src = torch.randint(0, 255, (8, int(ii.size)), dtype=torch.uint8, device=dev)
        dst = torch.zeros((8, int(oi.size)), dtype=torch.uint8, device=dev)
        srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for i in range(8):
            conv.frame(srcs[i], dsts[i], st)
        conv.frames(srcs, dsts, st)
        sync()
        ev[0].record()
        for i in range(80):
            conv.frame(srcs[i % 8], dsts[i % 8], st)
        ev[1].record()
        ev[2].record()
        for i in range(20):
            conv.frames(srcs, dsts, st)
        ev[3].record()
        sync()
        one, lst = ev[0].elapsed_time(ev[1]) * 1e3 / 80, ev[2].elapsed_time(ev[3]) * 1e3 / 160
        alg = conv.algorithmic_bytes()
        rows.append({"pair": "%s 3840x2160 -> %s %dx%d%s" % (ifmt, ofmt, ow, oh, " bilinear" if cfg else ""), "plan": conv.describe()[:90],
                     "us_per_frame_single": round(one, 2), "frac_single": round(alg / (one * 1e-6) / 1e9 / HBM_PEAK_GBS, 3),
                     "us_per_frame_lists_of_8": round(lst, 2), "frac_lists": round(alg / (lst * 1e-6) / 1e9 / HBM_PEAK_GBS, 3), "list_launches": conv.list_launches()})
        conv.free()
        del src, dst, srcs, dsts
    return rows


def spawn_ranks(args):
    """`--gpus N` given without a torchrun environment: run N ranks of this file under torch.distributed.run."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="c2", help="c2 (headline) | c1 | c3 | c4 | c4opaque | c4audio | c4audiomany | c5 | c4a | f2gamma | f2p010out | f2p010in | f4audioconv | f5encode16 | f8scale | f8pack | f8swizzle | f6p010nv12 | f6p010bgra | f6p010p010 | f6nv12i420")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="c2: skip the batch-1 / element numbers reported beside the headline")
    ap.add_argument("--preheat-ms", type=float, default=60.0,
                    help="untimed sustained load before the W warmup steps: from idle an MI355X needs 20-30 ms of load "
                         "to reach its steady rate (scripts/clock_ramp.py, profiles/r01_clock_ramp.log)")
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per kernel launch (gstamd_video_converter_frames, the GstBufferList analogue); 1 = one launch per frame")
    ap.add_argument("--opaque-hint", default="map", choices=["map", "all", "none"], help="c4opaque: where the pads' opacity is known from")
    ap.add_argument("--audio-block", type=int, default=1024, help="c4audio: frames per resample call")
    ap.add_argument("--size", default=None, help="experiments only: frame size (the headline metric is 3840x2160)")
    ap.add_argument("--backend", default=None, help=argparse.SUPPRESS)      # tests: gloo + the stub workload on CPU
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE) - refusing to report a %d-GPU number"
                         % (args.gpus, world, args.gpus))
    cpu_mode = args.backend == "gloo"
    shared_gpu = False
    if cpu_mode:
        if args.config != "stub":
            raise SystemExit("--backend gloo is for the stub workload (tests) only: the HIP path has no CPU fallback")
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        if torch.cuda.device_count() < world and world > 1 and os.environ.get("GSTAMD_BENCH_SHARE_GPU") != "1":
            raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (world, torch.cuda.device_count()))
        # dry run of the N-rank job on fewer devices (GSTAMD_BENCH_SHARE_GPU=1): ranks share devices, which RCCL refuses ("Duplicate GPU detected") - the
        # barrier and the two reductions of the line (never the data path) go over gloo on host tensors, and the line says so
        shared_gpu = world > 1 and torch.cuda.device_count() < world
        local_dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_dev)
        dev = torch.device("cuda", local_dev)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cpu_mode or shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    wl = make_workload(args)
    steps = args.steps if args.steps is not None else {"c2": 400, "c1": 800, "c3": 150, "c5": 150, "c4": 150, "c4audio": 20, "c4audiomany": 4, "f4audioconv": 10, "stub": 5}.get(args.config, 50)
    warmup = args.warmup if args.warmup is not None else {"c4audio": 2, "c4audiomany": 1, "f4audioconv": 2, "stub": 1}.get(args.config, 20)
    wl.setup(dev, rank)

    def sync():
        if not cpu_mode:
            torch.cuda.synchronize()

    # untimed: bring the device from idle to its steady state, then the W warmup steps of the contract
    t_pre, s_pre = time.perf_counter(), 0
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms and not cpu_mode:
        wl.step(s_pre)
        s_pre += 1
        if s_pre % 8 == 0:
            sync()
    for s in range(warmup):
        wl.step(s)
    sync()
    if distributed:
        dist.barrier()
    sync()
    if not cpu_mode:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if not cpu_mode:
        ev0.record()
    for s in range(steps):
        wl.step(s)
    if not cpu_mode:
        ev1.record()
    sync()
    t_rank = time.perf_counter() - t0               # this rank's own time (before the closing barrier)
    if distributed:
        dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1) if not cpu_mode else t_rank * 1e3

    ctl_dev = torch.device("cpu") if shared_gpu else dev
    wall_max, total_units = reduce_job(wall, steps * wl.units_per_step, ctl_dev, distributed)
    launches = steps * wl.launches_per_step
    per_launch_us = ev_ms * 1e3 / launches
    achieved = wl.alg_bytes_per_launch / (per_launch_us * 1e-6) / 1e9
    per_rank = gather_ranks([steps * wl.units_per_step / t_rank, achieved], ctl_dev, distributed, world)

    if rank == 0:
        assert total_units == steps * wl.units_per_step * world
        traffic, traffic_src = stored_traffic(args.config, wl.config(world).get("frames_per_launch"))
        line = {
            "metric": wl.metric,
            "value": round(total_units / wall_max, 1),
            "unit": wl.unit,
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(wall_max * 1e3 / steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl.dtype,
            "data": "synthetic",
            "config": dict(wl.config(world), preheat_ms=args.preheat_ms),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": wl.kernel, "algorithmic_bytes_per_launch": int(wl.alg_bytes_per_launch),
                         "algorithmic_bytes_per_unit": wl.alg_bytes, "avg_launch_us": round(per_launch_us, 3)},
            "per_rank": [{"rank": i, "value": round(v[0], 1), "hbm_gbs": round(v[1], 1)} for i, v in enumerate(per_rank)],
        }
        if shared_gpu:
            line["dry_run"] = ("%d ranks on %d device(s) (GSTAMD_BENCH_SHARE_GPU=1): the ranks' streams share a GPU, control-plane collectives over gloo - "
                               "a launcher check, NOT a %d-GPU number" % (world, torch.cuda.device_count(), world))
        if hasattr(wl, "flops_per_step"):
            line["roofline"]["gflops"] = round(wl.flops_per_step * steps / (ev_ms * 1e-3) / 1e9, 1)
        if world == 1 and args.config == "c2" and not args.no_secondary and not cpu_mode and not args.size and not args.batch:
            line["secondary"] = secondary_c2(wl, sync)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = wl.cpu_baseline()
            except Exception as e:  # the baseline is a report, never a reason to fail the bench
                line["cpu_baseline"] = {"value": None, "unit": wl.unit, "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
