#!/usr/bin/env python3
"""Secondary measurements for BASELINE configs 3, 4 (video + audio) and 5 on one MI355X (bench.py keeps the
headline C2 line).  Prints one JSON line per config with the same roofline accounting:
achieved = algorithmic bytes per launch-group / measured time (HIP events on the launch stream).

  python scripts/bench_configs.py [--iters N] [--cpu]     (--cpu adds bounded reference timings)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases  # noqa: E402
from gstreamer_amd import audio as A  # noqa: E402
from gstreamer_amd import video as V  # noqa: E402

PEAK = 8000.0
dev = torch.device("cuda:0")


def timed(fn, iters, warmup=3, preheat_ms=60.0):
    # from idle the device needs 20-30 ms of sustained load to reach its steady rate (scripts/clock_ramp.py)
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < preheat_ms:
        fn()
        n += 1
        if n % 16 == 0:
            torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def video_case(name, ifmt, w, h, ofmt, ow, oh, cfg, iters, pool=None, cpu=False, cpu_frames=3):
    ii, oi = V.video_info(ifmt, w, h), V.video_info(ofmt, ow, oh)
    if pool is None:          # distinct frames worth >= 600 MB on the input side: defeats the 256 MiB Infinity Cache
        pool = max(4, int(600e6 // int(ii.size)) + 1)
    conv = V.VideoConverter(ii, oi, V.converter_config(**cfg))
    base = torch.from_numpy(cases.frame_bytes(int(ii.size), "random", 11)).to(dev)
    ins = [torch.roll(base, shifts=i * 4099) for i in range(pool)]
    opool = max(4, int(600e6 // int(oi.size)) + 1)
    outs = [torch.zeros(int(oi.size), dtype=torch.uint8, device=dev) for _ in range(opool)]
    stream = torch.cuda.current_stream().cuda_stream
    k = [0]

    def one():
        conv.frame(ins[k[0] % pool], outs[k[0] % opool], stream)
        k[0] += 1

    t = timed(one, iters)
    alg = conv.algorithmic_bytes()
    line = {"config": name, "plan": conv.describe(), "frames_per_s": round(1.0 / t, 1), "us_per_frame": round(t * 1e6, 2),
            "algorithmic_bytes_per_frame": alg, "pool_frames_in_out": [pool, opool],
            "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 1), "peak": PEAK, "unit": "GB/s",
                         "frac": round(alg / t / 1e9 / PEAK, 4)}}
    if cpu:
        try:
            from oracle import ref
            src = cases.frame_bytes(ref.video_info(ifmt, w, h)["size"], "random", 11)
            rc = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, config=cases.ref_config_string(ref, cfg))
            rc.bench(src, 1)
            secs = rc.bench(src, cpu_frames)
            line["cpu_baseline"] = {"value": round(cpu_frames / secs, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                                    "sample": "%d frames, n-threads=1, -O2 C-backup ORC" % cpu_frames}
        except Exception as e:
            line["cpu_baseline"] = {"value": None, "sample": repr(e)}
    conv.free()
    print(json.dumps(line), flush=True)


def compositor_case(iters, cpu=False):
    dw, dh, pw, ph, n = 3840, 2160, 1920, 1080, 16
    base = torch.from_numpy(cases.frame_bytes(pw * ph * 4, "random", 31)).to(dev)
    pads = [torch.roll(base, shifts=i * 4099) for i in range(n * 5)]      # 80 x 8.3 MB = 663 MB of distinct pad frames
    arrs = []
    for s_ in range(5):
        arr = (V.CompositorPad * n)()
        for i in range(n):
            arr[i].data, arr[i].width, arr[i].height, arr[i].stride = pads[s_ * n + i].data_ptr(), pw, ph, pw * 4
            arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = (i % 4) * 640, (i // 4) * 360, 0.25 + 0.05 * i, 1
        arrs.append(arr)
    outs = [torch.zeros(dw * dh * 4, dtype=torch.uint8, device=dev) for _ in range(16)]
    stream = torch.cuda.current_stream().cuda_stream
    L = V.lib()
    k = [0]

    def one():
        assert L.gstamd_compositor_aggregate(V.FORMATS["BGRA"], 0, arrs[k[0] % 5], n, outs[k[0] % 16].data_ptr(), dw, dh, dw * 4, stream) == 0
        k[0] += 1

    t = timed(one, iters)
    alg = n * pw * ph * 4 + dw * dh * 4        # every pad read once + one canvas write
    line = {"config": "C4 compositor: 16 x 1080p BGRA (alpha 0.25+0.05i, checker bg) -> 4K BGRA, fused aggregate",
            "frames_per_s": round(1.0 / t, 1), "us_per_frame": round(t * 1e6, 2), "algorithmic_bytes_per_frame": alg,
            "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 1), "peak": PEAK, "unit": "GB/s",
                         "frac": round(alg / t / 1e9 / PEAK, 4)}}
    if cpu:
        try:
            from oracle import ref
            pad_np = cases.frame_bytes(pw * ph * 4, "random", 31)
            canvas = np.zeros(dw * dh * 4, np.uint8)
            t0 = time.perf_counter()
            ref.compositor_fill(0, "bgra", "BGRA", canvas, dw, dh, 0, dh)
            for i in range(n):
                ref.compositor_blend("blend_bgra", "BGRA", pad_np, pw, ph, (i % 4) * 640, (i // 4) * 360, 0.25 + 0.05 * i,
                                     canvas, dw, dh, 0, dh, 1)
            secs = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(1.0 / secs, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                                    "sample": "1 output frame: fill_checker + 16 x blend_bgra of the reference, single thread"}
        except Exception as e:
            line["cpu_baseline"] = {"value": None, "sample": repr(e)}
    print(json.dumps(line), flush=True)


def audio_case(iters, cpu=False):
    ch, n = 2, 48000 * 10                  # 10 s per call
    sig = torch.from_numpy(cases.audio_buffer("F32LE", ch, n, 4242)).to(dev)
    r = A.AudioResampler("F32LE", ch, 48000, 44100, "kaiser", None)
    no = r.get_out_frames(n)
    out = torch.zeros((no + 64, ch), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one():
        m = r.get_out_frames(n)
        r.resample(sig, n, out, m, stream)

    t = timed(one, iters)
    alg = n * ch * 4 + no * ch * 4
    flops = 2.0 * 72 * no * ch
    line = {"config": "C4 audio: 48k->44.1k F32 stereo, Kaiser q4 (72 taps x 147 phases), 10 s per call",
            "in_frames_per_s": round(n / t, 1), "x_realtime": round(n / t / 48000.0, 1), "us_per_call": round(t * 1e6, 2),
            "algorithmic_bytes_per_call": alg, "gflops": round(flops / t / 1e9, 1),
            "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 1), "peak": PEAK, "unit": "GB/s",
                         "frac": round(alg / t / 1e9 / PEAK, 4)}}
    if cpu:
        try:
            from oracle import ref
            rr = ref.AudioResampler("F32LE", ch, 48000, 44100, quality=4)
            data = cases.audio_buffer("F32LE", ch, n, 4242)
            t0 = time.perf_counter()
            rr.resample(data, in_frames=n, out_frames=rr.get_out_frames(n))
            secs = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(n / secs, 1), "unit": "in-frames/s", "cores": 1, "kind": "reference",
                                    "sample": "10 s of stereo F32, gst_audio_resampler_resample (C inner product, no SSE)"}
        except Exception as e:
            line["cpu_baseline"] = {"value": None, "sample": repr(e)}
    r.free()
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    video_case("C1-size: 1920x1080 NV12->BGRA", "NV12", 1920, 1080, "BGRA", 1920, 1080, {}, a.iters * 4, cpu=a.cpu, cpu_frames=20)
    video_case("C3: 7680x4320 I420 -> 1920x1080 RGBA, Lanczos", "I420", 7680, 4320, "RGBA", 1920, 1080, cases.LAN, a.iters, cpu=a.cpu, cpu_frames=1)
    video_case("C5 (per GPU): 7680x4320 NV12 -> 3840x2160 BGRA, bilinear", "NV12", 7680, 4320, "BGRA", 3840, 2160, cases.LIN, a.iters, cpu=a.cpu, cpu_frames=2)
    video_case("8K same-size: 7680x4320 NV12 -> BGRA", "NV12", 7680, 4320, "BGRA", 7680, 4320, {}, a.iters, cpu=False)
    video_case("encoder feed: 3840x2160 BGRA -> NV12", "BGRA", 3840, 2160, "NV12", 3840, 2160, {}, a.iters, cpu=a.cpu, cpu_frames=3)
    video_case("inference feed: 3840x2160 NV12 -> RGB (24-bit)", "NV12", 3840, 2160, "RGB", 3840, 2160, {}, a.iters, cpu=False)
    video_case("capture: 1920x1080 YUY2 -> BGRA", "YUY2", 1920, 1080, "BGRA", 1920, 1080, {}, a.iters * 2, cpu=False)
    video_case("decoder output: 3840x2160 I420 -> BGRA (the reference's convert_I420_BGRA fastpath)", "I420", 3840, 2160, "BGRA", 3840, 2160, {}, a.iters, cpu=False)
    compositor_case(a.iters, cpu=a.cpu)
    audio_case(max(3, a.iters // 3), cpu=a.cpu)


if __name__ == "__main__":
    main()
