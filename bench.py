#!/usr/bin/env python3
"""bench.py - headline benchmark: 4K frames/s of videoconvertscale's NV12->BGRA path per GPU.

Workload (BASELINE.json configs[1]): 3840x2160 NV12 (bt709, 16-235, chroma-site mpeg2) -> BGRA, frames
resident in HBM, HIP kernels behind the C ABI of include/gstamd_video.h.  One "step" converts
FRAMES_PER_STEP frames, cycling through an input pool and an output pool that together exceed the
256 MiB Infinity Cache, so the kernel really streams from/to HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

Independent streams shard one-per-GPU (no collective on the data path); the only torch.distributed
use is the timing barrier.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 3840, 2160
FRAMES_PER_STEP = 32
POOL_IN = 32          # 32 x 12.4 MB = 398 MB of distinct input frames
POOL_OUT = 16         # 16 x 33.2 MB = 531 MB of distinct output frames
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(sample_frames=40):
    """The reference's own gst_video_converter_frame (oracle/_ref, C-backup ORC, -O2) on this host,
    element-default n-threads=1, bounded sample of the same 4K workload."""
    try:
        import cases
        from oracle import ref
        if not ref.available():
            return None
        src = cases.frame_bytes(ref.video_info("NV12", W, H)["size"], "random", 1)
        rc = ref.VideoConverter("NV12", W, H, "BGRA", W, H, config=ref.config_string(GstVideoConverter__threads=1))
        rc.bench(src, 3)
        secs = rc.bench(src, sample_frames)
        out = {"value": round(sample_frames / secs, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": "%d frames of 3840x2160 NV12->BGRA, gst_video_converter_frame of the reference built "
                         "from /root/reference with -O2 -DDISABLE_ORC (ORC C backups, no JIT SIMD), n-threads=1 "
                         "(element default)" % sample_frames}
        ncpu = os.cpu_count() or 1
        if ncpu > 1:
            rc2 = ref.VideoConverter("NV12", W, H, "BGRA", W, H,
                                     config=ref.config_string(GstVideoConverter__threads=ncpu))
            rc2.bench(src, 3)
            s2 = rc2.bench(src, sample_frames)
            out["all_cores"] = {"value": round(sample_frames / s2, 3), "cores": ncpu,
                                "note": "n-threads=%d; NB the reference's output for 4:2:0 input changes with "
                                        "n-threads (tests/test_video_host.py)" % ncpu}
        return out
    except Exception as e:  # the baseline is a report, never a reason to fail the bench
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference", "sample": "unavailable: %r" % (e,)}


def reduce_job(wall_s, frames_this_rank, device, distributed):
    """Whole-job numbers from per-rank measurements: ranks convert independent streams (no data-path
    collective), so the job time is the MAX over ranks and the job's frames are the SUM over ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([wall_s], dtype=torch.float64, device=device)
    n = torch.tensor([float(frames_this_rank)], dtype=torch.float64, device=device)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(float(n.item())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat-ms", type=float, default=60.0,
                    help="untimed sustained load before the W warmup steps: from idle an MI355X needs 20-30 ms of load "
                         "to reach its steady rate (scripts/clock_ramp.py, profiles/r01_clock_ramp.log)")
    ap.add_argument("--batch", type=int, default=32,
                    help="frames per kernel launch (gstamd_video_converter_frames, the GstBufferList analogue); "
                         "1 = one launch per frame")
    ap.add_argument("--size", default="3840x2160", help="experiments only: frame size (the headline metric is 3840x2160)")
    args = ap.parse_args()
    global W, H
    W, H = [int(v) for v in args.size.split("x")]

    import torch
    import torch.distributed as dist

    import cases
    from gstreamer_amd import video as V

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    ii, oi = V.video_info("NV12", W, H), V.video_info("BGRA", W, H)
    conv = V.VideoConverter(ii, oi)          # element defaults: no scaling here, generic fused path
    alg_bytes = conv.algorithmic_bytes()

    # synthetic frames: full-range xorshift bytes, a different seed per pool slot and per rank
    pool_in = torch.empty((POOL_IN, int(ii.size)), dtype=torch.uint8, device=dev)
    base = torch.from_numpy(cases.frame_bytes(int(ii.size), "random", 2000 + rank)).to(dev)
    for i in range(POOL_IN):
        pool_in[i] = torch.roll(base, shifts=i * 4099)
    pool_out = torch.zeros((POOL_OUT, int(oi.size)), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream   # kernels go on torch's current stream
    in_ptrs = [pool_in[i].data_ptr() for i in range(POOL_IN)]
    out_ptrs = [pool_out[i].data_ptr() for i in range(POOL_OUT)]

    B = max(1, min(args.batch, FRAMES_PER_STEP))
    assert FRAMES_PER_STEP % B == 0

    def step(s):
        for f in range(0, FRAMES_PER_STEP, B):
            n = s * FRAMES_PER_STEP + f
            if B == 1:
                conv.frame(in_ptrs[n % POOL_IN], out_ptrs[n % POOL_OUT], stream)
            else:
                conv.frames([in_ptrs[(n + i) % POOL_IN] for i in range(B)],
                            [out_ptrs[(n + i) % POOL_OUT] for i in range(B)], stream)

    # untimed: bring the device from idle to its steady state, then the W warmup steps of the contract
    t_pre = time.perf_counter()
    s_pre = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        step(s_pre)
        s_pre += 1
        if s_pre % 8 == 0:
            torch.cuda.synchronize()
    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for s in range(args.steps):
        step(s)
    ev1.record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)

    wall_max, total_frames = reduce_job(wall, args.steps * FRAMES_PER_STEP, dev, distributed)

    if rank == 0:
        launches = args.steps * FRAMES_PER_STEP // B
        per_launch_us = ev_ms * 1e3 / launches
        achieved = alg_bytes * B / (per_launch_us * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_c2.json")
        if os.path.exists(tpath):
            try:
                traffic = int(json.load(open(tpath)).get("hbm_bytes_per_frame") * B)
            except Exception:
                traffic = None
        assert total_frames == launches * B * world
        line = {
            "metric": "4K frames/s (videoconvertscale NV12->BGRA) per GPU; % HBM roofline",
            "value": round(total_frames / wall_max, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall_max * 1e3 / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "C2: 3840x2160 NV12 (bt709 limited, chroma-site mpeg2) -> BGRA, fused unpack+chroma "
                                   "upsample+matrix+pack, %d frames/step, pools %d in / %d out resident in HBM, "
                                   "1 stream per GPU, %d frame(s) per kernel launch" % (FRAMES_PER_STEP, POOL_IN, POOL_OUT, B),
                       "plan": conv.describe(), "frames_per_step": FRAMES_PER_STEP, "frames_per_launch": B, "preheat_ms": args.preheat_ms,
                       "parallelism": "stream-per-gpu x%d" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "k_convert_strip<CHROMA_H_H2_CS, layout BGRA, 0>", "algorithmic_bytes_per_launch": alg_bytes * B,
                         "algorithmic_bytes_per_frame": alg_bytes,
                         "avg_launch_us": round(per_launch_us, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
