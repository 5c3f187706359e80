#!/usr/bin/env python3
"""One of the secondary configs (see scripts/bench_configs.py):  bench_one.py c1|c3|c5|8k|rgb24|rgb24s|yuy2|i420|nv12enc|c4|audio [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_configs as B
import cases
which = sys.argv[1]
it = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "c1":
    B.video_case("C1-size: 1920x1080 NV12->BGRA", "NV12", 1920, 1080, "BGRA", 1920, 1080, {}, it * 4)
elif which == "c3":
    B.video_case("C3: 7680x4320 I420 -> 1920x1080 RGBA, Lanczos", "I420", 7680, 4320, "RGBA", 1920, 1080, cases.LAN, it)
elif which == "c5":
    B.video_case("C5 (per GPU): 7680x4320 NV12 -> 3840x2160 BGRA, bilinear", "NV12", 7680, 4320, "BGRA", 3840, 2160, cases.LIN, it)
elif which == "8k":
    B.video_case("8K same-size: 7680x4320 NV12 -> BGRA", "NV12", 7680, 4320, "BGRA", 7680, 4320, {}, it)
elif which == "rgb24":
    B.video_case("4K NV12 -> RGB (24-bit), same size", "NV12", 3840, 2160, "RGB", 3840, 2160, {}, it * 2)
elif which == "rgb24s":
    B.video_case("4K NV12 -> RGB 224x224 bilinear (inference pre-processing)", "NV12", 3840, 2160, "RGB", 224, 224, cases.LIN, it * 2)
elif which == "yuy2":
    B.video_case("1080p YUY2 -> BGRA", "YUY2", 1920, 1080, "BGRA", 1920, 1080, {}, it * 4)
elif which == "nv12enc":
    B.video_case("4K BGRA -> NV12 (encoder feed)", "BGRA", 3840, 2160, "NV12", 3840, 2160, {}, it * 2)
elif which == "i420":
    B.video_case("4K I420 -> BGRA (decoder output -> display; the reference's convert_I420_BGRA fastpath)", "I420", 3840, 2160, "BGRA", 3840, 2160, {}, it * 2)
elif which == "i420bil":
    B.video_case("4K I420 -> 1080p BGRA, bilinear", "I420", 3840, 2160, "BGRA", 1920, 1080, cases.LIN, it * 2)
elif which == "nv12bil":
    B.video_case("4K NV12 -> 1080p BGRA, bilinear", "NV12", 3840, 2160, "BGRA", 1920, 1080, cases.LIN, it * 2)
elif which == "c4":
    B.compositor_case(it)
else:
    B.audio_case(it)
