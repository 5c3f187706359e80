"""the open 40-pad finding of the compositor fuzz: for failing draws, is the first launch (32 pads) already wrong, and was the draw a scaled one?"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases, test_compositor_fuzz as T
from gstreamer_amd import video as V
from oracle import ref
gpu = torch.device("cuda:0")
for seed, want in ((7018, 1), (7022, 10), (7045, 7), (7085, 2)):
    rnd = random.Random(seed)
    for it in range(12):
        fmt = rnd.choice(["BGRA", "RGBA", "ARGB", "ABGR", "AYUV", "ARGB64", "AYUV64"])
        wide = fmt.endswith("64"); bpp = 8 if wide else 4
        dw, dh = rnd.randint(8, 200), rnd.randint(8, 120)
        background = rnd.randint(0, 3)
        n = rnd.choice([1, 2, 5, 17, 40])
        scaled_ok = not wide and rnd.random() < 0.5
        pads = []
        for i in range(n):
            w, h = rnd.randint(1, 90), rnd.randint(1, 60)
            ow = oh = 0
            if scaled_ok and rnd.random() < 0.4:
                ow, oh = rnd.randint(1, 120), rnd.randint(1, 80)
                if (ow, oh) == (w, h):
                    ow = oh = 0
            method = rnd.choice(T.METHODS)
            x, y = rnd.randint(-60, dw + 10), rnd.randint(-40, dh + 10)
            alpha = rnd.choice([1.0, 1.0, 0.75, 0.5, 0.3, 0.004, 0.0])
            pads.append((w, h, ow, oh, method, x, y, alpha, rnd.randint(0, 2)))
        if it != want:
            continue
        frames = [cases.frame_bytes(p[0] * p[1] * bpp, "random", seed * 10000 + it * 100 + i) for i, p in enumerate(pads)]
        d_frames = [torch.from_numpy(f).to(gpu) for f in frames]
        scaled = any(p[2] for p in pads)
        print("seed", seed, "draw", it, fmt, "bg", background, "canvas", dw, dh, "scaled draw:", scaled, "modes", sorted(set(p[8] for p in pads)), "dw%4", dw % 4, flush=True)
        if scaled:
            continue
        for m in (8, 16, 24, 32, 33, 36, 40):
            exp = T.expected(ref, fmt, background, pads[:m], frames[:m], dw, dh)
            arr = (V.CompositorPad * m)()
            for i, (w, h, ow, oh, method, x, y, alpha, mode) in enumerate(pads[:m]):
                arr[i].data, arr[i].width, arr[i].height, arr[i].stride = d_frames[i].data_ptr(), w, h, w * bpp
                arr[i].xpos, arr[i].ypos, arr[i].alpha, arr[i].blend_mode = x, y, alpha, mode
            d = torch.empty(dw * dh * bpp, dtype=torch.uint8, device=gpu)
            V._check(V.lib().gstamd_compositor_aggregate(V.FORMATS[fmt], background, arr, m, d.data_ptr(), dw, dh, dw * bpp, None))
            torch.cuda.synchronize()
            got = d.cpu().numpy()
            bad = np.nonzero(got != exp)[0]
            print("   first %2d pads: %d bytes differ%s" % (m, len(bad), "" if not len(bad) else "  first at pixel (%d, %d) byte %d" % ((bad[0] // bpp) % dw, (bad[0] // bpp) // dw, bad[0] % bpp)), flush=True)
