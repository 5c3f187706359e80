#!/usr/bin/env python3
"""Randomised differential test of the converter: kernel bodies on the host emulator (tests/emu) against the reference (oracle/_ref) over
random format pairs, sizes and options.  A plan is either refused or must match the reference byte for byte (bytes outside the picture -
stride padding - excepted).  python scripts/fuzz_video.py <seed> <count> [-v] [--rects]; tests/test_video_fuzz.py runs a fixed slice of it."""
import ctypes as C
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from gstreamer_amd import video as V  # noqa: E402


def random_case(rnd, rects=None, more=None):
    """rects: a second generator; when given, 60 % of the cases also get a source crop and / or a destination rectangle with borders.
    more: a third generator (round 5) for the options the first two never drew: dither methods (error diffusion on 8- and 16-bit lines),
    gamma-mode = remap, primaries-mode"""
    fm = sorted(V.FORMATS)
    ifmt, ofmt = rnd.choice(fm), rnd.choice(fm)
    w, h = rnd.randint(1, 70), rnd.randint(1, 40)
    ow, oh = (w, h) if rnd.random() < 0.5 else (rnd.randint(1, 90), rnd.randint(1, 50))
    cfg = {}
    if rnd.random() < 0.5:
        cfg["resampler_method"] = rnd.choice(["nearest", "linear", "cubic", "sinc", "lanczos"])
    if rnd.random() < 0.2:
        cfg["max_taps"] = rnd.choice([2, 4, 8])
    if rnd.random() < 0.2:
        cfg["alpha_mode"] = rnd.choice(["copy", "set", "mult"])
        cfg["alpha_value"] = rnd.choice([0.25, 0.5, 1.0])
    if rnd.random() < 0.15:
        cfg["chroma_mode"] = rnd.choice(["full", "upsample-only", "downsample-only", "none"])
    if rnd.random() < 0.1:
        cfg["dither_quantization"] = rnd.choice([2, 8, 16])
    if rnd.random() < 0.1:
        cfg["matrix_mode"] = rnd.choice(["full", "input-only", "output-only", "none"])
    col = rnd.choice([None, None, "bt601", "bt709"])
    site = rnd.choice([None, None, "jpeg", "mpeg2", "cosited"])
    if rects is not None:
        r = rects
        if r.random() < 0.4 and w > 2 and h > 2:
            cfg["src_x"], cfg["src_y"] = r.randint(0, w // 2), r.randint(0, h // 2)
            cfg["src_width"], cfg["src_height"] = r.randint(1, w - cfg["src_x"]), r.randint(1, h - cfg["src_y"])
        if r.random() < 0.4 and ow > 2 and oh > 2:
            cfg["dest_x"], cfg["dest_y"] = r.randint(0, ow // 2), r.randint(0, oh // 2)
            cfg["dest_width"], cfg["dest_height"] = r.randint(1, ow - cfg["dest_x"]), r.randint(1, oh - cfg["dest_y"])
            if r.random() < 0.5:
                cfg["border_argb"] = r.randint(0, 0xffffffff)
            if r.random() < 0.2:
                cfg["fill_border"] = 0
    if more is not None:
        m = more
        if m.random() < 0.5:
            cfg["dither_method"] = m.choice(["none", "verterr", "floyd-steinberg", "sierra-lite", "bayer"])
            if m.random() < 0.5:
                cfg["dither_quantization"] = m.choice([2, 4, 16, 64, 256, 1024])
        if m.random() < 0.25:
            cfg["gamma_mode"] = "remap"
        if m.random() < 0.15:
            cfg["primaries_mode"] = m.choice(["merge-only", "fast"])
    return ifmt, w, h, ofmt, ow, oh, cfg, col, site


VERDICTS = ("ok", "refused", "bad", "defined-staged", "defined-masked", "defined-unchecked")


def emu_diverges(emu):
    """-> diverges (case): does the plan of that conversion announce a divergence (or refuse)?  Asked of the planner through the host emulator."""
    def diverges(case):
        ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
        ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site)
        oi = V.video_info(ofmt, ow, oh)
        c = V.converter_config(**cfg)
        src, dst = np.zeros(ii.size, np.uint8), np.zeros(oi.size, np.uint8)
        desc = C.create_string_buffer(256)
        r = emu.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(c), src.ctypes.data, dst.ctypes.data, 1, desc, 256)
        return r != 0 or emu.emu_video_last_divergence().decode() != ""
    return diverges


def check_defined(ref, case, src, got, divergence, diverges):
    """A draw whose plan announces a divergence (the reference's own one-step output is undefined there): compared with the reference run STAGE BY
    STAGE (tests/staged.py), or - where the split cannot express the draw - with the one-step reference on the bytes its undefined part cannot reach.
    -> ("defined-staged" | "defined-masked" | "defined-unchecked" | "bad", text)"""
    import staged
    notes = [d.strip() for d in divergence.split(". ") if d.strip()]
    # (the class whose one-step reference has no line order to follow: the plan pairs the chroma lines in frame order, which is what the split computes)
    canonical = all(n.startswith("vertical-first N-tap scaling fed by the 4:2:0 chroma upsampler") for n in notes)
    if any(n.startswith("packed 4:2:2 of odd width scaled vertically") for n in notes):
        return "defined-unchecked", " (the reference's plane scaler, convert_scale_planes: not the chain the split models)"
    r = staged.staged_expected(ref, case, src, diverges, canonical)
    if r is not None:
        same, text = staged.compare(ref, case, got, r[0], r[1])
        return ("defined-staged" if same else "bad"), text
    checked, same, text = staged.masked_check(ref, case, src, got, divergence)
    if not checked:
        return "defined-unchecked", " (" + staged.stageable(ref, case, diverges, canonical)[1] + ")"
    return ("defined-masked" if same else "bad"), text


def divergence_class(divergence):
    return " + ".join(sorted({d.strip()[:48] for d in divergence.split(". ") if d.strip()}))


def run_case(emu, ref, case, seed):
    """-> (one of VERDICTS, description).  A plan that announces a divergence (gstamd_video_converter_divergence: the reference's own output is
    undefined for this conversion) is checked by check_defined."""
    ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
    ii = V.video_info(ifmt, w, h, colorimetry=col, chroma_site=site)
    oi = V.video_info(ofmt, ow, oh)
    src = cases.frame_bytes(ii.size, "random", seed, w)
    c = V.converter_config(**cfg)
    dst = np.zeros(oi.size, np.uint8)
    desc = C.create_string_buffer(256)
    r = emu.emu_video_convert(C.byref(ii), C.byref(oi), C.byref(c), src.ctypes.data, dst.ctypes.data, 1, desc, 256)
    if r != 0:
        return "refused", desc.value.decode()
    div = emu.emu_video_last_divergence().decode()
    if div:
        verdict, text = check_defined(ref, case, src, dst, div, emu_diverges(emu))
        return verdict, desc.value.decode() + text + " | " + divergence_class(div)
    ok, text = matches_reference(ref, case, src, dst, oi)
    return ("ok" if ok else "bad"), desc.value.decode() + text


def matches_reference(ref, case, src, dst, oi):
    """dst against the reference's frame for the same case -> (equal, what differs)"""
    ifmt, w, h, ofmt, ow, oh, cfg, col, site = case
    want = ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site, config=cases.ref_config_string(ref, cfg)).frame(src)
    if (dst == want).all():
        return True, ""
    if cfg.get("fill_border", 1) == 0 and "dest_x" in cfg:
        # without a border line the reference's generic chain packs whatever its line buffers held left and right of the rectangle
        # (video_converter_generic packs out_maxwidth pixels a line), we leave those bytes alone: compare what the picture alone decides =
        # the bytes three border colours agree on
        ab = [ref.VideoConverter(ifmt, w, h, ofmt, ow, oh, in_colorimetry=col, in_chroma_site=site,
                                 config=cases.ref_config_string(ref, dict(cfg, fill_border=1, border_argb=b))).frame(src) for b in (0x00000000, 0xffffffff, 0x80aa5533)]
        inside = (ab[0] == ab[1]) & (ab[1] == ab[2])
        if (dst[inside] == want[inside]).all():
            return True, ""
        return False, ": %d of %d bytes inside the rectangle differ" % (int((dst[inside] != want[inside]).sum()), int(inside.sum()))
    vb = lambda b: cases.visible_bytes(ofmt, ow, oh, list(oi.stride), list(oi.offset), b)
    try:
        if (vb(dst) == vb(want)).all():
            return True, ""
    except Exception:
        pass
    return False, ": %d of %d bytes differ" % (int((dst != want).sum()), dst.size)


def load_emu():
    emu = C.CDLL(os.path.join(ROOT, "tests", "emu", "libgstamdemu.so"))
    emu.emu_video_convert.argtypes = [C.POINTER(V.VideoInfo), C.POINTER(V.VideoInfo), C.POINTER(V.ConverterConfig), C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_char_p, C.c_int]
    emu.emu_video_last_divergence.restype = C.c_char_p
    return emu


def main():
    from oracle import ref
    seed, n = int(sys.argv[1]), int(sys.argv[2])
    rnd = random.Random(seed)
    emu = load_emu()
    rects = random.Random(seed + 77) if "--rects" in sys.argv else None
    more = random.Random(seed + 313) if "--more" in sys.argv else None
    count = {v: 0 for v in VERDICTS}
    classes = {}
    for it in range(n):
        case = random_case(rnd, rects, more)
        verdict, text = run_case(emu, ref, case, seed * 1000 + it)
        count[verdict] += 1
        if verdict.startswith("defined"):
            k = (text.rsplit(" | ", 1)[1], verdict)
            classes[k] = classes.get(k, 0) + 1
        if verdict == "bad" or "-v" in sys.argv:
            print(verdict.upper(), case, text)
    print("seed %d: %s" % (seed, count))
    for k in sorted(classes):
        print("   %5d  %-18s %s" % (classes[k], k[1], k[0]))
    return 1 if count["bad"] else 0


if __name__ == "__main__":
    sys.exit(main())
