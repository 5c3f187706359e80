#!/usr/bin/env python3
"""Randomised differential test of the audio converter: kernel bodies on the host emulator against the reference's
gst_audio_converter_samples over random formats, channel counts, rates, dither and noise-shaping methods and buffer sizes.
python scripts/fuzz_audio_convert.py <seed> <count> [-v] [-t]   (late round 2: 650 draws, none differing)"""
import sys, random, ctypes as C, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_audio_convert as T
from gstreamer_amd import audio as A
from oracle import ref
emu = C.CDLL(os.path.join(ROOT, 'tests', 'emu', 'libgstamdemu.so'))
seed=int(sys.argv[1]); n=int(sys.argv[2])
rnd=random.Random(seed)
FM=list(T.BYTES)
ok=bad=refused=0
for it in range(n):
    ifmt, ofmt = rnd.choice(FM), rnd.choice(FM)
    ic = rnd.choice([1,1,2,2,2,3,4,6,8]); oc = ic if rnd.random()<0.6 else rnd.choice([1,2,4,6])
    ir = rnd.choice([8000,22050,44100,48000,96000]); orr = ir if rnd.random()<0.7 else rnd.choice([8000,16000,44100,48000])
    kw={}
    if rnd.random()<0.6: kw["dither_method"]=rnd.choice(["none","rpdf","tpdf","tpdf-hf"])
    if rnd.random()<0.3: kw["noise_shaping"]=rnd.choice(["none","error-feedback","simple","medium","high"])
    bufs=tuple(rnd.choice([1,7,256,500,1024] if ir==orr else [256,500,1024]) for _ in range(rnd.randint(1,3)))
    case=("fz%d"%it, ifmt, ir, ic, ofmt, orr, oc, kw, None, bufs)
    if "-t" in sys.argv: print("CASE", case, flush=True)
    try:
        ii, oi = T.infos(case)
        cfg = A.audio_converter_config(mix_matrix=None, **T.config_kw(kw))
    except Exception as e:
        print("SETUP", case, e); continue
    cv = T.EmuConverter(emu, ii, oi, cfg)
    if not cv.h:
        refused+=1
        if "-v" in sys.argv: print("REFUSED", case, cv.err.value.decode()[:100])
        continue
    try:
        srcs, exp = T.reference_stream(ref, case)
    except Exception as e:
        print("REFERR", case, str(e)[:100]); emu.emu_aconv_free(cv.h); continue
    good=True
    for k, src in enumerate(srcs):
        got = cv.samples(src, T.BYTES[ifmt]*ic, T.BYTES[ofmt]*oc)
        if got.size != exp[k].size or not (got == exp[k]).all():
            good=False
            print("MISMATCH", case, "buf", k, got.size, exp[k].size, int((got[:min(got.size,exp[k].size)] != exp[k][:min(got.size,exp[k].size)]).sum()))
            break
    emu.emu_aconv_free(cv.h)
    ok+=good; bad+=(not good)
print("seed",seed,"ok",ok,"refused",refused,"bad",bad)
