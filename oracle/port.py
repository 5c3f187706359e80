"""oracle/port.py - numpy RESTATEMENT of the reference's algorithms for the hot path (TEST INFRASTRUCTURE ONLY).

Independent of the product code (it shares nothing with gstreamer_amd/csrc) and written from the reference
sources; every function cites the reference lines it follows (paths under
/root/reference/subprojects/gst-plugins-base/).  It is PINNED: tests/test_oracle_port.py checks it against the
golden vectors generated from the reference itself (oracle/_ref, see tests/golden/make_golden.py) - the
reference's own tests hold no pixel/sample goldens for this path.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import it; the product never does.

Covered: the generic GstVideoConverter path for 4:2:0 YUV (NV12/NV21/I420) -> 4-byte RGB with the regular
chroma pairing (unpack, chroma upsample, AYUV->ARGB matrix, pack), the separable scalers (nearest, 2-tap
"bilinear", N-tap LQ with Lanczos/cubic/linear taps) in the reference's pass order, compositor blend_bgra/argb,
and the float polyphase FIR in FULL mode (Kaiser, cubic table interpolation, C summation order).
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------------------------
# colour matrix: gst-libs/gst/video/video-converter.c:901-1066 (4x4 helpers), :1372-1442, prepare_matrix :1323
# ------------------------------------------------------------------------------------------------------------
KR_KB = {"bt709": (0.2126, 0.0722), "bt601": (0.2990, 0.1140)}        # video-color.c:423-459


def ayuv_to_argb_params(matrix="bt709"):
    """p1..p5 of video_orc_convert_AYUV_ARGB for limited-range YUV -> full-range RGB."""
    f32 = np.float32
    m = np.eye(4)

    def mul(a, b):
        out = np.zeros((4, 4))
        for i in range(4):
            for j in range(4):
                x = 0.0
                for k in range(4):
                    x += a[i][k] * b[k][j]
                out[i][j] = x
        return out

    off = np.eye(4)
    off[0][3], off[1][3], off[2][3] = -16.0, -128.0, -128.0            # color_matrix_offset_components
    m = mul(off, m)
    sc = np.eye(4)
    sc[0][0], sc[1][1], sc[2][2] = 1 / float(f32(219)), 1 / float(f32(224)), 1 / float(f32(224))
    m = mul(sc, m)
    kr, kb = KR_KB[matrix]
    kg = 1.0 - kr - kb
    k = np.array([[1., 0., 2 * (1 - kr), 0.], [1., -2 * kb * (1 - kb) / kg, -2 * kr * (1 - kr) / kg, 0.],
                  [1., 2 * (1 - kb), 0., 0.], [0., 0., 0., 1.]])
    m = mul(k, m)                                                       # color_matrix_YCbCr_to_RGB
    sc = np.eye(4)
    sc[0][0] = sc[1][1] = sc[2][2] = float(f32(255))                    # compute_matrix_to_YUV, RGB full range
    m = mul(sc, m)
    sc[0][0] = sc[1][1] = sc[2][2] = 256.0                              # SCALE_F in prepare_matrix
    m = mul(sc, m)
    im = np.rint(m).astype(np.int64)
    return int(im[0][0]), int(im[0][2]), int(im[2][1]), int(im[1][1]), int(im[1][2])


def _i16(x):
    return ((x + 32768) & 0xffff) - 32768


def matrix_ayuv_argb(ayuv, p):
    """video_orc_convert_AYUV_ARGB, C backup semantics (video-orc-dist.c:22162-22318).  ayuv: [...,4] uint8."""
    a = ayuv[..., 0].astype(np.int64)
    b = ((ayuv[..., 1:].astype(np.int64) - 128) & 0xff)                 # subb 128, bytewise
    s = _i16((b << 8) | b)                                              # splatbw as int16
    wy = _i16((s[..., 0] * p[0]) >> 16)                                 # mulhsw
    r = _i16(wy + _i16((s[..., 2] * p[1]) >> 16))
    bl = _i16(wy + _i16((s[..., 1] * p[2]) >> 16))
    g = _i16(_i16(wy + _i16((s[..., 1] * p[3]) >> 16)) + _i16((s[..., 2] * p[4]) >> 16))
    out = np.stack([a, np.clip(r, -128, 127) + 128, np.clip(g, -128, 127) + 128, np.clip(bl, -128, 127) + 128], axis=-1)
    return out.astype(np.uint8)


# ------------------------------------------------------------------------------------------------------------
# unpack + chroma upsample: video-format.c:92-151, 1593-1640; video-chroma.c:277-327, 687-699;
# pairing (2k-1, 2k): video-converter.c:2991-3021 for sequential requests from line 0
# ------------------------------------------------------------------------------------------------------------
def unpack_420(frame, fmt, w, h):
    """-> AYUV [h, w, 4] uint8 with nearest-duplicated chroma (default strides of video-info.c:997-1062)."""
    s0 = (w + 3) // 4 * 4
    h2 = (h + 1) // 2 * 2
    Y = frame[: s0 * h].reshape(h, s0)[:, :w]
    cw, ch = (w + 1) // 2, (h + 1) // 2
    if fmt in ("NV12", "NV21"):
        uv = frame[s0 * h2: s0 * h2 + s0 * ch].reshape(ch, s0)[:, : 2 * cw].reshape(ch, cw, 2)
        U, V = (uv[..., 0], uv[..., 1]) if fmt == "NV12" else (uv[..., 1], uv[..., 0])
    else:
        s1 = ((w + 1) // 2 * 2 // 2 + 3) // 4 * 4
        p1 = frame[s0 * h2: s0 * h2 + s1 * (h2 // 2)].reshape(h2 // 2, s1)[:ch, :cw]
        p2 = frame[s0 * h2 + s1 * (h2 // 2): s0 * h2 + 2 * s1 * (h2 // 2)].reshape(h2 // 2, s1)[:ch, :cw]
        U, V = (p1, p2) if fmt == "I420" else (p2, p1)
    out = np.empty((h, w, 4), np.uint8)
    out[..., 0] = 255
    out[..., 1] = Y
    rows = np.arange(h) >> 1
    cols = np.arange(w) >> 1
    out[..., 2] = U[rows][:, cols]
    out[..., 3] = V[rows][:, cols]
    return out


def chroma_upsample_420(ayuv, h_cosited):
    """In-place semantics of video_chroma_up_h2(_cs)_u8 per line, then video_chroma_up_v2_u8 on the pairs
    (2k-1, 2k); line 0 and (for even h) line h-1 pair with a clamped copy of themselves -> unchanged."""
    h, w, _ = ayuv.shape
    c = ayuv[..., 2:].astype(np.int64)                                   # [h, w, 2]
    o = c.copy()
    if h_cosited:                                                        # FILT_1_1 on odd i < w-1
        idx = np.arange(1, w - 1, 2)
        o[:, idx] = (c[:, idx - 1] + c[:, idx + 1] + 1) >> 1
    else:                                                                # FILT_3_1 / FILT_1_3, originals as inputs
        idx = np.arange(1, w - 1, 2)
        tr0, tr1 = c[:, idx - 1], c[:, idx + 1]
        o[:, idx] = (3 * tr0 + tr1 + 2) >> 2
        o[:, idx + 1] = (tr0 + 3 * tr1 + 2) >> 2
    v = o.copy()
    ks = np.arange(1, h // 2 + (h % 2 == 1), 1)                          # pairs (2k-1, 2k) with 2k <= h-1
    ks = ks[2 * ks <= h - 1]
    a, b = o[2 * ks - 1], o[2 * ks]
    v[2 * ks - 1] = (3 * a + b + 2) >> 2
    v[2 * ks] = (a + 3 * b + 2) >> 2
    out = ayuv.copy()
    out[..., 2:] = v.astype(np.uint8)
    return out


PACK_POS = {"ARGB": (0, 1, 2, 3), "xRGB": (0, 1, 2, 3), "BGRA": (3, 2, 1, 0), "BGRx": (3, 2, 1, 0),
            "RGBA": (3, 0, 1, 2), "RGBx": (3, 0, 1, 2), "ABGR": (0, 3, 2, 1), "xBGR": (0, 3, 2, 1)}   # video-orc.orc:334-411


def pack_rgb(argb, fmt):
    out = np.empty_like(argb)
    for comp, pos in enumerate(PACK_POS[fmt]):
        out[..., pos] = argb[..., comp]
    return out


# ------------------------------------------------------------------------------------------------------------
# resampler taps: video-resampler.c:144-429; quantisation video-scaler.c:339-388
# ------------------------------------------------------------------------------------------------------------
def _sinc(x):
    return 1.0 if x == 0 else math.sin(math.pi * x) / (math.pi * x)


def resampler(method, in_size, out_size, n_taps=0, max_taps=128, envelope=2.0, sharpness=1.0, sharpen=0.0,
              cubic_b=1 / 3.0, cubic_c=1 / 3.0):
    scale = in_size / float(out_size)
    fx = (1.0 / scale) * sharpness if scale > 1.0 else sharpness
    n_taps = min(n_taps, max_taps)
    env = {"nearest": envelope, "linear": 1.0, "cubic": 2.0, "sinc": envelope, "lanczos": envelope}[method]
    if method == "nearest" and n_taps == 0:
        n_taps = 1
    if n_taps == 0:
        dx = math.ceil(2.0 * env / fx)
        n_taps = int(min(max(dx, 0), max_taps))
    fx = 2.0 * env / n_taps
    ex = 2.0 / n_taps
    n_taps = min(n_taps, in_size)
    tap_offs = (n_taps - 1) // 2
    corr = 0.0 if n_taps == 1 else 0.5
    offsets, taps = [], []
    for j in range(out_size):
        x = (0.5 + j) / out_size * in_size - corr
        x = min(max(x, 0), in_size - 1)
        xi = int(math.floor(x - tap_offs))
        t = []
        for l in range(n_taps):
            d = x - (xi + l)
            if method == "nearest":
                t.append(1.0)
            elif method == "linear":
                a = abs(d) * fx
                t.append(1.0 - a if a < 1.0 else 0.0)
            elif method == "cubic":
                a = abs(d) * fx
                a2, a3, b, c = a * a, a * a * a, cubic_b, cubic_c
                if a <= 1.0:
                    t.append(((12.0 - 9.0 * b - 6.0 * c) * a3 + (-18.0 + 12.0 * b + 6.0 * c) * a2 + (6.0 - 2.0 * b)) / 6.0)
                elif a <= 2.0:
                    t.append(((-b - 6.0 * c) * a3 + (6.0 * b + 30.0 * c) * a2 + (-12.0 * b - 48.0 * c) * a + (8.0 * b + 24.0 * c)) / 6.0)
                else:
                    t.append(0.0)
            elif method == "sinc":
                t.append(_sinc(d * fx))
            else:
                e = d * ex
                t.append((_sinc(d * fx) - sharpen) * (0.0 if (e <= -1 or e >= 1) else _sinc(e)))
        weight = 0.0
        for v in t:
            weight += v
        t = [v / weight for v in t]
        off = xi
        if xi < 0:
            sh = -xi
            for l in range(sh):
                t[sh] += t[l]
            t = t[sh:] + [0.0] * sh
            off += sh
        if xi > in_size - n_taps:
            sh = xi - (in_size - n_taps)
            for l in range(sh):
                t[n_taps - sh - 1] += t[n_taps - sh + l]
            t = [0.0] * sh + t[: n_taps - sh]
            off -= sh
        offsets.append(off)
        taps.append(t)
    return np.array(offsets), np.array(taps), n_taps


def quantise(taps, precision):
    out = np.zeros(taps.shape, np.int64)
    for i, src in enumerate(taps):
        lo, hi, offset = 0.0, 1.0, 0.5
        for _ in range(64):
            q = [_i16(int(math.floor(offset + s * (1 << precision)))) for s in src]
            total = sum(q)
            if total == (1 << precision) or lo == hi:
                break
            if total < (1 << precision):
                if offset > lo:
                    lo = offset
                offset += (hi - lo) / 2
            else:
                if offset < hi:
                    hi = offset
                offset -= (hi - lo) / 2
        out[i] = q
    return out


def scale_axis(img, out_size, axis, method, **opt):
    """One scaler pass over a [h, w, 4] uint8 image; dispatch of video-scaler.c:1202-1342 for 4x8-bit pixels."""
    img = np.moveaxis(img, axis, 0)
    in_size = img.shape[0]
    offs, taps, n = resampler(method, in_size, out_size, **opt)
    src = img.astype(np.int64)
    if n == 1:
        out = src[offs]
    elif n == 2 and axis == 1:                                            # ldreslinl (video-orc-dist.c:26162-26195)
        inc = 0 if out_size == 1 else ((in_size - 1) << 16) // (out_size - 1) - 1
        tmp = np.arange(out_size) * inc
        i0, f = tmp >> 16, ((tmp >> 8) & 0xff)[:, None, None]
        out = (src[i0] * (256 - f) + src[i0 + 1] * f) >> 8
    elif n == 2:                                                          # video_orc_resample_v_2tap_u8_lq
        p1 = quantise(taps, 8)[:, 1][:, None, None]
        s1, s2 = src[offs], src[offs + 1]
        w2 = _i16(_i16(_i16(s2 - s1) * p1) + 128)
        out = ((w2 >> 8) & 0xff) + s1 & 0xff
    else:                                                                 # N-tap LQ: video-orc.orc:2388-2480, 2557-2655
        q = quantise(taps, 6)
        acc = np.zeros((out_size,) + src.shape[1:], np.int64)
        for l in range(n):
            acc = (acc + src[offs + l] * q[:, l][:, None, None]) & 0xffff
        out = np.clip(_i16(acc + 32) >> 6, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def convert_420_to_rgb(frame, fmt, w, h, out_fmt, out_w=None, out_h=None, matrix=None, h_cosited=None, method="cubic", **opt):
    """The generic path for 4:2:0 -> RGB (regular chroma pairing; scale passes placed/ordered as chain_scale,
    video-converter.c:1685-1717).  Defaults by height as the elements negotiate them (video-info.c:155-225)."""
    out_w, out_h = out_w or w, out_h or h
    matrix = matrix or ("bt709" if h > 576 else "bt601")
    h_cosited = (h > 576) if h_cosited is None else h_cosited
    img = chroma_upsample_420(unpack_420(frame, fmt, w, h), h_cosited)
    p = ayuv_to_argb_params(matrix)
    down = out_w * out_h <= w * h

    def scale(im):
        if (out_w, out_h) == (w, h):
            return im
        steps = [(1, out_w), (0, out_h)] if out_w * h <= w * out_h else [(0, out_h), (1, out_w)]
        for axis, size in steps:
            if im.shape[axis] != size:
                im = scale_axis(im, size, axis, method, **opt)
        return im

    img = matrix_ayuv_argb(scale(img), p) if down else scale(matrix_ayuv_argb(img, p))
    return pack_rgb(img, out_fmt).reshape(-1)


# ------------------------------------------------------------------------------------------------------------
# compositor: gst/compositor/compositororc.orc:158-265 (C: compositororc-dist.c), clipping blend.c:41-99
# ------------------------------------------------------------------------------------------------------------
def blend_a32(src, sw, sh, xpos, ypos, alpha, dst, dw, dh, alpha_byte):
    """compositor_orc_blend_argb (alpha_byte 0) / _bgra (alpha_byte 3) with BLEND_A32's clipping, mode OVER."""
    s_alpha = min(max(int(alpha * 255), 0), 255)
    if s_alpha == 0:
        return dst
    S = src.reshape(sh, sw, 4).astype(np.int64)
    D = dst.reshape(dh, dw, 4)
    x0, y0, x1, y1 = max(xpos, 0), max(ypos, 0), min(xpos + sw, dw), min(ypos + sh, dh)
    if x1 <= x0 or y1 <= y0:
        return dst
    s = S[y0 - ypos: y1 - ypos, x0 - xpos: x1 - xpos]
    d = D[y0:y1, x0:x1].astype(np.int64)
    div255 = lambda x: ((x & 0xffff) * 0x8081) >> 23
    a = div255(s[..., alpha_byte:alpha_byte + 1] * s_alpha)
    o = div255(((s * a) & 0xffff) + ((d * (255 - a)) & 0xffff)) & 0xff
    o[..., alpha_byte] = 255
    D[y0:y1, x0:x1] = o.astype(np.uint8)
    return dst


# ------------------------------------------------------------------------------------------------------------
# audio: gst-libs/gst/audio/audio-resampler.c (Kaiser q4 defaults :60-72, taps :206-217, :1063-1208, FULL-mode
# phase build :503-552, inner product :693-707, stream bookkeeping :1466-1488, 1649-1806)
# ------------------------------------------------------------------------------------------------------------
def _bessel_i0(x):
    """Power series; agrees with Ooura's dbesi0 to ~1 ulp, enough for identical float32 taps (pinned by test)."""
    s, term, k = 1.0, 1.0, 1
    q = x * x / 4.0
    while term > 1e-20 * s:
        term *= q / (k * k)
        s += term
        k += 1
    return s


class FloatResampler:
    """48k->44.1k style F32 polyphase FIR: Kaiser window, quality 4, FULL filter mode, cubic table interpolation."""

    def __init__(self, in_rate, out_rate, channels, cutoff=0.94, down_factor=0.97979, atten=85.0, tr_bw=0.087, oversample=8):
        g = math.gcd(in_rate, out_rate)
        self.in_rate, self.out_rate, self.ch = in_rate // g, out_rate // g, channels
        fc = cutoff * (down_factor if out_rate < in_rate else 1.0)
        beta = 0.1102 * (atten - 8.7)
        n = int((atten - 8.0) / (2.285 * 2 * math.pi * tr_bw)) + 1
        if self.out_rate < self.in_rate:
            fc = fc * self.out_rate / self.in_rate
            n = n * self.in_rate // self.out_rate
        n = (n + 7) // 8 * 8
        self.n_taps = n
        mult, osamp = 2, oversample
        while osamp > 1 and mult * self.out_rate < self.in_rate:
            mult *= 2
            osamp >>= 1
        rows = []
        for i in range(osamp + 4):
            x = -(n // 2) + i / float(osamp)
            t = []
            for k in range(n):
                xx = x + k
                y = math.pi * xx
                s = fc if y == 0.0 else math.sin(y * fc) / y
                w = 2.0 * xx / n
                t.append(s * _bessel_i0(beta * math.sqrt(max(1 - w * w, 0))))
            weight = 0.0
            for v in t:
                weight += v
            rows.append(np.array([v / weight for v in t]).astype(np.float32))
        f32 = np.float32
        self.table = np.zeros((self.out_rate, n), f32)
        for phase in range(self.out_rate):
            pos = phase * osamp
            offset, frac = (osamp - 1) - pos // self.out_rate, pos % self.out_rate
            x = f32(frac) / f32(self.out_rate)
            x2 = x * x
            x3 = x2 * x
            c0 = f32(0.16667) * (x3 - x)
            c1 = x + f32(0.5) * (x2 - x3)
            c3 = f32(-0.33333) * x + f32(0.5) * x2 - f32(0.16667) * x3
            c2 = f32(1.0) - c0 - c1 - c3
            r = rows[offset: offset + 4]
            self.table[phase] = ((r[0] * c0 + r[1] * c1) + r[2] * c2) + r[3] * c3
        self.hist = np.zeros((n // 2 - 1, channels), f32)
        self.phase = 0

    def get_out_frames(self, in_frames):
        avail = len(self.hist) + in_frames
        if avail < self.n_taps:
            return 0
        out = (avail - self.n_taps) * self.out_rate
        return 0 if out < self.phase else (out - self.phase) // self.in_rate + 1

    def resample(self, data, out_frames):
        buf = np.concatenate([self.hist, data.astype(np.float32)])
        out = np.zeros((out_frames, self.ch), np.float32)
        idx, phase = 0, self.phase
        for j in range(out_frames):
            win = buf[idx: idx + self.n_taps] * self.table[phase][:, None]        # exact float32 products
            r = win.reshape(self.n_taps // 4, 4, self.ch)
            acc = np.zeros((4, self.ch), np.float32)
            for blk in r:                                                          # 4 interleaved partial sums
                acc = acc + blk
            out[j] = ((acc[0] + acc[1]) + acc[2]) + acc[3]
            phase += self.in_rate % self.out_rate
            idx += self.in_rate // self.out_rate
            if phase >= self.out_rate:
                phase -= self.out_rate
                idx += 1
        self.hist, self.phase = buf[idx:], phase
        return out
