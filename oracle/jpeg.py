"""ORACLE (test infrastructure only - never imported by the product).

CPU restatement of `tf.image.adjust_jpeg_quality(img_u8, q)` as the reference calls it (train.py:86-92): a baseline
JPEG encode (4:2:0 chroma subsampling, slow-integer DCT, quality-scaled Annex-K tables, `force_baseline`) followed by a
decode (slow-integer IDCT, "fancy" triangle chroma upsampling), i.e. TensorFlow's `encode_jpeg` / `decode_jpeg` defaults,
which are thin wrappers over libjpeg(-turbo).  Entropy coding is lossless and therefore omitted.  The arithmetic below
restates libjpeg's published integer algorithms (file names refer to libjpeg / libjpeg-turbo, a third-party dependency
that is not part of /root/reference): jccolor.c (RGB->YCbCr, 16-bit fixed point), jcsample.c h2v2_downsample
(alternating bias 1,2), jfdctint.c + jcdctmgr.c (forward DCT scaled by 8, divisors q<<3, round-half-away), jidctint.c,
jdsample.c h2v2_fancy_upsample, jdcolor.c.

PINNED: `tests/test_jpeg_cpu.py` compares this restatement bit for bit with libjpeg itself (through Pillow's JPEG
codec, which uses the same defaults) on random, smooth and saturated images at every quality the reference uses.
"""
import numpy as np

STD_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99],
    np.int64).reshape(8, 8)
STD_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99],
    np.int64).reshape(8, 8)

CONST_BITS, PASS1_BITS = 13, 2
F_0_298, F_0_390, F_0_541, F_0_765, F_0_899, F_1_175 = 2446, 3196, 4433, 6270, 7373, 9633
F_1_501, F_1_847, F_1_961, F_2_053, F_2_562, F_3_072 = 12299, 15137, 16069, 16819, 20995, 25172


def quality_table(std, quality):
    """jpeg_set_quality(q, force_baseline=TRUE): jcparam.c jpeg_quality_scaling + jpeg_add_quant_table."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    return np.clip((std * scale + 50) // 100, 1, 255)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _fdct_1d(d, first):
    """One pass of jfdctint.c over the LAST axis of d[..., 8] (int64)."""
    d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., i] for i in range(8))
    t0, t7, t1, t6, t2, t5, t3, t4 = d0 + d7, d0 - d7, d1 + d6, d1 - d6, d2 + d5, d2 - d5, d3 + d4, d3 - d4
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    n = CONST_BITS - PASS1_BITS if first else CONST_BITS + PASS1_BITS
    o = [None] * 8
    if first:
        o[0], o[4] = (t10 + t11) << PASS1_BITS, (t10 - t11) << PASS1_BITS
    else:
        o[0], o[4] = _descale(t10 + t11, PASS1_BITS), _descale(t10 - t11, PASS1_BITS)
    z1 = (t12 + t13) * F_0_541
    o[2] = _descale(z1 + t13 * F_0_765, n)
    o[6] = _descale(z1 - t12 * F_1_847, n)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * F_1_175
    t4, t5, t6, t7 = t4 * F_0_298, t5 * F_2_053, t6 * F_3_072, t7 * F_1_501
    z1, z2, z3, z4 = -z1 * F_0_899, -z2 * F_2_562, -z3 * F_1_961 + z5, -z4 * F_0_390 + z5
    o[7], o[5], o[3], o[1] = _descale(t4 + z1 + z3, n), _descale(t5 + z2 + z4, n), _descale(t6 + z2 + z3, n), _descale(t7 + z1 + z4, n)
    return np.stack(o, -1)


def _idct_1d(c, first):
    """One pass of jidctint.c over the LAST axis of c[..., 8] (int64)."""
    i0, i1, i2, i3, i4, i5, i6, i7 = (c[..., i] for i in range(8))
    z1 = (i2 + i6) * F_0_541
    t2, t3 = z1 - i6 * F_1_847, z1 + i2 * F_0_765
    t0, t1 = (i0 + i4) << CONST_BITS, (i0 - i4) << CONST_BITS
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F_1_175
    t0, t1, t2, t3 = t0 * F_0_298, t1 * F_2_053, t2 * F_3_072, t3 * F_1_501
    z1, z2, z3, z4 = -z1 * F_0_899, -z2 * F_2_562, -z3 * F_1_961 + z5, -z4 * F_0_390 + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    n = CONST_BITS - PASS1_BITS if first else CONST_BITS + PASS1_BITS + 3
    o = [t10 + t3, t11 + t2, t12 + t1, t13 + t0, t13 - t0, t12 - t1, t11 - t2, t10 - t3]
    return np.stack([_descale(v, n) for v in o], -1)


def _blocks(plane):
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)       # [by, bx, row, col]


def _unblocks(b):
    by, bx = b.shape[:2]
    return b.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


def codec_plane(plane, qtbl):
    """One component: level shift, FDCT, quantise, dequantise, IDCT, range limit.  plane: int [h, w], multiples of 8."""
    d = _blocks(plane.astype(np.int64) - 128)
    d = _fdct_1d(d, True)                                                   # rows
    d = _fdct_1d(d.swapaxes(-1, -2), False).swapaxes(-1, -2)                # columns
    div = qtbl.astype(np.int64) << 3
    coef = np.sign(d) * ((np.abs(d) + (div >> 1)) // div)                   # jcdctmgr.c: round half away from zero
    c = coef * qtbl
    c = _idct_1d(c.swapaxes(-1, -2), True).swapaxes(-1, -2)                 # columns first (jidctint.c pass 1)
    c = _idct_1d(c, False)                                                  # rows
    return np.clip(_unblocks(c) + 128, 0, 255)


def rgb_to_ycc(rgb):
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    half, off = 1 << 15, 128 << 16
    y = (19595 * r + 38470 * g + 7471 * b + half) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + off + half - 1) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + off + half - 1) >> 16
    return y, cb, cr


def h2v2_downsample(p):
    h, w = p.shape
    s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
    bias = np.where(np.arange(w // 2) % 2 == 0, 1, 2)[None, :]
    return (s + bias) >> 2


def h2v2_fancy_upsample(p):
    h, w = p.shape
    up = np.concatenate([p[:1], p[:-1]], 0)       # row above (replicated at the top edge)
    dn = np.concatenate([p[1:], p[-1:]], 0)       # row below (replicated at the bottom edge)
    out = np.empty((2 * h, 2 * w), np.int64)
    for v, other in ((0, up), (1, dn)):
        cs = 3 * p + other                          # column sums of the vertically weighted pair
        last = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        # interior: (3*this + neighbour + 8 | 7) >> 4; at the edges the neighbour is `this` itself (4*this)
        out[v::2, 0::2] = (3 * cs + last + 8) >> 4
        out[v::2, 1::2] = (3 * cs + nxt + 7) >> 4
    return out


def ycc_to_rgb(y, cb, cr):
    half = 1 << 15
    cbx, crx = cb - 128, cr - 128
    r = y + ((91881 * crx + half) >> 16)
    g = y + ((-22554 * cbx - 46802 * crx + half) >> 16)
    b = y + ((116130 * cbx + half) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255)


def adjust_jpeg_quality(img_u8, quality):
    """img_u8: [H, W, 3] uint8 RGB, any size.  Returns the decoded uint8 RGB image.
    Partial 16x16 MCUs follow libjpeg: full-resolution rows / columns are replicated up to an even row count and to
    whole MCU columns before the chroma downsampling (jcsample.c expand_right_edge, jcprepct.c expand_bottom_edge), the
    DOWNSAMPLED planes are then padded downwards by replicating their last row (jcprepct.c), the decoder crops; chroma
    planes of at most two columns are upsampled by plain replication instead of the triangle filter (jdsample.c
    jinit_upsampler: fancy upsampling needs downsampled_width > 2)."""
    h, w, _ = img_u8.shape
    H16, W16 = -(-h // 16) * 16, -(-w // 16) * 16
    he = h + (h & 1)
    pad = np.pad(img_u8, ((0, he - h), (0, W16 - w), (0, 0)), mode="edge")
    y, cb, cr = rgb_to_ycc(pad)
    ql, qc = quality_table(STD_LUMA, quality), quality_table(STD_CHROMA, quality)
    yd = codec_plane(np.pad(y, ((0, H16 - he), (0, 0)), mode="edge"), ql)[:h, :w]
    hc, wc = -(-h // 2), -(-w // 2)
    planes = []
    for c in (cb, cr):
        dec = codec_plane(np.pad(h2v2_downsample(c), ((0, H16 // 2 - he // 2), (0, 0)), mode="edge"), qc)[:hc, :wc]
        up = h2v2_fancy_upsample(dec) if wc > 2 else np.repeat(np.repeat(dec, 2, 0), 2, 1)
        planes.append(up[:h, :w])
    return ycc_to_rgb(yd, planes[0], planes[1]).astype(np.uint8)


def batch_qualities(b):
    """train.py:89: int(round(i / (b - 1) * 10 + 90)) for sample i of a batch of b."""
    return [int(round(float(i) / float(b - 1) * 10.0 + 90.0)) if b > 1 else 90 for i in range(b)]


def jpeg_batch(ldr, order="rgb"):
    """ldr: float [B, H, W, 3] holding k/255 (train.py:83-92).  Returns float32 [B, H, W, 3] = decoded / 255."""
    b = ldr.shape[0]
    out = np.empty(ldr.shape, np.float32)
    for i, q in enumerate(batch_qualities(b)):
        u8 = np.rint(ldr[i] * 255.0).astype(np.uint8)
        if order == "bgr":
            u8 = u8[..., ::-1]
        dec = adjust_jpeg_quality(u8, q)
        if order == "bgr":
            dec = dec[..., ::-1]
        out[i] = dec.astype(np.float32) / 255.0
    return out
