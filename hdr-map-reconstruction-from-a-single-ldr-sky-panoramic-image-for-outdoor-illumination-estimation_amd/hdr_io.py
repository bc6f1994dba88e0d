"""Radiance RGBE (.hdr) reader / writer - replaces cv2.imread / cv2.imwrite of the reference (inference.py:142,156,
utils.py:61-84).  Arrays are float32 [H,W,3] in BGR order (OpenCV convention, which the reference's tensors follow);
the file stores RGB.  Flat (non run-length) scanlines are written; the reader handles flat and new-style RLE."""
import numpy as np


def float_to_rgbe(rgb):
    rgb = np.maximum(np.asarray(rgb, np.float32), 0.0)
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    nz = m > 1e-32
    mant, exp = np.frexp(m[nz])
    scale = (mant * 256.0 / m[nz])[..., None]
    out[nz, :3] = np.clip(rgb[nz] * scale, 0, 255).astype(np.uint8)
    out[nz, 3] = (exp + 128).astype(np.uint8)
    return out


def rgbe_to_float(rgbe):
    e = rgbe[..., 3].astype(np.int32)
    f = np.where(e > 0, np.ldexp(1.0, e - (128 + 8)), 0.0).astype(np.float32)
    return rgbe[..., :3].astype(np.float32) * f[..., None]


def write_hdr(path, bgr):
    bgr = np.asarray(bgr, np.float32)
    h, w, _ = bgr.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(("-Y %d +X %d\n" % (h, w)).encode())
        f.write(float_to_rgbe(bgr[..., ::-1]).tobytes())


def read_hdr(path):
    with open(path, "rb") as f:
        data = f.read()
    pos = data.index(b"\n\n") + 2
    eol = data.index(b"\n", pos)
    dims = data[pos:eol].split()
    h, w = int(dims[1]), int(dims[3])
    buf = np.frombuffer(data, np.uint8, offset=eol + 1)
    if buf.size == h * w * 4 and not (w >= 8 and buf[0] == 2 and buf[1] == 2):
        rgbe = buf.reshape(h, w, 4)
    else:   # new-style RLE: per scanline 2,2,hi(w),lo(w) then the four channels run-length coded
        rgbe = np.zeros((h, w, 4), np.uint8)
        p = 0
        for y in range(h):
            assert buf[p] == 2 and buf[p + 1] == 2 and (int(buf[p + 2]) << 8 | int(buf[p + 3])) == w
            p += 4
            for c in range(4):
                x = 0
                while x < w:
                    n = int(buf[p]); p += 1
                    if n > 128:
                        n -= 128
                        rgbe[y, x:x + n, c] = buf[p]; p += 1
                    else:
                        rgbe[y, x:x + n, c] = buf[p:p + n]; p += n
                    x += n
    return rgbe_to_float(rgbe)[..., ::-1].copy()
