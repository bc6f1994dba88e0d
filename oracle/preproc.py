"""Host augmentation of the reference restated in numpy (train.py:42-94, tf_utils.py:191-255).
TEST INFRASTRUCTURE; PARITY UNPINNED (see oracle/__init__.py).  The JPEG round trip that follows (train.py:86-92,
`tf.image.adjust_jpeg_quality` = a libjpeg encode/decode) is restated - and pinned to libjpeg - in oracle/jpeg.py."""
import numpy as np

F32 = np.float32


def preprocessing(hdr, t, sigma_s, sigma_c, noise_s, noise_c, crf):
    """`_preprocessing` (train.py:54-94) with the random draws passed in.  hdr [B,H,W,3]; t [B]; sigma_* [B,3] (the
    reference's 0.08/6*U(0,1) and 0.005*U(0,1), shape [B,1,1,3]); noise_* [B,H,W,3] standard normals; crf [B,K].
    Returns (hdr_t, ldr) with ldr = round(255*CRF(clip(hdr_t,0,1)))/255 (tf.round: half to even)."""
    hdr = np.asarray(hdr, F32)
    b = hdr.shape[0]
    x = hdr * np.asarray(t, F32).reshape(b, 1, 1, 1)                                    # :63
    ss = np.asarray(sigma_s, F32).reshape(b, 1, 1, 3); sc = np.asarray(sigma_c, F32).reshape(b, 1, 1, 3)
    noise_s_map = ss * x                                                                # :69
    x = x + np.asarray(noise_s, F32) * noise_s_map                                      # :70-71
    x = x + sc * np.asarray(noise_c, F32)                                               # :72-73
    hdr_t = np.maximum(x, F32(0))                                                       # :74
    clipped = np.clip(hdr_t, F32(0), F32(1))                                            # :77
    ldr = apply_rf(clipped, crf)                                                        # :80
    return hdr_t.astype(F32), (np.round(ldr * F32(255.0)) / F32(255.0)).astype(F32)     # :83 (np.round: half to even)


def apply_rf(x, rf):
    """tf_utils.apply_rf / interp_1d / sample_1d (tf_utils.py:191-255): per-sample LUT, linear interpolation at
    (k-1)*x, gather indices clipped to [0, k-1]."""
    x = np.asarray(x, F32); rf = np.asarray(rf, F32)
    b, k = rf.shape
    pos = F32(k - 1) * x.reshape(b, -1)
    y0 = np.floor(pos); y1 = y0 + F32(1)
    i0 = np.clip(y0.astype(np.int32), 0, k - 1); i1 = np.clip(y1.astype(np.int32), 0, k - 1)
    v0 = np.take_along_axis(rf, i0, axis=1); v1 = np.take_along_axis(rf, i1, axis=1)
    out = (y1 - pos) * v0 + (pos - y0) * v1
    return out.reshape(x.shape).astype(F32)


def sunpose_bins(h, w):
    """tf_utils.sunpose_init (tf_utils.py:112-129) for every bin index."""
    i = np.arange(h * w, dtype=np.float64)
    row = np.floor(i / w)
    x = ((i + 1.0) - row * w - 1.0) * (360.0 / w) + (360.0 / (w * 2.0))
    y = row * (90.0 / h) + (90.0 / (2.0 * h))
    phi = y * (np.pi / 180.0); theta = (x - 180.0) * (np.pi / 180.0)
    return np.stack([np.cos(phi) * np.cos(theta), np.sin(phi), np.cos(phi) * np.sin(theta)], axis=1)


def vmf(azimuth, elevation, h, w, kappa=80.0):
    """train.py:42-52 with tf_utils.sphere2world (tf_utils.py:95-110, skydome=True)."""
    theta = (azimuth - 0.5 * w) * (2.0 * np.pi / w)
    phi = (h - elevation) * (np.pi / (h * 2))
    v = np.array([np.cos(phi) * np.cos(theta), np.sin(phi), np.cos(phi) * np.sin(theta)])
    pdf = np.exp(kappa * (sunpose_bins(h, w) @ v))
    return (pdf / pdf.sum()).astype(F32)
