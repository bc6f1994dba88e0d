"""Distortion-aware panoramic conv / deconv restated in numpy float32.
TEST INFRASTRUCTURE; PARITY UNPINNED (see oracle/__init__.py).

Follows distortion_aware_ops.py: `conv2d` (:5-270) and `deconv2d` (:272-542, the same call
preceded by a bilinear resize, :322).  The offset table is evaluated in float32 in the
reference's operation order - at row 0 float32(pi)/2 rounds above pi/2, cos(phi) < 0 and the
`ur_i[0] < 0` branch (:241-245) is taken.
"""
import numpy as np

F32 = np.float32


def make_grid(k):
    """distortion_aware_ops.py:186-196."""
    r = k // 2
    return [(x, y) for y in range(r, -r - 1, -1) for x in range(r, -r - 1, -1)]


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=F32)


def distortion(h, w, k=3, dilation_rate=1, skydome=True):
    """distortion_aware_ops.py:198-270 -> offsets [h, k*k, 2] (y, x), identical for every column
    (the reference stacks this w times to [1,h,w,k*k,2], :266-268)."""
    pi = F32(np.pi)
    n = k // 2
    middle = n * (k + 1)
    unit_w = F32(2 * np.pi) / F32(w)
    unit_h = pi / F32(h * 2 if skydome else h)
    rho = F32(np.tan(unit_w, dtype=F32) * F32(dilation_rate))
    v = np.array([0.0, 1.0, 0.0], dtype=F32)
    grid = make_grid(k)
    x = int(w * 0.5)
    res = np.zeros((h, k * k, 2), dtype=F32)
    for y in range(h):
        theta = F32((x - 0.5 * w)) * unit_w
        phi = F32(h - y) * unit_h if skydome else F32(h * 0.5 - y) * unit_h
        p_u = np.array([np.cos(phi, dtype=F32) * np.cos(theta, dtype=F32), np.sin(phi, dtype=F32),
                        np.cos(phi, dtype=F32) * np.sin(theta, dtype=F32)], dtype=F32)
        t_x = _cross(v, p_u)
        t_y = _cross(p_u, t_x)
        kk = np.zeros((k * k, 2), dtype=F32)
        for i, (gx, gy) in enumerate(grid):
            ur = (p_u + rho * (F32(gx) * t_x + F32(gy) * t_y)).astype(F32)
            if ur[0] > 0:
                theta_r = np.arctan2(ur[2], ur[0], dtype=F32)
            elif ur[0] < 0:
                if ur[2] >= 0:
                    theta_r = F32(np.arctan2(ur[2], ur[0], dtype=F32) + pi)
                else:
                    theta_r = F32(np.arctan2(ur[2], ur[0], dtype=F32) - pi)
            else:
                if ur[2] > 0:
                    theta_r = F32(pi * F32(0.5))
                elif ur[2] < 0:
                    theta_r = F32(-pi * F32(0.5))
                else:
                    raise ValueError("undefined coordinates")
            phi_r = np.arcsin(ur[1], dtype=F32)
            x_r = F32((F32(theta_r / pi) + F32(1)) * F32(0.5) * F32(w))
            if skydome:
                y_r = F32((F32(1.0) - F32(F32(2) * phi_r) / pi) * F32(h))
            else:
                y_r = F32((F32(0.5) - phi_r / pi) * F32(h))
            kk[i] = (y_r, x_r)
        res[y] = kk - kk[middle]
    return res


def _pad_amounts(size, k, s):
    """conv2d._pad_input (:125-150)."""
    same_out = (size + s - 1) // s
    valid_out = (size - k + s) // s
    if same_out == valid_out:
        return 0, 0
    p = k - 1
    return p // 2, p - p // 2


def da_conv2d(x, kernel, bias, offsets, k=3, stride=1):
    """distortion_aware_ops.conv2d.call (:50-123).  x [B,h,w,C] fp32, kernel [k*k*C, F] (row = tap*C + c),
    bias [F], offsets [h,k*k,2].  stride must be 1 (base grid / offset shapes only agree then)."""
    assert stride == 1
    x = np.asarray(x, F32)
    b, h, w, c = x.shape
    pt, pb = _pad_amounts(h, k, stride)
    pl, pr = _pad_amounts(w, k, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    in_h, in_w = xp.shape[1], xp.shape[2]
    out_h, out_w = offsets.shape[0], w
    # _get_conv_indices (:152-168): VALID patches of the padded meshgrid, tap order row-major
    oy = np.arange(out_h).reshape(out_h, 1, 1)
    ox = np.arange(out_w).reshape(1, out_w, 1)
    ty = (np.arange(k * k) // k).reshape(1, 1, k * k)
    tx = (np.arange(k * k) % k).reshape(1, 1, k * k)
    base_y = (oy + ty + 0 * ox).astype(F32)
    base_x = (ox + tx + 0 * oy).astype(F32)
    y = base_y + offsets[:, None, :, 0]
    xx = base_x + offsets[:, None, :, 1]
    y = np.clip(y, F32(0), F32(in_h - 1))
    xx = np.where(xx < 0, xx + F32(in_w), xx)
    xx = np.where(xx > in_w - 1, xx - F32(in_w), xx)
    y0 = np.floor(y).astype(np.int32)
    x0 = np.floor(xx).astype(np.int32)
    y1, x1 = y0 + 1, x0 + 1
    y0 = np.clip(y0, 0, in_h - 1)
    y1 = np.clip(y1, 0, in_h - 1)
    x0_w, x1_w = x0, x1
    x0 = np.where(x0 < 0, x0 + in_w, x0); x1 = np.where(x1 < 0, x1 + in_w, x1)
    x0 = np.where(x0 > in_w - 1, x0 - in_w, x0); x1 = np.where(x1 > in_w - 1, x1 - in_w, x1)
    p0 = xp[:, y0, x0]; p1 = xp[:, y0, x1]; p2 = xp[:, y1, x0]; p3 = xp[:, y1, x1]  # [B,h,w,k2,C]
    y0f, y1f = y0.astype(F32), y1.astype(F32)
    x0f, x1f = x0_w.astype(F32), x1_w.astype(F32)
    w0 = ((y1f - y) * (x1f - xx))[None, ..., None]
    w1 = ((y1f - y) * (xx - x0f))[None, ..., None]
    w2 = ((y - y0f) * (x1f - xx))[None, ..., None]
    w3 = ((y - y0f) * (xx - x0f))[None, ..., None]
    pixels = (w0 * p0 + w1 * p1 + w2 * p2 + w3 * p3).astype(F32)
    pixels = pixels.reshape(b, out_h * out_w, k * k * c)
    out = pixels @ np.asarray(kernel, F32) + np.asarray(bias, F32)
    return out.reshape(b, out_h, out_w, -1).astype(F32)


def resize_bilinear(x, oh, ow):
    """tf.image.resize BILINEAR, half-pixel centres, explicit numpy (independent of torch)."""
    x = np.asarray(x, F32)
    _, h, w, _ = x.shape

    def axis(n_in, n_out):
        src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        f = np.floor(src)
        lo = np.maximum(f, 0).astype(np.int64)
        hi = np.minimum(np.ceil(src), n_in - 1).astype(np.int64)
        return lo, hi, (src - f).astype(F32)

    ylo, yhi, ty = axis(h, oh)
    xlo, xhi, tx = axis(w, ow)
    top = x[:, ylo][:, :, xlo] + (x[:, ylo][:, :, xhi] - x[:, ylo][:, :, xlo]) * tx[None, None, :, None]
    bot = x[:, yhi][:, :, xlo] + (x[:, yhi][:, :, xhi] - x[:, yhi][:, :, xlo]) * tx[None, None, :, None]
    return (top + (bot - top) * ty[None, :, None, None]).astype(F32)


def da_deconv2d(x, kernel, bias, offsets, out_h, out_w, k=3):
    """distortion_aware_ops.deconv2d.call (:321-395): bilinear resize to output_imshape, then the
    conv2d call with strides forced to 1 (:323); offsets are built for (out_h, out_w) (:315)."""
    return da_conv2d(resize_bilinear(x, out_h, out_w), kernel, bias, offsets, k=k, stride=1)


def da_conv2d_grads(x, kernel, offsets, dy, k=3):
    """Gradients of y = da_conv2d(x, kernel, bias, offsets) for an upstream dy (what tf.GradientTape returns for
    distortion_aware_ops.conv2d.call :50-123): the layer is linear in x and in the kernel, y = G(x) W + b with G the
    bilinear gather, so dW = G(x)^T dy, db = sum dy and dx = G^T (dy W^T) - the transpose of the gather is a scatter-add
    with the same corner indices and weights (the zero padding receives, and drops, its share).
    Returns (dx [B,h,w,C], dkernel [k*k*C, F], dbias [F])."""
    x = np.asarray(x, F32); dy = np.asarray(dy, F32); kernel = np.asarray(kernel, F32)
    b, h, w, c = x.shape
    pt, pb = _pad_amounts(h, k, 1)
    pl, pr = _pad_amounts(w, k, 1)
    in_h, in_w = h + pt + pb, w + pl + pr
    oy = np.arange(h).reshape(h, 1, 1); ox = np.arange(w).reshape(1, w, 1)
    ty = (np.arange(k * k) // k).reshape(1, 1, k * k); tx = (np.arange(k * k) % k).reshape(1, 1, k * k)
    y = (oy + ty + 0 * ox).astype(F32) + offsets[:, None, :, 0]
    xx = (ox + tx + 0 * oy).astype(F32) + offsets[:, None, :, 1]
    y = np.clip(y, F32(0), F32(in_h - 1))
    xx = np.where(xx < 0, xx + F32(in_w), xx)
    xx = np.where(xx > in_w - 1, xx - F32(in_w), xx)
    y0 = np.floor(y).astype(np.int32); x0 = np.floor(xx).astype(np.int32)
    y1, x1 = y0 + 1, x0 + 1
    y0 = np.clip(y0, 0, in_h - 1); y1 = np.clip(y1, 0, in_h - 1)
    x0_w, x1_w = x0, x1
    x0 = np.where(x0 < 0, x0 + in_w, x0); x1 = np.where(x1 < 0, x1 + in_w, x1)
    x0 = np.where(x0 > in_w - 1, x0 - in_w, x0); x1 = np.where(x1 > in_w - 1, x1 - in_w, x1)
    y0f, y1f, x0f, x1f = y0.astype(F32), y1.astype(F32), x0_w.astype(F32), x1_w.astype(F32)
    ws = [(y1f - y) * (x1f - xx), (y1f - y) * (xx - x0f), (y - y0f) * (x1f - xx), (y - y0f) * (xx - x0f)]
    corners = [(y0, x0), (y0, x1), (y1, x0), (y1, x1)]
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0))).astype(np.float64)
    G = sum(wk[None, ..., None].astype(np.float64) * xp[:, yk, xk] for wk, (yk, xk) in zip(ws, corners))   # [B,h,w,k2,C]
    dyf = dy.reshape(b * h * w, -1).astype(np.float64)
    dkernel = G.reshape(b * h * w, k * k * c).T @ dyf
    dbias = dyf.sum(axis=0)
    dG = (dyf @ kernel.astype(np.float64).T).reshape(b, h, w, k * k, c)
    dxp = np.zeros_like(xp)
    for wk, (yk, xk) in zip(ws, corners):
        for bi in range(b):
            np.add.at(dxp[bi], (yk, xk), wk[..., None].astype(np.float64) * dG[bi])
    dx = dxp[:, pt:pt + h, pl:pl + w]
    return dx.astype(F32), dkernel.astype(F32), dbias.astype(F32)
