"""Generator / sun-pose / sun-radiance / discriminator / VGG16 networks restated on the
torch-CPU operator set of oracle/tfsem.py.  TEST INFRASTRUCTURE; PARITY UNPINNED.

Every network is a pure function of (params: dict[str, Tensor], inputs).  Parameter
names follow the reference attribute paths (they define its checkpoint keys):
generator.py:51-90, sunpose_net.py:32-52, sunrad_net.py:30-44, discriminator.py:29-40,
vgg16.py:107-119.
"""
import math

import torch

from . import tfsem as T


# ----------------------------------------------------------------------------
# generator.py
# ----------------------------------------------------------------------------
def _conv(p, name, x, stride=1):
    """ops.conv2d instance (ops.py:4-42): vars `w`, `b`."""
    return T.conv2d(x, p[name + ".w"], p[name + ".b"], stride=stride, padding="SAME")


def _deconv(p, name, x, out_h, out_w):
    """ops.deconv2d(method='resize') instance (ops.py:44-126): vars kernel_deconv2d / bias_deconv2d."""
    return T.deconv2d_resize(x, p[name + ".kernel_deconv2d"], p[name + ".bias_deconv2d"], out_h, out_w)


def _inorm(p, name, x):
    return T.instance_norm(x, p[name + ".gamma"], p[name + ".beta"])


def res_block(p, prefix, x):
    """generator.resBlock.call (generator.py:26-35); identity is a lambda (filter_in == filter_out)."""
    c1 = _conv(p, prefix + ".conv1", x)
    a1 = T.leaky_relu(_inorm(p, prefix + ".norm1", c1), 0.1)
    c2 = _conv(p, prefix + ".conv2", a1)
    return x + _inorm(p, prefix + ".norm2", c2)


class _DAConv(torch.autograd.Function):
    """distortion_aware_ops.conv2d.call (:50-123) as a differentiable torch op: forward = the numpy restatement
    oracle/da_ops.da_conv2d, backward = da_ops.da_conv2d_grads (what the tape returns: the layer is linear in x and in
    the kernel).  The HWIO filter is the layer's [k*k*C, F] kernel reshaped."""

    @staticmethod
    def forward(ctx, x, w, b, offs):
        import numpy as np
        from . import da_ops
        c = x.shape[-1]
        k = w.shape[0]
        ctx.save_for_backward(x, w)
        ctx.offs, ctx.k = offs, k
        y = da_ops.da_conv2d(x.detach().numpy(), w.detach().numpy().reshape(k * k * c, -1), b.detach().numpy(), offs, k=k)
        return torch.from_numpy(y.astype(np.float32))

    @staticmethod
    def backward(ctx, dy):
        from . import da_ops
        x, w = ctx.saved_tensors
        k, c = ctx.k, x.shape[-1]
        dx, dk, db = da_ops.da_conv2d_grads(x.detach().numpy(), w.detach().numpy().reshape(k * k * c, -1), ctx.offs,
                                            dy.detach().numpy(), k=k)
        return torch.from_numpy(dx), torch.from_numpy(dk).view_as(w), torch.from_numpy(db), None


DA_PARTS = ("res", "sunpose", "decoders")


def da_parts(spec):
    """Which layer families run as distortion_aware_ops layers: False/None -> none; True -> the res blocks (generator.py:14,18,
    the switch earlier rounds had); "all" or a comma list / iterable of DA_PARTS: "sunpose" = sunposeLayer.conv1/conv2
    (sunpose_net.py:11,16), "decoders" = distortion_aware_ops.deconv2d (:272-542) in place of ops.deconv2d (generator.py:110-156)."""
    if not spec:
        return frozenset()
    if spec is True:
        return frozenset({"res"})
    if isinstance(spec, str):
        spec = DA_PARTS if spec == "all" else [t.strip() for t in spec.split(",") if t.strip()]
    parts = frozenset(spec)
    if not parts <= frozenset(DA_PARTS):
        raise ValueError("distortion_aware: unknown part(s) %s" % sorted(parts - frozenset(DA_PARTS)))
    return parts


def _da_conv(p, wname, bname, x, dilation_rate=1):
    """distortion_aware_ops.conv2d(filters, kernel_size=k) on x with the HWIO filter p[wname] as its [k*k*C, F] kernel."""
    from . import da_ops
    _, h, w, _ = x.shape
    k = p[wname].shape[0]
    return _DAConv.apply(x, p[wname], p[bname], da_ops.distortion(h, w, k, dilation_rate))


def _deconv_da(p, name, x, out_h, out_w):
    """distortion_aware_ops.deconv2d.call (:318-390): bilinear resize to output_imshape, then the distortion-aware conv at
    the output resolution; same variables as ops.deconv2d (kernel_deconv2d / bias_deconv2d)."""
    return _da_conv(p, name + ".kernel_deconv2d", name + ".bias_deconv2d", T.resize_bilinear(x, out_h, out_w))


def res_block_da(p, prefix, x, dilation_rate=1):
    """The res block with the two lines generator.py:14,18 keeps commented out switched on: conv1 / conv2 are
    distortion_aware_ops.conv2d(filter_out, kernel_size=3, dilation_rate) (numpy restatement oracle/da_ops.py; its
    [k*k*C, F] kernel is the HWIO filter reshaped).  Differentiable (see _DAConv)."""
    from . import da_ops
    _, h, w, c = x.shape
    offs = da_ops.distortion(h, w, 3, dilation_rate)

    def da(name, t):
        return _DAConv.apply(t, p[name + ".w"], p[name + ".b"], offs)
    c1 = da(prefix + ".conv1", x)
    a1 = T.leaky_relu(_inorm(p, prefix + ".norm1", c1), 0.1)
    c2 = da(prefix + ".conv2", a1)
    return x + _inorm(p, prefix + ".norm2", c2)


def gen_encode(p, x, distortion_aware=False):
    """generator.model.encode (generator.py:92-108)."""
    a = T.leaky_relu(_inorm(p, "norm1_d", _conv(p, "conv1_d", x, 1)), 0.1)
    a = T.leaky_relu(_inorm(p, "norm2_d", _conv(p, "conv2_d", a, 2)), 0.1)
    a = T.leaky_relu(_inorm(p, "norm3_d", _conv(p, "conv3_d", a, 2)), 0.1)
    for i in range(6):
        a = res_block_da(p, "res.%d" % i, a) if "res" in da_parts(distortion_aware) else res_block(p, "res.%d" % i, a)
    return a


def _decode(p, sfx, x, h, w, da=False):
    dc = _deconv_da if da else _deconv
    a = T.leaky_relu(_inorm(p, "norm3_" + sfx, dc(p, "conv3_" + sfx, x, h // 2, w // 2)), 0.1)
    a = T.leaky_relu(_inorm(p, "norm2_" + sfx, dc(p, "conv2_" + sfx, a, h, w)), 0.1)
    return T.leaky_relu(_conv(p, "conv1_" + sfx, a), 0.1)


def gen_sky_decode(p, x, inp, distortion_aware=False):
    """generator.model.sky_decode (generator.py:110-125)."""
    h, w = inp.shape[1], inp.shape[2]
    return torch.relu(inp + _decode(p, "f", x, h, w, "decoders" in da_parts(distortion_aware)))


def gen_sun_decode(p, x, sun_rad, distortion_aware=False):
    """generator.model.sun_decode (generator.py:127-156); sun_cam* args are unused there."""
    h, w = sun_rad.shape[1], sun_rad.shape[2]
    return torch.relu(sun_rad + _decode(p, "u", x, h, w, "decoders" in da_parts(distortion_aware)))


def _down(p, prefix, x, stride, apply_norm, training, new_stats):
    """downsampling.call (sunrad_net.py:21-28 == discriminator.py:20-27): Conv2D(k=4,'same',
    use_bias=False) -> [BatchNormalization] -> LeakyReLU() (alpha 0.3)."""
    y = T.conv2d(x, p[prefix + ".conv.kernel"], None, stride=stride, padding="SAME")
    if apply_norm:
        y, mm, mv = T.batch_norm(y, p[prefix + ".norm.gamma"], p[prefix + ".norm.beta"],
                                 p[prefix + ".norm.moving_mean"], p[prefix + ".norm.moving_variance"],
                                 training)
        if new_stats is not None:
            new_stats[prefix + ".norm.moving_mean"] = mm
            new_stats[prefix + ".norm.moving_variance"] = mv
    return T.leaky_relu(y, 0.3)


def _down_stack(p, prefix, x, training, new_stats):
    x = _down(p, prefix + "d1", x, 2, False, training, new_stats)
    x = _down(p, prefix + "d2", x, 2, True, training, new_stats)
    x = _down(p, prefix + "d3", x, 2, True, training, new_stats)
    x = _down(p, prefix + "d4", x, 1, True, training, new_stats)
    return x


def sun_rad_net(p, x, actv_map, training, new_stats=None, prefix="sun."):
    """sunrad_net.sunRadNet.call (sunrad_net.py:46-70).  `x` is the normalised cmf map."""
    d4 = _down_stack(p, prefix, actv_map, training, new_stats)
    flat = T.flatten_nhwc(d4)
    gamma = torch.sigmoid(T.dense(flat, p[prefix + "gamma.kernel"], p[prefix + "gamma.bias"])).view(-1, 1, 1, 1)
    beta = torch.sigmoid(T.dense(flat, p[prefix + "beta.kernel"], p[prefix + "beta.bias"])).view(-1, 1, 1, 1)
    eps = 1e-5
    # deltafunc_const = tf.sqrt(pi) evaluated in float32 (sunrad_net.py:35)
    const = float(torch.sqrt(torch.tensor(math.pi, dtype=torch.float32)))
    y = -torch.pow(1.0 - x, 2.0)
    y = y / (beta + eps)
    y = torch.exp(y)
    y = y * gamma
    y = y / (beta * const + eps)
    y = torch.where(y > 30000.0, torch.full_like(y, 30000.0), y)
    return y, gamma, beta


def gen_sun_rad_estimation(p, ldr, cam1, cam2, cam3, sunpose_pred, training, new_stats=None):
    """generator.model.sun_rad_estimation (generator.py:158-169).  NOTE reduce_max is over the
    WHOLE batch tensor (generator.py:160)."""
    h, w = ldr.shape[1], ldr.shape[2]
    normed = sunpose_pred / sunpose_pred.max()
    r2 = T.resize_bilinear(cam2, h, w)
    r3 = T.resize_bilinear(cam3, h, w)
    plz = torch.cat([ldr, cam1, r2, r3], dim=-1)
    rad, gamma, beta = sun_rad_net(p, normed, plz, training, new_stats)
    return rad.repeat(1, 1, 1, 3), gamma, beta


# ----------------------------------------------------------------------------
# sunpose_net.py
# ----------------------------------------------------------------------------
def _sunpose_layer(p, prefix, x, da=False):
    """sunposeLayer.call (sunpose_net.py:20-30); da: the commented-out lines :11,16 (distortion_aware_ops.conv2d with
    kernel_size = k_h) instead of ops.conv2d."""
    cv = (lambda n, t: _da_conv(p, n + ".w", n + ".b", t)) if da else (lambda n, t: _conv(p, n, t))
    a = torch.relu(_inorm(p, prefix + ".norm1", cv(prefix + ".conv1", x)))
    return torch.relu(_inorm(p, prefix + ".norm2", cv(prefix + ".conv2", a)))


def sunpose_estimation(p, x, distortion_aware=False):
    """sunpose_net.model.sunposeEstimation (sunpose_net.py:54-72).
    Returns (softmax [B,H*W], [A1,A2,A3], logits-after-relu)."""
    da = "sunpose" in da_parts(distortion_aware)
    a1 = _sunpose_layer(p, "sunlayer1", x, da)
    a2 = _sunpose_layer(p, "sunlayer2", T.maxpool2x2(a1), da)
    a3 = _sunpose_layer(p, "sunlayer3", T.maxpool2x2(a2), da)
    flat = T.flatten_nhwc(T.maxpool2x2(a3))
    f1 = torch.relu(T.dense(flat, p["fc1.kernel"], p["fc1.bias"]))
    f2 = torch.relu(T.dense(f1, p["fc2.kernel"], p["fc2.bias"]))
    return torch.softmax(f2, dim=-1), [a1, a2, a3]


def grad_cam_layer(y_c, a_k, create_graph=False):
    """grad_cam.layer (grad_cam.py:29-44): tf.gradients(y_c, A_k) (sums y_c over the batch),
    GAP over (H,W), channel-weighted sum, ReLU, expand last dim."""
    (grad,) = torch.autograd.grad(y_c.sum(), a_k, retain_graph=True, create_graph=create_graph)
    weights = grad.mean(dim=(1, 2))
    cam = torch.einsum("bc,bhwc->bhw", weights, a_k)
    return torch.relu(cam).unsqueeze(-1)


# ----------------------------------------------------------------------------
# discriminator.py
# ----------------------------------------------------------------------------
def discriminator(p, ldr, hdr, training, new_stats=None):
    """discriminator.model.call (discriminator.py:42-50): concat -> d1..d4 -> Conv2D(1,4) VALID + bias."""
    x = torch.cat([ldr, hdr], dim=-1)
    x = _down_stack(p, "", x, training, new_stats)
    return T.conv2d(x, p["out.kernel"], p["out.bias"], stride=1, padding="VALID")


# ----------------------------------------------------------------------------
# vgg16.py
# ----------------------------------------------------------------------------
VGG_MEAN = (103.939, 116.779, 123.68)
VGG_LAYERS = ("conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3")


def vgg16_pools(p, bgr):
    """Vgg16.call (vgg16.py:127-165): x*255 - mean, conv3x3+bias+relu chain, returns pool1..3."""
    mean = torch.tensor(VGG_MEAN, dtype=bgr.dtype).view(1, 1, 1, 3)
    x = bgr * 255.0 - mean

    def c(name, x):
        return torch.relu(T.conv2d(x, p[name + ".w"], p[name + ".b"], 1, "SAME"))

    x = c("conv1_2", c("conv1_1", x))
    p1 = T.maxpool2x2(x)
    x = c("conv2_2", c("conv2_1", p1))
    p2 = T.maxpool2x2(x)
    x = c("conv3_3", c("conv3_2", c("conv3_1", p2)))
    p3 = T.maxpool2x2(x)
    return p1, p2, p3
