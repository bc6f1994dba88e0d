"""Parameter inventories and Keras-equivalent initialisers for the four networks.

Names follow the reference attribute paths, which are its checkpoint keys
(generator.py:51-90, sunpose_net.py:32-52, sunrad_net.py:30-44,
discriminator.py:29-40, vgg16.py:107-119).  Host-side numpy only: weights are
created on the CPU with a seeded ``numpy.random.Generator`` and uploaded once.

Initialisers restated (Keras defaults the reference relies on):
  glorot_uniform : U(-l, l), l = sqrt(6/(fan_in+fan_out)); conv [kh,kw,Cin,Cout] ->
                   fan_in = kh*kw*Cin, fan_out = kh*kw*Cout; 2-D [in,out] -> (in, out)
                   (ops.py:30-34, Keras Dense in sunpose_net.py:48-51, DA kernel
                   distortion_aware_ops.py:32-37)
  normal002      : N(0, 0.02)  (discriminator.py:12, sunrad_net.py:13)
  zeros / ones   : biases, IN/BN beta / gamma, BN moving stats (0 / 1)
"""
from collections import OrderedDict

import numpy as np


def _glorot(rng, shape):
    if len(shape) == 4:
        rf = shape[0] * shape[1]
        fan_in, fan_out = rf * shape[2], rf * shape[3]
    else:
        fan_in, fan_out = shape[0], shape[1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _normal002(rng, shape):
    return (rng.standard_normal(size=shape) * 0.02).astype(np.float32)


def _he(rng, shape):
    fan_in = shape[0] * shape[1] * shape[2]
    return (rng.standard_normal(size=shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)


_INIT = {
    "glorot": _glorot,
    "normal002": _normal002,
    "he": _he,
    "zeros": lambda rng, s: np.zeros(s, np.float32),
    "ones": lambda rng, s: np.ones(s, np.float32),
}


def _conv(spec, name, kh, kw, cin, cout):
    spec[name + ".w"] = ((kh, kw, cin, cout), "glorot")
    spec[name + ".b"] = ((cout,), "zeros")


def _deconv(spec, name, kh, kw, cin, cout):
    spec[name + ".kernel_deconv2d"] = ((kh, kw, cin, cout), "glorot")
    spec[name + ".bias_deconv2d"] = ((cout,), "zeros")


def _inorm(spec, name, c):
    spec[name + ".gamma"] = ((c,), "ones")
    spec[name + ".beta"] = ((c,), "zeros")


def _down_stack(spec, prefix, cin):
    for name, ci, co, norm in (("d1", cin, 64, False), ("d2", 64, 128, True),
                               ("d3", 128, 256, True), ("d4", 256, 512, True)):
        spec[prefix + name + ".conv.kernel"] = ((4, 4, ci, co), "normal002")
        if norm:
            spec[prefix + name + ".norm.gamma"] = ((co,), "ones")
            spec[prefix + name + ".norm.beta"] = ((co,), "zeros")
            spec[prefix + name + ".norm.moving_mean"] = ((co,), "zeros")
            spec[prefix + name + ".norm.moving_variance"] = ((co,), "ones")


def generator_spec(im_height=32, im_width=128):
    """generator.model.__init__ (generator.py:52-90) incl. sunRadNet (sunrad_net.py:31-44)."""
    s = OrderedDict()
    _conv(s, "conv1_d", 7, 7, 3, 32); _inorm(s, "norm1_d", 32)
    _conv(s, "conv2_d", 3, 3, 32, 64); _inorm(s, "norm2_d", 64)
    _conv(s, "conv3_d", 3, 3, 64, 128); _inorm(s, "norm3_d", 128)
    for i in range(6):
        _conv(s, "res.%d.conv1" % i, 3, 3, 128, 128); _inorm(s, "res.%d.norm1" % i, 128)
        _conv(s, "res.%d.conv2" % i, 3, 3, 128, 128); _inorm(s, "res.%d.norm2" % i, 128)
    for sfx in ("f", "u"):
        _deconv(s, "conv3_" + sfx, 3, 3, 128, 64); _inorm(s, "norm3_" + sfx, 64)
        _deconv(s, "conv2_" + sfx, 3, 3, 64, 32); _inorm(s, "norm2_" + sfx, 32)
        _conv(s, "conv1_" + sfx, 7, 7, 32, 3)
    _down_stack(s, "sun.", 6)
    flat = (im_height // 8) * (im_width // 8) * 512
    for head in ("gamma", "beta"):
        s["sun.%s.kernel" % head] = ((flat, 1), "glorot")
        s["sun.%s.bias" % head] = ((1,), "zeros")
    return s


def sunpose_spec(im_height=32, im_width=128):
    """sunpose_net.model.__init__ (sunpose_net.py:33-52)."""
    s = OrderedDict()
    for name, k, cin, cout in (("sunlayer1", 7, 3, 32), ("sunlayer2", 3, 32, 64), ("sunlayer3", 3, 64, 128)):
        _conv(s, name + ".conv1", k, k, cin, cout); _inorm(s, name + ".norm1", cout)
        _conv(s, name + ".conv2", k, k, cout, cout); _inorm(s, name + ".norm2", cout)
    fc_dim = im_height * im_width
    flat = (im_height // 8) * (im_width // 8) * 128
    s["fc1.kernel"] = ((flat, fc_dim), "glorot"); s["fc1.bias"] = ((fc_dim,), "zeros")
    s["fc2.kernel"] = ((fc_dim, fc_dim), "glorot"); s["fc2.bias"] = ((fc_dim,), "zeros")
    return s


def discriminator_spec():
    """discriminator.model.__init__ (discriminator.py:30-40)."""
    s = OrderedDict()
    _down_stack(s, "", 6)
    s["out.kernel"] = ((4, 4, 512, 1), "normal002")
    s["out.bias"] = ((1,), "zeros")
    return s


VGG_CHANNELS = (("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128),
                ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256))


def vgg_spec():
    """Vgg16.__init__ (vgg16.py:107-119): conv1_1 .. conv3_3, frozen."""
    s = OrderedDict()
    for name, cin, cout in VGG_CHANNELS:
        s[name + ".w"] = ((3, 3, cin, cout), "he")
        s[name + ".b"] = ((cout,), "zeros")
    return s


def init_params(spec, seed):
    """Materialise a spec as ``OrderedDict[name -> np.float32 array]`` (seeded)."""
    rng = np.random.default_rng(seed)
    return OrderedDict((k, _INIT[kind](rng, shape)) for k, (shape, kind) in spec.items())


def is_trainable(name):
    """BN moving statistics are the only non-trainable variables of gen/sun/dis."""
    return "moving_" not in name


def load_vgg_npy(path):
    """vgg16.py:99: ``np.load(path, encoding='latin1', allow_pickle=True).item()`` ->
    dict name -> [W[3,3,Cin,Cout], b[Cout]]."""
    d = np.load(path, encoding="latin1", allow_pickle=True).item()
    out = OrderedDict()
    for name, _, _ in VGG_CHANNELS:
        out[name + ".w"] = np.asarray(d[name][0], np.float32)
        out[name + ".b"] = np.asarray(d[name][1], np.float32)
    return out


def count_params(spec, trainable_only=True):
    return sum(int(np.prod(shape)) for k, (shape, _) in spec.items() if is_trainable(k) or not trainable_only)
