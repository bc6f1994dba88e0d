"""TF / TFA / Keras operator semantics restated with torch-CPU ops (NHWC at the surface).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED at the TF boundary.

All tensors are NHWC ``torch.Tensor`` on CPU; dtype follows the input (fp32 for
parity tests, fp64 for finite-difference gradient checks).  Third-party
semantics mirrored here are listed in SURVEY.md section 8c appendix.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------
def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def same_pad(in_size, k, s):
    """TF 'SAME' padding: out = ceil(in/s); total = max((out-1)*s + k - in, 0);
    before = total // 2, after = rest (tf.nn.conv2d / Keras Conv2D)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d(x, w, b=None, stride=1, padding="SAME"):
    """``tf.nn.conv2d(x, w[kh,kw,Cin,Cout], strides=[1,s,s,1], padding)`` + bias_add.

    Follows reference ops.py:41-42 (`ops.conv2d.call`), Keras Conv2D as used in
    discriminator.py:11-13 / sunrad_net.py:12-14 and vgg16.py:32-36.
    """
    kh, kw, _, _ = w.shape
    xn = _nchw(x)
    if padding == "SAME":
        pt, pb = same_pad(x.shape[1], kh, stride)
        pl, pr = same_pad(x.shape[2], kw, stride)
        xn = F.pad(xn, (pl, pr, pt, pb))
    elif padding != "VALID":
        raise ValueError(padding)
    wt = w.permute(3, 2, 0, 1)  # HWIO -> OIHW
    y = F.conv2d(xn, wt, bias=b, stride=stride)
    return _nhwc(y)


def resize_bilinear(x, out_h, out_w):
    """``tf.image.resize(x, (out_h,out_w), BILINEAR)`` in TF2: half-pixel centres,
    antialias=False (ops.py:122, generator.py:161-162, tf_utils.py:64)."""
    y = F.interpolate(_nchw(x), size=(out_h, out_w), mode="bilinear", align_corners=False)
    return _nhwc(y)


def deconv2d_resize(x, w, b, out_h, out_w):
    """ops.deconv2d(method='resize').call: ops.py:121-124."""
    return conv2d(resize_bilinear(x, out_h, out_w), w, b, stride=1, padding="SAME")


def maxpool2x2(x):
    """tf.nn.max_pool ksize 2 stride 2 SAME on even dims (ops.py:299-300, vgg16.py:85-86)."""
    assert x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
    return _nhwc(F.max_pool2d(_nchw(x), 2, 2))


def leaky_relu(x, alpha):
    return torch.where(x >= 0, x, x * alpha)


def instance_norm(x, gamma, beta, eps=1e-3):
    """tfa.layers.InstanceNormalization() == GroupNormalization(groups=C):
    biased moments over (H,W) per (b,c); tf.nn.batch_normalization form
    ``x*inv + (beta - mean*inv)`` with ``inv = gamma*rsqrt(var+eps)``; eps=1e-3.
    Call sites: generator.py:15,19,61-85; sunpose_net.py:12,17."""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    inv = gamma.view(1, 1, 1, -1) * torch.rsqrt(var + eps)
    return x * inv + (beta.view(1, 1, 1, -1) - mean * inv)


def batch_norm(x, gamma, beta, moving_mean, moving_var, training, momentum=0.99, eps=1e-3):
    """Keras BatchNormalization() (discriminator.py:16, sunrad_net.py:17).

    training=True : normalise with batch mean / biased variance over (N,H,W);
                    returns the updated moving stats (moving_var fed with the
                    Bessel-corrected variance, as on the fused NHWC path).
    training=False: normalise with the moving stats.
    Returns (y, new_moving_mean, new_moving_var).
    """
    if training:
        n = x.shape[0] * x.shape[1] * x.shape[2]
        mean = x.mean(dim=(0, 1, 2))
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
        inv = gamma * torch.rsqrt(var + eps)
        y = x * inv + (beta - mean * inv)
        unbiased = var * (n / max(n - 1, 1))
        new_mm = moving_mean * momentum + mean.detach() * (1 - momentum)
        new_mv = moving_var * momentum + unbiased.detach() * (1 - momentum)
        return y, new_mm, new_mv
    inv = gamma * torch.rsqrt(moving_var + eps)
    y = x * inv + (beta - moving_mean * inv)
    return y, moving_mean, moving_var


def dense(x, kernel, bias):
    """Keras Dense: x @ W[in,out] + b (sunpose_net.py:48-51, sunrad_net.py:42-43)."""
    return x @ kernel + bias


def flatten_nhwc(x):
    """Keras Flatten on NHWC: (h, w, c) row-major."""
    return x.reshape(x.shape[0], -1)


# ----------------------------------------------------------------------------
# tf_utils.py live helpers
# ----------------------------------------------------------------------------
def hdr_log_compression(x, valid_dr=10.0):
    """tf_utils.py:263-271."""
    return torch.log(1.0 + valid_dr * x) / math.log(1.0 + valid_dr)


def hdr_log_decompression(x, valid_dr=10.0):
    """tf_utils.py:273-280."""
    return (torch.exp(x * math.log(1.0 + valid_dr)) - 1.0) / valid_dr


def rgb2bgr(x):
    """tf_utils.py:85-93 (channel reversal; its own inverse)."""
    return x.flip(-1)


bgr2rgb = rgb2bgr


def gaussian_kernel_1d(sigma, dtype):
    """TFA gaussian_filter2d 1-D kernel for filter_shape 3: softmax(-x^2/(2 sigma^2)), x=-1,0,1."""
    xs = torch.tensor([-1.0, 0.0, 1.0], dtype=dtype)
    return torch.softmax(-(xs ** 2) / (2.0 * sigma * sigma), dim=0)


def gaussian_filter2d_3x3(x, sigma):
    """tfa.image.gaussian_filter2d(filter_shape=(3,3), sigma, padding='REFLECT'):
    reflect-pad by 1 (no edge repeat) then depthwise VALID conv with the outer
    product of the 1-D kernels (tf_utils.py:65,69-70)."""
    k1 = gaussian_kernel_1d(sigma, x.dtype)
    k2 = torch.outer(k1, k1)
    c = x.shape[-1]
    xn = F.pad(_nchw(x), (1, 1, 1, 1), mode="reflect")
    w = k2.view(1, 1, 3, 3).repeat(c, 1, 1, 1)
    return _nhwc(F.conv2d(xn, w, groups=c))


DOG_SIGMA_BASE = 1.2489996
DOG_SIGMAS_1 = (1.2262735, 1.5450078, 1.9465878, 2.452547)
DOG_SIGMAS_2 = (1.5450078, 1.9465878, 2.452547, 3.0900156)


def dog(img):
    """tf_utils.DoG (tf_utils.py:61-73): 2x bilinear upsample, base blur, then the
    4 differences blur(base, s2_i) - blur(base, s1_i)."""
    _, h, w, _ = img.shape
    up = resize_bilinear(img, 2 * h, 2 * w)
    base = gaussian_filter2d_3x3(up, DOG_SIGMA_BASE)
    g1 = [gaussian_filter2d_3x3(base, s) for s in DOG_SIGMAS_1]
    g2 = [gaussian_filter2d_3x3(base, s) for s in DOG_SIGMAS_2]
    return [b - a for a, b in zip(g1, g2)]


def kl_divergence(y_true, y_pred, eps=1e-7):
    """tf.keras.losses.KLDivergence() (train.py:232, :305): clip both to [eps,1],
    sum_j y*log(y/yhat), mean over the batch."""
    yt = y_true.clamp(eps, 1.0)
    yp = y_pred.clamp(eps, 1.0)
    return (yt * torch.log(yt / yp)).sum(dim=-1).mean()


def rmsprop_update(w, g, ms, lr, rho=0.9, eps=1e-7):
    """Keras-2 OptimizerV2 RMSprop, momentum=0, centered=False (train.py:201-202):
    ms <- rho*ms + (1-rho)*g^2 ; w <- w - lr*g/(sqrt(ms)+eps).  Returns (w, ms)."""
    r = torch.tensor(rho, dtype=torch.float32)       # OptimizerV2 forms 1-rho in float32
    ms = r * ms + (1.0 - r) * g * g
    w = w - lr * g / (torch.sqrt(ms) + eps)
    return w, ms


def adam_update(w, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-7):
    """Keras-2 OptimizerV2 Adam (train_sun.py:191 via tf_utils.py:324), non-amsgrad:
    m <- b1*m + (1-b1)*g ; v <- b2*v + (1-b2)*g^2 ; lr_t = lr*sqrt(1-b2^t)/(1-b1^t) ;
    w <- w - lr_t*m/(sqrt(v)+eps).  `step` is the 1-based update count.  Returns (w, m, v)."""
    # OptimizerV2 keeps the hyper-parameters as float32 tensors and forms 1-beta, beta^t and lr_t in float32
    f = lambda x: torch.tensor(x, dtype=torch.float32)
    b1, b2 = f(beta1), f(beta2)
    lr_t = f(lr) * torch.sqrt(1.0 - torch.pow(b2, f(float(step)))) / (1.0 - torch.pow(b1, f(float(step))))
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    w = w - lr_t * m / (torch.sqrt(v) + eps)
    return w, m, v


def conv2d_transpose(x, w, b, out_h, out_w, stride, padding="SAME"):
    """tf.nn.conv2d_transpose(x, w[kh,kw,Cout,Cin], output_shape=[B,out_h,out_w,Cout], strides, padding) + bias
    (ops.py:116-119).  By definition the gradient of conv2d(z[B,out_h,out_w,Cout], w, strides, padding) wrt z,
    evaluated with x as the upstream gradient."""
    z = torch.zeros(x.shape[0], out_h, out_w, w.shape[2], dtype=x.dtype, requires_grad=True)
    y = conv2d(z, w, None, stride, padding)
    assert tuple(y.shape) == tuple(x.shape), (tuple(y.shape), tuple(x.shape))
    (g,) = torch.autograd.grad(y, z, x)
    return g + b if b is not None else g
