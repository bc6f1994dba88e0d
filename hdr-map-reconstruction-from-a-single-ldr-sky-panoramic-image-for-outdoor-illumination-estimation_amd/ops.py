"""Mirror of the reference's live `ops` layers on libhdrsky (ops.py:4-126, 287-300, 324-329).

    conv2d(output_channels, strides, k_h, k_w, padding="SAME", ...)        variables  w [kh,kw,Cin,Cout], b [Cout]
    deconv2d(output_channels, output_imshape, k_h, k_w, strides=1, padding="SAME", method='resize' | 'upsample', ...)
                                                                           variables  kernel_deconv2d, bias_deconv2d
    maxpool2d(kernel_size, strides=None, padding="SAME")   relu()

Layers are built lazily on the first call (Keras `build`), hold fp32 masters on the GPU and a packed bf16 MFMA
image of the kernel; `assign()` replaces the variables and re-packs.  NHWC float32 CUDA tensors in and out.
Stand-alone layers run un-fused (one conv launch, bias in its epilogue); the fused plan of the whole network is
`engine.py` / `trainer.py`.  Dead layers of the reference (fc2d, dfc2d, batch_normalization, avgpool2d, elu, tanh,
sigmoid, dropout: never instantiated by a live model) are not built.
"""
import numpy as np
import torch

from . import kernels as K
from .params import _glorot

_INITS = {"glorot_uniform": _glorot, "zeros": lambda rng, s: np.zeros(s, np.float32)}


def _init(kind, rng, shape):
    if kind not in _INITS:
        raise ValueError("unsupported initializer %r (live models use glorot_uniform / zeros)" % (kind,))
    return _INITS[kind](rng, shape)


class conv2d:
    WNAME, BNAME = "w", "b"

    def __init__(self, output_channels, strides, k_h, k_w, padding="SAME", kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", seed=0, compute=K.BF16, precise=False):
        if padding not in ("SAME", "VALID"):
            raise ValueError("padding must be 'SAME' or 'VALID'")
        self.output_channels, self.strides, self.k_h, self.k_w, self.padding = output_channels, strides, k_h, k_w, padding
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
        self.seed, self.compute, self.precise, self.built = seed, compute, precise or compute == K.BF16X3, False

    def build(self, input_shape, device):
        cin = int(input_shape[-1])
        rng = np.random.default_rng(self.seed)
        w = _init(self.kernel_initializer, rng, (self.k_h, self.k_w, cin, self.output_channels))
        b = _init(self.bias_initializer, rng, (self.output_channels,))
        self.assign(torch.from_numpy(w).to(device), torch.from_numpy(b).to(device))

    def assign(self, w, b=None):
        w = torch.as_tensor(w, dtype=torch.float32).contiguous()
        if tuple(w.shape[:2]) != (self.k_h, self.k_w) or w.shape[3] != self.output_channels:
            raise ValueError("kernel shape %s does not match the layer" % (tuple(w.shape),))
        setattr(self, self.WNAME, w)
        if b is not None:
            setattr(self, self.BNAME, torch.as_tensor(b, dtype=torch.float32, device=w.device).contiguous())
        self._pw = K.PackedConv(w, self.precise)
        self.built = True

    @property
    def variables(self):
        return {self.WNAME: getattr(self, self.WNAME), self.BNAME: getattr(self, self.BNAME)}

    def __call__(self, x):
        if not self.built:
            self.build(tuple(x.shape), x.device)
        y, _ = K.conv2d(x, self._pw, getattr(self, self.BNAME), stride=self.strides, same=self.padding == "SAME",
                        compute=self.compute)
        return y


class deconv2d(conv2d):
    """method='resize': tf.image.resize(BILINEAR, half-pixel centres) to output_imshape, then a stride-1 conv
    (ops.py:90-109, 121-124); the resize is fused into the conv's operand load and the live model only ever doubles
    the resolution, which is what the kernel implements (identity resize = plain conv).
    method='upsample': tf.nn.conv2d_transpose with strides = output_height // input_height (ops.py:69-88, 116-119),
    i.e. the data gradient of a strided conv: it runs on the same zero-stuffed-input kernel path the training step
    uses for its stride-2 data gradients.  Kernel variable [kh, kw, output_channels, Cin] as in the reference."""
    WNAME, BNAME = "kernel_deconv2d", "bias_deconv2d"

    def __init__(self, output_channels, output_imshape, k_h, k_w, strides=1, padding="SAME", method="resize", **kw):
        if method not in ("resize", "upsample"):
            raise ValueError("method must be 'resize' or 'upsample'")
        super().__init__(output_channels, 1, k_h, k_w, padding, **kw)
        self.method = method
        self.output_imshape = tuple(int(v) for v in output_imshape)

    def build(self, input_shape, device):
        if self.method == "resize":
            return super().build(input_shape, device)
        cin = int(input_shape[-1])
        rng = np.random.default_rng(self.seed)
        w = _init(self.kernel_initializer, rng, (self.k_h, self.k_w, self.output_channels, cin))
        b = _init(self.bias_initializer, rng, (self.output_channels,))
        self.assign(torch.from_numpy(w).to(device), torch.from_numpy(b).to(device))

    def assign(self, w, b=None):
        if self.method == "resize":
            return super().assign(w, b)
        w = torch.as_tensor(w, dtype=torch.float32).contiguous()
        if tuple(w.shape[:3]) != (self.k_h, self.k_w, self.output_channels):
            raise ValueError("kernel shape %s does not match the layer" % (tuple(w.shape),))
        self.kernel_deconv2d = w
        if b is not None:
            self.bias_deconv2d = torch.as_tensor(b, dtype=torch.float32, device=w.device).contiguous()
        self._pw = K.PackedConv(w, self.precise, transpose_flip=True)
        self.built = True

    def __call__(self, x):
        B, h, w, cin = x.shape
        if not self.built:
            self.build(tuple(x.shape), x.device)
        same = self.padding == "SAME"
        if self.method == "upsample":
            oh, ow = self.output_imshape
            stride = oh // h
            if stride not in (1, 2):
                raise ValueError("conv2d_transpose stride %d: only 1 and 2 are built" % stride)
            fwd = K.conv_desc(B, oh, ow, self.output_channels, cin, self.k_h, self.k_w, stride, same, 1)
            if (fwd.Ho, fwd.Wo) != (h, w):
                raise ValueError("output_imshape %s is not consistent with input %s, stride %d, padding %s" %
                                 ((oh, ow), (h, w), stride, self.padding))
            y, _ = K.conv2d(x, self._pw, self.bias_deconv2d, desc=K.conv_dgrad_desc(fwd), compute=self.compute)
            return y
        if self.output_imshape == (2 * h, 2 * w):
            up = 2
        elif self.output_imshape == (h, w):
            up = 1
        else:
            raise ValueError("resize %s -> %s: only 1x and 2x are built" % ((h, w), self.output_imshape))
        y, _ = K.conv2d(x, self._pw, self.bias_deconv2d, stride=1, same=same, upsample=up, compute=self.compute)
        return y


class maxpool2d:
    def __init__(self, kernel_size, strides=None, padding="SAME"):
        strides = kernel_size if strides is None else strides
        if kernel_size != 2 or strides != 2:
            raise ValueError("only the 2x2 / stride-2 pool of the live models is built")
        self.kernel_size, self.strides, self.padding = kernel_size, strides, padding

    def __call__(self, x):
        if (x.shape[1] | x.shape[2]) & 1:
            raise ValueError("odd spatial size: SAME padding of the pool is not built")
        return K.maxpool(x)


class relu:
    def __call__(self, x):
        return K.leaky_relu(x, 0.0)
