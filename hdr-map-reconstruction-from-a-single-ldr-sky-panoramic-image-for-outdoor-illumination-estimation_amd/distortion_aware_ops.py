"""Mirror of the reference's `distortion_aware_ops` layer API on libhdrsky (distortion_aware_ops.py:5-542).

    conv2d(filters, kernel_size=3, strides=1, padding='VAILD', dilation_rate=1, kernel_initializer='glorot_uniform',
           bias_initializer='zeros', skydome=True)                                   reference :7-25
    deconv2d(filters, kernel_size=3, strides=1, output_imshape=[], ...)              reference :274-296

Variables keep the reference's names/shapes: ``kernel [k*k*Cin, filters]`` (row = tap*Cin + c), ``bias [filters]``,
attribute ``offset [1, h, w, k*k, 2]``.  Layers are built lazily on the first call, like Keras' ``build``.
strides must be 1 (the reference's base grid / offset shapes only agree then).
"""
import numpy as np
import torch

from . import kernels as K


class conv2d:
    def __init__(self, filters, kernel_size=3, strides=1, padding="VAILD", dilation_rate=1,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", skydome=True, seed=0, compute=K.BF16):
        if strides != 1:
            raise ValueError("distortion-aware conv supports strides == 1 only")
        self.filters, self.kernel_size, self.dilation_rate, self.skydome = filters, kernel_size, dilation_rate, skydome
        self.seed, self.compute, self.built = seed, compute, False

    def build(self, input_shape, device):
        _, h, w, c = input_shape
        k = self.kernel_size
        rng = np.random.default_rng(self.seed)
        lim = np.sqrt(6.0 / (k * k * c + self.filters))      # glorot_uniform on the 2-D [k*k*C, F] kernel
        self.kernel = torch.from_numpy(rng.uniform(-lim, lim, (k * k * c, self.filters)).astype(np.float32)).to(device)
        self.bias = torch.zeros(self.filters, dtype=torch.float32, device=device)
        self._offs_np = K.da_offsets(h, w, k, self.dilation_rate, self.skydome)
        self._offs = K.da_offsets_device(h, w, k, self.dilation_rate, self.skydome, device)
        self.offset = np.broadcast_to(self._offs_np[None, :, None], (1, h, w, k * k, 2))   # reference attribute
        self._cin = c
        self.repack()
        self.built = True

    def repack(self):
        k = self.kernel_size
        self._pw = K.PackedConv(self.kernel.view(k, k, self._cin, self.filters), precise=True)

    def __call__(self, inputs):
        if not self.built:
            self.build(tuple(inputs.shape), inputs.device)
        return K.da_conv2d(inputs, self._pw, self.bias, self._offs, self.compute)

    def backward(self, inputs, dy, want_dx=True):
        """What a tape returns for this layer: (d inputs, d kernel [k*k*Cin, filters], d bias)."""
        _, h, w, _ = inputs.shape
        table = K.da_transpose_table(h, w, self.kernel_size, self.dilation_rate, self.skydome, inputs.device) if want_dx else None
        return K.da_conv2d_bwd(inputs, dy, self.kernel, self._offs, self.kernel_size, self.compute, want_dx, table=table)


class deconv2d(conv2d):
    """Resize-deconv: tf.image.resize(BILINEAR) to output_imshape (2x), then the distortion-aware conv (:321-395)."""

    def __init__(self, filters, kernel_size=3, strides=1, output_imshape=(), **kw):
        super().__init__(filters, kernel_size, 1, **kw)
        self.output_imshape = tuple(int(v) for v in output_imshape)

    def __call__(self, inputs):
        b, h, w, c = inputs.shape
        if self.output_imshape != (2 * h, 2 * w):
            raise ValueError("only the 2x resize of the live configuration is built (got %s from %s)" %
                             (self.output_imshape, (h, w)))
        up = K.up2x(inputs)
        if not self.built:
            self.build(tuple(up.shape), inputs.device)
        return K.da_conv2d(up, self._pw, self.bias, self._offs, self.compute)

    def backward(self, inputs, dy, want_dx=True):
        dup, dk, db = super().backward(K.up2x(inputs), dy, want_dx)
        return (K.up2x_bwd(dup) if want_dx else None), dk, db
