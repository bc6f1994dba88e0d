"""Mirror of `sunrad_net.sunRadNet` (sunrad_net.py:30-70) on libhdrsky."""
from collections import OrderedDict

import torch

from . import engine, kernels as K, params as P


class sunRadNet:
    """call(x, actv_map, training): x = normalised sun-position map [B,H,W,1] (peak 1), actv_map = concat(LDR, 3 CAMs)
    [B,H,W,6] -> (radiance [B,H,W,1], gamma [B,1,1,1], beta [B,1,1,1])."""

    def __init__(self, epsilon=1e-5, pi=None, variables=None, compute=K.BF16):
        if abs(epsilon - 1e-5) > 1e-12:
            raise ValueError("epsilon is fixed at the reference's 1e-5 in the kernel")
        self.compute = compute
        self.p = variables            # OrderedDict of d1..d4 / gamma / beta tensors (names of params.generator_spec 'sun.*')
        self._pk = None

    def _pack(self):
        self._pk = {d: K.PackedConv(self.p["%s.conv.kernel" % d], self.compute == K.BF16X3) for d in ("d1", "d2", "d3", "d4")}

    def __call__(self, x, actv_map, training="training"):
        if self._pk is None:
            self._pack()
        B, H, W, _ = actv_map.shape
        raw, xf = engine.down_stack(actv_map, self._pk, self.p, "", self.compute, training=bool(training) and training != "inference")
        part = K.dense_heads(raw, xf.scale, xf.shift, 0.3, self.p["gamma.kernel"], self.p["beta.kernel"])
        one = torch.ones(1, dtype=torch.float32, device=x.device).view(torch.int32)   # x is already divided by its max
        lin, _, gamma, beta = K.sun_rad(x.reshape(B, H * W).contiguous(), one, part, self.p["gamma.bias"], self.p["beta.bias"], H, W)
        return lin[..., :1].contiguous(), gamma, beta
