"""Mirror of `discriminator.model` (discriminator.py:29-50) on libhdrsky."""
import torch

from . import engine, kernels as K, params as P, _lib as L


class model:
    def __init__(self, im_height=32, im_width=128, da_kernel_size=3, dilation_rate=1, seed=2, device="cuda",
                 compute=K.BF16, weights=None):
        self.compute, self.device = compute, torch.device(device)
        w = weights if weights is not None else P.init_params(P.discriminator_spec(), seed)
        self.p = engine._dev(w, self.device)
        self._repack()

    def _repack(self):
        pr = self.compute == K.BF16X3
        self._pk = {d: K.PackedConv(self.p["%s.conv.kernel" % d], pr) for d in ("d1", "d2", "d3", "d4")}
        self._pk["out"] = K.PackedConv(self.p["out.kernel"], pr)

    @property
    def variables(self):
        return self.p

    def assign(self, weights):
        for k, v in weights.items():
            self.p[k].copy_(torch.as_tensor(v))
        self._repack()

    def __call__(self, inputs, training="training"):
        """inputs = [ldr, hdr] (each [B,H,W,3]) -> patch logits [B,1,13,1] at 32x128 (no sigmoid).  training=True uses
        the batch statistics in the three BatchNorm layers and updates their moving averages."""
        ldr, hdr = inputs
        x = K.concat2(ldr, hdr)
        raw, xf = engine.down_stack(x, self._pk, self.p, "", self.compute,
                                    training=bool(training) and training != "inference")
        y, _ = K.conv2d(raw, self._pk["out"], self.p["out.bias"], same=False, xf=xf, compute=self.compute)
        return y
