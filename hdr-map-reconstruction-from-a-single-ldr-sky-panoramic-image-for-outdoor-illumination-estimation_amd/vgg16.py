"""Mirror of `vgg16.Vgg16` (vgg16.py:88-165) on libhdrsky: frozen conv1_1 .. conv3_3, returns (pool1, pool2, pool3)."""
import torch

from . import engine, kernels as K, params as P

BLOCKS = (("conv1_1", "conv1_2"), ("conv2_1", "conv2_2"), ("conv3_1", "conv3_2", "conv3_3"))


class Vgg16:
    def __init__(self, vgg16_npy_path=None, VGG_MEAN=(103.939, 116.779, 123.68), weights=None, device="cuda",
                 compute=K.BF16):
        if tuple(round(float(v), 3) for v in VGG_MEAN) != (103.939, 116.779, 123.68):
            raise ValueError("the BGR mean is fixed in the pre-processing kernel")
        if weights is None:
            if vgg16_npy_path is None:
                raise ValueError("vgg16_npy_path is required (or pass weights=params.init_params(params.vgg_spec(), seed) "
                                 "for synthetic weights)")
            weights = P.load_vgg_npy(vgg16_npy_path)
        self.compute = compute
        self.p = engine._dev(weights, torch.device(device))
        self._pk = {n: K.PackedConv(self.p[n + ".w"], compute == K.BF16X3) for n, _, _ in P.VGG_CHANNELS}

    def __call__(self, bgr01):
        """bgr01: gamma-domain BGR in [0,1] -> *255 - mean (vgg16.py:133-141) -> three pooled feature maps."""
        x = K.vgg_pre(bgr01)
        pools = []
        # single-product mode: the chain's (ReLU-only, hence final) activations are stored as bf16 - what the next conv
        # rounds its operand to anyway, so its result is bit-identical; the pooled features come back in fp32
        b16 = self.compute == K.BF16
        for blk in BLOCKS:
            for name in blk:
                x, _ = K.conv2d(x, self._pk[name], self.p[name + ".b"], out_slope=0.0, compute=self.compute, out_bf16=b16)
            if b16:
                p32, x = K.maxpool(x, want_bf16=blk is not BLOCKS[-1])
                pools.append(p32)
            else:
                x = K.maxpool(x)
                pools.append(x)
        return tuple(pools)
