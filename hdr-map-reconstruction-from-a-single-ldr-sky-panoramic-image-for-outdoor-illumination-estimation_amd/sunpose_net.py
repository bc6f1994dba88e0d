"""Mirror of `sunpose_net.model` (sunpose_net.py:32-72) on libhdrsky's fused plan (engine.sunpose_forward)."""
from collections import OrderedDict

import torch

from . import engine, kernels as K, params as P


class model:
    def __init__(self, im_height=32, im_width=128, da_kernel_size=3, dilation_rate=1, seed=1, device="cuda",
                 compute=K.BF16, weights=None, distortion_aware=False):
        """distortion_aware: every sunposeLayer convolution is distortion_aware_ops.conv2d(filter_out, kernel_size=k_h) - the
        lines sunpose_net.py:11,16 keep commented out (same variables: an HWIO filter is the [k*k*C, F] kernel reshaped)."""
        self.distortion_aware = "sunpose" in engine.da_parts("sunpose" if distortion_aware is True else distortion_aware)
        self.im_height, self.im_width, self.fc_dim = im_height, im_width, im_height * im_width
        self.compute, self.device = compute, torch.device(device)
        w = weights if weights is not None else P.init_params(P.sunpose_spec(im_height, im_width), seed)
        self.nets = engine.Nets(None, w, device=self.device, precise=compute == K.BF16X3, im_height=im_height,
                                im_width=im_width)

    @property
    def variables(self):
        return self.nets.sun

    def assign(self, weights):
        """Copy new values into the device-resident variables and re-pack the MFMA images."""
        for k, v in weights.items():
            self.nets.sun[k].copy_(torch.as_tensor(v))
        self.nets.repack_all()

    def sunposeEstimation(self, x, training="training"):
        """-> (softmax cmf [B, H*W], [A1, A2, A3]).  InstanceNorm has no training/inference difference."""
        t = engine.sunpose_forward(self.nets, x, self.compute, "sunpose" if self.distortion_aware else False)
        cmf = t["cmf"]
        cmf._hdrsky_ctx = (self, t)       # what grad_cam.layer differentiates through
        return cmf, [t["A1"], t["A2"], t["A3"]]
