"""Mirror of `generator.model` (generator.py:51-175) on libhdrsky's fused plan (engine.py)."""
from collections import OrderedDict

import torch

from . import engine, kernels as K, params as P
from .sunrad_net import sunRadNet


class model:
    def __init__(self, batch_size=32, im_height=32, im_width=128, da_kernel_size=3, dilation_rate=1, seed=0,
                 device="cuda", compute=K.BF16, weights=None, distortion_aware=False):
        """distortion_aware (engine.da_parts): True / "res" builds the res blocks from distortion_aware_ops.conv2d, the
        variant generator.py:14,18 keeps commented out (da_kernel_size must be 3, the res-block filter size); "decoders"
        makes the two resize-deconvolutions of each decoder distortion_aware_ops.deconv2d (:272-542)."""
        self.da_parts = engine.da_parts(distortion_aware)
        if self.da_parts and da_kernel_size != 3:
            raise ValueError("the res blocks / decoder deconvolutions are 3x3")
        self.distortion_aware, self.dilation_rate = "res" in self.da_parts, dilation_rate
        self.im_height, self.im_width, self.fc_dim = im_height, im_width, im_height * im_width
        self.compute, self.device = compute, torch.device(device)
        w = weights if weights is not None else P.init_params(P.generator_spec(im_height, im_width), seed)
        self.nets = engine.Nets(w, None, device=self.device, precise=compute == K.BF16X3, im_height=im_height,
                                im_width=im_width)
        self.sun = sunRadNet(variables=OrderedDict((k[4:], v) for k, v in self.nets.gen.items() if k.startswith("sun.")),
                             compute=compute)

    @property
    def variables(self):
        return self.nets.gen

    def assign(self, weights):
        for k, v in weights.items():
            self.nets.gen[k].copy_(torch.as_tensor(v))
        self.nets.repack_all()
        self.sun._pk = None

    @staticmethod
    def _training(flag):
        return bool(flag) and flag != "inference"

    def encode(self, x, training="training"):
        return engine.encode(self.nets, x, self.compute, distortion_aware=self.distortion_aware,
                             dilation_rate=self.dilation_rate)

    def sky_decode(self, x, _input, training="training"):
        return engine.decode(self.nets, x, "f", _input, self.compute, self.da_parts)

    def sun_decode(self, x, sun_cam1, sun_cam2, sun_cam3, sun_rad, training="training"):
        """The CAM arguments are unused, as in the reference (generator.py:130-149 commented skips)."""
        return engine.decode(self.nets, x, "u", sun_rad, self.compute, self.da_parts)

    def sun_rad_estimation(self, jpeg_img_float, sun_cam1, sun_cam2, sun_cam3, sunpose_pred, training="training"):
        """-> (sun radiance tiled to 3 channels (linear), gamma, beta).  The normalisation divides by the maximum over
        the WHOLE batch tensor (generator.py:160)."""
        B = jpeg_img_float.shape[0]
        cmf = sunpose_pred.reshape(B, -1).contiguous()
        t = {"cmf": cmf, "gmax": cmf.max().reshape(1).view(torch.int32)}
        lin, _, gamma, beta = engine.sun_rad_estimation(self.nets, jpeg_img_float, (sun_cam1, sun_cam2, sun_cam3), t,
                                                        self.compute, training=self._training(training))
        return lin, gamma, beta

    def blending(self, sky_pred, sun_pred, training="training"):
        return K.axpby(sky_pred, 1.0, sun_pred, 1.0)
