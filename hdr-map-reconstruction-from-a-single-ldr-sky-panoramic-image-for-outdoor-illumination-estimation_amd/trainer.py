"""The reference's joint GAN training step (train.py:382-415) on libhdrsky kernels.

`Trainer.step(ldr, hdr_t, sunpose_gt)` =
    generator_in_step(training=True)  train.py:239-349   (generator + sun-pose + Grad-CAM + sun radiance, BN in
                                                          train mode, discriminator in inference mode, KL + 1000*DoG
                                                          + LSGAN + 10*L1 + 0.01*VGG-perceptual)
    discriminator_in_step(training=True) train.py:351-380 (real / generated passes, BN batch statistics)
    both gradient sets on the SAME pre-update weights, then RMSprop x2 (train.py:402-406).
Forward and backward are written out explicitly (no autograd tape): every FLOP is a libhdrsky launch, so the whole
step is capturable into one hipGraph.  Parameters of one optimizer live in ONE flat fp32 buffer (weights, grads,
RMS slots) so zeroing the gradients and the optimizer update are a single memset / launch each.
"""
from collections import OrderedDict

import os

import numpy as np
import torch

from . import _lib as L
from . import hooks as HOOKS
from . import engine as E
from . import kernels as K
from . import params as P
from .kernels import BF16, InXf, PackedConv, PackedFC

LOSS_SLOTS = ("kl", "perceptual", "dog", "l1", "adv", "disc_generated", "disc_real")


class FlatParams:
    """Named fp32 tensors as views of one contiguous buffer: trainables first ([0:ntrain), padded to 4), then the
    non-trainable BN moving statistics.  `grad` / `ms` mirror the trainable range."""

    def __init__(self, named, device):
        train = [(k, v) for k, v in named.items() if P.is_trainable(k)]
        frozen = [(k, v) for k, v in named.items() if not P.is_trainable(k)]
        self.offsets = OrderedDict()
        off = 0
        for k, v in train + frozen:
            n = int(np.prod(v.shape))
            self.offsets[k] = (off, n, tuple(v.shape))
            off += (n + 3) // 4 * 4
            if k == train[-1][0]:
                self.ntrain = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.ntrain, dtype=torch.float32, device=device)
        self.ms = torch.zeros(self.ntrain, dtype=torch.float32, device=device)
        self.w, self.g = OrderedDict(), OrderedDict()
        for k, v in train + frozen:
            o, n, shape = self.offsets[k]
            self.w[k] = self.flat[o:o + n].view(shape)
            self.w[k].copy_(torch.as_tensor(v))
            if o < self.ntrain:
                self.g[k] = self.grad[o:o + n].view(shape)

    def named_grads(self):
        return self.g


class _Conv:
    """One trainable convolution: master weights (views), packed images, geometry."""

    def __init__(self, w, b, wkey, bkey, stride=1, same=True, upsample=1, need_dgrad=True, precise=True):
        self.w, self.b, self.wkey, self.bkey = w, b, wkey, bkey
        self.kh, self.kw, self.cin, self.cout = w.shape
        self.stride, self.same, self.upsample = stride, same, upsample
        self.need_dgrad, self.precise = need_dgrad, precise
        self.pk = PackedConv(w, precise)
        self.pkT = PackedConv(w, precise, transpose_flip=True) if need_dgrad else None
        # (the layers conv_wgrad2_kernel takes: csrc/conv_wgrad.hip v2_eligible)
        self.emit_xb = (not precise) and upsample == 1 and self.cin >= 32 and self.cin % 32 == 0 and self.cout >= 32 and \
            self.cout % 32 == 0 and stride in (1, 2) and self.kh * self.kw <= 64
        # the layer's handle for operands its forward launch writes for its weight gradient (kernels.Operand): the bf16
        # act(norm(x)) of conv2d(emit_xb=), the gathered operand of a distortion-aware layer
        self.op = K.Operand()

    def repack(self):
        self.pk.repack(self.w)
        if self.pkT is not None:
            self.pkT.repack(self.w)

    def desc(self, x):
        B, H, W, C = x.shape
        return K.conv_desc(B, H, W, C, self.cout, self.kh, self.kw, self.stride, self.same, self.upsample)

    def fwd(self, x, xf=None, compute=BF16, **kw):
        K.label(self.wkey)
        # a training-mode forward (statistics wanted) in front of a weight gradient on the LDS-DMA kernel: the launch also writes
        # its transformed operand as bf16 (kernels.conv2d(emit_xb=True)) - no hdrsky_act_bf16 launch in the backward pass
        if xf is not None and kw.get("want_stats") and self.emit_xb and compute == BF16 and HOOKS.H.emit_xb:
            kw.setdefault("emit_xb", self.op)
        return K.conv2d(x, self.pk, self.b, stride=self.stride, same=self.same, upsample=self.upsample, xf=xf,
                        compute=compute, **kw)

    def wgrad_job(self, x, xf, dy, gw, gb, compute):
        return K.wgrad_job(x, dy, self.kh, self.kw, gw, gb, stride=self.stride, same=self.same, upsample=self.upsample,
                           xf=xf, compute=compute, operand=self.op)

    def dgrad(self, x, dy, compute, residual=None, want_stats=False, out=None, out_bf16=False, up_bf16=False):
        """Gradient wrt the (transformed, pre-resize) conv operand.  out_bf16: stored as bf16 (a gradient whose only reader is
        the InstanceNorm backward, kernels.norm_act_bwd)."""
        K.label(self.wkey + " (data gradient)")
        # (a resize-deconvolution: the gradient at the doubled resolution is read by the resize adjoint below only - bf16
        # when the caller says so with up_bf16)
        d, st = K.conv2d_dgrad(dy, self.pkT, self.desc(x), residual=None if self.upsample == 2 else residual,
                               compute=compute, want_stats=want_stats,
                               out_bf16=(up_bf16 and not want_stats) if self.upsample == 2 else out_bf16)
        if self.upsample == 2:
            d = K.up2x_bwd(d, 1.0, out=out)
        return (d, st) if want_stats else d


class Trainer:
    def __init__(self, gen_params, sun_params, dis_params, vgg_params, device="cuda", lr=1e-4, im_height=32,
                 im_width=128, precise=False, compute=BF16, world_size=1, resconv=True, distortion_aware=False,
                 fused_dense=True, sunpose="net", defer_dense=False):
        """sunpose="external": the step takes the sun-pose net's outputs - cmf [B,H*W] and the three Grad-CAM maps - as
        INPUTS (step(..., cmf=, cams=)) instead of owning the net: SURVEY.md section 8d's substitution for the 128x512
        configuration, whose faithful sun-pose net has 12.9 G parameters (sun_params may be None).  Everything else of
        train.py:382-415 - generator, sun-radiance head, discriminator step, VGG16 / DoG / L1 / LSGAN terms, both
        backward passes, RMSprop x2 - is the same plan; the KL term is reported but is a constant of such a step."""
        if sunpose not in ("net", "external"):
            raise ValueError("sunpose: 'net' or 'external'")
        self.ext_sun = sunpose == "external"
        self.device = torch.device(device)
        self.h, self.w = im_height, im_width
        self.lr, self.compute, self.precise, self.world = lr, compute, precise, world_size
        # res-block chain on bf16 activations through the sample-resident conv + InstanceNorm launches (HDRSKY_BF16
        # mode, 8x32 maps); otherwise the generic conv / norm launches on fp32 activations
        # distortion_aware: the res blocks' 3x3 convolutions are distortion_aware_ops.conv2d (the variant generator.py:14,18
        # keeps commented out; same HWIO weights = its [k*k*C, F] kernel), forward and backward
        # ("res" / "sunpose" / "decoders" parts: engine.da_parts; True = the res blocks)
        self.da_parts = E.da_parts(distortion_aware)
        self.da, self.da_sun, self.da_dec = ("res" in self.da_parts), ("sunpose" in self.da_parts), ("decoders" in self.da_parts)
        if self.ext_sun and self.da_sun:
            raise ValueError("sunpose='external': there is no sun-pose net to make distortion-aware")
        self._da_geo = {}                   # (h, w, k) -> (offsets on the device, transposed sample table)
        self.use_resconv = bool(resconv) and not self.da and compute == BF16 and not precise and \
            K.resconv_supported(im_height // 4, im_width // 4, 128, 128)
        self._da_offs, self._da_table = self._da(im_height // 4, im_width // 4, 3) if self.da else (None, None)
        self._rc = {}
        self.dense_wgrad_external = False   # a data-parallel driver recomputes the Dense weight gradients (parallel.py)
        # fused_dense (HDRSKY_BF16 mode): on an updating step the two Dense kernels' gradients are never written - their
        # RMSprop launch recomputes x^T dy tile by tile (hdrsky_rmsprop_fc_fused).  gs.g["sun.fc*.kernel"] then holds
        # nothing of that step; step(update=False) / replay(update=False) materialise them as before.  A data-parallel
        # driver that all-reduces gradients switches this off; one that gathers the operands sets `dense_operands`.
        # (the matrix-core Dense gradient wants N = im_height * im_width in multiples of 256: other sizes keep the fp32 FMA
        # weight gradient + the plain Dense RMSprop)
        hw = im_height * im_width
        self.dense_mfma = compute == BF16 and not precise and not self.ext_sun and \
            K.fc_xtdy_supported(hw // 64 * 128, hw) and K.fc_xtdy_supported(hw, hw)
        self.fused_dense = bool(fused_dense) and self.dense_mfma
        # defer_dense (captured steps of a fused_dense trainer): the Dense kernels' update - 1 GB of HBM traffic, 0.3 ms, nothing
        # in the step waits for it - does not close the step on stream 2 any more.  The step ends with its first launch only (the
        # operands' bf16 images into a persistent workspace + the bias vectors' step: `apply_fc`), the update itself
        # (`apply_fc_run`) opens the NEXT replay on stream 2, which idles there for ~0.4 ms beside the forward pass; the sun-pose
        # net's Dense layers (`fwd_sun_fc`) wait for it.  Same launches, same arguments, same results; what changes is when the
        # Dense weights are current: after replay() only once flush() - or the next replay - has run (step(), test_step(),
        # capture() and loss_dict() flush themselves).  Off by default; bench.py and train.py's loop switch it on.
        self.defer_dense = bool(defer_dense) and self.fused_dense
        self._fc_pending, self._fc_ws = False, {}
        self.dense_operands = None          # (flat, df1, f1, dz) of the GLOBAL batch, set by parallel.GradientExchange
        self.on_bind = None                 # callable(B) run when a step is bound to a batch (static exchange buffers)
        self.sync = None                    # parallel.BatchSync: batch statistics / the batch maximum over every replica's batch
        named = OrderedDict(("gen." + k, v) for k, v in gen_params.items())
        if not self.ext_sun:
            named.update(("sun." + k, v) for k, v in sun_params.items())
        self.gs = FlatParams(named, self.device)          # optimizer_gen: _gen + _sun variables (train.py:402-403)
        self.ds = FlatParams(OrderedDict(("dis." + k, v) for k, v in dis_params.items()), self.device)
        self.vgg = E._dev(vgg_params, self.device)
        # Three streams: 0 = the critical chain; 1 and 2 = the sun-pose branch / the two perceptual half-batches and,
        # once those are done, the work that only has to finish by the end of the step (discriminator step, Dense-layer
        # optimizer | sun-pose backward, weight gradients).  More streams than that end up sharing one of the runtime's
        # four hardware queues with each other or with RCCL's stream (measured: 4.6 instead of 4.1 ms per step).
        self._streams = [torch.cuda.Stream(device=self.device, priority=HOOKS.H.stream_prio[i]) for i in range(3)]
        if HOOKS.H.apply_fc_cus:      # (experiment: an HBM-bound segment beside the backward pass on a slice of the chip)
            c = HOOKS.H.apply_fc_cus
            self._streams.append(torch.cuda.Stream(device=self.device) if c[0] < 0 else K.masked_stream(c[0], c[1], step=c[2] if len(c) > 2 else 1))     # (-1: a plain fourth stream)
        self._graphs, self._gscale = None, 1.0 / world_size
        self._bn_training = True            # False only inside test_step (sun-radiance head BatchNorm in inference mode)
        self.losses = torch.zeros(len(LOSS_SLOTS), dtype=torch.float32, device=self.device)
        self._wjobs = {}
        self._build_layers()
        if self.da_sun:
            self._init_da_sun()

    # -------------------------------------------------------------------------------------------------
    def _build_layers(self):
        w, pr = self.gs.w, self.precise
        c = self.conv = {}

        def add(name, wkey, bkey, **kw):
            src = self.ds.w if name.startswith("dis.") else w
            c[name] = _Conv(src[wkey], src[bkey] if bkey else None, wkey, bkey, precise=pr, **kw)

        add("gen.conv1_d", "gen.conv1_d.w", "gen.conv1_d.b", need_dgrad=False)
        add("gen.conv2_d", "gen.conv2_d.w", "gen.conv2_d.b", stride=2)
        add("gen.conv3_d", "gen.conv3_d.w", "gen.conv3_d.b", stride=2)
        for i in range(6):
            for j in (1, 2):
                n = "gen.res.%d.conv%d" % (i, j)
                add(n, n + ".w", n + ".b")
        for sfx in ("f", "u"):
            add("gen.conv3_" + sfx, "gen.conv3_%s.kernel_deconv2d" % sfx, "gen.conv3_%s.bias_deconv2d" % sfx, upsample=2)
            add("gen.conv2_" + sfx, "gen.conv2_%s.kernel_deconv2d" % sfx, "gen.conv2_%s.bias_deconv2d" % sfx, upsample=2)
            add("gen.conv1_" + sfx, "gen.conv1_%s.w" % sfx, "gen.conv1_%s.b" % sfx)
        for net in ("gen.sun.", "dis."):
            add(net + "d1", net + "d1.conv.kernel", None, stride=2, need_dgrad=(net == "dis."))
            add(net + "d2", net + "d2.conv.kernel", None, stride=2)
            add(net + "d3", net + "d3.conv.kernel", None, stride=2)
            add(net + "d4", net + "d4.conv.kernel", None, stride=1)
        add("dis.out", "dis.out.kernel", "dis.out.bias", same=False)
        for l in (1, 2, 3) if not self.ext_sun else ():
            for j in (1, 2):
                n = "sun.sunlayer%d.conv%d" % (l, j)
                add(n, n + ".w", n + ".b", need_dgrad=not (l == 1 and j == 1))
        self.fc1 = PackedFC(w["sun.fc1.kernel"], pr) if not self.ext_sun else None
        self.fc2 = PackedFC(w["sun.fc2.kernel"], pr) if not self.ext_sun else None
        # frozen VGG16: forward filters + data-gradient filters
        self.vgg_pk, self.vgg_pkT = {}, {}
        for name, _, _ in P.VGG_CHANNELS:
            self.vgg_pk[name] = PackedConv(self.vgg[name + ".w"], pr)
            self.vgg_pkT[name] = PackedConv(self.vgg[name + ".w"], pr, transpose_flip=True)
        # The preprocessing x*255 - mean (vgg16.py:133-141) rides in conv1_1's operand staging as a per-channel affine (SAME padding
        # pads the PREPROCESSED image with zeros: so does the staging), its derivative 255 in conv1_1's data-gradient filter:
        # two launches less per VGG pass on the perceptual term's critical path.  HDRSKY_VGG_FOLD=0: separate launches (A/B hook)
        self._vgg_fold = HOOKS.H.vgg_fold
        dev_ = self.vgg["conv1_1.w"].device
        self._vgg_xf = InXf(mode=L.IN_AFFINE, slope=1.0, scale=torch.full((3,), 255.0, dtype=torch.float32, device=dev_),
                            shift=torch.tensor([-103.939, -116.779, -123.68], dtype=torch.float32, device=dev_))
        self._vgg_pkT255 = PackedConv(self.vgg["conv1_1.w"] * 255.0, pr, transpose_flip=True)
        self._make_packer()

    def _make_packer(self):
        pairs = []
        for name, cv in self.conv.items():
            pairs.append((cv.w, cv.pk))
            if cv.pkT is not None:
                pairs.append((cv.w, cv.pkT))
        if getattr(self, "da_sun", False) and hasattr(self, "_w1pad"):
            pairs.append((self._w1pad, self._pk1pad))
        self._packer = K.MultiPacker(pairs)      # uploads its job table: must not happen inside a graph capture

    def repack(self, fc=True):
        if getattr(self, "_packer", None) is None:
            self._make_packer()
        if getattr(self, "da_sun", False) and hasattr(self, "_w1pad"):
            cv = self.conv["sun.sunlayer1.conv1"]
            K.pad_channels(cv.w.reshape(cv.kh * cv.kw, cv.cin * cv.cout), 32 * cv.cout, out=self._w1pad.view(cv.kh * cv.kw, 32 * cv.cout))
        self._packer.run()
        if fc and self.fc1 is not None:
            self.fc1.repack(self.gs.w["sun.fc1.kernel"])
            self.fc2.repack(self.gs.w["sun.fc2.kernel"])

    # ---- small helpers ------------------------------------------------------------------------------
    def _inxf(self, stats, name, slope, partials=False):
        w = self.gs.w
        if partials:      # consumers that derive the tables themselves (hdrsky_up2x_xf_bf16)
            return InXf(mode=L.IN_PARTIALS, slope=slope, stats=stats, gamma=w[name + ".gamma"], beta=w[name + ".beta"])
        return K.in_xf(stats, w[name + ".gamma"], w[name + ".beta"], slope)

    # InstanceNorm layers whose (d gamma, d beta) are produced by hdrsky_norm_act_bwd, by the plan segment that
    # differentiates them: their per-sample terms land in persistent tables and one fixed-order launch per segment adds
    # them to the gradient vectors (bit-reproducible; fp32 atomics would sum in arrival order)
    NORM_GROUPS = {
        "bwd_sunpose": ["sun.sunlayer%d.norm%d" % (l, j) for l in (3, 2, 1) for j in (2, 1)],
        "bwd_dec": ["gen.norm%d_%s" % (k, sfx) for sfx in ("f", "u") for k in (2, 3)],
        "bwd_res": ["gen.res.%d.norm%d" % (i, j) for i in range(6) for j in (1, 2)],
        "bwd_enc": ["gen.norm3_d", "gen.norm2_d", "gen.norm1_d"],
    }

    RESCONV_NORMS = ("sun.sunlayer3.norm1", "sun.sunlayer3.norm2")

    def _sun3_ok(self):
        # (mirrors the forward's condition: a distortion-aware sun-pose net never takes the sample-resident launches)
        return self.compute == BF16 and not self.precise and not self.da_sun and \
            K.resconv_supported(self.h // 4, self.w // 4, 64, 128) and HOOKS.H.sun3

    def _norm_state(self, B):
        st = getattr(self, "_nstate", None)
        if st is None:
            st = self._nstate = {}
        if B not in st:
            w, g, sums, red = self.gs.w, self.gs.g, {}, {}
            pairs = {}      # the two decoders' norm layers of one depth share a table of 2 B samples (paired launches)
            for k in (2, 3) if "gen.norm2_f.gamma" in w else ():      # (the sun-pose pre-trainer has no generator)
                tp = torch.zeros((2 * B, 2, w["gen.norm%d_f.gamma" % k].numel()), dtype=torch.float32, device=self.device)
                pairs["gen.norm%d_f" % k], pairs["gen.norm%d_u" % k] = tp[:B], tp[B:]
                sums["gen.norm%d_f+u" % k] = tp
            for seg, names in self.NORM_GROUPS.items():
                entries = []
                for n in names:
                    if n + ".gamma" not in w:
                        continue
                    t = pairs[n] if n in pairs else torch.zeros((B, 2, w[n + ".gamma"].numel()), dtype=torch.float32, device=self.device)
                    sums[n] = t
                    if n in self.RESCONV_NORMS and self._sun3_ok():
                        entries.append((t, g[n + ".gamma"], g[n + ".beta"]))  # hdrsky_resconv's table: (d gamma, d beta)
                    else:
                        entries.append((t, g[n + ".beta"], g[n + ".gamma"]))  # hdrsky_norm_act_bwd's table: (d beta, d gamma)
                if entries:
                    red[seg] = K.DgbReducer(entries)
            st[B] = (sums, red)
        return st[B]

    def _sun3_bwd(self, t, dP, B):
        """Training backward through sunlayer3 on the sample-resident launches (engine.sun3_backward) + its two weight
        gradients on bf16 operands; the norm layers' per-sample (d gamma, d beta) terms go to the segment's tables."""
        n, w, c = "sun.sunlayer3", self.gs.w, self.conv
        sums = self._norm_state(B)[0]
        dc2, dc1, dx = E.sun3_backward(t["s3"], dP, c[n + ".conv1"].pkT, c[n + ".conv2"].pkT, w[n + ".norm1.gamma"],
                                       w[n + ".norm1.beta"], w[n + ".norm2.gamma"], w[n + ".norm2.beta"],
                                       dgb1=sums[n + ".norm1"], dgb2=sums[n + ".norm2"])
        self._wg(n + ".conv2", t["s3"]["o1"]["bf16"], None, dc2)
        self._wg(n + ".conv1", t["s3"]["xb"], None, dc1)
        return dx

    def _in_bwd(self, x, stats, name, slope, dy, pooled=False, f32=False):
        """Gradient wrt the raw conv output in front of InstanceNorm `name`.  Its readers are that conv's data gradient and
        weight gradient, which round it to bf16 while staging: in the single-product mode it is stored as bf16 (f32=True:
        a reader that needs fp32, i.e. the distortion-aware kernels)."""
        w = self.gs.w
        return K.norm_act_bwd(x, stats, w[name + ".gamma"], w[name + ".beta"], slope, dy, pooled,
                              sums=self._norm_state(x.shape[0])[0][name], out_bf16=self._act_bf16() and not f32)

    def _da_f32(self, cv, raw):
        """Whether the gradient wrt the raw output `raw` of the distortion-aware layer `cv` must stay fp32: the fused data-gradient
        kernel takes fp32 sources; where the data gradient runs on a written gathered operand (kernels.da_mat_ok) it reads
        bf16, like the kernel gradient."""
        return not K.da_mat_ok(self.compute, cv.kh, cv.cout, raw.shape[1] * raw.shape[2], "dgrad")

    def _norm_grads(self, seg, B):
        """Adds the per-sample (d gamma, d beta) terms of segment `seg`'s norm layers to the gradient vectors."""
        self._norm_state(B)[1][seg].run()

    def _rc_state(self, B):
        """Per-batch-size state of the sample-resident res chain: the per-sample (d gamma, d beta) terms of its 12 norm
        layers and the one-launch reducer into the gradient vectors (its pointer table is uploaded here, outside any
        graph capture)."""
        st = self._rc.get(B)
        if st is None:
            dgb = torch.zeros((12, B, 2, 128), dtype=torch.float32, device=self.device)
            g, entries = self.gs.g, []
            for i in range(6):
                for j in (1, 2):
                    n = "gen.res.%d.norm%d" % (i, j)
                    entries.append((dgb[2 * i + j - 1], g[n + ".gamma"], g[n + ".beta"]))
            st = self._rc[B] = (dgb, K.DgbReducer(entries))
        return st

    def _wg(self, name, x, xf, dy):
        """Queues the weight gradient of one conv layer.  Nothing in the backward chain consumes it, and ~40 of them
        one by one would each need the whole chip: they are launched together (per stream, `_flush_wgrads`) so that
        layers of similar geometry share a launch - at the end of the segment that queued them, or in a segment of
        their own on the wgrad stream.  The queued job keeps its operands referenced until then."""
        cv = self.conv[name]
        grads = self.ds.g if name.startswith("dis.") else self.gs.g
        q = self._wjobs.setdefault(torch.cuda.current_stream().cuda_stream, [])
        K.WG_NAMES[grads[cv.wkey].data_ptr()] = name
        q.append(cv.wgrad_job(x, xf, dy, grads[cv.wkey], grads[cv.bkey] if cv.bkey else None, self.compute))

    def _da(self, h, w, k=3):
        """(offsets on the device, transposed sample table) of a k x k distortion-aware layer on an h x w map (cached)."""
        key = (h, w, k)
        if key not in self._da_geo:
            tab = K.da_transpose_table(h, w, k, 1, True, self.device)
            if tab is None:
                raise ValueError("distortion-aware training: the transposed sample table of a %dx%d map (k=%d) needs more "
                                 "than %d readers per (pixel, tap)" % (h, w, k, K.DA_KMAX))
            self._da_geo[key] = (K.da_offsets_device(h, w, k, 1, True, self.device), tab)
        return self._da_geo[key]

    def _wg_plain(self, name, x, dy):
        """Queues the weight gradient of layer `name` computed on a materialised operand x (its final, already resized bf16
        activation) as a plain stride-1 layer - what a resize-deconvolution's gradient is with respect to its conv."""
        cv = self.conv[name]
        g = self.gs.g
        q = self._wjobs.setdefault(torch.cuda.current_stream().cuda_stream, [])
        K.WG_NAMES[g[cv.wkey].data_ptr()] = name
        q.append(K.wgrad_job(x, dy, cv.kh, cv.kw, g[cv.wkey], g[cv.bkey] if cv.bkey else None, stride=1, same=True, upsample=1,
                             xf=None, compute=self.compute))

    def _wg_da(self, name, x, dy, dw=None):
        """Queues the kernel gradient of a distortion-aware layer (same HWIO weights viewed [k*k*C, F]); the gathered
        operand is recomputed inside the launch.  dw: another destination (the channel-padded first sun-pose layer)."""
        cv = self.conv[name]
        g = self.gs.g
        q = self._wjobs.setdefault(torch.cuda.current_stream().cuda_stream, [])
        offs = self._da(x.shape[1], x.shape[2], cv.kh)[0]
        q.append(K.da_wgrad_job(x, dy, cv.kh, offs, dw if dw is not None else g[cv.wkey].view(cv.kh * cv.kw * cv.cin, cv.cout),
                                g[cv.bkey], self.compute, operand=cv.op))

    def _flush_wgrads(self):
        """Launches the weight gradients queued on the current stream."""
        q = self._take_wgrads()
        if q:
            K.conv2d_wgrad_multi(q)

    def _sunpose_forward(self, ldr, pick=None, convs_only=False):
        w, c, cp = self.gs.w, self.conv, self.compute
        t, x = {"da": self.da_sun}, ldr
        for l in (1, 2, 3):
            n = "sun.sunlayer%d" % l
            if self.da_sun:       # sunpose_net.py:11,16: distortion_aware_ops.conv2d(filter_out, kernel_size=k_h)
                cv1, cv2 = c[n + ".conv1"], c[n + ".conv2"]
                offs = self._da(x.shape[1], x.shape[2], cv1.kh)[0]
                xin, pk1 = (K.pad_channels(x, 32), self._pk1pad) if l == 1 else (x, cv1.pk)
                r1, st1 = K.da_conv2d(xin, pk1, cv1.b, offs, cp, want_stats=True, train=True, operand=cv1.op)
                a1 = K.norm_apply(r1, st1, w[n + ".norm1.gamma"], w[n + ".norm1.beta"], slope=0.0)
                r2, st2 = K.da_conv2d(a1, cv2.pk, cv2.b, offs, cp, want_stats=True, train=True, operand=cv2.op)
                a, pooled = K.norm_apply(r2, st2, w[n + ".norm2.gamma"], w[n + ".norm2.beta"], slope=0.0, pool=True)
                t["in%d" % l], t["r%da" % l], t["st%da" % l], t["a%da" % l] = xin, r1, st1, a1
                t["r%db" % l], t["st%db" % l], t["A%d" % l], t["P%d" % l] = r2, st2, a, pooled
                x = pooled
                continue
            if l == 3 and E.sun3_supported(x, cp) and not self.precise:
                t["in3"] = x
                t["s3"] = E.sun3_forward(x, c[n + ".conv1"].pk, c[n + ".conv2"].pk, w[n + ".norm1.gamma"], w[n + ".norm1.beta"],
                                         w[n + ".norm2.gamma"], w[n + ".norm2.beta"])
                t["A3"], t["P3"] = t["s3"]["A"], t["s3"]["P"]
                x = t["P3"]
                continue
            r1, st1 = c[n + ".conv1"].fwd(x, compute=cp, want_stats=True, out_bf16=self._raw_bf16())
            xf = self._inxf(st1, n + ".norm1", 0.0)
            r2, st2 = c[n + ".conv2"].fwd(r1, xf, cp, want_stats=True)
            a, pooled = K.norm_apply(r2, st2, w[n + ".norm2.gamma"], w[n + ".norm2.beta"], slope=0.0, pool=True)
            t["in%d" % l], t["r%da" % l], t["st%da" % l], t["xf%d" % l] = x, r1, st1, xf
            t["r%db" % l], t["st%db" % l], t["A%d" % l], t["P%d" % l] = r2, st2, a, pooled
            x = pooled
        B = ldr.shape[0]
        t["flat"] = x.reshape(B, -1)
        return t if convs_only else self._sunpose_dense(t, pick)

    def _sunpose_dense(self, t, pick=None):
        """The Dense layers + soft-max head of sunposeEstimation (sunpose_net.py:64-72) on the record of the conv layers."""
        w, cp = self.gs.w, self.compute
        t["gmax"] = torch.empty(1, dtype=torch.int32, device=t["flat"].device)      # cleared by the finalize launch below
        t["f1"] = K.fc_fwd_fin(t["flat"], self.fc1, cp, w["sun.fc1.bias"], relu=True, zero_word=t["gmax"])
        if pick is None:
            t["z"], t["cmf"] = K.softmax_head(K.fc_fwd(t["f1"], self.fc2, cp), w["sun.fc2.bias"], t["gmax"])
        else:     # Grad-CAM seed of the class pick[m].argmax() from the same launch (train.py:265-267)
            t["z"], t["cmf"], t["dz_pick"] = K.softmax_head_pick(K.fc_fwd(t["f1"], self.fc2, cp), w["sun.fc2.bias"], t["gmax"], pick)
        return t

    def _init_da_sun(self):
        """sunlayer1.conv1 reads the RGB image: the distortion-aware kernels work on 32-channel groups, so it runs on a
        zero-padded copy of the image with a zero-padded copy of its filter (refreshed with every re-pack); its kernel
        gradient lands in a padded buffer whose three real input-channel rows are then added to the gradient."""
        cv = self.conv["sun.sunlayer1.conv1"]
        self._w1pad = torch.zeros((cv.kh, cv.kw, 32, cv.cout), dtype=torch.float32, device=self.device)
        self._pk1pad = PackedConv(self._w1pad, self.precise)
        self._dw1pad = torch.zeros((cv.kh * cv.kw * 32, cv.cout), dtype=torch.float32, device=self.device)
        self._packer = None
        self.repack(fc=False)

    def _sunpose_bwd_da(self, t, dP, B):
        """Backward pass of the distortion-aware sun-pose layers from dP = d loss / d pool3 output: data gradients by
        hdrsky_da_conv2d_dgrad, kernel gradients queued (the gather is recomputed inside their launch) and launched here."""
        c, cp, g = self.conv, self.compute, self.gs.g
        for l in (3, 2, 1):
            n = "sun.sunlayer%d" % l
            k = c[n + ".conv1"].kh
            tab = self._da(self.h >> (l - 1), self.w >> (l - 1), k)[1]
            dr2 = self._in_bwd(t["r%db" % l], t["st%db" % l], n + ".norm2", 0.0, dP, pooled=True, f32=self._da_f32(c[n + ".conv2"], t["r%db" % l]))
            self._wg_da(n + ".conv2", t["a%da" % l], dr2)
            da = K.da_conv2d_dgrad(dr2, c[n + ".conv2"].pkT, tab, k, cp)
            dr1 = self._in_bwd(t["r%da" % l], t["st%da" % l], n + ".norm1", 0.0, da, f32=self._da_f32(c[n + ".conv1"], t["r%da" % l]))
            if l > 1:
                self._wg_da(n + ".conv1", t["in%d" % l], dr1)
                dP = K.da_conv2d_dgrad(dr1, c[n + ".conv1"].pkT, tab, k, cp)
            else:
                K.zero_(self._dw1pad)
                self._wg_da(n + ".conv1", t["in1"], dr1, dw=self._dw1pad)
        self._norm_grads("bwd_sunpose", B)
        self._flush_wgrads()
        cv = c["sun.sunlayer1.conv1"]      # rows of the three real input channels out of the padded kernel gradient
        K.slice_channels(self._dw1pad.view(cv.kh * cv.kw, 32 * cv.cout), 0, cv.cin * cv.cout,
                         out=g[cv.wkey].view(cv.kh * cv.kw, cv.cin * cv.cout))

    def _gradcam(self, t, pick_src):
        """grad_cam.layer x3 under gen_tape.stop_recording() (train.py:257-271): constants for the gradient."""
        w, c, cp = self.gs.w, self.conv, self.compute
        B, h, wd = t["cmf"].shape[0], self.h, self.w
        dz = t["dz_pick"] if "dz_pick" in t else K.softmax_pick_bwd(t["cmf"], t["z"], pick_src)[0]
        df1 = K.fc_dgrad_fin(dz, self.fc2, cp, mask_src=t["f1"])
        dP3 = K.fc_dgrad_fin(df1, self.fc1, cp).reshape(B, h // 8, wd // 8, 128)
        small = (h // 8) * (wd // 8) <= 256       # cam3's GAP weights: summed inside its own launch when the map is small
        w3 = dP3 if small else K.spatial_sum(dP3, 1.0 / ((h // 4) * (wd // 4)))
        s3 = 1.0 / ((h // 4) * (wd // 4)) if small else 1.0
        if self.da_sun:
            dP, sums = dP3, {}
            for l in (3, 2):
                n = "sun.sunlayer%d" % l
                hl, wl = h >> (l - 1), wd >> (l - 1)
                tab = self._da(hl, wl, 3)[1]
                g_ = K.norm_act_bwd(t["r%db" % l], t["st%db" % l], w[n + ".norm2.gamma"], w[n + ".norm2.beta"], 0.0, dP, True)
                g_ = K.da_conv2d_dgrad(g_, c[n + ".conv2"].pkT, tab, 3, cp)
                g_ = K.norm_act_bwd(t["r%da" % l], t["st%da" % l], w[n + ".norm1.gamma"], w[n + ".norm1.beta"], 0.0, g_, False)
                dP = K.da_conv2d_dgrad(g_, c[n + ".conv1"].pkT, tab, 3, cp)
                sums[l - 1] = K.spatial_sum(dP, 1.0 / ((2 * hl) * (2 * wl)))
            return K.grad_cam_maps([(t["A1"], sums[1], 1.0), (t["A2"], sums[2], 1.0), (t["A3"], w3, s3)])
        n3, n2 = "sun.sunlayer3", "sun.sunlayer2"
        if "s3" in t:
            _, _, dP2 = E.sun3_backward(t["s3"], dP3, c[n3 + ".conv1"].pkT, c[n3 + ".conv2"].pkT, w[n3 + ".norm1.gamma"],
                                        w[n3 + ".norm1.beta"], w[n3 + ".norm2.gamma"], w[n3 + ".norm2.beta"])
            sP2 = K.spatial_sum(dP2, 1.0 / ((h // 2) * (wd // 2)))
        else:
            g = K.norm_act_bwd(t["r3b"], t["st3b"], w[n3 + ".norm2.gamma"], w[n3 + ".norm2.beta"], 0.0, dP3, True, out_bf16=self._act_bf16())
            g = c[n3 + ".conv2"].dgrad(t["r3a"], g, cp)
            g = K.norm_act_bwd(t["r3a"], t["st3a"], w[n3 + ".norm1.gamma"], w[n3 + ".norm1.beta"], 0.0, g, False, out_bf16=self._act_bf16())
            dP2, sP2 = c[n3 + ".conv1"].dgrad(t["in3"], g, cp, want_stats=True)
        g = K.norm_act_bwd(t["r2b"], t["st2b"], w[n2 + ".norm2.gamma"], w[n2 + ".norm2.beta"], 0.0, dP2, True, out_bf16=self._act_bf16())
        g = c[n2 + ".conv2"].dgrad(t["r2a"], g, cp)
        g = K.norm_act_bwd(t["r2a"], t["st2a"], w[n2 + ".norm1.gamma"], w[n2 + ".norm1.beta"], 0.0, g, False, out_bf16=self._act_bf16())
        _, sP1 = c[n2 + ".conv1"].dgrad(t["in2"], g, cp, want_stats=True)
        sc2 = 1.0 if "s3" in t else 1.0 / ((h // 2) * (wd // 2))
        return K.grad_cam_maps([(t["A1"], sP1, 1.0 / (h * wd)), (t["A2"], sP2, sc2), (t["A3"], w3, s3)])

    def _down_stack(self, net, params, x, training, update_moving=True, eval_affine=None):
        """downsampling x4 (discriminator.py:20-27 == sunrad_net.py:21-28).  Returns records for the backward pass:
        training=True -> BN batch statistics (+ moving update; update_moving=False leaves the moving statistics alone and
        keeps the layer's statistics partials in the record - `_moving_update` applies them later, in program order),
        else the moving statistics as a constant affine."""
        c, cp = self.conv, self.compute
        B = x.shape[0]
        R = {"in": x}
        # d1 has no norm: its LeakyReLU output is a final activation - stored as bf16 in the single-product mode (what d2's
        # conv and weight gradient would round it to anyway)
        R["d1"], _ = c[net + "d1"].fwd(x, compute=cp, out_slope=0.3, out_bf16=self._act_bf16())
        cur, xf = R["d1"], None
        for d in ("d2", "d3", "d4"):
            # (d4's readers - the Dense heads / the 1-channel output conv - take fp32)
            raw, st = c[net + d].fwd(cur, xf, cp, want_stats=training, out_bf16=self._raw_bf16() and d != "d4")
            n = net + d + ".norm."
            if training:
                if self.sync is not None:      # batch statistics over the batch of every replica
                    st = self.sync.gather_stats(st)
                mean, rstd, sc, sh = K.bn_train_finalize(st, params[n + "gamma"], params[n + "beta"], st.part.shape[0], raw.shape[-1],
                                                         params[n + "moving_mean"] if update_moving else None,
                                                         params[n + "moving_variance"] if update_moving else None)
            else:
                # (eval_affine: the inference-mode affines of the three layers, computed ahead of time by the caller)
                sc, sh = eval_affine[d] if eval_affine is not None else K.bn_eval_affine(
                    params[n + "gamma"], params[n + "beta"], params[n + "moving_mean"], params[n + "moving_variance"])
                mean = rstd = None
            R[d] = dict(x=cur, xf=xf, raw=raw, mean=mean, rstd=rstd, scale=sc, shift=sh)
            if training and not update_moving:
                R[d]["stats"] = st
            cur, xf = raw, InXf(mode=L.IN_AFFINE, slope=0.3, scale=sc, shift=sh)
        R["xf_out"] = xf
        return R

    def _moving_update(self, net, params, R):
        """The moving-statistics update of a `_down_stack(..., update_moving=False)` pass (the same launch on the same
        partials: identical batch moments)."""
        for d in ("d2", "d3", "d4"):
            n, st = net + d + ".norm.", R[d]["stats"]
            K.bn_train_finalize(st, params[n + "gamma"], params[n + "beta"], st.part.shape[0], R[d]["raw"].shape[-1],
                                params[n + "moving_mean"], params[n + "moving_variance"])

    def _down_stack_bwd(self, net, params, grads, R, dact4, training, want_input_grad, do_wgrad=True):
        """Backward of _down_stack from the gradient wrt the ACTIVATED d4 output."""
        c, cp = self.conv, self.compute
        dy = dact4
        for d in ("d4", "d3", "d2"):
            r = R[d]
            n = net + d + ".norm."
            b16 = self._act_bf16()      # gradients wrt raw conv outputs: read by a data-gradient conv / a weight gradient only
            if training:
                draw = K.bn_act_bwd(r["raw"], dy, r["mean"], r["rstd"], params[n + "gamma"], params[n + "beta"], 0.3,
                                    grads[n + "gamma"] if do_wgrad else None, grads[n + "beta"] if do_wgrad else None,
                                    out_bf16=b16, sync=self.sync)
            else:
                draw = K.affine_act_bwd(r["raw"], dy, r["scale"], r["shift"], 0.3, out_bf16=b16)
            if do_wgrad:
                self._wg(net + d, r["x"], r["xf"], draw)
            # gradient wrt the activated input of this layer: read by the next BatchNorm / activation backward only - bf16
            dy = c[net + d].dgrad(r["x"], draw, cp, out_bf16=self._nab_bf16() and self.sync is None)
        d1pre = K.affine_act_bwd(R["d1"], dy, None, None, 0.3, out_bf16=self._act_bf16())
        if do_wgrad:
            self._wg(net + "d1", R["in"], None, d1pre)
        if want_input_grad:
            return c[net + "d1"].dgrad(R["in"], d1pre, cp)
        return None

    # ---- the two training-mode discriminator passes (real | generated) as ONE batch of 2B through the convolutions; the
    # BatchNorm layers keep the two halves apart (their own batch statistics, moving averages updated real first, then
    # generated - the reference's program order, train.py:360-361) through per-sample affine tables
    def _down_stack_pair(self, net, params, x2):
        c, cp = self.conv, self.compute
        B = x2.shape[0] // 2
        R = {"in": x2}
        R["d1"], _ = c[net + "d1"].fwd(x2, compute=cp, out_slope=0.3, out_bf16=self._act_bf16())
        cur, xf = R["d1"], None
        for d in ("d2", "d3", "d4"):
            raw, st = c[net + d].fwd(cur, xf, cp, want_stats=True, out_bf16=self._raw_bf16() and d != "d4")
            n, C = net + d + ".norm.", raw.shape[-1]
            sc2 = torch.empty((2 * B, C), dtype=torch.float32, device=raw.device)
            sh2 = torch.empty_like(sc2)
            halves = []
            for hf in (0, 1):
                sth = K.Stats(st.part[hf * B:(hf + 1) * B], st.nparts, st.count)
                if self.sync is not None:
                    sth = self.sync.gather_stats(sth)
                mean, rstd, _, _ = K.bn_train_finalize(sth, params[n + "gamma"], params[n + "beta"], sth.part.shape[0], C,
                                                       params[n + "moving_mean"], params[n + "moving_variance"],
                                                       scale_rows=sc2[hf * B:(hf + 1) * B], shift_rows=sh2[hf * B:(hf + 1) * B])
                halves.append((mean, rstd))
            R[d] = dict(x=cur, xf=xf, raw=raw, halves=halves)
            cur, xf = raw, InXf(mode=L.IN_AFFINE, slope=0.3, scale=sc2, shift=sh2)
        R["xf_out"] = xf
        return R

    def _down_stack_pair_bwd(self, net, params, grads, R, dact4):
        c, cp = self.conv, self.compute
        dy = dact4
        B = dy.shape[0] // 2
        for d in ("d4", "d3", "d2"):
            r = R[d]
            n = net + d + ".norm."
            b16 = self._act_bf16()
            draw = torch.empty(r["raw"].shape, dtype=torch.bfloat16 if b16 else torch.float32, device=r["raw"].device)
            for hf, (mean, rstd) in enumerate(r["halves"]):
                sl = slice(hf * B, (hf + 1) * B)
                K.bn_act_bwd(r["raw"][sl], dy[sl], mean, rstd, params[n + "gamma"], params[n + "beta"], 0.3,
                             grads[n + "gamma"], grads[n + "beta"], out=draw[sl], out_bf16=b16, sync=self.sync)
            self._wg(net + d, r["x"], r["xf"], draw)
            dy = c[net + d].dgrad(r["x"], draw, cp, out_bf16=self._nab_bf16() and self.sync is None)
        d1pre = K.affine_act_bwd(R["d1"], dy, None, None, 0.3, out_bf16=self._act_bf16())
        self._wg(net + "d1", R["in"], None, d1pre)

    def _sunrad_forward(self, ldr, cams, t, S):
        w = self.gs.w
        plz = K.plz_build(ldr, *cams)
        R = self._down_stack("gen.sun.", w, plz, training=self._bn_training)
        xf = R["xf_out"]
        part = K.dense_heads(R["d4"]["raw"], xf.scale, xf.shift, 0.3, w["gen.sun.gamma.kernel"], w["gen.sun.beta.kernel"])
        if self.sync is not None:          # tf.reduce_max(sunpose_pred) (generator.py:160) over the batch of every replica
            self.sync.max_word(t["gmax"])
        rad_lin, rad_gamma, gamma, beta = K.sun_rad(t["cmf"], t["gmax"], part, w["gen.sun.gamma.bias"], w["gen.sun.beta.bias"],
                                                    self.h, self.w)
        S["sunrad"] = R
        return rad_lin, rad_gamma, gamma, beta

    # ---- VGG16 perceptual term (vgg16.py:127-165, train.py:308-313) ---------------------------------------
    VGG_BLOCKS = (("conv1_1", "conv1_2"), ("conv2_1", "conv2_2"), ("conv3_1", "conv3_2", "conv3_3"))

    def _act_bf16(self):
        """Final activations of ReLU / LeakyReLU-only stretches are stored as bf16 (HDRSKY_BF16 mode; HDRSKY_VGG_BF16=0: A/B hook)."""
        return self.compute == BF16 and not self.precise and HOOKS.H.vgg_bf16

    def _nab_bf16(self):
        """Gradients that go from a data-gradient conv straight into an InstanceNorm backward (and nowhere else) travel as
        bf16 in the single-product mode: the norm backward's output is stored as bf16 anyway, and its two passes over this
        tensor are half as long (HDRSKY_NAB_DY_BF16=0: A/B hook)."""
        return self._act_bf16() and HOOKS.H.nab_dy_bf16

    def _raw_bf16(self):
        """Raw conv outputs in front of an InstanceNorm / BatchNorm layer are stored as bf16 in the single-product mode: the
        statistics come from the fp32 accumulators (conv epilogue), every reader - the next conv's staging, the weight
        gradient's operand, the norm backward - widens while loading (include/hdrsky.h, "RAW CONV OUTPUTS AS bf16").  Not
        the tensors a max-pool argmax is taken on (the sun-pose net's second convs: rounding makes ties), not beside the
        distortion-aware kernels or the cross-replica BatchNorm (fp32 readers).  OFF by default (tuning hook HDRSKY_RAW_BF16=1):
        profiles/r04_raw_bf16_ab.txt - the 32x128 step is 0.3 % shorter (its tensors live in the MALL; the readers are not
        bandwidth-bound) and the loss terms move 1.5x further from the fp32 oracle."""
        return self._act_bf16() and HOOKS.H.raw_bf16 and self.sync is None and not self.da_parts

    def _deconv_mat(self):
        return not self.precise and K.deconv_materialised(self.compute)

    def _vgg_forward(self, x_gamma, keep):
        """pool1..3 of a gamma-domain BGR batch; `keep` collects what the backward pass re-reads."""
        cp = self.compute
        x = x_gamma if self._vgg_fold else K.vgg_pre(x_gamma)
        pools = []
        # HDRSKY_BF16: the chain's activations live in bf16 (ReLU only, so they are final: the next conv would round them to
        # bf16 anyway - its result is bit-identical - and the fp32 input / output bursts of these launches halve); the
        # pooled features of the perceptual term are returned in fp32
        b16 = self._act_bf16()
        for blk in self.VGG_BLOCKS:
            for name in blk:
                if keep is not None:
                    keep[name + ".in"] = x
                K.label("vgg." + name)
                x, _ = K.conv2d(x, self.vgg_pk[name], self.vgg[name + ".b"], out_slope=0.0, compute=cp, out_bf16=b16,
                                xf=self._vgg_xf if (self._vgg_fold and name == "conv1_1") else None)
                if keep is not None:
                    keep[name] = x
            if b16:
                p32, x = K.maxpool(x, want_bf16=blk is not self.VGG_BLOCKS[-1])
                pools.append(p32)
            else:
                x = K.maxpool(x)
                pools.append(x)
        return pools

    def _vgg_target(self, hdr_t):
        """vgg2(hdr_t_gamma) (train.py:309): does not depend on the generator, so it runs beside the forward pass."""
        return self._vgg_forward(K.tonemap(hdr_t, False), None)

    def _vgg_loss_and_grad(self, y_gamma, target_pools, share=1.0, out=None):
        """Perceptual term of the samples in y_gamma; `share` = their fraction of the batch (the L1 terms are batch
        means, so a part of the batch contributes share x its own mean)."""
        cp, B = self.compute, y_gamma.shape[0]
        acts = {}
        pools = self._vgg_forward(y_gamma, acts)
        # 0.01 * sum_i mean|pool_i(pred) - pool_i(target)|, gradient wrt the prediction.  The L1 term of a block's pooled output is
        # issued when the backward pass reaches that block and adds its gradient INTO the gradient arriving from the block above
        # (fp32 there) - no separate gradient tensor + sum launch per block
        g = None
        for bi, blk in reversed(list(enumerate(self.VGG_BLOCKS))):
            if self._vgg_fold and acts[blk[-1]].dtype == torch.bfloat16:
                # bf16 chain: the block's L1 term inside its pool backward (one launch instead of hdrsky_l1 + the pool backward)
                g = K.maxpool_relu_l1_bwd(acts[blk[-1]], pools[bi], target_pools[bi], g, share, 0.01 * share, self.losses[1:2])
                dp = None
            elif g is None or not self._vgg_fold:
                dp = torch.empty_like(pools[bi])
                K.l1(pools[bi], target_pools[bi], share, 0.01 * share, self.losses[1:2], da=dp)
                if g is not None:
                    dp = K.axpby(dp, 1.0, g, 1.0)
            else:
                K.l1(pools[bi], target_pools[bi], share, 0.01 * share, self.losses[1:2], da=g, accumulate=True)
                dp = g
            b16 = acts[blk[-1]].dtype == torch.bfloat16     # bf16 chain: gradients between the data-gradient convs are final
            if dp is not None:
                g = K.maxpool_relu_bwd(acts[blk[-1]], dp, out_bf16=b16)   # wrt the pre-ReLU output of the block's last conv
            for k in range(len(blk) - 1, -1, -1):
                name = blk[k]
                xin = acts[name + ".in"]
                d = K.conv_desc(B, xin.shape[1], xin.shape[2], xin.shape[3], self.vgg_pk[name].Cout, 3, 3, 1, True, 1)
                K.label("vgg." + name + " (data gradient)")
                if b16 and k > 0:     # the ReLU mask of the layer below rides in this conv's epilogue, bf16 out
                    g, _ = K.conv2d_dgrad(g, self.vgg_pkT[name], d, compute=cp, mask_bf16=acts[blk[k - 1]], mask_slope=0.0,
                                          out_bf16=True)
                    continue
                if self._vgg_fold and name == "conv1_1":   # x 255 in the filter, straight into the caller's gradient slice
                    g, _ = K.conv2d_dgrad(g, self._vgg_pkT255, d, compute=cp, out=out)
                    return g
                g, _ = K.conv2d_dgrad(g, self.vgg_pkT[name], d, compute=cp)   # wrt this conv's (post-ReLU) input
                if k > 0:
                    g = K.affine_act_bwd(acts[blk[k - 1]], g, None, None, 0.0)
        return K.axpby(g, 255.0, out=out)   # d/d y_gamma of the 0.01-weighted perceptual term

    # ---- one training step -----------------------------------------------------------------------------------
    # The step is a DAG of linear SEGMENTS, each bound to one of three HIP streams.  Eagerly they are enqueued in plan
    # order with event waits; for replay every segment is captured into its own hipGraph and the graphs are launched
    # on their streams with the same event waits.  (One hipGraph of the whole multi-stream step executes its
    # branches almost serially on this runtime - measured: one kernel in flight 75 % of the time - whereas separate
    # single-stream graphs on separate streams do run side by side.)
    def _plan(self):
        w, g, c, cp = self.gs.w, self.gs.g, self.conv, self.compute
        T = self._T                                   # tensors that cross segment boundaries
        B, h, wd = T["ldr"].shape[0], self.h, self.w
        segs = []

        def seg(name, stream, deps=()):
            def deco(fn):
                segs.append((name, stream, tuple(deps), fn))
                return fn
            return deco

        def decode_head(sfx):
            """The two resize-deconvolutions of a decoder (generator.py:110-156) - everything that does not need the
            residual input of its last layer; decode_tail finishes it.  (The sun decoder's residual is the sun-radiance map,
            the last thing fwd_sun produces: its two deconvolutions run in fwd_enc, while stream 0 would wait for it.)"""
            res_out = T["x"][-1]
            if self.da_dec:       # distortion_aware_ops.deconv2d (:272-542): bilinear 2x resize, then the distortion-aware 3x3
                c3, c2 = c["gen.conv3_" + sfx], c["gen.conv2_" + sfx]
                # where a layer runs on its written gathered operand (kernels.da_mat_ok) the resized map feeds nothing but that
                # gather: written as bf16 by the resize launch (with the InstanceNorm + LeakyReLU in front of it folded in),
                # and the two decoders share the resized encoder output and its gathered operand
                px3 = 4 * res_out.shape[1] * res_out.shape[2]
                if K.da_mat_ok(cp, 3, c3.cin, px3, "fwd"):
                    if sfx == "f":      # (the first decoder of every pass writes it: T outlives a pass)
                        T["u3_da"] = K.up2x_act_bf16(res_out)
                    u3 = T["u3_da"]
                else:
                    u3 = K.up2x(res_out)
                # (the two decoders' first layers read the same resized map: the sun decoder's takes - and hands its kernel
                # gradient - the gathered operand the sky decoder's wrote, through one shared handle)
                if sfx == "u":
                    c3.op = c["gen.conv3_f"].op
                d3, s3 = K.da_conv2d(u3, c3.pk, c3.b, self._da(u3.shape[1], u3.shape[2])[0], cp, want_stats=True, train=True,
                                     reuse_operand=(sfx == "u"), operand=c3.op)
                if K.da_mat_ok(cp, 3, c2.cin, 4 * px3, "fwd"):
                    u2 = K.up2x_act_bf16(d3, self._inxf(s3, "gen.norm3_" + sfx, 0.1, partials=True))
                else:
                    u2 = K.up2x(K.norm_apply(d3, s3, w["gen.norm3_%s.gamma" % sfx], w["gen.norm3_%s.beta" % sfx], slope=0.1))
                d2, s2 = K.da_conv2d(u2, c2.pk, c2.b, self._da(u2.shape[1], u2.shape[2])[0], cp, want_stats=True, train=True, operand=c2.op)
                xf1 = self._inxf(s2, "gen.norm2_" + sfx, 0.1)
                T["dech_" + sfx] = (d3, s3, u3, d2, s2, xf1, u2)
            elif self._deconv_mat():
                # single-product mode: the operand of each resize-deconvolution is written once as bf16 (hdrsky_up2x_xf_bf16:
                # the fused staging's own arithmetic); the plain conv and - later - the plain weight gradient run on it, and
                # the two decoders share the upsampled encoder output
                c3, c2 = c["gen.conv3_" + sfx], c["gen.conv2_" + sfx]
                u3 = T["u3"]          # written by fwd_enc, in front of the first decoder
                K.label(c3.wkey)
                d3, s3 = K.conv2d(u3, c3.pk, c3.b, compute=cp, want_stats=True, out_bf16=self._raw_bf16())
                xf2 = self._inxf(s3, "gen.norm3_" + sfx, 0.1, partials=True)
                u2 = K.up2x_act_bf16(d3, xf2)
                K.label(c2.wkey)
                d2, s2 = K.conv2d(u2, c2.pk, c2.b, compute=cp, want_stats=True, out_bf16=self._raw_bf16())
                xf1 = self._inxf(s2, "gen.norm2_" + sfx, 0.1)
                T["dech_" + sfx] = (d3, s3, xf2, d2, s2, xf1, u3, u2)
            else:
                d3, s3 = c["gen.conv3_" + sfx].fwd(res_out, compute=cp, want_stats=True)
                xf2 = self._inxf(s3, "gen.norm3_" + sfx, 0.1)
                d2, s2 = c["gen.conv2_" + sfx].fwd(d3, xf2, cp, want_stats=True)
                xf1 = self._inxf(s2, "gen.norm2_" + sfx, 0.1)
                T["dech_" + sfx] = (d3, s3, xf2, d2, s2, xf1, None, None)

        early_head = HOOKS.H.dec_head_early     # (tuning hook)
        # Round 5: the two decoders run the same layer shapes on the same encoder output with their own weights
        # (generator.py:110-156).  Their two resize-deconvolutions and - backwards - the whole chain from the 7x7 tails' data
        # gradients to the gradient with respect to the encoder output are issued as PAIRED launches on a batch of 2 B (first
        # half: sky decoder, second half: sun decoder; kernels.ConvPair, include/hdrsky.h "PAIRED LAUNCHES"): 3 instead of 6
        # launches forward, 9 instead of 18 backward, every value bit-identical to the unpaired launches
        # (tests/test_pair_gpu.py).  HDRSKY_DEC_PAIR=0: the unpaired plan (A/B hook).
        dec_pair = self._deconv_mat() and not self.da_dec and HOOKS.H.dec_pair

        def decode_heads_pair():
            c3f, c3u, c2f, c2u = (c["gen.conv%d_%s" % (k, sfx)] for k in (3, 2) for sfx in ("f", "u"))
            u3 = T["u3"]
            K.label("gen.conv3_f + gen.conv3_u")
            d3, s3 = K.conv2d(u3, c3f.pk, c3f.b, compute=cp, want_stats=True, out_bf16=self._raw_bf16(), pair=K.ConvPair(c3u.pk, c3u.b),
                              x_shared=True)
            xf2 = InXf(mode=L.IN_PARTIALS, slope=0.1, stats=s3, gamma=w["gen.norm3_f.gamma"], beta=w["gen.norm3_f.beta"],
                       gamma2=w["gen.norm3_u.gamma"], beta2=w["gen.norm3_u.beta"])
            u2 = K.up2x_act_bf16(d3, xf2)
            K.label("gen.conv2_f + gen.conv2_u")
            d2, s2 = K.conv2d(u2, c2f.pk, c2f.b, compute=cp, want_stats=True, out_bf16=self._raw_bf16(), pair=K.ConvPair(c2u.pk, c2u.b))
            xf1 = K.in_xf(s2, w["gen.norm2_f.gamma"], w["gen.norm2_f.beta"], 0.1, pair=(w["gen.norm2_u.gamma"], w["gen.norm2_u.beta"]))
            T["dec_pair"] = dict(d3=d3, s3=s3, u2=u2, d2=d2, s2=s2, u3=u3)
            for i, sfx in enumerate(("f", "u")):      # each decoder's own record: views of the paired tensors
                lo, hi = i * B, (i + 1) * B
                half = lambda st: K.Stats(st.part[lo:hi], st.nparts, st.count)
                T["dech_" + sfx] = (d3[lo:hi], half(s3), K._half_xf(xf2, lo, hi, i == 1), d2[lo:hi], half(s2), K._half_xf(xf1, lo, hi, i == 1),
                                    u3, u2[lo:hi])

        def decode_tail(sfx, residual):
            hd = T["dech_" + sfx]
            d2, xf1 = hd[3], hd[5]
            y, _ = c["gen.conv1_" + sfx].fwd(d2, xf1, cp, out_slope=0.1, residual=residual, final_relu=True)
            if self.da_dec:
                d3, s3, u3, d2, s2, xf1, u2 = hd
                T["dec_" + sfx] = (d3, s3, u3, d2, s2, xf1, y, residual, u2)
            else:
                d3, s3, xf2, d2, s2, xf1, u3, u2 = hd
                T["dec_" + sfx] = (d3, s3, xf2, d2, s2, xf1, y, residual, u3, u2)
            return y

        # ------------------------------------------------------------------ forward (train.py:239-299)
        # (the longest independent chain is enqueued first; segment order = host launch order)
        defer = self._defer
        sun_split = HOOKS.H.fwd_sun_split and not defer and not self.ext_sun      # (experiment: the sun branch as two segments)
        sun_done = "fwd_sun_fc" if (defer or sun_split) else "fwd_sun"      # the segment that completes the sun branch
        if defer:
            # the PREVIOUS step's Dense update (see __init__: defer_dense), in the window stream 2 idles in beside the forward pass
            M = (self.dense_operands[0] if self.dense_operands else T["ldr"]).shape[0]
            for name, pf in (("sun.fc2", self.fc2), ("sun.fc1", self.fc1)):
                if name not in self._fc_ws or self._fc_ws[name][0] != M:
                    self._fc_ws[name] = (M, K.fc_xtdy_ws(M, pf.K, pf.N, self.device))

            @seg("apply_fc_run", 2)
            def _():
                for name, pf in (("sun.fc2", self.fc2), ("sun.fc1", self.fc1)):
                    o, n, shape = self.gs.offsets[name + ".kernel"]
                    K.rmsprop_fc_fused_apply(w[name + ".kernel"], self.gs.ms[o:o + n].view(shape), self._fc_ws[name][0], pf, self.lr,
                                             self._fc_ws[name][1], gscale=self._gscale)

            @seg("fwd_sun", 1)
            def _():       # the sun-pose net's conv layers: nothing here reads the Dense kernels
                T["t"] = self._sunpose_forward(T["ldr"], pick=T["gt"], convs_only=True)

            @seg("fwd_sun_fc", 1, ["apply_fc_run"])
            def _():       # Dense layers + soft-max head on the updated kernels, Grad-CAM, sun radiance head
                t = self._sunpose_dense(T["t"], T["gt"])
                T["cams"] = self._gradcam(t, T["gt"])
                T["rad"] = self._sunrad_forward(T["ldr"], T["cams"], t, T)
        elif sun_split:
            @seg("fwd_sun", 1)
            def _():
                T["t"] = self._sunpose_forward(T["ldr"], pick=T["gt"], convs_only=True)

            @seg("fwd_sun_fc", 1)
            def _():
                t = self._sunpose_dense(T["t"], T["gt"])
                T["cams"] = self._gradcam(t, T["gt"])
                T["rad"] = self._sunrad_forward(T["ldr"], T["cams"], t, T)
        else:
            @seg("fwd_sun", 1)
            def _():       # sun-pose net, Grad-CAM (constants for the gradient: train.py:257-271), sun radiance head
                if self.ext_sun:      # the net's outputs are inputs of the step; tf.reduce_max(sunpose_pred) from its own launch
                    t = T["t"] = {"cmf": T["cmf_in"], "gmax": K.global_max(T["cmf_in"])}
                    T["cams"] = T["cams_in"]
                else:
                    t = T["t"] = self._sunpose_forward(T["ldr"], pick=T["gt"])
                    T["cams"] = self._gradcam(t, T["gt"])
                T["rad"] = self._sunrad_forward(T["ldr"], T["cams"], t, T)

        # the discriminator's pass over the REAL pairs needs nothing the generator produces: it runs beside the forward
        # pass, in the window stream 2 otherwise idles in (between the VGG target features and the perceptual term, ~0.45 ms)
        # instead of inside the backward pass where all three streams are busy.  Its BatchNorm moving-statistics update waits
        # for its place in the reference's program order (after the adversarial term's inference-mode call, in front of the
        # generated pairs: train.py:302,360-361).  Measured: no gain - the forward pass it runs beside slows down by what the
        # backward pass wins (2.77 ms either way, profiles/LABNOTES.md r3 5.0) - so the default stays ONE batch of 2B in disc_step; HDRSKY_DISC_SPLIT=1
        # selects the split (A/B hook, covered by tests/test_train_gpu.py)
        split_disc = HOOKS.H.disc_split

        def zero_grads():
            # (the two Dense kernels + biases, 201 of the 222 MB, are overwritten by their weight-gradient launches)
            K.zero_(self.ds.grad); K.zero_(self.losses); K.zero_(self.gs.grad[:self.fc_grad_range()[0]])
            # the discriminator's inference-mode BatchNorm affines for the adversarial term (moving statistics as they stand
            # at the start of the step: disc_step updates them later): three tiny launches that need nothing of this step, off
            # the dependent chain
            dw_ = self.ds.w
            T["dis_eval_affine"] = {d: K.bn_eval_affine(dw_["dis.%s.norm.gamma" % d], dw_["dis.%s.norm.beta" % d],
                                                        dw_["dis.%s.norm.moving_mean" % d], dw_["dis.%s.norm.moving_variance" % d])
                                    for d in ("d2", "d3", "d4")}

        if split_disc:            # disc_real accumulates into the zeroed buffers: they are cleared first
            seg("zero", 0)(zero_grads)

        @seg("fwd_enc", 0)
        def _():
            ldr = T["ldr"]
            r16 = self._raw_bf16()
            T["c1"], T["s1"] = c["gen.conv1_d"].fwd(ldr, compute=cp, want_stats=True, out_bf16=r16)        # generator.py:92-108
            T["xf2"] = self._inxf(T["s1"], "gen.norm1_d", 0.1)
            T["c2"], T["s2"] = c["gen.conv2_d"].fwd(T["c1"], T["xf2"], cp, want_stats=True, out_bf16=r16)
            T["xf3"] = self._inxf(T["s2"], "gen.norm2_d", 0.1)
            T["c3"], T["s3"] = c["gen.conv3_d"].fwd(T["c2"], T["xf3"], cp, want_stats=True, out_bf16=r16)
            x = K.norm_apply(T["c3"], T["s3"], w["gen.norm3_d.gamma"], w["gen.norm3_d.beta"], slope=0.1)
            T["x"] = [x]
            if self.use_resconv:      # generator.py:26-35 x6: two launches per block, InstanceNorm inside them
                xb = K.to_bf16(x)
                T["xb"] = [xb]
                for i in range(6):
                    p = "gen.res.%d." % i
                    o1 = K.resconv_fwd(xb, c[p + "conv1"].pk, None, w[p + "norm1.gamma"], w[p + "norm1.beta"], 0.1, save=True)
                    o2 = K.resconv_fwd(o1["bf16"], c[p + "conv2"].pk, None, w[p + "norm2.gamma"], w[p + "norm2.beta"], 1.0,
                                       residual=x, want_f32=True, want_bf16=(i < 5), save=True)
                    T["res%d" % i] = (o1, o2)
                    x, xb = o2["f32"], o2.get("bf16")
                    T["x"].append(x); T["xb"].append(xb)
            if self.da:               # generator.py:14,18: both 3x3 convolutions of a block are distortion-aware
                offs = self._da_offs
                for i in range(6):
                    p = "gen.res.%d." % i
                    cv1, cv2 = c[p + "conv1"], c[p + "conv2"]
                    c1, t1 = K.da_conv2d(x, cv1.pk, cv1.b, offs, cp, want_stats=True, train=True, operand=cv1.op)
                    a1 = K.norm_apply(c1, t1, w[p + "norm1.gamma"], w[p + "norm1.beta"], slope=0.1)
                    c2, t2 = K.da_conv2d(a1, cv2.pk, cv2.b, offs, cp, want_stats=True, train=True, operand=cv2.op)
                    xn = K.norm_apply(c2, t2, w[p + "norm2.gamma"], w[p + "norm2.beta"], slope=1.0, residual=x)
                    T["res%d" % i] = (c1, t1, a1, c2, t2)
                    x = xn
                    T["x"].append(x)
            for i in range(0 if (self.use_resconv or self.da) else 6):   # generic launches (BF16X3, other image sizes)
                p = "gen.res.%d." % i
                r1, t1 = c[p + "conv1"].fwd(x, compute=cp, want_stats=True, out_bf16=r16)
                xf = self._inxf(t1, p + "norm1", 0.1)
                r2, t2 = c[p + "conv2"].fwd(r1, xf, cp, want_stats=True, out_bf16=r16)
                x = K.norm_apply(r2, t2, w[p + "norm2.gamma"], w[p + "norm2.beta"], slope=1.0, residual=x)
                T["res%d" % i] = (r1, t1, xf, r2, t2)
                T["x"].append(x)
            if self._deconv_mat() and not self.da_dec:
                T["u3"] = K.up2x_act_bf16(T["x"][-1])      # the resized encoder output, shared by both decoders
            if dec_pair:
                decode_heads_pair()
            else:
                decode_head("f")
            T["sky_gamma"] = decode_tail("f", ldr)
            if early_head and not dec_pair:
                decode_head("u")

        if not split_disc:
            # nothing in the forward pass touches the gradient / loss buffers: they are cleared behind the encoder-decoder
            # chain, while stream 0 waits for the sun branch, instead of in front of it (~35 us of the step's critical path)
            seg("zero", 0)(zero_grads)

        # The perceptual term's target pass has no input from the forward pass and used to open stream 2 beside it ("for free").
        # It is not free: its seven launches are the LARGEST of the forward phase (batch 32, 64-256 channels) and they take the chip
        # from the two dependent chains of small launches the step is waiting for (fwd_sun 335 us alone, 585 us beside them).
        # Behind the encoder-decoder chain - stream 2 idles until the perceptual term anyway - the step is 1.5 % shorter
        # (2.390 -> 2.355 ms, profiles/r05_plan_deps_ab.txt; behind the sun branch instead: +3 %, the term then waits for it).
        @seg("vgg_target", 2, ["fwd_enc"] if HOOKS.H.vgg_target_late else [])
        def _():
            T["vgg_tgt"] = self._vgg_target(T["hdr_t"])

        if split_disc:
            @seg("disc_real", 2, ["zero"])
            def _():       # discriminator_in_step's real half (train.py:351-364): forward, loss, backward, weight gradients
                cvo = c["dis.out"]
                Rr = T["disc_real"] = self._down_stack("dis.", self.ds.w, K.concat2(T["ldr"], T["hdr_t"]), training=True,
                                                       update_moving=False)
                lg, _ = cvo.fwd(Rr["d4"]["raw"], Rr["xf_out"], cp)
                dl = torch.empty_like(lg)
                K.mse(lg, 1.0, 1.0, 0.5, self.losses[6:7], out=dl)
                self._wg("dis.out", Rr["d4"]["raw"], Rr["xf_out"], dl)
                da4 = cvo.dgrad(Rr["d4"]["raw"], dl, cp)
                self._down_stack_bwd("dis.", self.ds.w, self.ds.g, Rr, da4, training=True, want_input_grad=False)
                # its weight gradients are launched by disc_step: with them this segment outlasts the window (0.49 ms) and
                # delays the perceptual term's second half, which the backward pass waits for
                T["disc_real_wg"] = self._take_wgrads()

        @seg("fwd_blend", 0, [sun_done])
        def _():
            rad_lin, rad_gamma, gamma, beta = T["rad"]
            if not early_head and not dec_pair:
                decode_head("u")
            sun_gamma = decode_tail("u", rad_gamma)
            y_gamma, y_lin, alpha, sky_lin, sun_lin = K.blend(T["sky_gamma"], sun_gamma, E.THRESHOLD)
            T.update(y_gamma=y_gamma, y_lin=y_lin, alpha=alpha, sun_gamma=sun_gamma, gamma=gamma, beta=beta,
                     rad_gamma=rad_gamma, rad_lin=rad_lin, sky_lin=sky_lin, sun_lin=sun_lin,
                     dyg=torch.empty_like(y_gamma))

        # ------------------------------------------------------------------ losses (train.py:301-331)
        # (L1 / DoG / KL are issued at the head of the adversarial term's segment: nothing waits for them alone, and every
        # segment boundary on the dependent chain is a graph launch of its own, ~15 us of idle stream)
        def loss_main():
            dyl = T["dyl"] = torch.empty_like(T["y_lin"])
            K.l1(T["y_lin"], T["hdr_t"], 1.0, 10.0, self.losses[3:4], da=dyl)                       # 10 * L1
            K.dog_loss(T["y_lin"], T["hdr_t"], 1000.0, self.losses[2:3], dyl)                       # 1000 * DoG
            T["dcmf"] = K.kl(T["gt"], T["t"]["cmf"], self.losses[0:1])                              # KL

        # 0.01 * perceptual: the longest pole between the forward pass and the backward pass, and the main chain idles
        # meanwhile - the batch is split in two halves that run on two streams (the L1 terms are batch means)
        half = B // 2 if HOOKS.H.vgg_split else 0      # (HDRSKY_VGG_SPLIT=0: one pass over the whole batch on stream 1, A/B hook)

        @seg("loss_vgg", 1, ["fwd_blend", "vgg_target"])
        def _():
            n = half if half > 0 else B
            self._vgg_loss_and_grad(T["y_gamma"][:n], [p[:n] for p in T["vgg_tgt"]], n / B, out=T["dyg"][:n])

        if half > 0:
            @seg("loss_vgg_b", 2, ["fwd_blend"])
            def _():
                self._vgg_loss_and_grad(T["y_gamma"][half:], [p[half:] for p in T["vgg_tgt"]], (B - half) / B,
                                        out=T["dyg"][half:])

        @seg("loss_adv", 0)
        def _():       # adversarial term: discriminator with inference-mode BN (train.py:302)
            loss_main()
            cvo = c["dis.out"]
            Rg = self._down_stack("dis.", self.ds.w, K.concat2(T["ldr"], T["y_lin"]), training=False,
                                  eval_affine=T["dis_eval_affine"])
            logits, _ = cvo.fwd(Rg["d4"]["raw"], Rg["xf_out"], cp)
            dlog = K.mse(logits, 1.0, 1.0, 1.0, self.losses[4:5])
            dact4 = cvo.dgrad(Rg["d4"]["raw"], dlog, cp)
            din = self._down_stack_bwd("dis.", self.ds.w, None, Rg, dact4, training=False, want_input_grad=True, do_wgrad=False)
            T["din_adv"] = din        # (channels 3..5 = d / d prediction: picked up by the head's backward launch)

        # ------------------------------------------------------------------ backward, first stretch
        @seg("bwd_head", 0, ["loss_vgg", "loss_vgg_b"] if half > 0 else ["loss_vgg"])
        def _():       # blend -> decoder tails -> sun radiance head -> dcmf complete -> sun-pose Dense layers
            t, dyl = T["t"], T["dyl"]
            (yf, rf), (yu, ru) = (T["dec_f"][6], T["dec_f"][7]), (T["dec_u"][6], T["dec_u"][7])
            out = None
            if dec_pair:      # the two tails' gradients as the halves of one paired tensor (the paired data gradient reads it)
                T["dc_pair"] = torch.empty((2 * B,) + tuple(yf.shape[1:]), dtype=torch.float32, device=self.device)
                out = (T["dc_pair"][:B], T["dc_pair"][B:], torch.empty_like(yu))
            dc_f, dc_u, dres_u = K.head_bwd(T["y_gamma"], T["alpha"], T["dyg"], dyl, T["din_adv"], yf, rf, yu, ru, out=out)
            tails = T["tails"] = {"f": (dc_f, None), "u": (dc_u, dres_u)}
            T["dpre"] = K.sun_rad_bwd(t["cmf"], t["gmax"], T["gamma"], T["beta"], tails["u"][1], T["dcmf"], sync=self.sync)

        # The sun-pose net's soft-max and Dense layers backwards (cmf is an input of a sunpose='external' step: no consumer).
        # Nothing on the decoder / res-block / encoder chain reads their results - the sun-pose conv layers (stream 2), the Dense
        # weight gradients and the Dense update (stream 1) do: in front of bwd_sunpose on stream 2 they are ~75 us (five launches,
        # two of them streaming the 67 MB Dense image) that the dependent chain on stream 0 no longer waits for.
        # HDRSKY_BWD_DENSE_STREAM=0: on stream 0 behind bwd_head, the old order (A/B hook)
        @seg("bwd_dense", HOOKS.H.bwd_dense_stream, ["bwd_head"])
        def _():
            t = T["t"]
            dz = T["dz"] = K.softmax_bwd(t["cmf"], T["dcmf"], t["z"])       # KL + the sun-radiance path meet in dcmf
            df1 = T["df1"] = K.fc_dgrad_fin(dz, self.fc2, cp, mask_src=t["f1"])
            T["dP3"] = K.fc_dgrad_fin(df1, self.fc1, cp).reshape(B, h // 8, wd // 8, 128)

        # ------------------------------------------------------------------ discriminator step (train.py:351-380)
        @seg("disc_step", 1, ["loss_adv", "disc_real"] if split_disc else ["loss_adv"])
        def _():       # real and generated pass as one batch of 2B (after the inference-mode call of loss_adv)
            cvo = c["dis.out"]
            if split_disc:     # the generated half; the real half's moving-statistics update first (program order)
                self._moving_update("dis.", self.ds.w, T["disc_real"])
                if T["disc_real_wg"]:      # (a launch of its own: both halves accumulate into the same gradients)
                    K.conv2d_wgrad_multi(T["disc_real_wg"])
                Rf = self._down_stack("dis.", self.ds.w, K.concat2(T["ldr"], T["y_lin"]), training=True)
                lg, _ = cvo.fwd(Rf["d4"]["raw"], Rf["xf_out"], cp)
                dl = torch.empty_like(lg)
                K.mse(lg, 0.0, 1.0, 0.5, self.losses[5:6], out=dl)      # generated (train.py:365)
                self._wg("dis.out", Rf["d4"]["raw"], Rf["xf_out"], dl)
                da4 = cvo.dgrad(Rf["d4"]["raw"], dl, cp)
                self._down_stack_bwd("dis.", self.ds.w, self.ds.g, Rf, da4, training=True, want_input_grad=False)
                self._flush_wgrads()
                return
            x2 = torch.empty((2 * B,) + tuple(T["ldr"].shape[1:3]) + (6,), dtype=torch.float32, device=self.device)
            K.concat2(T["ldr"], T["hdr_t"], out=x2[:B]); K.concat2(T["ldr"], T["y_lin"], out=x2[B:])
            Rd = self._down_stack_pair("dis.", self.ds.w, x2)
            lg, _ = cvo.fwd(Rd["d4"]["raw"], Rd["xf_out"], cp)
            dl = torch.empty_like(lg)
            K.mse(lg[:B], 1.0, 1.0, 0.5, self.losses[6:7], out=dl[:B])      # real      (train.py:364)
            K.mse(lg[B:], 0.0, 1.0, 0.5, self.losses[5:6], out=dl[B:])      # generated (train.py:365)
            self._wg("dis.out", Rd["d4"]["raw"], Rd["xf_out"], dl)
            da4 = cvo.dgrad(Rd["d4"]["raw"], dl, cp)
            self._down_stack_pair_bwd("dis.", self.ds.w, self.ds.g, Rd, da4)
            self._flush_wgrads()

        # ------------------------------------------------------------------ sun-pose conv layers (sunpose_net.py:54-62)
        @seg("bwd_sunpose", 2, ["bwd_dense"])
        def _():
            t, dP = T["t"], T["dP3"]
            if self.da_sun:
                self._sunpose_bwd_da(t, dP, B)
                return
            for l in (3, 2, 1):
                n = "sun.sunlayer%d" % l
                if l == 3 and "s3" in t:
                    dP = self._sun3_bwd(t, dP, B)
                    continue
                dr2 = self._in_bwd(t["r%db" % l], t["st%db" % l], n + ".norm2", 0.0, dP, pooled=True)
                self._wg(n + ".conv2", t["r%da" % l], t["xf%d" % l], dr2)
                da = c[n + ".conv2"].dgrad(t["r%da" % l], dr2, cp, out_bf16=self._nab_bf16())
                dr1 = self._in_bwd(t["r%da" % l], t["st%da" % l], n + ".norm1", 0.0, da)
                self._wg(n + ".conv1", t["in%d" % l], None, dr1)
                if l > 1:
                    dP = c[n + ".conv1"].dgrad(t["in%d" % l], dr1, cp, out_bf16=self._nab_bf16())
            self._norm_grads("bwd_sunpose", B)
            self._flush_wgrads()

        # ------------------------------------------------------------------ generator backward; its weight gradients
        # are launched in groups on stream 3 as soon as each stretch of the chain has produced their operands
        @seg("bwd_dec", 0)
        def _():
            dres = T["dres"] = K.zero_(torch.empty_like(T["x"][-1]))
            for sfx in ("f", "u") if self.da_dec else ():
                d3, s3, u3, d2, s2, xf1, y, residual, u2 = T["dec_" + sfx]
                dc = T["tails"][sfx][0]
                self._wg("gen.conv1_" + sfx, d2, xf1, dc)
                da2 = c["gen.conv1_" + sfx].dgrad(d2, dc, cp)
                dd2 = self._in_bwd(d2, s2, "gen.norm2_" + sfx, 0.1, da2, f32=self._da_f32(c["gen.conv2_" + sfx], d2))
                self._wg_da("gen.conv2_" + sfx, u2, dd2)
                du2 = K.da_conv2d_dgrad(dd2, c["gen.conv2_" + sfx].pkT, self._da(u2.shape[1], u2.shape[2])[1], 3, cp)
                dd3 = self._in_bwd(d3, s3, "gen.norm3_" + sfx, 0.1, K.up2x_bwd(du2), f32=self._da_f32(c["gen.conv3_" + sfx], d3))
                self._wg_da("gen.conv3_" + sfx, u3, dd3)
                du3 = K.da_conv2d_dgrad(dd3, c["gen.conv3_" + sfx].pkT, self._da(u3.shape[1], u3.shape[2])[1], 3, cp)
                K.up2x_bwd(du3, 1.0, out=dres)
            if dec_pair:
                PR, nab, sums = T["dec_pair"], self._nab_bf16(), self._norm_state(B)[0]
                c1f, c1u, c2f, c2u, c3f, c3u = (c["gen.conv%d_%s" % (k, sfx)] for k in (1, 2, 3) for sfx in ("f", "u"))
                d2, s2, d3, s3, u2, u3 = PR["d2"], PR["s2"], PR["d3"], PR["s3"], PR["u2"], PR["u3"]
                for i, sfx in enumerate(("f", "u")):
                    self._wg("gen.conv1_" + sfx, T["dec_" + sfx][3], T["dec_" + sfx][5], T["tails"][sfx][0])
                K.label("gen.conv1_f + gen.conv1_u (data gradients)")
                da2, _ = K.conv2d_dgrad(T["dc_pair"], c1f.pkT, K.conv_desc(2 * B, d2.shape[1], d2.shape[2], d2.shape[3], c1f.cout, c1f.kh, c1f.kw,
                                                                       1, True, 1), compute=cp, out_bf16=nab, pair=K.ConvPair(c1u.pkT))
                dd2 = K.norm_act_bwd(d2, s2, w["gen.norm2_f.gamma"], w["gen.norm2_f.beta"], 0.1, da2, False, sums=sums["gen.norm2_f+u"],
                                     out_bf16=self._act_bf16(), pair=(w["gen.norm2_u.gamma"], w["gen.norm2_u.beta"]))
                for i, sfx in enumerate(("f", "u")):
                    self._wg_plain("gen.conv2_" + sfx, u2[i * B:(i + 1) * B], dd2[i * B:(i + 1) * B])
                K.label("gen.conv2_f + gen.conv2_u (data gradients)")
                g2, _ = K.conv2d_dgrad(dd2, c2f.pkT, K.conv_desc(2 * B, u2.shape[1], u2.shape[2], u2.shape[3], c2f.cout, c2f.kh, c2f.kw, 1, True, 1),
                                       compute=cp, out_bf16=nab, pair=K.ConvPair(c2u.pkT))
                da3 = K.up2x_bwd(g2, 1.0)
                dd3 = K.norm_act_bwd(d3, s3, w["gen.norm3_f.gamma"], w["gen.norm3_f.beta"], 0.1, da3, False, sums=sums["gen.norm3_f+u"],
                                     out_bf16=self._act_bf16(), pair=(w["gen.norm3_u.gamma"], w["gen.norm3_u.beta"]))
                for i, sfx in enumerate(("f", "u")):
                    self._wg_plain("gen.conv3_" + sfx, u3, dd3[i * B:(i + 1) * B])
                K.label("gen.conv3_f + gen.conv3_u (data gradients)")
                g3, _ = K.conv2d_dgrad(dd3, c3f.pkT, K.conv_desc(2 * B, u3.shape[1], u3.shape[2], u3.shape[3], c3f.cout, c3f.kh, c3f.kw, 1, True, 1),
                                       compute=cp, out_bf16=nab, pair=K.ConvPair(c3u.pkT))
                K.up2x_bwd(g3, 1.0, out=dres, pair_sum=True)      # both decoders' gradients w.r.t. the encoder output they share
            for sfx in () if (self.da_dec or dec_pair) else ("f", "u"):
                d3, s3, xf2, d2, s2, xf1, y, residual, u3, u2 = T["dec_" + sfx]
                dc = T["tails"][sfx][0]
                self._wg("gen.conv1_" + sfx, d2, xf1, dc)
                da2 = c["gen.conv1_" + sfx].dgrad(d2, dc, cp, out_bf16=self._nab_bf16())
                dd2 = self._in_bwd(d2, s2, "gen.norm2_" + sfx, 0.1, da2)
                if u2 is not None:       # weight gradients on the materialised (bf16, already resized) operands
                    self._wg_plain("gen.conv2_" + sfx, u2, dd2)
                else:
                    self._wg("gen.conv2_" + sfx, d3, xf2, dd2)
                da3 = c["gen.conv2_" + sfx].dgrad(d3, dd2, cp, up_bf16=self._nab_bf16())
                dd3 = self._in_bwd(d3, s3, "gen.norm3_" + sfx, 0.1, da3)
                if u3 is not None:
                    self._wg_plain("gen.conv3_" + sfx, u3, dd3)
                else:
                    self._wg("gen.conv3_" + sfx, T["x"][-1], None, dd3)
                c["gen.conv3_" + sfx].dgrad(T["x"][-1], dd3, cp, out=dres, up_bf16=self._nab_bf16())
            self._norm_grads("bwd_dec", B)
            T["wq_dec"] = self._take_wgrads()

        @seg("wg_dec", 1, ["bwd_dec"])
        def _():
            K.conv2d_wgrad_multi(T["wq_dec"])

        @seg("bwd_sunrad", 2, ["bwd_head"])   # independent of the decoder / res-block chain: off the main stream
        def _():       # sun radiance head (generator.py:158-169, sunrad_net.py:46-70)
            R = T["sunrad"]
            xf = R["xf_out"]
            dact4 = K.dense_heads_bwd(R["d4"]["raw"], xf.scale, xf.shift, 0.3, w["gen.sun.gamma.kernel"],
                                      w["gen.sun.beta.kernel"], T["dpre"], g["gen.sun.gamma.kernel"],
                                      g["gen.sun.beta.kernel"], g["gen.sun.gamma.bias"], g["gen.sun.beta.bias"])
            self._down_stack_bwd("gen.sun.", w, g, R, dact4, training=True, want_input_grad=False)
            T["wq_sunrad"] = self._take_wgrads()

        @seg("wg_sunrad", 2, ["bwd_sunrad"])
        def _():
            K.conv2d_wgrad_multi(T["wq_sunrad"])

        @seg("bwd_res", 0)
        def _():       # encoder res blocks (generator.py:26-35)
            dx = T["dres"]
            if self.use_resconv:
                # every data gradient carries the norm (+ activation) backward that follows it in its epilogue and hands
                # the next one a final bf16 operand; the identity branch's gradient stays fp32 (`skip`)
                dgb, reducer = self._rc_state(B)

                def nd(i, j, slope):
                    o, n = T["res%d" % i][j - 1], "gen.res.%d.norm%d" % (i, j)
                    return dict(xhat=o["xhat"], inv=o["inv"], gamma=w[n + ".gamma"], beta=w[n + ".beta"], slope=slope,
                                dgb=dgb[2 * i + j - 1])
                cur = K.resconv_bwd(None, None, skip=dx, norm=nd(5, 2, 1.0))          # through norm2 of the last block
                for i in range(5, -1, -1):
                    p = "gen.res.%d." % i
                    o1, _ = T["res%d" % i]
                    dc2 = cur["bf16"]
                    self._wg(p + "conv2", o1["bf16"], None, dc2)
                    dc1 = K.resconv_bwd(dc2, c[p + "conv2"].pkT, norm=nd(i, 1, 0.1))["bf16"]
                    self._wg(p + "conv1", T["xb"][i], None, dc1)
                    if i > 0:    # + identity branch, then through norm2 of the previous block
                        cur = K.resconv_bwd(dc1, c[p + "conv1"].pkT, skip=dx, norm=nd(i - 1, 2, 1.0), want_f32=True)
                        dx = cur["f32"]
                    else:
                        dx = K.resconv_bwd(dc1, c[p + "conv1"].pkT, skip=dx, want_f32=True, want_bf16=False)["f32"]
                reducer.run()
            if self.da:
                # y = G(x) W + b with G the bilinear gather.  dx = G^T (dY W^T): the transpose of the gather is again a
                # (table-driven) gather, one launch with the flipped filter image (hdrsky_da_conv2d_dgrad); dW = G^T dY is
                # queued like every other weight gradient - the twelve layers share one launch that recomputes G tile by tile
                for i in range(5, -1, -1):
                    p = "gen.res.%d." % i
                    c1, t1, a1, c2, t2 = T["res%d" % i]
                    dr2 = self._in_bwd(c2, t2, p + "norm2", 1.0, dx, f32=self._da_f32(c[p + "conv2"], c2))
                    self._wg_da(p + "conv2", a1, dr2)
                    da1 = K.da_conv2d_dgrad(dr2, c[p + "conv2"].pkT, self._da_table, 3, cp)
                    dr1 = self._in_bwd(c1, t1, p + "norm1", 0.1, da1, f32=self._da_f32(c[p + "conv1"], c1))
                    self._wg_da(p + "conv1", T["x"][i], dr1)
                    dxx = K.da_conv2d_dgrad(dr1, c[p + "conv1"].pkT, self._da_table, 3, cp)
                    dx = K.axpby(dx, 1.0, dxx, 1.0)                                     # + identity branch
                self._norm_grads("bwd_res", B)
            for i in range(-1 if (self.use_resconv or self.da) else 5, -1, -1):
                p = "gen.res.%d." % i
                r1, t1, xf, r2, t2 = T["res%d" % i]
                dr2 = self._in_bwd(r2, t2, p + "norm2", 1.0, dx)
                self._wg(p + "conv2", r1, xf, dr2)
                da1 = c[p + "conv2"].dgrad(r1, dr2, cp)
                dr1 = self._in_bwd(r1, t1, p + "norm1", 0.1, da1)
                self._wg(p + "conv1", T["x"][i], None, dr1)
                dx = c[p + "conv1"].dgrad(T["x"][i], dr1, cp, residual=dx)      # + identity branch
            if not self.use_resconv and not self.da:
                self._norm_grads("bwd_res", B)
            T["dx_enc"] = dx
            T["wq_res"] = self._take_wgrads()

        # (stream 1, behind wg_dec.  Re-timed whenever the balance of the tail changed: stream 2 was better (-0.5 %) while the
        # sun-pose Dense backward sat on the main chain; with that in front of bwd_sunpose on stream 2 (bwd_dense) stream 2 is the
        # last side stream to finish again and this segment on stream 1 makes the step 1.3 % shorter (2.736 against 2.772 ms, two
        # runs each).  Other orders of the tail that were timed - the sun-radiance backward in front of the sun-pose backward with
        # its weight gradients on stream 1, the Dense update at the end of stream 2, wg_res behind bwd_enc on stream 0, the
        # discriminator step split into an early real half and a generated half - all lengthened the step or changed nothing.
        # HDRSKY_WG_RES_STREAM / HDRSKY_APPLY_FC_STREAM / HDRSKY_BWD_DENSE_STREAM: A/B hooks.)
        @seg("wg_res", HOOKS.H.wg_res_stream, ["bwd_res"])
        def _():
            K.conv2d_wgrad_multi(T["wq_res"])

        # The encoder head closes the backward pass on stream 0.  The gradients of its two stride-2 layers are complete two
        # data gradients before the chain ends: their weight gradients (act_bf16 + conv_wgrad2 + reduce, ~100 us) run as
        # segment wg_enc on stream 1 - idle by then - beside the rest of the chain instead of behind it; only the stem's
        # (conv_wgrad3) stays at the end.  Rounds 3-4: HDRSKY_WG_ENC_SPLIT=1.  Round 5 (paired decoders: stream 0 is the FIRST to
        # finish its backward chain, stream 1 the last): one segment at the end of stream 0 again, the default; =1: A/B hook.
        split_enc = HOOKS.H.wg_enc_split

        @seg("bwd_enc", 0)
        def _():       # encoder head (generator.py:92-108)
            dc3 = self._in_bwd(T["c3"], T["s3"], "gen.norm3_d", 0.1, T["dx_enc"])
            self._wg("gen.conv3_d", T["c2"], T["xf3"], dc3)
            da2 = c["gen.conv3_d"].dgrad(T["c2"], dc3, cp, out_bf16=self._nab_bf16())
            dc2 = self._in_bwd(T["c2"], T["s2"], "gen.norm2_d", 0.1, da2)
            self._wg("gen.conv2_d", T["c1"], T["xf2"], dc2)
            T["dc2_enc"] = dc2
            if split_enc:
                T["wq_enc"] = self._take_wgrads()

        if split_enc:
            @seg("wg_enc", 1, ["bwd_enc"])
            def _():
                K.conv2d_wgrad_multi(T["wq_enc"])

        @seg("bwd_enc2", 0)
        def _():
            da1 = c["gen.conv2_d"].dgrad(T["c1"], T["dc2_enc"], cp, out_bf16=self._nab_bf16())
            dc1 = self._in_bwd(T["c1"], T["s1"], "gen.norm1_d", 0.1, da1)
            self._wg("gen.conv1_d", T["ldr"], None, dc1)
            self._norm_grads("bwd_enc", B)
            self._flush_wgrads()

        # Materialised Dense weight gradients (skipped on an updating step of a fused_dense trainer, see _skip)
        @seg("wg_dense", 1, ["bwd_dense"])
        def _():
            if self.dense_wgrad_external:
                return
            fn = K.fc_wgrad_bf16 if self.dense_mfma else K.fc_wgrad
            fn(T["t"]["f1"], T["dz"], g["sun.fc2.kernel"], g["sun.fc2.bias"])
            fn(T["t"]["flat"], T["df1"], g["sun.fc1.kernel"], g["sun.fc1.bias"])

        # The two Dense layers hold 50.3 M of the 58.3 M parameters and nothing reads their weights or gradients after
        # bwd_dense: their RMSprop update and bf16 re-packing run here, beside the rest of the backward pass, instead of
        # at the end of the step.  (Data-parallel: after the all-reduce of that slice / the all-gather of the operands.)
        # (Round 2 kept it on stream 1 beside the backward chains.  Since the weight-gradient segments moved there it sat at
        # the END of stream 1, with the conv-side update waiting behind it: now it closes stream 2 - idle from wg_sunrad on -
        # and `apply` no longer waits for it (disjoint parameters): the two updates overlap, step -1 %
        # (profiles/r03_plan_ab2.txt; HDRSKY_APPLY_FC_STREAM / HDRSKY_APPLY_AFTER_FC are the A/B hooks))
        @seg("apply_fc", 3 if HOOKS.H.apply_fc_cus else HOOKS.H.apply_fc_stream, ["bwd_dense", "wg_dense"])
        def _():
            fc0, fc1 = self.fc_grad_range()
            if defer:      # the operands' images + the bias vectors' step; the kernels' update opens the next replay (apply_fc_run)
                flat, df1, f1, dz = self.dense_operands or (T["t"]["flat"], T["df1"], T["t"]["f1"], T["dz"])
                for name, pf, x_, dy_ in (("sun.fc2", self.fc2, f1, dz), ("sun.fc1", self.fc1, flat, df1)):
                    ob, nb_, _ = self.gs.offsets[name + ".bias"]
                    K.rmsprop_fc_fused_prepare(x_, dy_, pf.K, pf.N, self.lr, self._fc_ws[name][1], g[name + ".bias"], gscale=self._gscale,
                                               bias=self.gs.flat[ob:ob + nb_], bias_ms=self.gs.ms[ob:ob + nb_])
                return
            if self.fused_dense:
                flat, df1, f1, dz = self.dense_operands or (T["t"]["flat"], T["df1"], T["t"]["f1"], T["dz"])
                for name, pf, x_, dy_ in (("sun.fc2", self.fc2, f1, dz), ("sun.fc1", self.fc1, flat, df1)):
                    o, n, shape = self.gs.offsets[name + ".kernel"]
                    ob, nb_, _ = self.gs.offsets[name + ".bias"]      # (the bias vector's step rides in the operand launch)
                    K.rmsprop_fc_fused(w[name + ".kernel"], self.gs.ms[o:o + n].view(shape), x_, dy_, pf, self.lr,
                                       db=g[name + ".bias"], gscale=self._gscale,
                                       bias=self.gs.flat[ob:ob + nb_], bias_ms=self.gs.ms[ob:ob + nb_])
                return
            if self.precise:      # BF16X3 keeps residual planes: plain update, then re-pack
                K.rmsprop(self.gs.flat[fc0:fc1], self.gs.grad[fc0:fc1], self.gs.ms[fc0:fc1], self.lr, gscale=self._gscale)
                self.fc1.repack(w["sun.fc1.kernel"])
                self.fc2.repack(w["sun.fc2.kernel"])
                return
            for name, pf in (("sun.fc1", self.fc1), ("sun.fc2", self.fc2)):
                o, n, shape = self.gs.offsets[name + ".kernel"]
                K.rmsprop_fc(w[name + ".kernel"], g[name + ".kernel"], self.gs.ms[o:o + n].view(shape), pf, self.lr,
                             gscale=self._gscale)
                o, n, _ = self.gs.offsets[name + ".bias"]
                K.rmsprop(self.gs.flat[o:o + n], self.gs.grad[o:o + n], self.gs.ms[o:o + n], self.lr, gscale=self._gscale)

        # every gradient is complete here: a data-parallel driver hooks its all-reduce onto this (empty) segment
        segs.append(("grads_ready", 0, ("disc_step", "wg_dec", "bwd_sunpose", "wg_res", "wg_sunrad") + (("wg_enc",) if split_enc else ()), None))

        # ------------------------------------------------------------------ optimizers (train.py:403,406)
        # (starts behind grads_ready, beside the Dense update; HDRSKY_APPLY_AFTER_FC=1: the round-2 order, A/B hook)
        @seg("apply", 0, ["apply_fc"] if HOOKS.H.apply_after_fc else [])
        def _():
            gscale, fc0 = self._gscale, self.fc_grad_range()[0]
            if fc0 % 4 == 0 and self.ds.ntrain % 4 == 0:      # both optimizers' conv-side parameters: one launch
                K.rmsprop2(self.gs.flat[:fc0], self.gs.grad[:fc0], self.gs.ms[:fc0],
                           self.ds.flat[:self.ds.ntrain], self.ds.grad, self.ds.ms, self.lr, gscale=gscale)
            else:
                K.rmsprop(self.gs.flat[:fc0], self.gs.grad[:fc0], self.gs.ms[:fc0], self.lr, gscale=gscale)
                K.rmsprop(self.ds.flat[:self.ds.ntrain], self.ds.grad, self.ds.ms, self.lr, gscale=gscale)
            self.repack(fc=False)

        # HDRSKY_PLAN_MOVE="name=stream@after,...": scheduling experiments - segment `name` goes to `stream`, enqueued right
        # behind segment `after` (dependencies are unchanged: only where it waits changes)
        for ent in filter(None, HOOKS.H.plan_move.split(",")):
            name, rest = ent.split("=")
            si, after = rest.split("@")
            idx = [k for k, sg in enumerate(segs) if sg[0] == name]
            if not idx or after not in [sg[0] for sg in segs]:
                continue
            sg = segs.pop(idx[0])
            at = [k for k, q in enumerate(segs) if q[0] == after][0]
            segs.insert(at + 1, (sg[0], int(si), sg[2], sg[3]))

        # HDRSKY_PLAN_DEPS="name:dep,...": scheduling experiments - segment `name` additionally waits for segment `dep`
        for ent in filter(None, HOOKS.H.plan_deps.split(",")):
            name, dep = ent.split(":")
            order = [sg[0] for sg in segs]
            if name not in order or dep not in order or order.index(dep) >= order.index(name):
                # (a dependency must be ENQUEUED before its waiter: the wait is on an event the earlier segment records)
                raise ValueError("HDRSKY_PLAN_DEPS: %r is not a segment pair in enqueue order (%s)" % (ent, ", ".join(order)))
            k = order.index(name)
            segs[k] = (segs[k][0], segs[k][1], tuple(segs[k][2]) + (dep,), segs[k][3])

        if self.ext_sun:          # no sun-pose net: its backward, Dense weight gradients and Dense optimizer segments go
            gone = ("bwd_dense", "bwd_sunpose", "wg_dense", "apply_fc")
            segs[:] = [(n, si, tuple(d for d in deps if d not in gone), fn) for n, si, deps, fn in segs if n not in gone]

        # Default merges (round 5): the sun-side backward chain of stream 2 (the step's critical chain) as ONE graph - single GPU only:
        # a data-parallel exchange starts the Dense slice's collective behind bwd_dense - and the two encoder segments that close
        # the main chain; nothing outside waits for a member of either before its last one.  Step -1.0 % (profiles/
        # r05_plan_merge_ab.txt).  NOT merged: bwd_dec + bwd_res and wg_dec + wg_res - another -0.3 % on the plain 32x128 plan,
        # where the decoders' weight gradients wait for the discriminator step anyway, but +2.5 ... +4.5 % with distortion-aware
        # layers, whose weight gradients are due as soon as bwd_dec is done (profiles/r05_plan_merge_da_ab.txt).
        if HOOKS.H.plan_merge == "auto":
            auto = (["bwd_dense+bwd_sunpose+bwd_sunrad+wg_sunrad"] if self.world == 1 else []) + \
                   ["bwd_sunrad+wg_sunrad", "bwd_enc+bwd_enc2"]
            self._merge_plan(segs, ",".join(auto), strict=False)
        else:
            self._merge_plan(segs, HOOKS.H.plan_merge)
        return segs

    @staticmethod
    def _merge_plan(segs, spec, strict=True):
        """HDRSKY_PLAN_MERGE / the default merges: the segments a+b+c of ONE stream become one segment - one hipGraph replay instead
        of three (a graph launch costs the stream ~13-19 us of idle time: profiles/r05_segment_timeline.txt).  The merged segment
        keeps the first name, waits for every member's outside dependencies and sits where the LAST member sat in the enqueue
        order (all those dependencies are enqueued by then); segments that waited for a member wait for the merged one.  Refused:
        members on different streams, another segment of that stream between them, a segment between them that waits for one."""
        for grp in filter(None, spec.split(",")):
            names = grp.split("+")
            order = [sg[0] for sg in segs]
            if any(n not in order for n in names):
                continue                          # (a plan without these segments: the sun-pose trainer, the external sun-pose net)
            idx = [order.index(n) for n in names]
            si = segs[idx[0]][1]
            between = [sg for k, sg in enumerate(segs) if idx[0] < k < idx[-1] and k not in idx]
            if idx != sorted(idx) or any(segs[k][1] != si for k in idx) or any(sg[1] == si for sg in between) or \
               any(set(sg[2]) & set(names) for sg in between):
                if not strict:
                    continue                      # (a default merge that another hook's plan does not allow)
                raise ValueError("HDRSKY_PLAN_MERGE: %r cannot be merged in this plan" % grp)
            fns = [segs[k][3] for k in idx]
            deps = tuple(d for k in idx for d in segs[k][2] if d not in names)
            merged = (names[0], si, tuple(dict.fromkeys(deps)), (lambda fns=fns: [f() for f in fns if f is not None] and None))
            last = idx[-1]
            segs[last] = merged
            for k in reversed(idx[:-1]):
                del segs[k]
            for k, sg in enumerate(segs):         # waiters of a member wait for the merged segment
                if set(sg[2]) & set(names[1:]):
                    segs[k] = (sg[0], sg[1], tuple(dict.fromkeys(names[0] if d in names[1:] else d for d in sg[2])), sg[3])

    GRADS_READY = "grads_ready"                                  # hook point of a data-parallel driver
    DISC_GRADS_READY = "disc_step"                               # the discriminator's gradients are complete behind it
    @property
    def APPLY(self):                                             # the optimizer segments
        return ("apply",) if self.ext_sun else ("apply_fc", "apply")

    @property
    def FC_GRADS_READY(self):
        """Segment after which the Dense slice of the gradients (fused_dense: the operands of its contraction) exists."""
        return "bwd_dense" if self.fused_dense or self.dense_wgrad_external else "wg_dense"

    def _skip(self, update):
        """Segments an `update` / gradient-only step leaves out (apply_fc_run: only while a deferred Dense update is pending)."""
        late = () if self._fc_pending else ("apply_fc_run",)
        if not update:
            return tuple(self.APPLY) + late
        return (("wg_dense",) if self.fused_dense else ()) + late

    @property
    def _defer(self):
        """The Dense update of a captured step is deferred into the next replay (see __init__: defer_dense)."""
        return bool(getattr(self, "defer_dense", False)) and self.fused_dense and not self.ext_sun

    def flush(self):
        """Applies a deferred Dense update now (defer_dense): after it the trainer's weights are those the reference holds after
        optimizer.apply_gradients.  A no-op otherwise."""
        if not self._fc_pending:
            return
        captured = self._graphs is not None and getattr(self, "_captured", None) is not None and self._T is self._captured[0]
        self._execute(["apply_fc_run"], graphs=self._graphs if captured else None)
        self._fc_pending = False

    def _take_wgrads(self):
        return self._wjobs.pop(torch.cuda.current_stream().cuda_stream, [])

    def _execute(self, names=None, graphs=None, hooks=None, pre_hooks=None):
        """Enqueues the plan's segments (all, or those in `names`) on the three streams; `graphs` replays captured
        segments instead of re-issuing their launches.  pre_hooks[name]() / hooks[name]() run on the segment's stream
        right before / after it."""
        caller = torch.cuda.current_stream()
        st = self._streams
        for s in st:
            s.wait_stream(caller)
        on_stream = {name: si for name, si, _, _ in self._segs}
        for name, si, deps, fn in self._segs:
            if names is not None and name not in names:
                continue
            s = st[si]
            for d_ in deps:
                # (a dependency on the same stream is its order: no event.  Found while capturing the whole step as ONE
                # hipGraph: there a wait for an event recorded on the waiting stream itself crashes hipStreamEndCapture on
                # ROCm 7.2; that experiment - profiles/LABNOTES.md r3 section 5, third part - replayed at the SUM of the kernel times and was dropped)
                if d_ in self._events and on_stream.get(d_) != si:
                    s.wait_event(self._events[d_])
            with torch.cuda.stream(s):
                if pre_hooks and name in pre_hooks:
                    pre_hooks[name]()
                if fn is None:
                    pass
                elif graphs is not None:
                    graphs[name].replay()
                else:
                    fn()
                if hooks and name in hooks:
                    hooks[name]()
                ev = self._events.get(name)
                if ev is None:
                    ev = self._events[name] = torch.cuda.Event()
                ev.record(s)
        for s in st:
            caller.wait_stream(s)

    def event(self, name):
        """The HIP event recorded after segment `name` in the current step (for drivers that order their own work)."""
        return self._events[name]

    def _bind(self, ldr, hdr_t, sunpose_gt, cmf=None, cams=None):
        if self.ext_sun:
            B, h, w = ldr.shape[0], self.h, self.w
            if cmf is None or cams is None or len(cams) != 3:
                raise ValueError("sunpose='external': step(..., cmf=[B,H*W], cams=(cam1, cam2, cam3)) expected")
            want = [(B, h * w), (B, h, w, 1), (B, h // 2, w // 2, 1), (B, h // 4, w // 4, 1)]
            for t, shp in zip((cmf,) + tuple(cams), want):
                if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous():
                    raise ValueError("sunpose='external': expected contiguous fp32 %s, got %s" % (shp, tuple(t.shape)))
        elif cmf is not None or cams is not None:
            raise ValueError("cmf / cams are inputs of a sunpose='external' trainer only")
        if self.on_bind is not None:
            self.on_bind(ldr.shape[0])
        self._norm_state(ldr.shape[0])            # pointer tables are uploaded here, never inside a graph capture
        if self.use_resconv:
            self._rc_state(ldr.shape[0])
        self._T = dict(ldr=ldr, hdr_t=hdr_t, gt=sunpose_gt)
        if self.ext_sun:
            self._T.update(cmf_in=cmf, cams_in=tuple(cams))
        self._segs = self._plan()
        self._events = {}

    def _outputs(self):
        T = self._T
        return dict(y_final_lin=T["y_lin"], y_final_gamma=T["y_gamma"], sky_pred_lin=T["sky_lin"], sun_pred_lin=T["sun_lin"],
                    gamma=T["gamma"], beta=T["beta"], alpha_c3=T["alpha"], sunpose_cmf=T["t"]["cmf"],
                    sun_cam1=T["cams"][0], sun_cam2=T["cams"][1], sun_cam3=T["cams"][2], sun_rad_lin=T["rad_lin"])

    def step(self, ldr, hdr_t, sunpose_gt, update=True, cmf=None, cams=None):
        """ldr / hdr_t [B,H,W,3] BGR (train.py:386-387 rgb2bgr already applied), sunpose_gt [B,H*W]; cmf / cams: the
        sun-pose net's outputs for a sunpose='external' trainer.
        Returns the dict generator_in_step returns (train.py:349) - losses are in self.losses (device)."""
        self.flush()                      # (a pending deferred update belongs to the previous binding's workspace)
        self._bind(ldr, hdr_t, sunpose_gt, cmf, cams)
        skip = self._skip(update)
        self._execute([n for n, *_ in self._segs if n not in skip])
        if update and self._defer:      # an eager step is never left half applied
            self._fc_pending = True
            self.flush()
        return self._outputs()

    def test_step(self, ldr, hdr_t, sunpose_gt, cmf=None, cams=None):
        """`test_step` (train.py:417-442): the validation pass - the training graph (ground-truth-bin Grad-CAM pick) with
        every BatchNorm in inference mode, all loss terms (generator_in_step / discriminator_in_step with
        training=False) and no update.  Returns the output dict of `step`; the loss terms are in self.losses /
        loss_dict().  Issued eagerly (not part of the captured step); gradients buffers are left zeroed / partial."""
        self.flush()
        self._bn_training = False
        saved = (getattr(self, "_T", None), getattr(self, "_segs", None), getattr(self, "_events", None))
        try:
            self._bind(ldr, hdr_t, sunpose_gt, cmf, cams)
            self._execute(["zero", "fwd_sun", "fwd_sun_fc", "fwd_enc", "vgg_target", "fwd_blend", "loss_vgg", "loss_vgg_b", "loss_adv"])
            T, cvo = self._T, self.conv["dis.out"]
            for other, target, slot in ((hdr_t, 1.0, 6), (T["y_lin"], 0.0, 5)):        # train.py:351-369, training=False
                R = self._down_stack("dis.", self.ds.w, K.concat2(ldr, other), training=False)
                logits, _ = cvo.fwd(R["d4"]["raw"], R["xf_out"], self.compute)
                K.mse(logits, target, 1.0, 0.5, self.losses[slot:slot + 1])
            outs = self._outputs()
        finally:
            self._bn_training = True
            if self._graphs is not None and saved[1] is not None:
                # a captured step keeps replaying ITS plan (segment list, static tensors, events): the validation plan -
                # possibly another batch size, hence other segments - must not replace it
                self._T, self._segs, self._events = saved
        return outs

    def refresh_eval(self):
        """Hook for callers that changed the BatchNorm moving statistics from outside (parallel.sync_moving_stats_):
        nothing is cached on this side - the inference-mode affines are recomputed from the flat buffers at every use."""
        return None

    def apply_gradients(self, gscale=None):
        """optimizer_gen / optimizer_disc .apply_gradients (train.py:403,406): RMSprop(lr) on gradients scaled by
        gscale (default 1/world: data-parallel sum -> mean), then refresh the packed bf16 weight images."""
        if gscale is not None:
            self._gscale = float(gscale)
        self._execute(self.APPLY)
        if self._defer:
            self._fc_pending = True
            self.flush()

    def fc_grad_range(self):
        """[start, end) of the two Dense layers' gradients inside gs.grad - contiguous, the last trainables of the
        sun-pose net (50.3 M of the 58.3 M parameters); complete once segment FC_GRADS_READY has run, so a data-parallel
        driver starts their all-reduce there (hook) and it overlaps the rest of the backward pass."""
        if self.ext_sun:
            return self.gs.ntrain, self.gs.ntrain
        o = self.gs.offsets["sun.fc1.kernel"][0]
        return o, self.gs.ntrain

    def capture(self, ldr, hdr_t, sunpose_gt, warmup=2, cmf=None, cams=None):
        """Captures every segment of the step on (ldr, hdr_t, sunpose_gt[, cmf, cams]) - static input buffers the caller
        refills - into its own hipGraph.  `replay()` then runs one step."""
        if self.sync is not None:
            raise RuntimeError("Trainer.capture: a step with batch statistics over several replicas (parallel.BatchSync) holds "
                               "collectives inside its segments - issue it eagerly (step / reduce_all / apply_gradients)")
        self.flush()
        self._bind(ldr, hdr_t, sunpose_gt, cmf, cams)
        # the warm-up steps (lazy kernel attributes, allocator) must not train: weights, RMSprop slots and BatchNorm
        # moving statistics are put back afterwards (and replicas of a data-parallel job stay identical)
        state = [(t, t.clone()) for t in (self.gs.flat, self.gs.ms, self.ds.flat, self.ds.ms)]
        for _ in range(warmup):
            self._execute([n for n, *_ in self._segs if n != "apply_fc_run"])
            if self._defer:      # (the deferred half, so that its launches are warm too)
                self._execute(["apply_fc_run"])
        torch.cuda.synchronize()
        for t, saved in state:
            t.copy_(saved)
        del state
        self._fc_pending = False
        self.repack()
        torch.cuda.synchronize()
        self._graphs = {}
        with K.no_gc():      # (no finaliser of an old graph / event in the middle of a capture)
            for name, si, deps, fn in self._segs:
                if fn is None:
                    continue
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=self._streams[si], capture_error_mode="thread_local"):
                    fn()
                self._graphs[name] = gr
        torch.cuda.synchronize()
        self._captured = (self._T, self._segs)    # an eager step() / test_step() in between re-binds both (see replay)
        return self._outputs()

    def replay(self, update=True, hooks=None, pre_hooks=None):
        """One step from the captured graphs; hooks / pre_hooks: {segment name: callable run on that segment's stream
        after / before it}."""
        if getattr(self, "_graphs", None) is None:
            raise RuntimeError("Trainer.replay: capture() first")
        if self._T is not self._captured[0]:      # an eager pass ran since: the graphs (and outputs()) belong to the captured binding
            self._T, self._segs = self._captured
            self._events = {}
        skip = self._skip(update)
        self._execute([n for n, *_ in self._segs if n not in skip], graphs=self._graphs, hooks=hooks, pre_hooks=pre_hooks)
        if self._defer:      # apply_fc_run (if it was pending) has run; an updating step leaves a new one
            self._fc_pending = bool(update)

    def loss_dict(self):
        """Host copy of the loss terms with the reference's names (train.py:480-489) - synchronises."""
        v = dict(zip(LOSS_SLOTS, self.losses.tolist()))
        v["total_gen_loss"] = v["kl"] + 1000.0 * v["dog"] + v["adv"] + 10.0 * v["l1"] + 0.01 * v["perceptual"]
        v["total_disc_loss"] = 0.5 * (v["disc_generated"] + v["disc_real"])
        return v


class SunPoseTrainer(Trainer):
    """Sun-pose pre-training step (train_sun.py:220-264): loss = KLDivergence(gt, cmf) + DoG-L1 between the cmf image
    and the target image; tf.keras Adam(lr).  Re-uses the sun-pose forward / Grad-CAM / backward chains of `Trainer`;
    only the parameter set (sun-pose net alone), the loss head and the optimizer differ.  (The reference script
    reads `utils.utils.str2bool` and `args.dorfpath`, neither of which exists - train_sun.py:155,167; the step itself
    is what is mirrored here.)"""

    LOSSES = ("kl", "dog")

    def __init__(self, sun_params, device="cuda", lr=1e-4, im_height=32, im_width=128, precise=False, compute=BF16,
                 world_size=1, distortion_aware=False):
        self.device = torch.device(device)
        self.h, self.w = im_height, im_width
        self.lr, self.compute, self.precise, self.world = lr, compute, precise, world_size
        # distortion_aware: sunpose_net.py:11,16 - every sunposeLayer convolution is distortion_aware_ops.conv2d
        self.da_sun, self._da_geo = bool(distortion_aware), {}
        self.da_parts, self.sync = (("sunpose",) if self.da_sun else ()), None
        self.defer_dense, self._fc_pending, self.fused_dense, self.ext_sun = False, False, False, False
        hw = im_height * im_width
        self.dense_mfma = compute == BF16 and not precise and K.fc_xtdy_supported(hw // 64 * 128, hw) and K.fc_xtdy_supported(hw, hw)
        self.gs = FlatParams(OrderedDict(("sun." + k, v) for k, v in sun_params.items()), self.device)
        self.adam_m, self.adam_v = torch.zeros_like(self.gs.grad), torch.zeros_like(self.gs.grad)
        self.steps_done = 0
        self.losses = torch.zeros(len(self.LOSSES), dtype=torch.float32, device=self.device)
        self._wjobs, self._packer = {}, None
        w, c = self.gs.w, {}
        for l in (1, 2, 3):
            for j in (1, 2):
                n = "sun.sunlayer%d.conv%d" % (l, j)
                c[n] = _Conv(w[n + ".w"], w[n + ".b"], n + ".w", n + ".b", precise=precise, need_dgrad=not (l == 1 and j == 1))
        self.conv = c
        self.fc1, self.fc2 = PackedFC(w["sun.fc1.kernel"], precise), PackedFC(w["sun.fc2.kernel"], precise)
        if self.da_sun:
            self._init_da_sun()

    def step(self, ldr, sunpose_gt, update=True, want_cams=True, dog_weight=1.0):
        """ldr [B,H,W,3] BGR in [0,1], sunpose_gt [B,H*W].  Returns (pred [B,H,W,1], target image, [cam1,cam2,cam3]).
        dog_weight is 1 in the reference (train_sun.py:255)."""
        w, g, c, cp = self.gs.w, self.gs.g, self.conv, self.compute
        B = ldr.shape[0]
        K.zero_(self.gs.grad); K.zero_(self.losses)
        t = self._sunpose_forward(ldr)
        cams = self._gradcam(t, sunpose_gt) if want_cams else None              # under stop_recording: constants
        dcmf = K.kl(sunpose_gt, t["cmf"], self.losses[0:1])                     # d KL / d cmf
        pred, gt_img = t["cmf"].view(B, self.h, self.w, 1), sunpose_gt.view(B, self.h, self.w, 1)
        if dog_weight != 0.0:
            K.dog_loss(pred, gt_img, float(dog_weight), self.losses[1:2], dcmf.view(B, self.h, self.w, 1))
        dz = K.softmax_bwd(t["cmf"], dcmf, t["z"])
        fc_wgrad = K.fc_wgrad_bf16 if self.dense_mfma else K.fc_wgrad
        fc_wgrad(t["f1"], dz, g["sun.fc2.kernel"], g["sun.fc2.bias"])
        df1 = K.fc_dgrad_fin(dz, self.fc2, cp, mask_src=t["f1"])
        fc_wgrad(t["flat"], df1, g["sun.fc1.kernel"], g["sun.fc1.bias"])
        dP = K.fc_dgrad_fin(df1, self.fc1, cp).reshape(B, self.h // 8, self.w // 8, 128)
        for l in (3, 2, 1) if not self.da_sun else ():
            n = "sun.sunlayer%d" % l
            if l == 3 and "s3" in t:
                dP = self._sun3_bwd(t, dP, B)
                continue
            dr2 = self._in_bwd(t["r%db" % l], t["st%db" % l], n + ".norm2", 0.0, dP, pooled=True)
            self._wg(n + ".conv2", t["r%da" % l], t["xf%d" % l], dr2)
            da = c[n + ".conv2"].dgrad(t["r%da" % l], dr2, cp)
            dr1 = self._in_bwd(t["r%da" % l], t["st%da" % l], n + ".norm1", 0.0, da)
            self._wg(n + ".conv1", t["in%d" % l], None, dr1)
            if l > 1:
                dP = c[n + ".conv1"].dgrad(t["in%d" % l], dr1, cp)
        if self.da_sun:
            self._sunpose_bwd_da(t, dP, B)
        else:
            self._norm_grads("bwd_sunpose", B)
            self._flush_wgrads()
        if update:
            self.apply_gradients()
        return pred, gt_img, cams

    def apply_gradients(self, gscale=None):
        """optimizer_sun.apply_gradients (train_sun.py:259): Adam(lr) on gradients scaled by gscale (default 1/world)."""
        self.steps_done += 1
        K.adam(self.gs.flat[:self.gs.ntrain], self.gs.grad, self.adam_m, self.adam_v, self.lr, self.steps_done,
               gscale=(1.0 / self.world) if gscale is None else gscale)
        self.repack()

    def loss_dict(self):
        v = dict(zip(self.LOSSES, self.losses.tolist()))
        v["sun_loss"] = v["kl"] + v["dog"]
        return v
