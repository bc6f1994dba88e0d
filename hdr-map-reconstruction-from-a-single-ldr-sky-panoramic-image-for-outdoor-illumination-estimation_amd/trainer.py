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

import numpy as np
import torch

from . import _lib as L
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

    def repack(self):
        self.pk.repack(self.w)
        if self.pkT is not None:
            self.pkT.repack(self.w)

    def desc(self, x):
        B, H, W, C = x.shape
        return K.conv_desc(B, H, W, C, self.cout, self.kh, self.kw, self.stride, self.same, self.upsample)

    def fwd(self, x, xf=None, compute=BF16, **kw):
        return K.conv2d(x, self.pk, self.b, stride=self.stride, same=self.same, upsample=self.upsample, xf=xf,
                        compute=compute, **kw)

    def wgrad_job(self, x, xf, dy, gw, gb, compute):
        return K.wgrad_job(x, dy, self.kh, self.kw, gw, gb, stride=self.stride, same=self.same, upsample=self.upsample,
                           xf=xf, compute=compute)

    def dgrad(self, x, dy, compute, residual=None, want_stats=False, out=None):
        """Gradient wrt the (transformed, pre-resize) conv operand."""
        d, st = K.conv2d_dgrad(dy, self.pkT, self.desc(x), residual=None if self.upsample == 2 else residual,
                               compute=compute, want_stats=want_stats)
        if self.upsample == 2:
            d = K.up2x_bwd(d, 1.0, out=out)
        return (d, st) if want_stats else d


class Trainer:
    def __init__(self, gen_params, sun_params, dis_params, vgg_params, device="cuda", lr=1e-4, im_height=32,
                 im_width=128, precise=False, compute=BF16, world_size=1):
        self.device = torch.device(device)
        self.h, self.w = im_height, im_width
        self.lr, self.compute, self.precise, self.world = lr, compute, precise, world_size
        named = OrderedDict(("gen." + k, v) for k, v in gen_params.items())
        named.update(("sun." + k, v) for k, v in sun_params.items())
        self.gs = FlatParams(named, self.device)          # optimizer_gen: _gen + _sun variables (train.py:402-403)
        self.ds = FlatParams(OrderedDict(("dis." + k, v) for k, v in dis_params.items()), self.device)
        self.vgg = E._dev(vgg_params, self.device)
        self.side_stream = torch.cuda.Stream(device=self.device)
        self.side_stream2 = torch.cuda.Stream(device=self.device)
        self.losses = torch.zeros(len(LOSS_SLOTS), dtype=torch.float32, device=self.device)
        self._wjobs, self._wkeep = {}, []
        self.wgrad_stream = torch.cuda.Stream(device=self.device)
        self._build_layers()

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
        for l in (1, 2, 3):
            for j in (1, 2):
                n = "sun.sunlayer%d.conv%d" % (l, j)
                add(n, n + ".w", n + ".b", need_dgrad=not (l == 1 and j == 1))
        self.fc1 = PackedFC(w["sun.fc1.kernel"], pr)
        self.fc2 = PackedFC(w["sun.fc2.kernel"], pr)
        # frozen VGG16: forward filters + data-gradient filters
        self.vgg_pk, self.vgg_pkT = {}, {}
        for name, _, _ in P.VGG_CHANNELS:
            self.vgg_pk[name] = PackedConv(self.vgg[name + ".w"], pr)
            self.vgg_pkT[name] = PackedConv(self.vgg[name + ".w"], pr, transpose_flip=True)

    def repack(self):
        if getattr(self, "_packer", None) is None:
            pairs = []
            for cv in self.conv.values():
                pairs.append((cv.w, cv.pk))
                if cv.pkT is not None:
                    pairs.append((cv.w, cv.pkT))
            self._packer = K.MultiPacker(pairs)
        self._packer.run()
        self.fc1.repack(self.gs.w["sun.fc1.kernel"])
        self.fc2.repack(self.gs.w["sun.fc2.kernel"])

    # ---- small helpers ------------------------------------------------------------------------------
    def _inxf(self, stats, name, slope):
        w = self.gs.w
        return InXf(mode=L.IN_PARTIALS, slope=slope, stats=stats, gamma=w[name + ".gamma"], beta=w[name + ".beta"])

    def _in_bwd(self, x, stats, name, slope, dy, pooled=False):
        w, g = self.gs.w, self.gs.g
        return K.norm_act_bwd(x, stats, w[name + ".gamma"], w[name + ".beta"], slope, dy, pooled,
                              dgamma=g[name + ".gamma"], dbeta=g[name + ".beta"])

    def _wg(self, name, x, xf, dy):
        """Queues the weight gradient of one conv layer.  Nothing in the backward chain consumes it, and ~40 of them
        one by one would each need the whole chip: they are launched together (per stream, `_flush_wgrads`) so that
        layers of similar geometry share a launch.  The queued job keeps its operands referenced until then."""
        cv = self.conv[name]
        grads = self.ds.g if name.startswith("dis.") else self.gs.g
        q = self._wjobs.setdefault(torch.cuda.current_stream().cuda_stream, [])
        q.append(cv.wgrad_job(x, xf, dy, grads[cv.wkey], grads[cv.bkey] if cv.bkey else None, self.compute))

    def _flush_wgrads(self, handoff=False):
        """Launches the weight gradients queued on the current stream - on this stream, or (handoff) on the wgrad
        stream behind an event, so that they run beside the rest of this stream's backward chain instead of at its
        tail.  Handed-off jobs stay referenced until `_join_wgrads`."""
        cur = torch.cuda.current_stream()
        q = self._wjobs.pop(cur.cuda_stream, None)
        if not q:
            return
        if not handoff:
            K.conv2d_wgrad_multi(q)
            return
        self.wgrad_stream.wait_stream(cur)
        with torch.cuda.stream(self.wgrad_stream):
            K.conv2d_wgrad_multi(q)
        self._wkeep.extend(q)

    def _join_wgrads(self):
        if self._wkeep:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self._wkeep.clear()

    # ---- generator forward (training mode) --------------------------------------------------------------
    def _gen_forward(self, ldr, pick_src):
        S, w, c, cp = {}, self.gs.w, self.conv, self.compute
        # encoder (generator.py:92-108)
        S["c1"], S["s1"] = c["gen.conv1_d"].fwd(ldr, compute=cp, want_stats=True)
        S["xf2"] = self._inxf(S["s1"], "gen.norm1_d", 0.1)
        S["c2"], S["s2"] = c["gen.conv2_d"].fwd(S["c1"], S["xf2"], cp, want_stats=True)
        S["xf3"] = self._inxf(S["s2"], "gen.norm2_d", 0.1)
        S["c3"], S["s3"] = c["gen.conv3_d"].fwd(S["c2"], S["xf3"], cp, want_stats=True)
        x = K.norm_apply(S["c3"], S["s3"], w["gen.norm3_d.gamma"], w["gen.norm3_d.beta"], slope=0.1)
        S["x"] = [x]
        for i in range(6):
            p = "gen.res.%d." % i
            r1, t1 = c[p + "conv1"].fwd(x, compute=cp, want_stats=True)
            xf = self._inxf(t1, p + "norm1", 0.1)
            r2, t2 = c[p + "conv2"].fwd(r1, xf, cp, want_stats=True)
            x = K.norm_apply(r2, t2, w[p + "norm2.gamma"], w[p + "norm2.beta"], slope=1.0, residual=x)
            S["res%d" % i] = (r1, t1, xf, r2, t2)
            S["x"].append(x)
        res_out = x

        def decode(sfx, residual):
            d3, s3 = c["gen.conv3_" + sfx].fwd(res_out, compute=cp, want_stats=True)
            xf2 = self._inxf(s3, "gen.norm3_" + sfx, 0.1)
            d2, s2 = c["gen.conv2_" + sfx].fwd(d3, xf2, cp, want_stats=True)
            xf1 = self._inxf(s2, "gen.norm2_" + sfx, 0.1)
            y, _ = c["gen.conv1_" + sfx].fwd(d2, xf1, cp, out_slope=0.1, residual=residual, final_relu=True)
            S["dec_" + sfx] = (d3, s3, xf2, d2, s2, xf1, y, residual)
            return y

        # the sun-pose branch runs beside the encoder / sky decoder on a second stream
        main, side = torch.cuda.current_stream(), self.side_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            t = self._sunpose_forward(ldr)
            cams = self._gradcam(t, pick_src)
            rad = self._sunrad_forward(ldr, cams, t, S)
        sky_gamma = decode("f", ldr)
        main.wait_stream(side)
        S["t"], S["cams"] = t, cams
        rad_lin, rad_gamma, gamma, beta = rad
        sun_gamma = decode("u", rad_gamma)
        y_gamma, y_lin, alpha, sky_lin, sun_lin = K.blend(sky_gamma, sun_gamma, E.THRESHOLD)
        S.update(y_gamma=y_gamma, y_lin=y_lin, alpha=alpha, sky_gamma=sky_gamma, sun_gamma=sun_gamma, gamma=gamma, beta=beta,
                 rad_gamma=rad_gamma, rad_lin=rad_lin, sky_lin=sky_lin, sun_lin=sun_lin, ldr=ldr)
        return S

    def _sunpose_forward(self, ldr):
        w, c, cp = self.gs.w, self.conv, self.compute
        t, x = {}, ldr
        for l in (1, 2, 3):
            n = "sun.sunlayer%d" % l
            r1, st1 = c[n + ".conv1"].fwd(x, compute=cp, want_stats=True)
            xf = self._inxf(st1, n + ".norm1", 0.0)
            r2, st2 = c[n + ".conv2"].fwd(r1, xf, cp, want_stats=True)
            a, pooled = K.norm_apply(r2, st2, w[n + ".norm2.gamma"], w[n + ".norm2.beta"], slope=0.0, pool=True)
            t["in%d" % l], t["r%da" % l], t["st%da" % l], t["xf%d" % l] = x, r1, st1, xf
            t["r%db" % l], t["st%db" % l], t["A%d" % l], t["P%d" % l] = r2, st2, a, pooled
            x = pooled
        B = ldr.shape[0]
        t["flat"] = x.reshape(B, -1)
        t["f1"] = K.fc_finalize(K.fc_fwd(t["flat"], self.fc1, cp), w["sun.fc1.bias"], relu=True)
        t["gmax"] = torch.zeros(1, dtype=torch.int32, device=ldr.device)
        t["z"], t["cmf"] = K.softmax_head(K.fc_fwd(t["f1"], self.fc2, cp), w["sun.fc2.bias"], t["gmax"])
        return t

    def _gradcam(self, t, pick_src):
        """grad_cam.layer x3 under gen_tape.stop_recording() (train.py:257-271): constants for the gradient."""
        w, c, cp = self.gs.w, self.conv, self.compute
        B, h, wd = t["cmf"].shape[0], self.h, self.w
        dz, _ = K.softmax_pick_bwd(t["cmf"], t["z"], pick_src)
        df1 = K.fc_finalize(K.fc_dgrad(dz, self.fc2, cp), None, relu=False, mask_src=t["f1"])
        dP3 = K.fc_finalize(K.fc_dgrad(df1, self.fc1, cp)).reshape(B, h // 8, wd // 8, 128)
        w3 = K.spatial_sum(dP3, 1.0 / ((h // 4) * (wd // 4)))
        n3, n2 = "sun.sunlayer3", "sun.sunlayer2"
        g = K.norm_act_bwd(t["r3b"], t["st3b"], w[n3 + ".norm2.gamma"], w[n3 + ".norm2.beta"], 0.0, dP3, True)
        g = c[n3 + ".conv2"].dgrad(t["r3a"], g, cp)
        g = K.norm_act_bwd(t["r3a"], t["st3a"], w[n3 + ".norm1.gamma"], w[n3 + ".norm1.beta"], 0.0, g, False)
        dP2, sP2 = c[n3 + ".conv1"].dgrad(t["in3"], g, cp, want_stats=True)
        g = K.norm_act_bwd(t["r2b"], t["st2b"], w[n2 + ".norm2.gamma"], w[n2 + ".norm2.beta"], 0.0, dP2, True)
        g = c[n2 + ".conv2"].dgrad(t["r2a"], g, cp)
        g = K.norm_act_bwd(t["r2a"], t["st2a"], w[n2 + ".norm1.gamma"], w[n2 + ".norm1.beta"], 0.0, g, False)
        _, sP1 = c[n2 + ".conv1"].dgrad(t["in2"], g, cp, want_stats=True)
        return (K.grad_cam_map(t["A1"], sP1, 1.0 / (h * wd)), K.grad_cam_map(t["A2"], sP2, 1.0 / ((h // 2) * (wd // 2))),
                K.grad_cam_map(t["A3"], w3))

    def _down_stack(self, net, params, x, training):
        """downsampling x4 (discriminator.py:20-27 == sunrad_net.py:21-28).  Returns records for the backward pass:
        training=True -> BN batch statistics (+ moving update), else the moving statistics as a constant affine."""
        c, cp = self.conv, self.compute
        B = x.shape[0]
        R = {"in": x}
        R["d1"], _ = c[net + "d1"].fwd(x, compute=cp, out_slope=0.3)
        cur, xf = R["d1"], None
        for d in ("d2", "d3", "d4"):
            raw, st = c[net + d].fwd(cur, xf, cp, want_stats=training)
            n = net + d + ".norm."
            if training:
                mean, rstd, sc, sh = K.bn_train_finalize(st, params[n + "gamma"], params[n + "beta"], B, raw.shape[-1],
                                                         params[n + "moving_mean"], params[n + "moving_variance"])
            else:
                sc, sh = K.bn_eval_affine(params[n + "gamma"], params[n + "beta"], params[n + "moving_mean"],
                                          params[n + "moving_variance"])
                mean = rstd = None
            R[d] = dict(x=cur, xf=xf, raw=raw, mean=mean, rstd=rstd, scale=sc, shift=sh)
            cur, xf = raw, InXf(mode=L.IN_AFFINE, slope=0.3, scale=sc, shift=sh)
        R["xf_out"] = xf
        return R

    def _down_stack_bwd(self, net, params, grads, R, dact4, training, want_input_grad, do_wgrad=True):
        """Backward of _down_stack from the gradient wrt the ACTIVATED d4 output."""
        c, cp = self.conv, self.compute
        dy = dact4
        for d in ("d4", "d3", "d2"):
            r = R[d]
            n = net + d + ".norm."
            if training:
                draw = K.bn_act_bwd(r["raw"], dy, r["mean"], r["rstd"], params[n + "gamma"], params[n + "beta"], 0.3,
                                    grads[n + "gamma"] if do_wgrad else None, grads[n + "beta"] if do_wgrad else None)
            else:
                draw = K.affine_act_bwd(r["raw"], dy, r["scale"], r["shift"], 0.3)
            if do_wgrad:
                self._wg(net + d, r["x"], r["xf"], draw)
            dy = c[net + d].dgrad(r["x"], draw, cp)   # gradient wrt the activated input of this layer
        d1pre = K.affine_act_bwd(R["d1"], dy, None, None, 0.3)
        if do_wgrad:
            self._wg(net + "d1", R["in"], None, d1pre)
        if want_input_grad:
            return c[net + "d1"].dgrad(R["in"], d1pre, cp)
        return None

    def _sunrad_forward(self, ldr, cams, t, S):
        w = self.gs.w
        plz = K.plz_build(ldr, *cams)
        R = self._down_stack("gen.sun.", w, plz, training=True)
        xf = R["xf_out"]
        part = K.dense_heads(R["d4"]["raw"], xf.scale, xf.shift, 0.3, w["gen.sun.gamma.kernel"], w["gen.sun.beta.kernel"])
        rad_lin, rad_gamma, gamma, beta = K.sun_rad(t["cmf"], t["gmax"], part, w["gen.sun.gamma.bias"], w["gen.sun.beta.bias"],
                                                    self.h, self.w)
        S["sunrad"] = R
        return rad_lin, rad_gamma, gamma, beta

    # ---- VGG16 perceptual term (vgg16.py:127-165, train.py:308-313) ---------------------------------------
    def _vgg_loss_and_grad(self, y_gamma, hdr_t):
        cp, B = self.compute, y_gamma.shape[0]
        both = torch.empty((2 * B,) + tuple(y_gamma.shape[1:]), dtype=torch.float32, device=y_gamma.device)
        K.axpby(y_gamma, 1.0, out=both[:B])
        K.axpby(K.tonemap(hdr_t, False), 1.0, out=both[B:])
        x = K.vgg_pre(both)
        acts, pools = {}, []
        for blk in (("conv1_1", "conv1_2"), ("conv2_1", "conv2_2"), ("conv3_1", "conv3_2", "conv3_3")):
            for name in blk:
                acts[name + ".in"] = x
                x, _ = K.conv2d(x, self.vgg_pk[name], self.vgg[name + ".b"], out_slope=0.0, compute=cp)
                acts[name] = x
            x = K.maxpool(x)
            pools.append(x)
        dps = []
        for p in pools:   # L1 between the prediction half and the target half; gradient wrt the prediction half
            dp = torch.empty_like(p[:B])
            K.l1(p[:B], p[B:], 1.0, 0.01, self.losses[1:2], da=dp)
            dps.append(dp)
        # backward through the prediction half only
        g = None
        for bi, blk in reversed(list(enumerate((("conv1_1", "conv1_2"), ("conv2_1", "conv2_2"), ("conv3_1", "conv3_2", "conv3_3"))))):
            dp = dps[bi] if g is None else K.axpby(dps[bi], 1.0, g, 1.0)
            g = K.maxpool_relu_bwd(acts[blk[-1]][:B], dp)          # wrt the pre-ReLU output of the block's last conv
            for k in range(len(blk) - 1, -1, -1):
                name = blk[k]
                xin = acts[name + ".in"][:B]
                d = K.conv_desc(B, xin.shape[1], xin.shape[2], xin.shape[3], self.vgg_pk[name].Cout, 3, 3, 1, True, 1)
                g, _ = K.conv2d_dgrad(g, self.vgg_pkT[name], d, compute=cp)   # wrt this conv's (post-ReLU) input
                if k > 0:
                    g = K.affine_act_bwd(acts[blk[k - 1]][:B], g, None, None, 0.0)
        return K.axpby(g, 255.0)   # d/d y_gamma of the 0.01-weighted perceptual term

    # ---- one training step -----------------------------------------------------------------------------------
    def step(self, ldr, hdr_t, sunpose_gt, update=True):
        """ldr / hdr_t [B,H,W,3] BGR (train.py:386-387 rgb2bgr already applied), sunpose_gt [B,H*W].
        Returns the dict generator_in_step returns (train.py:349) - losses are in self.losses (device)."""
        out = self.step_a1(ldr, hdr_t, sunpose_gt)
        self.step_a2()
        if update:
            self.apply_gradients()
        return out

    def fc_grad_range(self):
        """[start, end) of the two Dense layers' gradients inside gs.grad - contiguous, the last trainables of the
        sun-pose net (50.3 M of the 58.3 M parameters): the slice whose all-reduce overlaps step_a2."""
        o = self.gs.offsets["sun.fc1.kernel"][0]
        return o, self.gs.ntrain

    def step_a1(self, ldr, hdr_t, sunpose_gt):
        """Part 1: forward, losses, and the backward pass down to the gradients of the sun-pose Dense layers
        (which are 86 % of the gradient bytes: the data-parallel driver starts their all-reduce right after this)."""
        w, g, c, cp = self.gs.w, self.gs.g, self.conv, self.compute
        B = ldr.shape[0]
        self.gs.grad.zero_(); self.ds.grad.zero_(); self.losses.zero_()

        S = self._gen_forward(ldr, sunpose_gt)
        t = S["t"]
        y_lin, y_gamma = S["y_lin"], S["y_gamma"]

        # Three independent chains run on three HIP streams (fork/join; captured as such in the hipGraph):
        #   main : L1 + DoG + KL            sA : VGG16 perceptual forward/backward
        #   sB   : adversarial term (discriminator with inference-mode BN, train.py:302)
        main, sA, sB = torch.cuda.current_stream(), self.side_stream, self.side_stream2
        sA.wait_stream(main); sB.wait_stream(main)
        dyl = torch.empty_like(y_lin)
        K.l1(y_lin, hdr_t, 1.0, 10.0, self.losses[3:4], da=dyl)                       # 10 * L1
        K.dog_loss(y_lin, hdr_t, 1000.0, self.losses[2:3], dyl)                       # 1000 * DoG
        dcmf = K.kl(sunpose_gt, t["cmf"], self.losses[0:1])                            # KL
        with torch.cuda.stream(sA):
            dyg = self._vgg_loss_and_grad(y_gamma, hdr_t)                              # 0.01 * perceptual
        cvo = c["dis.out"]
        with torch.cuda.stream(sB):
            Rg = self._down_stack("dis.", self.ds.w, K.concat2(ldr, y_lin), training=False)
            logits, _ = cvo.fwd(Rg["d4"]["raw"], Rg["xf_out"], cp)
            dlog = K.mse(logits, 1.0, 1.0, 1.0, self.losses[4:5])
            dact4 = cvo.dgrad(Rg["d4"]["raw"], dlog, cp)
            din = self._down_stack_bwd("dis.", self.ds.w, None, Rg, dact4, training=False, want_input_grad=True, do_wgrad=False)
            d_adv = K.slice_channels(din, 3, 3, 1.0)
        main.wait_stream(sB); main.wait_stream(sA)
        K.axpby(dyl, 1.0, d_adv, 1.0, out=dyl)

        # generator backward, first stretch: blend -> decoder tails -> sun radiance head -> dcmf complete
        dsky, dsun = K.blend_bwd(y_gamma, S["alpha"], dyg, dyl)
        tails = {}
        for sfx, dy in (("f", dsky), ("u", dsun)):
            y, residual = S["dec_" + sfx][6], S["dec_" + sfx][7]
            tails[sfx] = K.decoder_tail_bwd(y, residual, dy, want_dres=(sfx == "u"))
        dpre = K.sun_rad_bwd(t["cmf"], t["gmax"], S["gamma"], S["beta"], tails["u"][1], dcmf)
        # sun-pose Dense layers (sunpose_net.py:64-70): KL + the sun-radiance path meet in dcmf
        dz = K.softmax_bwd(t["cmf"], dcmf, t["z"])
        K.fc_wgrad(t["f1"], dz, g["sun.fc2.kernel"], g["sun.fc2.bias"])
        df1 = K.fc_finalize(K.fc_dgrad(dz, self.fc2, cp), None, relu=False, mask_src=t["f1"])
        K.fc_wgrad(t["flat"], df1, g["sun.fc1.kernel"], g["sun.fc1.bias"])
        dP3 = K.fc_finalize(K.fc_dgrad(df1, self.fc1, cp)).reshape(B, self.h // 8, self.w // 8, 128)
        self._S = dict(S=S, tails=tails, dpre=dpre, dP3=dP3, ldr=ldr, hdr_t=hdr_t)
        return dict(y_final_lin=y_lin, y_final_gamma=y_gamma, sky_pred_lin=S["sky_lin"], sun_pred_lin=S["sun_lin"],
                    gamma=S["gamma"], beta=S["beta"], alpha_c3=S["alpha"], sunpose_cmf=t["cmf"], sun_cam1=S["cams"][0],
                    sun_cam2=S["cams"][1], sun_cam3=S["cams"][2], sun_rad_lin=S["rad_lin"])

    def step_a2(self):
        """Part 2: everything else - discriminator step (sB), sun-pose conv layers backward (sA), generator backward
        (main).  The three chains write disjoint gradient ranges."""
        w, g, c, cp = self.gs.w, self.gs.g, self.conv, self.compute
        S, tails, dpre, dP, ldr, hdr_t = (self._S[k] for k in ("S", "tails", "dpre", "dP3", "ldr", "hdr_t"))
        t, y_lin = S["t"], S["y_lin"]
        main, sA, sB = torch.cuda.current_stream(), self.side_stream, self.side_stream2
        sA.wait_stream(main); sB.wait_stream(main)
        cvo = c["dis.out"]
        # ---- discriminator step (train.py:351-380): real then generated, BN batch statistics (their moving-stat
        # updates come after the generator step's inference-mode call of part 1, as in the reference's program order)
        with torch.cuda.stream(sB):
            for which, img, target, slot in (("real", hdr_t, 1.0, 6), ("fake", y_lin, 0.0, 5)):
                Rd = self._down_stack("dis.", self.ds.w, K.concat2(ldr, img), training=True)
                lg, _ = cvo.fwd(Rd["d4"]["raw"], Rd["xf_out"], cp)
                dl = K.mse(lg, target, 1.0, 0.5, self.losses[slot:slot + 1])
                self._wg("dis.out", Rd["d4"]["raw"], Rd["xf_out"], dl)
                da4 = cvo.dgrad(Rd["d4"]["raw"], dl, cp)
                self._down_stack_bwd("dis.", self.ds.w, self.ds.g, Rd, da4, training=True, want_input_grad=False)
            self._flush_wgrads()
        # ---- sun-pose conv layers (sunpose_net.py:54-62) --------------------------------------------------------
        with torch.cuda.stream(sA):
            for l in (3, 2, 1):
                n = "sun.sunlayer%d" % l
                dr2 = self._in_bwd(t["r%db" % l], t["st%db" % l], n + ".norm2", 0.0, dP, pooled=True)
                self._wg(n + ".conv2", t["r%da" % l], t["xf%d" % l], dr2)
                da = c[n + ".conv2"].dgrad(t["r%da" % l], dr2, cp)
                dr1 = self._in_bwd(t["r%da" % l], t["st%da" % l], n + ".norm1", 0.0, da)
                self._wg(n + ".conv1", t["in%d" % l], None, dr1)
                if l > 1:
                    dP = c[n + ".conv1"].dgrad(t["in%d" % l], dr1, cp)
            self._flush_wgrads()
        # ---- generator: decoders, sun radiance stack, encoder -----------------------------------------------------
        dres = torch.zeros_like(S["x"][-1])
        for sfx in ("f", "u"):
            d3, s3, xf2, d2, s2, xf1, y, residual = S["dec_" + sfx]
            dc = tails[sfx][0]
            self._wg("gen.conv1_" + sfx, d2, xf1, dc)
            da2 = c["gen.conv1_" + sfx].dgrad(d2, dc, cp)
            dd2 = self._in_bwd(d2, s2, "gen.norm2_" + sfx, 0.1, da2)
            self._wg("gen.conv2_" + sfx, d3, xf2, dd2)
            da3 = c["gen.conv2_" + sfx].dgrad(d3, dd2, cp)
            dd3 = self._in_bwd(d3, s3, "gen.norm3_" + sfx, 0.1, da3)
            self._wg("gen.conv3_" + sfx, S["x"][-1], None, dd3)
            c["gen.conv3_" + sfx].dgrad(S["x"][-1], dd3, cp, out=dres)
        self._flush_wgrads(handoff=True)
        R = S["sunrad"]    # sun radiance head (generator.py:158-169, sunrad_net.py:46-70)
        xf = R["xf_out"]
        dact4 = K.dense_heads_bwd(R["d4"]["raw"], xf.scale, xf.shift, 0.3, w["gen.sun.gamma.kernel"], w["gen.sun.beta.kernel"],
                                  dpre, g["gen.sun.gamma.kernel"], g["gen.sun.beta.kernel"], g["gen.sun.gamma.bias"],
                                  g["gen.sun.beta.bias"])
        self._down_stack_bwd("gen.sun.", w, g, R, dact4, training=True, want_input_grad=False)
        self._flush_wgrads(handoff=True)
        dx = dres          # encoder (generator.py:92-108, resBlock :26-35)
        for i in range(5, -1, -1):
            p = "gen.res.%d." % i
            r1, t1, xf, r2, t2 = S["res%d" % i]
            dr2 = self._in_bwd(r2, t2, p + "norm2", 1.0, dx)
            self._wg(p + "conv2", r1, xf, dr2)
            da1 = c[p + "conv2"].dgrad(r1, dr2, cp)
            dr1 = self._in_bwd(r1, t1, p + "norm1", 0.1, da1)
            self._wg(p + "conv1", S["x"][i], None, dr1)
            dx = c[p + "conv1"].dgrad(S["x"][i], dr1, cp, residual=dx)      # + identity branch
        self._flush_wgrads(handoff=True)
        dc3 = self._in_bwd(S["c3"], S["s3"], "gen.norm3_d", 0.1, dx)
        self._wg("gen.conv3_d", S["c2"], S["xf3"], dc3)
        da2 = c["gen.conv3_d"].dgrad(S["c2"], dc3, cp)
        dc2 = self._in_bwd(S["c2"], S["s2"], "gen.norm2_d", 0.1, da2)
        self._wg("gen.conv2_d", S["c1"], S["xf2"], dc2)
        da1 = c["gen.conv2_d"].dgrad(S["c1"], dc2, cp)
        dc1 = self._in_bwd(S["c1"], S["s1"], "gen.norm1_d", 0.1, da1)
        self._wg("gen.conv1_d", ldr, None, dc1)
        self._flush_wgrads()
        main.wait_stream(sA); main.wait_stream(sB)
        self._join_wgrads()

    def apply_gradients(self, gscale=1.0):
        """optimizer_gen / optimizer_disc .apply_gradients (train.py:403,406): RMSprop(lr), then refresh the packed
        bf16 weight images."""
        K.rmsprop(self.gs.flat[:self.gs.ntrain], self.gs.grad, self.gs.ms, self.lr, gscale=gscale)
        K.rmsprop(self.ds.flat[:self.ds.ntrain], self.ds.grad, self.ds.ms, self.lr, gscale=gscale)
        self.repack()

    def loss_dict(self):
        """Host copy of the loss terms with the reference's names (train.py:480-489) - synchronises."""
        v = dict(zip(LOSS_SLOTS, self.losses.tolist()))
        v["total_gen_loss"] = v["kl"] + 1000.0 * v["dog"] + v["adv"] + 10.0 * v["l1"] + 0.01 * v["perceptual"]
        v["total_disc_loss"] = 0.5 * (v["disc_generated"] + v["disc_real"])
        return v
