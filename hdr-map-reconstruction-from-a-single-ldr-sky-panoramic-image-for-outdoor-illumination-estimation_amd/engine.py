"""Fused execution plan of the generator graph on libhdrsky (inference.py:81-115, train.py:239-299).

The reference runs ~250 TF ops per forward (conv, bias, moments, normalise, activation, resize,
gather ... each its own kernel).  Here the same arithmetic is ~70 launches:
  * every InstanceNorm is split into (a) the producing conv's epilogue, which writes per-tile
    (sum, sumsq) partials, and (b) the consuming conv's operand load, which finalises the
    statistics and applies normalise + activation while staging its halo tile into LDS;
  * the resize-deconv's bilinear 2x resize happens in that same operand load;
  * the three `tf.gradients` sweeps of Grad-CAM are one shared backward sweep that stops at
    pool1's input (only the spatial MEANS of the activation gradients are needed, and the mean of a
    max-pool's input gradient equals the sum of its output gradient / (H*W)).
The whole sequence is enqueued on one HIP stream and is capturable into a hipGraph (no host
synchronisation, no allocation outside torch's caching allocator).
"""
from collections import OrderedDict
import os

import torch

from . import _lib as L
from . import hooks as HOOKS
from . import kernels as K
from .kernels import BF16, BF16X3, InXf, PackedConv, PackedFC

THRESHOLD = 0.12  # inference.py:36 / train.py:247


DA_PARTS = ("res", "sunpose", "decoders")


def da_parts(spec):
    """Which layer families run as distortion_aware_ops layers.  False / None: none.  True: the res blocks (generator.py:14,18).
    "all", a comma list or an iterable of DA_PARTS: "sunpose" = both convolutions of every sunposeLayer
    (sunpose_net.py:11,16: kernel_size = k_h, i.e. 7 / 3 / 3), "decoders" = distortion_aware_ops.deconv2d (:272-542: bilinear
    resize, then the distortion-aware conv at the output size) in place of ops.deconv2d in sky_decode / sun_decode
    (generator.py:110-156).  The variables are the plain layers' (an HWIO filter is the [k*k*C, F] kernel reshaped)."""
    if not spec:
        return frozenset()
    if spec is True:
        return frozenset({"res"})
    if isinstance(spec, str):
        spec = DA_PARTS if spec == "all" else [t.strip() for t in spec.split(",") if t.strip()]
    parts = frozenset(spec)
    if not parts <= frozenset(DA_PARTS):
        raise ValueError("distortion_aware: unknown part(s) %s (known: %s)" % (sorted(parts - frozenset(DA_PARTS)), DA_PARTS))
    return parts


def _dev(params, device):
    return OrderedDict((k, torch.as_tensor(v).to(device=device, dtype=torch.float32).contiguous())
                       for k, v in params.items())


class Nets:
    """Device-resident parameters + packed MFMA weight images of the generator (incl. sunRadNet)
    and the sun-pose net."""

    def __init__(self, gen_params, sun_params, device="cuda", precise=True, im_height=32, im_width=128):
        self.device = torch.device(device)
        self.h, self.w = im_height, im_width
        self.precise = precise
        # either network may be absent (the layer-API mirrors own one network each)
        self.gen = _dev(gen_params, self.device) if gen_params is not None else None
        self.sun = _dev(sun_params, self.device) if sun_params is not None else None
        self.pk = {}
        self._da_offs = {}
        self.side_stream = torch.cuda.Stream(device=self.device)      # (a higher stream priority for the longer sun branch: no effect)
        self.repack_all()

    def da_offsets(self, h, w, k=3, dilation_rate=1):
        """Device copy of the distortion-aware sampling offsets for an h x w map (cached)."""
        key = (h, w, k, dilation_rate)
        if key not in self._da_offs:
            self._da_offs[key] = K.da_offsets_device(h, w, k, dilation_rate, True, self.device)
        return self._da_offs[key]

    def da_table(self, h, w, k=3):
        """Transposed sample table of a k x k distortion-aware conv on an h x w map (kernels.da_transpose_table, cached)."""
        tab = K.da_transpose_table(h, w, k, 1, True, self.device)
        if tab is None:
            raise ValueError("distortion-aware data gradient: more than %d readers per (pixel, tap) at %dx%d, k=%d" % (K.DA_KMAX, h, w, k))
        return tab

    def da_pk(self, name, cpad=32):
        """Packed image of filter `name` ("sun.sunlayer1.conv1") with its input-channel axis zero-padded to cpad (the
        distortion-aware kernels read 32-channel groups; the padded input channels are zeros as well)."""
        key = name + ".da%d" % cpad
        if key not in self.pk:
            net, _, rest = name.partition(".")
            wgt = (self.gen if net == "gen" else self.sun)[rest + ".w"]
            kh, kw, cin, cout = wgt.shape
            self.pk[key] = PackedConv(K.pad_channels(wgt.reshape(kh * kw, cin * cout), cpad * cout).view(kh, kw, cpad, cout), self.precise)
        return self.pk[key]

    def repack_all(self):
        g, s = self.gen, self.sun
        pk = self.pk
        for key in [k for k in pk if k.rsplit(".", 1)[-1][:2] == "da" and k.rsplit(".", 1)[-1][2:].isdigit()]:
            del pk[key]                           # channel-padded images (da_pk): rebuilt from the current weights on demand
        for name in (["conv1_d", "conv2_d", "conv3_d", "conv1_f", "conv1_u"] +
                     ["res.%d.conv%d" % (i, j) for i in range(6) for j in (1, 2)]) if g is not None else ():
            pk["gen." + name] = PackedConv(g[name + ".w"], self.precise)
        for name in ("conv3_f", "conv2_f", "conv3_u", "conv2_u") if g is not None else ():
            pk["gen." + name] = PackedConv(g[name + ".kernel_deconv2d"], self.precise)
        for d in ("d1", "d2", "d3", "d4") if g is not None else ():
            pk["gen.sun." + d] = PackedConv(g["sun.%s.conv.kernel" % d], self.precise)
        for l in (1, 2, 3) if s is not None else ():
            for c in (1, 2):
                name = "sunlayer%d.conv%d" % (l, c)
                pk["sun." + name] = PackedConv(s[name + ".w"], self.precise)
                if not (l == 1):  # Grad-CAM sweep needs dgrad of layers 2..3 (all four convs)
                    pk["sun." + name + ".T"] = PackedConv(s[name + ".w"], self.precise, transpose_flip=True)
        if s is not None:
            pk["sun.fc1"] = PackedFC(s["fc1.kernel"], self.precise)
            pk["sun.fc2"] = PackedFC(s["fc2.kernel"], self.precise)
        if g is not None:
            self.refresh_eval_tables()

    def refresh_eval_tables(self):
        """Inference-mode BatchNorm of sunRadNet as per-channel affines (recompute after the moving stats change)."""
        g = self.gen
        self.bn_eval = {}
        for d in ("d2", "d3", "d4"):
            n = "sun.%s.norm." % d
            self.bn_eval[d] = K.bn_eval_affine(g[n + "gamma"], g[n + "beta"], g[n + "moving_mean"], g[n + "moving_variance"])


def _in_xf(stats, p, name, slope):
    return K.in_xf(stats, p[name + ".gamma"], p[name + ".beta"], slope)


def sun3_supported(x, compute):
    """sunlayer3 (3x3 64->128->128 on the 8x32 maps of a 32x128 image) can run on the sample-resident launches."""
    # Opt-in (HDRSKY_SUN3=1).  Measured on one box, same process order (profiles/ab_bench.sh): forward 0.621 vs 0.621 ms, training
    # step 3.64 vs 3.61 ms with / without - the two launches are each faster than the generic conv + norm pair alone
    # (6-7 us vs 15 + 8 us), but a sample-resident workgroup owns its CU (100 KB of LDS, 512 threads), so the kernels of
    # the other streams cannot run beside it and the step loses the overlap it gains in kernel time.
    if not HOOKS.H.sun3:
        return False
    return compute == BF16 and K.resconv_supported(x.shape[1], x.shape[2], 64, 128) and K.resconv_supported(x.shape[1], x.shape[2], 128, 128)


def sun3_forward(x, pk1, pk2, g1, b1, g2, b2):
    """sunposeLayer.call (sunpose_net.py:20-30) + the max-pool behind it for sunlayer3, on two sample-resident launches:
    conv -> InstanceNorm -> relu twice (bf16 between them), A3 in fp32 (Grad-CAM multiplies it), P3 = 2x2 max-pool.
    Returns the record the Grad-CAM sweep and the training backward re-read."""
    xb = K.to_bf16(x)
    o1 = K.resconv_fwd(xb, pk1, None, g1, b1, 0.0, save=True)
    o2 = K.resconv_fwd(o1["bf16"], pk2, None, g2, b2, 0.0, want_bf16=False, want_f32=True, save=True)
    return dict(xb=xb, o1=o1, o2=o2, A=o2["f32"], P=K.maxpool(o2["f32"]))


def sun3_backward(rec, dP, pkT1, pkT2, g1, b1, g2, b2, dgb1=None, dgb2=None):
    """Gradient chain through sunlayer3 from dP (w.r.t. the pooled output) to the layer's input: pool routing + relu mask
    (hdrsky_maxpool_relu_bwd on the post-relu map), then norm2 backward, the two data gradients and norm1 + relu backward
    inside sample-resident launches.  Returns (dc2, dc1, dx): bf16 gradients w.r.t. the two conv outputs (weight-gradient
    operands) and the fp32 gradient w.r.t. the layer input."""
    o1, o2 = rec["o1"], rec["o2"]
    dz = K.maxpool_relu_bwd(rec["A"], dP)
    dc2 = K.resconv_bwd(None, None, skip=dz, norm=dict(xhat=o2["xhat"], inv=o2["inv"], gamma=g2, beta=b2, slope=1.0, dgb=dgb2))["bf16"]
    dc1 = K.resconv_bwd(dc2, pkT2, norm=dict(xhat=o1["xhat"], inv=o1["inv"], gamma=g1, beta=b1, slope=0.0, dgb=dgb1))["bf16"]
    dx = K.resconv_bwd(dc1, pkT1, want_f32=True, want_bf16=False)["f32"]
    return dc2, dc1, dx


def sunpose_forward(nets, ldr, compute, distortion_aware=False, pick=None, convs_only=False, t=None):
    """sunpose_net.model.sunposeEstimation (sunpose_net.py:54-72) -> dict with cmf, z, A1..3 (+ what the
    Grad-CAM sweep re-reads: raw conv outputs and their IN partials).  distortion_aware ("sunpose" in da_parts): the
    convolutions are distortion_aware_ops.conv2d (sunpose_net.py:11,16)."""
    s, pk = nets.sun, nets.pk
    da = "sunpose" in da_parts(distortion_aware)
    if t is not None:      # (second half: the Dense layers + soft-max head on the record `t` of a convs_only call)
        return _sunpose_dense(nets, t, compute, pick)
    t = {"da": da}
    x = ldr
    for l in (1, 2, 3):
        n1, n2 = "sunlayer%d.conv1" % l, "sunlayer%d.conv2" % l
        if da:
            k = s[n1 + ".w"].shape[0]
            offs = nets.da_offsets(x.shape[1], x.shape[2], k)
            padded = x.shape[-1] % 32 != 0                          # the RGB image: zero channels up to 32
            xin = K.pad_channels(x, 32) if padded else x
            r1, st1 = K.da_conv2d(xin, nets.da_pk("sun." + n1) if padded else pk["sun." + n1], s[n1 + ".b"], offs, compute,
                                  want_stats=True)
            a1 = K.norm_apply(r1, st1, s["sunlayer%d.norm1.gamma" % l], s["sunlayer%d.norm1.beta" % l], slope=0.0)
            r2, st2 = K.da_conv2d(a1, pk["sun." + n2], s[n2 + ".b"], offs, compute, want_stats=True)
            a, pooled = K.norm_apply(r2, st2, s["sunlayer%d.norm2.gamma" % l], s["sunlayer%d.norm2.beta" % l], slope=0.0,
                                     pool=True)
            t["r%da" % l], t["st%da" % l], t["r%db" % l], t["st%db" % l], t["A%d" % l], t["P%d" % l] = r1, st1, r2, st2, a, pooled
            x = pooled
            continue
        if l == 3 and sun3_supported(x, compute):
            t["s3"] = sun3_forward(x, pk["sun." + n1], pk["sun." + n2], s["sunlayer3.norm1.gamma"], s["sunlayer3.norm1.beta"],
                                   s["sunlayer3.norm2.gamma"], s["sunlayer3.norm2.beta"])
            t["A3"], t["P3"] = t["s3"]["A"], t["s3"]["P"]
            x = t["P3"]
            continue
        r1, st1 = K.conv2d(x, pk["sun." + n1], s[n1 + ".b"], want_stats=True, compute=compute)
        r2, st2 = K.conv2d(r1, pk["sun." + n2], s[n2 + ".b"], want_stats=True, compute=compute,
                           xf=_in_xf(st1, s, "sunlayer%d.norm1" % l, 0.0))
        a, pooled = K.norm_apply(r2, st2, s["sunlayer%d.norm2.gamma" % l], s["sunlayer%d.norm2.beta" % l], slope=0.0,
                                 pool=True)
        t["r%da" % l], t["st%da" % l], t["r%db" % l], t["st%db" % l], t["A%d" % l], t["P%d" % l] = r1, st1, r2, st2, a, pooled
        x = pooled
    B = ldr.shape[0]
    t["flat"] = x.reshape(B, -1)
    return t if convs_only else _sunpose_dense(nets, t, compute, pick)


def _sunpose_dense(nets, t, compute, pick):
    """Dense layers + soft-max head of sunposeEstimation (sunpose_net.py:64-72) on the conv layers' record."""
    s, pk = nets.sun, nets.pk
    flat = t["flat"]
    t["gmax"] = torch.empty(1, dtype=torch.int32, device=flat.device)      # cleared by the finalize launch below
    t["f1"] = K.fc_fwd_fin(flat, pk["sun.fc1"], compute, s["fc1.bias"], relu=True, zero_word=t["gmax"])
    part2 = K.fc_fwd(t["f1"], pk["sun.fc2"], compute)
    if pick is None:
        t["z"], t["cmf"] = K.softmax_head(part2, s["fc2.bias"], t["gmax"])
    else:     # the Grad-CAM seed comes out of the same launch: pick = "self" (the row's own argmax) or the picking tensor
        t["z"], t["cmf"], t["dz_pick"] = K.softmax_head_pick(part2, s["fc2.bias"], t["gmax"], None if pick == "self" else pick)
    return t


def gradcam_sweep(nets, t, pick_src, compute):
    """grad_cam.layer x3 (grad_cam.py:29-44) as ONE backward sweep from y_c = cmf[b, argmax pick_src[b]]
    down to the input of pool1.  Returns (cam1, cam2, cam3).  (A distortion-aware sun-pose net - t["da"] - sweeps back
    through hdrsky_da_conv2d_dgrad.)"""
    s, pk = nets.sun, nets.pk
    B = t["cmf"].shape[0]
    h, w = nets.h, nets.w
    dz = t["dz_pick"] if "dz_pick" in t else K.softmax_pick_bwd(t["cmf"], t["z"], pick_src)[0]
    df1 = K.fc_dgrad_fin(dz, pk["sun.fc2"], compute, mask_src=t["f1"])
    dflat = K.fc_dgrad_fin(df1, pk["sun.fc1"], compute)
    dP3 = dflat.reshape(B, h // 8, w // 8, 128)
    small = (h // 8) * (w // 8) <= 256        # cam3's GAP weights: summed inside its own launch when the map is small
    w3 = dP3 if small else K.spatial_sum(dP3, 1.0 / ((h // 4) * (w // 4)))
    s3 = 1.0 / ((h // 4) * (w // 4)) if small else 1.0
    if t.get("da"):
        dP = dP3
        sums = {}
        for l in (3, 2):
            n = "sunlayer%d" % l
            hl, wl = h >> (l - 1), w >> (l - 1)
            tab = nets.da_table(hl, wl, 3)
            g = K.norm_act_bwd(t["r%db" % l], t["st%db" % l], s[n + ".norm2.gamma"], s[n + ".norm2.beta"], 0.0, dP, True)
            g = K.da_conv2d_dgrad(g, pk["sun." + n + ".conv2.T"], tab, 3, compute)
            g = K.norm_act_bwd(t["r%da" % l], t["st%da" % l], s[n + ".norm1.gamma"], s[n + ".norm1.beta"], 0.0, g, False)
            dP = K.da_conv2d_dgrad(g, pk["sun." + n + ".conv1.T"], tab, 3, compute)       # gradient at the pooled map below
            sums[l - 1] = K.spatial_sum(dP, 1.0 / ((2 * hl) * (2 * wl)))
        return K.grad_cam_maps([(t["A1"], sums[1], 1.0), (t["A2"], sums[2], 1.0), (t["A3"], w3, s3)])
    # layer 3 backward: pool3 + relu + IN2 -> dgrad conv2 -> relu + IN1 -> dgrad conv1
    if "s3" in t:
        _, _, dP2 = sun3_backward(t["s3"], dP3, pk["sun.sunlayer3.conv1.T"], pk["sun.sunlayer3.conv2.T"],
                                  s["sunlayer3.norm1.gamma"], s["sunlayer3.norm1.beta"], s["sunlayer3.norm2.gamma"],
                                  s["sunlayer3.norm2.beta"])
        sP2 = K.spatial_sum(dP2, 1.0 / ((h // 2) * (w // 2)))         # GAP numerator of d y_c / d A2 as a [B,C] table
    else:
        g = K.norm_act_bwd(t["r3b"], t["st3b"], s["sunlayer3.norm2.gamma"], s["sunlayer3.norm2.beta"], 0.0, dP3, True, out_bf16=compute == BF16)
        g, _ = K.conv2d(g, pk["sun.sunlayer3.conv2.T"], None, compute=compute)
        g = K.norm_act_bwd(t["r3a"], t["st3a"], s["sunlayer3.norm1.gamma"], s["sunlayer3.norm1.beta"], 0.0, g, False, out_bf16=compute == BF16)
        dP2, sP2 = K.conv2d(g, pk["sun.sunlayer3.conv1.T"], None, compute=compute, want_stats=True)
    g = K.norm_act_bwd(t["r2b"], t["st2b"], s["sunlayer2.norm2.gamma"], s["sunlayer2.norm2.beta"], 0.0, dP2, True, out_bf16=compute == BF16)
    g, _ = K.conv2d(g, pk["sun.sunlayer2.conv2.T"], None, compute=compute)
    g = K.norm_act_bwd(t["r2a"], t["st2a"], s["sunlayer2.norm1.gamma"], s["sunlayer2.norm1.beta"], 0.0, g, False, out_bf16=compute == BF16)
    _, sP1 = K.conv2d(g, pk["sun.sunlayer2.conv1.T"], None, compute=compute, want_stats=True)
    # GAP of d y_c / d A_k == sum of the pooled-map gradient / (H_k*W_k): the dgrad conv's per-tile sums
    sc2 = 1.0 if "s3" in t else 1.0 / ((h // 2) * (w // 2))
    return K.grad_cam_maps([(t["A1"], sP1, 1.0 / (h * w)), (t["A2"], sP2, sc2), (t["A3"], w3, s3)])


def encode(nets, ldr, compute, distortion_aware=False, dilation_rate=1):
    """generator.model.encode (generator.py:92-108) -> res_out [B,H/4,W/4,128].
    distortion_aware=True: the variant the reference keeps commented out (generator.py:14,18) - both 3x3 convolutions of
    every res block are distortion_aware_ops.conv2d (same weights, HWIO == its [k*k*C, F] kernel).  Forward only."""
    g, pk = nets.gen, nets.pk
    r1, s1 = K.conv2d(ldr, pk["gen.conv1_d"], g["conv1_d.b"], want_stats=True, compute=compute)
    r2, s2 = K.conv2d(r1, pk["gen.conv2_d"], g["conv2_d.b"], stride=2, want_stats=True, compute=compute,
                      xf=_in_xf(s1, g, "norm1_d", 0.1))
    r3, s3 = K.conv2d(r2, pk["gen.conv3_d"], g["conv3_d.b"], stride=2, want_stats=True, compute=compute,
                      xf=_in_xf(s2, g, "norm2_d", 0.1))
    x = K.norm_apply(r3, s3, g["norm3_d.gamma"], g["norm3_d.beta"], slope=0.1)
    if distortion_aware:
        _, h4, w4, _ = x.shape
        offs = nets.da_offsets(h4, w4, 3, dilation_rate)
        for i in range(6):
            p = "res.%d." % i
            c1, t1 = K.da_conv2d(x, pk["gen." + p + "conv1"], g[p + "conv1.b"], offs, compute, want_stats=True)
            a1 = K.norm_apply(c1, t1, g[p + "norm1.gamma"], g[p + "norm1.beta"], slope=0.1)
            c2, t2 = K.da_conv2d(a1, pk["gen." + p + "conv2"], g[p + "conv2.b"], offs, compute, want_stats=True)
            x = K.norm_apply(c2, t2, g[p + "norm2.gamma"], g[p + "norm2.beta"], slope=1.0, residual=x)
        return x
    if compute == BF16 and K.resconv_supported(x.shape[1], x.shape[2], 128, 128):
        # 8x32 maps, single-product mode: each half of a res block is one sample-resident launch with the InstanceNorm
        # (+ activation / identity add) in its epilogue; activations between them are final bf16 tensors
        xb = K.to_bf16(x)
        for i in range(6):
            p = "res.%d." % i
            a1 = K.resconv_fwd(xb, pk["gen." + p + "conv1"], None, g[p + "norm1.gamma"], g[p + "norm1.beta"], 0.1)["bf16"]
            o = K.resconv_fwd(a1, pk["gen." + p + "conv2"], None, g[p + "norm2.gamma"], g[p + "norm2.beta"], 1.0, residual=x,
                              want_f32=True, want_bf16=(i < 5))
            x, xb = o["f32"], o.get("bf16")
        return x
    for i in range(6):
        p = "res.%d." % i
        c1, t1 = K.conv2d(x, pk["gen." + p + "conv1"], g[p + "conv1.b"], want_stats=True, compute=compute)
        c2, t2 = K.conv2d(c1, pk["gen." + p + "conv2"], g[p + "conv2.b"], want_stats=True, compute=compute,
                          xf=_in_xf(t1, g, p + "norm1", 0.1))
        x = K.norm_apply(c2, t2, g[p + "norm2.gamma"], g[p + "norm2.beta"], slope=1.0, residual=x)
    return x


def decode_head(nets, res_out, sfx, compute, distortion_aware=False):
    """The two resize-deconvolutions of generator.model.sky_decode / sun_decode (generator.py:110-156) - all of a decoder that
    does not need the residual input of its last layer: (raw output of the second one, its InstanceNorm transform).
    distortion_aware ("decoders" in da_parts): they are distortion_aware_ops.deconv2d (:272-542) - bilinear 2x resize, then
    the distortion-aware 3x3 conv at the output size."""
    g, pk = nets.gen, nets.pk
    if "decoders" in da_parts(distortion_aware):
        u3 = K.up2x(res_out)
        r3, s3 = K.da_conv2d(u3, pk["gen.conv3_" + sfx], g["conv3_%s.bias_deconv2d" % sfx], nets.da_offsets(u3.shape[1], u3.shape[2]),
                             compute, want_stats=True)
        u2 = K.up2x(K.norm_apply(r3, s3, g["norm3_%s.gamma" % sfx], g["norm3_%s.beta" % sfx], slope=0.1))
        r2, s2 = K.da_conv2d(u2, pk["gen.conv2_" + sfx], g["conv2_%s.bias_deconv2d" % sfx], nets.da_offsets(u2.shape[1], u2.shape[2]),
                             compute, want_stats=True)
        return r2, _in_xf(s2, g, "norm2_" + sfx, 0.1)
    r3, s3 = K.conv2d(res_out, pk["gen.conv3_" + sfx], g["conv3_%s.bias_deconv2d" % sfx], upsample=2, want_stats=True,
                      compute=compute)
    r2, s2 = K.conv2d(r3, pk["gen.conv2_" + sfx], g["conv2_%s.bias_deconv2d" % sfx], upsample=2, want_stats=True,
                      compute=compute, xf=_in_xf(s3, g, "norm3_" + sfx, 0.1))
    return r2, _in_xf(s2, g, "norm2_" + sfx, 0.1)


def decode_tail(nets, head, sfx, residual, compute):
    """The decoder's last layer: 7x7 conv + `residual` (the LDR input for the sky, the log-compressed sun radiance for the
    sun decoder) + ReLU, on decode_head's result."""
    r2, xf = head
    y, _ = K.conv2d(r2, nets.pk["gen.conv1_" + sfx], nets.gen["conv1_%s.b" % sfx], compute=compute, xf=xf, out_slope=0.1,
                    residual=residual, final_relu=True)
    return y


def decode(nets, res_out, sfx, residual, compute, distortion_aware=False):
    """generator.model.sky_decode / sun_decode (generator.py:110-156)."""
    return decode_tail(nets, decode_head(nets, res_out, sfx, compute, distortion_aware), sfx, residual, compute)


def down_stack(x, pk, p, prefix, compute, training=False, bn_eval=None):
    """downsampling x4 (sunrad_net.py:21-28 == discriminator.py:20-27): 4x4 convs without bias, d1 without norm,
    LeakyReLU(0.3).  Returns the RAW d4 output and the (scale, shift) affine of its BatchNorm; the caller's kernel
    applies affine + LeakyReLU while loading.  training=True: batch statistics and the moving-average update
    (momentum 0.99); training=False: moving statistics."""
    B = x.shape[0]
    x, _ = K.conv2d(x, pk[prefix + "d1"], None, stride=2, out_slope=0.3, compute=compute)
    xf = None
    for d in ("d2", "d3", "d4"):
        x, st = K.conv2d(x, pk[prefix + d], None, stride=(1 if d == "d4" else 2), xf=xf, compute=compute,
                         want_stats=training)
        n = "%s.norm." % d
        if training:
            _, _, sc, sh = K.bn_train_finalize(st, p[n + "gamma"], p[n + "beta"], B, x.shape[-1], p[n + "moving_mean"],
                                               p[n + "moving_variance"])
        elif bn_eval is not None:
            sc, sh = bn_eval[d]
        else:
            sc, sh = K.bn_eval_affine(p[n + "gamma"], p[n + "beta"], p[n + "moving_mean"], p[n + "moving_variance"])
        xf = InXf(mode=L.IN_AFFINE, slope=0.3, scale=sc, shift=sh)
    return x, xf


def sun_rad_estimation(nets, ldr, cams, t, compute, training=False):
    """generator.model.sun_rad_estimation + sunRadNet (generator.py:158-169, sunrad_net.py:46-70); inference mode
    (BN moving statistics) unless training=True."""
    g, pk = nets.gen, nets.pk
    plz = K.plz_build(ldr, *cams)
    sunp = OrderedDict((k[4:], v) for k, v in g.items() if k.startswith("sun."))
    x, xf = down_stack(plz, pk, sunp, "gen.sun.", compute, training, None if training else nets.bn_eval)
    part = K.dense_heads(x, xf.scale, xf.shift, 0.3, g["sun.gamma.kernel"], g["sun.beta.kernel"])
    rad_lin, rad_gamma, gamma, beta = K.sun_rad(t["cmf"], t["gmax"], part, g["sun.gamma.bias"], g["sun.beta.bias"],
                                                nets.h, nets.w)
    return rad_lin, rad_gamma, gamma, beta


def _forward_sun(nets, ldr, pick_src, compute, da):
    """The sun branch of the generator graph: sun-pose net -> Grad-CAM sweep -> sun-radiance head."""
    t = sunpose_forward(nets, ldr, compute, da, pick="self" if pick_src is None else pick_src)
    cams = gradcam_sweep(nets, t, t["cmf"] if pick_src is None else pick_src, compute)
    rad_lin, rad_gamma, gamma, beta = sun_rad_estimation(nets, ldr, cams, t, compute)
    return dict(t=t, cams=cams, rad_lin=rad_lin, rad_gamma=rad_gamma, gamma=gamma, beta=beta)


def _forward_main(nets, ldr, compute, da):
    """The encoder branch: encoder, sky decoder, and the sun decoder up to its last layer (which needs the sun branch)."""
    res_out = encode(nets, ldr, compute, distortion_aware="res" in da)
    sky_gamma = decode(nets, res_out, "f", ldr, compute, da)
    sun_head = decode_head(nets, res_out, "u", compute, da)      # (only its last layer needs the sun branch's radiance map)
    return dict(res_out=res_out, sky_gamma=sky_gamma, sun_head=sun_head)


def _forward_tail(nets, ldr, S, M, compute):
    """Joins the two branches: the sun decoder's last layer on the radiance map, alpha mask, blending, tone mapping."""
    B, H, W, _ = ldr.shape
    t, cams = S["t"], S["cams"]
    sun_gamma = decode_tail(nets, M["sun_head"], "u", S["rad_gamma"], compute)
    y_gamma, y_lin, alpha, sky_lin, sun_lin = K.blend(M["sky_gamma"], sun_gamma, THRESHOLD)
    return dict(y_final_lin=y_lin, y_final_gamma=y_gamma, sky_pred_lin=sky_lin, sun_pred_lin=sun_lin, gamma=S["gamma"],
                beta=S["beta"], alpha_c3=alpha, sunpose_cmf=t["cmf"], sunpose_pred=t["cmf"].reshape(B, H, W, 1),
                sun_cam1=cams[0], sun_cam2=cams[1], sun_cam3=cams[2], sun_rad_lin=S["rad_lin"], res_out=M["res_out"],
                actv_maps=(t["A1"], t["A2"], t["A3"]))


def generator_forward(nets, ldr, pick_src=None, compute=BF16, distortion_aware=False):
    """inference.py:81-115 (pick_src=None: y_c = max cmf) / train.py:239-299 in test mode
    (pick_src = sunpose_gt).  ldr [B,H,W,3] BGR in [0,1].  Returns the reference's outputs as a dict.
    distortion_aware: see da_parts."""
    da = da_parts(distortion_aware)
    # two independent branches (generator encoder + sky decoder | sun-pose net + Grad-CAM + sun radiance) run on
    # two HIP streams; under hipGraph capture this becomes a fork/join in the graph (ForwardGraphs: one graph per branch).
    main = torch.cuda.current_stream()
    side = nets.side_stream
    side.wait_stream(main)
    with torch.cuda.stream(side):
        if HOOKS.H.fwd_stagger:
            # the encoder branch forks off BEHIND the sun-pose net's conv layers (an edge of the captured graph): it then runs beside
            # the sun branch's Dense layers (an HBM weight stream) and the small launches of the Grad-CAM sweep instead of beside its
            # full-resolution convolutions - the two branches' matrix-core launches no longer share the chip
            t = sunpose_forward(nets, ldr, compute, da, convs_only=True)
            forked = torch.cuda.Event(); forked.record(side)
            t = sunpose_forward(nets, ldr, compute, da, pick="self" if pick_src is None else pick_src, t=t)
            cams = gradcam_sweep(nets, t, t["cmf"] if pick_src is None else pick_src, compute)
            rad_lin, rad_gamma, gamma, beta = sun_rad_estimation(nets, ldr, cams, t, compute)
            S = dict(t=t, cams=cams, rad_lin=rad_lin, rad_gamma=rad_gamma, gamma=gamma, beta=beta)
        else:
            forked = None
            S = _forward_sun(nets, ldr, pick_src, compute, da)
    if forked is not None:
        main.wait_event(forked)
    M = _forward_main(nets, ldr, compute, da)
    main.wait_stream(side)
    return _forward_tail(nets, ldr, S, M, compute)


class ForwardGraphs:
    """generator_forward on a static input as THREE hipGraphs: the sun branch on the side stream, the encoder branch and the
    joining tail on the main stream, ordered by one event - the pattern of trainer.Trainer's segments.  One hipGraph with a
    fork / join inside executes its two branches almost one after the other on this runtime (profiles/r04_fwd_timeline.txt:
    the encoder branch starts 316 us into the pass although it depends on the input alone; the whole-step experiment of
    profiles/LABNOTES.md r3 section 5 saw the same); two single-stream graphs on two streams do run side by side.
    replay() enqueues one pass behind the caller's current stream; `out` is generator_forward's dict (static tensors)."""

    def __init__(self, nets, ldr, pick_src=None, compute=BF16, distortion_aware=False, warmup=2, main_stream=None, side_stream=None):
        da = da_parts(distortion_aware)
        self.nets, self.ldr = nets, ldr
        # (main_stream / side_stream: the caller's streams - e.g. streams with a compute-unit mask, kernels.masked_stream)
        self.main = main_stream if main_stream is not None else torch.cuda.Stream(device=nets.device)
        self.side = side_stream if side_stream is not None else nets.side_stream
        self.joined = torch.cuda.Event()
        cur = torch.cuda.current_stream()
        for _ in range(warmup):       # lazy kernel attributes, allocator
            self.main.wait_stream(cur)
            with torch.cuda.stream(self.main):
                generator_forward(nets, ldr, pick_src, compute, distortion_aware)
            cur.wait_stream(self.main)
        torch.cuda.synchronize()
        self.g_sun, self.g_main, self.g_tail = (torch.cuda.CUDAGraph() for _ in range(3))
        with K.no_gc():
            with torch.cuda.graph(self.g_sun, stream=self.side, capture_error_mode="thread_local"):
                S = _forward_sun(nets, ldr, pick_src, compute, da)
            with torch.cuda.graph(self.g_main, stream=self.main, capture_error_mode="thread_local"):
                M = _forward_main(nets, ldr, compute, da)
            with torch.cuda.graph(self.g_tail, stream=self.main, capture_error_mode="thread_local"):
                self.out = _forward_tail(nets, ldr, S, M, compute)
        self._keep = (S, M)
        torch.cuda.synchronize()

    def replay(self):
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        self.main.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self.g_sun.replay()
            self.joined.record(self.side)
        with torch.cuda.stream(self.main):
            self.g_main.replay()
            self.main.wait_event(self.joined)
            self.g_tail.replay()
        cur.wait_stream(self.main)
        return self.out
