"""GPU parity of the sample-resident conv + InstanceNorm kernel (csrc/res_conv.hip; generator.py:26-35 and its
backward, train.py:402) through the C ABI.

The kernel multiplies bf16 operands with fp32 accumulation, so the reference here is the oracle's conv2d / instance_norm
evaluated in float64 ON THE SAME bf16-rounded operands: what remains is fp32 accumulation order (1e-5 class), which
makes the tolerances tight enough to catch any indexing / tap / shift error (random, asymmetric filters)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tfsem as T

pytestmark = pytest.mark.gpu


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _mk(B, Cin, Cout, seed):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((B, 8, 32, Cin)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32))
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
    beta = torch.from_numpy((0.3 * rng.standard_normal(Cout)).astype(np.float32))
    return rng, x, w, bias, gamma, beta


def _close(got, ref, tol, what):
    got, ref = got.detach().cpu().double(), ref.double()
    assert got.shape == ref.shape, what
    assert torch.isfinite(got).all(), what
    e = float((got - ref).abs().max() / (ref.abs().max() + 1e-30))
    assert e <= tol, "%s: rel max err %.3e > %.1e" % (what, e, tol)


@pytest.mark.parametrize("B,Cin,Cout", [(1, 128, 128), (3, 128, 128), (2, 64, 128), (2, 128, 64)])
def test_resconv_forward_both_halves_of_the_res_block(dev, B, Cin, Cout):
    K = pkg("kernels")
    rng, x, w, bias, gamma, beta = _mk(B, Cin, Cout, 10 + B + Cin)
    res = torch.from_numpy(rng.standard_normal((B, 8, 32, Cout)).astype(np.float32))
    xq, wq = bf16_round(x), bf16_round(w)
    c = T.conv2d(xq, wq, bias.double(), 1, "SAME")
    mean = c.mean(dim=(1, 2), keepdim=True)
    var = ((c - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    inv = torch.rsqrt(var + 1e-3)
    xhat = (c - mean) * inv
    pw = K.PackedConv(w.to(dev), precise=False)
    xb = x.to(dev).to(torch.bfloat16)
    assert torch.equal(K.to_bf16(x.to(dev)), xb)                      # hdrsky_to_bf16 = round to nearest even
    d = lambda t: t.to(dev)
    # conv1 half: leaky(IN(conv + b), 0.1) as bf16 (+ what the backward pass saves)
    o = K.resconv_fwd(xb, pw, d(bias), d(gamma), d(beta), 0.1, want_bf16=True, want_f32=True, save=True)
    y = T.leaky_relu(T.instance_norm(c, gamma.double(), beta.double()), 0.1)
    _close(o["f32"], y, 2e-5, "act(IN(conv)) fp32")
    _close(o["bf16"].float(), y, 2.0 ** -8, "act(IN(conv)) bf16")
    _close(o["xhat"].float(), xhat, 2.0 ** -8, "xhat bf16")
    _close(o["inv"], inv.reshape(B, Cout), 1e-5, "rstd")
    # conv2 half: IN(conv + b) + identity, fp32 stream and its bf16 copy
    o2 = K.resconv_fwd(xb, pw, d(bias), d(gamma), d(beta), 1.0, residual=d(res), want_bf16=True, want_f32=True)
    y2 = res.double() + T.instance_norm(c, gamma.double(), beta.double())
    _close(o2["f32"], y2, 2e-5, "IN(conv) + identity fp32")
    assert torch.equal(o2["bf16"], o2["f32"].to(torch.bfloat16)), "bf16 copy = rounding of the fp32 stream"
    # no bias, relu (the sun-pose layer form)
    o3 = K.resconv_fwd(xb, pw, None, d(gamma), d(beta), 0.0, want_bf16=False, want_f32=True)
    _close(o3["f32"], torch.relu(T.instance_norm(c - bias.double(), gamma.double(), beta.double())), 2e-5, "relu(IN(conv))")


@pytest.mark.parametrize("B,C", [(1, 128), (3, 128)])
def test_resconv_backward_chain(dev, B, C):
    """dgrad conv (+ skip) -> through leaky(InstanceNorm) with the saved xhat / rstd: against the closed form evaluated on
    the same bf16 operands (tight) and against autograd through the oracle's instance_norm (loose: xhat is stored in
    bf16, 2^-9 relative)."""
    K = pkg("kernels")
    rng, dy, w, _, gamma, beta = _mk(B, C, C, 40 + B)
    skip = torch.from_numpy(rng.standard_normal((B, 8, 32, C)).astype(np.float32))
    craw = torch.from_numpy((rng.standard_normal((B, 8, 32, C)) * 1.7 + 0.6).astype(np.float64)).requires_grad_(True)
    d = lambda t: t.to(dev)
    dyq, wq = bf16_round(dy), bf16_round(w)
    # data gradient of y = conv(a, w) w.r.t. a, upstream dyq
    a0 = torch.zeros((B, 8, 32, C), dtype=torch.float64, requires_grad=True)
    (g_conv,) = torch.autograd.grad(T.conv2d(a0, wq, None, 1, "SAME"), a0, dyq)
    pwT = K.PackedConv(w.to(dev), precise=False, transpose_flip=True)
    dyb = dy.to(dev).to(torch.bfloat16)
    # (1) plain data gradient + skip, no norm behind it (exit of the chain)
    o = K.resconv_bwd(dyb, pwT, skip=d(skip), norm=None, want_f32=True, want_bf16=True)
    _close(o["f32"], g_conv + skip.double(), 2e-5, "dgrad + skip")
    _close(o["bf16"].float(), g_conv + skip.double(), 2.0 ** -8, "dgrad + skip (bf16)")
    # the forward quantities the backward launch re-reads
    mean = craw.mean(dim=(1, 2), keepdim=True)
    var = ((craw - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    inv = torch.rsqrt(var + 1e-3).detach()
    xhat_b = ((craw - mean) * inv).detach().to(torch.bfloat16)
    xq = xhat_b.to(torch.float64)
    for slope, use_skip in ((0.1, False), (1.0, True), (0.0, False)):
        g = g_conv + (skip.double() if use_skip else 0.0)
        dz = g * torch.where(gamma.double() * xq + beta.double() > 0, 1.0, slope)
        s1 = dz.sum(dim=(1, 2), keepdim=True); s2 = (dz * xq).sum(dim=(1, 2), keepdim=True)
        dc = gamma.double() * inv * (dz - s1 / 256.0 - xq * s2 / 256.0)
        dgb = torch.zeros((B, 2, C), device=dev)
        nd = dict(xhat=xhat_b.to(dev), inv=inv.reshape(B, C).float().to(dev), gamma=d(gamma), beta=d(beta), slope=slope, dgb=dgb)
        o = K.resconv_bwd(dyb, pwT, skip=d(skip) if use_skip else None, norm=nd, want_f32=use_skip, want_bf16=True)
        _close(o["bf16"].float(), dc, 2.0 ** -8, "dc (closed form, slope %g)" % slope)
        _close(dgb[:, 0], s2.reshape(B, C), 5e-5, "d gamma terms"); _close(dgb[:, 1], s1.reshape(B, C), 5e-5, "d beta terms")
        if use_skip:
            _close(o["f32"], g, 2e-5, "stream gradient")
        # autograd through the oracle's InstanceNorm + activation at the un-rounded pre-activation
        y = T.leaky_relu(T.instance_norm(craw, gamma.double(), beta.double()), slope)
        (dc_auto,) = torch.autograd.grad(y, craw, g, retain_graph=True)
        # (where the pre-activation is within bf16 rounding of zero the activation mask may differ: excluded)
        pre = (gamma.double() * ((craw - mean) * inv) + beta.double()).detach()
        far = pre.abs() > 0.05
        diff = (o["bf16"].float().cpu().double() - dc_auto).abs()
        e = float(diff[far].max() / dc_auto.abs().max())
        r = float(torch.sqrt((diff[far] ** 2).mean()) / torch.sqrt((dc_auto ** 2).mean()))
        assert e < 3e-2 and r < 3e-2, (slope, e, r)
    # (2) no convolution: the norm backward of `skip` alone (entry of the chain)
    dz = skip.double()
    s1 = dz.sum(dim=(1, 2), keepdim=True); s2 = (dz * xq).sum(dim=(1, 2), keepdim=True)
    dc = gamma.double() * inv * (dz - s1 / 256.0 - xq * s2 / 256.0)
    dgb = torch.zeros((B, 2, C), device=dev)
    nd = dict(xhat=xhat_b.to(dev), inv=inv.reshape(B, C).float().to(dev), gamma=d(gamma), beta=d(beta), slope=1.0, dgb=dgb)
    o = K.resconv_bwd(None, None, skip=d(skip), norm=nd)
    _close(o["bf16"].float(), dc, 2.0 ** -8, "norm backward without a conv")
    # (3) fixed-order reduction of the per-sample terms into the gradient vectors (added to their contents)
    dg0 = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(dev); db0 = torch.zeros(C, device=dev)
    ref_g, ref_b = dg0.cpu().double() + dgb[:, 0].cpu().double().sum(0), dgb[:, 1].cpu().double().sum(0)
    K.DgbReducer([(dgb, dg0, db0)]).run()
    _close(dg0, ref_g, 1e-6, "dgb_reduce gamma"); _close(db0, ref_b, 1e-6, "dgb_reduce beta")


def test_resconv_is_bit_reproducible_and_rejects_other_shapes(dev):
    K, L = pkg("kernels"), pkg("_lib")
    rng, x, w, bias, gamma, beta = _mk(4, 128, 128, 77)
    pw = K.PackedConv(w.to(dev), precise=False)
    xb = x.to(dev).to(torch.bfloat16)
    d = lambda t: t.to(dev)
    a = K.resconv_fwd(xb, pw, d(bias), d(gamma), d(beta), 0.1, want_f32=True)
    for _ in range(3):
        b = K.resconv_fwd(xb, pw, d(bias), d(gamma), d(beta), 0.1, want_f32=True)
        assert torch.equal(a["f32"], b["f32"]) and torch.equal(a["bf16"], b["bf16"])
    assert not K.resconv_supported(32, 128, 128, 128) and not K.resconv_supported(8, 32, 96, 128)
    assert not K.resconv_supported(8, 32, 128, 128, 5, 5) and K.resconv_supported(8, 32, 64, 128)
    args = L.ResconvArgs()
    args.B, args.Cin, args.Cout, args.mode = 1, 96, 128, L.RC_FWD
    assert L.load().hdrsky_resconv(args, None) == -2          # HDRSKY_EUNSUPPORTED before anything is launched


def test_sunlayer3_on_the_sample_resident_launches(dev, monkeypatch):
    """Opt-in path (HDRSKY_SUN3=1): sunlayer3 (3x3 64->128->128 + InstanceNorm + relu, sunpose_net.py:20-30) forward, its part
    of the Grad-CAM sweep and its training backward on hdrsky_resconv.  Same bf16 operands as the default path, so the
    two agree at fp32 accumulation level in the forward and at bf16-xhat level in the gradients."""
    params, synth, engine, trainer, K = pkg("params"), pkg("synth"), pkg("engine"), pkg("trainer"), pkg("kernels")
    gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
    dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
    batch = synth.make_batch(2, seed=1234)
    ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    nets = engine.Nets(gen, sun, device=dev, precise=False)
    base = engine.generator_forward(nets, ldr, compute=K.BF16)
    tr0 = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16)
    tr0.step(ldr, hdr, gt, update=False)
    g0 = {k: tr0.gs.g[k].clone() for k in tr0.gs.g if k.startswith("sun.")}
    monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1")
    monkeypatch.setenv("HDRSKY_SUN3", "1")
    out = engine.generator_forward(nets, ldr, compute=K.BF16)
    for k, tol in (("sunpose_cmf", 2e-3), ("sun_cam3", 2e-2), ("sun_cam2", 5e-2), ("y_final_gamma", 5e-3)):
        _close(out[k], base[k].cpu(), tol, k)
    tr1 = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16)
    tr1.step(ldr, hdr, gt, update=False)
    assert "s3" in tr1._T["t"]
    dot = na = nb = 0.0
    for k, v in g0.items():
        a, b = tr1.gs.g[k].double(), v.double()
        dot += float((a * b).sum()); na += float((a * a).sum()); nb += float((b * b).sum())
    cos = dot / (na * nb) ** 0.5
    print("sun-pose gradient cosine, sunlayer3 on resconv vs default: %.5f" % cos)
    assert cos > 0.995 and abs((na / nb) ** 0.5 - 1.0) < 2e-2
    for leaf in ("conv1.w", "conv2.w", "norm1.gamma", "norm2.beta"):
        a, b = tr1.gs.g["sun.sunlayer3." + leaf], g0["sun.sunlayer3." + leaf]
        assert float((a - b).abs().max()) <= 6e-2 * float(b.abs().max()), leaf
