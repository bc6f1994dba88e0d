"""Paired launches (round 5; include/hdrsky.h "PAIRED LAUNCHES"): the sky / sun decoder layers of generator.py:110-156 - identical
shapes, separate weights - and their whole backward chain as ONE launch per layer on a batch of 2 B.  The claim under test is
bit-identity: every sample of a paired launch runs the arithmetic of the unpaired launch on its own layer."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _g(dev, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return g


# (name, B per layer, H, W, Cin, Cout, k, x dtype, x_shared, flip (data gradient), want_stats, out_bf16)
CONV_CASES = [
    ("decoder conv3 forward (shared resized encoder output)", 4, 16, 64, 128, 64, 3, "bf16", True, False, True, False),
    ("decoder conv2 forward", 4, 32, 128, 64, 32, 3, "bf16", False, False, True, False),
    ("decoder conv1 data gradient (3 -> 32, narrow)", 4, 32, 128, 3, 32, 7, "f32", False, True, False, True),
    ("decoder conv2 data gradient", 4, 32, 128, 32, 64, 3, "bf16", False, True, False, True),
    ("decoder conv3 data gradient", 4, 16, 64, 64, 128, 3, "bf16", False, True, False, True),
    ("bench batch: conv2 forward at B = 32", 32, 32, 128, 64, 32, 3, "bf16", False, False, True, False),
    ("bench batch: conv3 data gradient at B = 32", 32, 16, 64, 64, 128, 3, "bf16", False, True, False, True),
    ("a tile without a paired instantiation (falls back to two launches)", 2, 4, 16, 256, 512, 4, "f32", False, False, True, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_paired_conv_equals_the_two_launches(dev, case):
    K = pkg("kernels")
    name, B, H, W, C, F, k, xdt, shared, flip, stats, ob = case
    g = _g(dev, B + H + C + F)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    xa, xb = mk(B, H, W, C), mk(B, H, W, C)
    if xdt == "bf16":
        xa, xb = xa.to(torch.bfloat16), xb.to(torch.bfloat16)
    wshape = (k, k, F, C) if flip else (k, k, C, F)
    w1, w2 = mk(*wshape) / (k * C ** 0.5), mk(*wshape) / (k * C ** 0.5)
    pw1, pw2 = K.PackedConv(w1, precise=False, transpose_flip=flip), K.PackedConv(w2, precise=False, transpose_flip=flip)
    b1, b2 = (None, None) if flip else (mk(F), mk(F))
    kw = dict(compute=K.BF16, want_stats=stats, out_bf16=ob)
    ya, sa = K.conv2d(xa, pw1, b1, **kw)
    yb, sb = K.conv2d(xa if shared else xb, pw2, b2, **kw)
    x2 = xa if shared else torch.cat([xa, xb]).contiguous()
    y2, s2 = K.conv2d(x2, pw1, b1, pair=K.ConvPair(pw2, b2), x_shared=shared, **kw)
    torch.cuda.synchronize()
    assert tuple(y2.shape) == (2 * B,) + tuple(ya.shape[1:]) and y2.dtype == ya.dtype
    assert torch.equal(y2[:B], ya) and torch.equal(y2[B:], yb), name
    if stats:
        assert s2.nparts == sa.nparts and torch.equal(s2.part[:B], sa.part) and torch.equal(s2.part[B:], sb.part), name


@pytest.mark.parametrize("mode", ["partials", "affine"])
def test_paired_conv_with_operand_transform_and_residual(dev, mode):
    """The decoder tail's form (7x7 32->3, InstanceNorm + LeakyReLU of the paired tensor in front, residual + ReLU behind) paired:
    PARTIALS transform with the second layer's gamma / beta, AFFINE tables of 2 B samples, per-layer residual tensors."""
    K, L = pkg("kernels"), pkg("_lib")
    B, H, W, C, F = 3, 32, 128, 32, 3
    g = _g(dev, 7)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = (mk(2 * B, H, W, C) * 1.5 + 0.3).contiguous()
    eye = torch.eye(C, device=dev).reshape(1, 1, C, C).contiguous()
    _, st = K.conv2d(x, K.PackedConv(eye, precise=False), None, want_stats=True, compute=K.BF16)      # statistics partials of x itself
    ga, ba, gb, bb = torch.rand(C, device=dev, generator=g) + 0.5, mk(C) * 0.2, torch.rand(C, device=dev, generator=g) + 0.5, mk(C) * 0.2
    half = lambda lo, hi: K.Stats(st.part[lo:hi], st.nparts, st.count)
    if mode == "partials":
        xf = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=ga, beta=ba, gamma2=gb, beta2=bb)
        xfa = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=half(0, B), gamma=ga, beta=ba)
        xfb = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=half(B, 2 * B), gamma=gb, beta=bb)
    else:
        nparts = st.nparts
        import importlib
        HK = pkg("hooks")
        sc = torch.empty(2 * B, C, device=dev); sh = torch.empty_like(sc)
        lib = L.load()
        L.check(lib.hdrsky_in_affine_pair(st.part.data_ptr(), nparts, 2 * B, C, st.count, ga.data_ptr(), ba.data_ptr(), gb.data_ptr(), bb.data_ptr(),
                                          K.IN_EPS, sc.data_ptr(), sh.data_ptr(), torch.cuda.current_stream().cuda_stream), "in_affine_pair")
        sa, ha = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
        sb_, hb = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
        for part, gm, bt, s_, h_ in ((st.part[:B], ga, ba, sa, ha), (st.part[B:], gb, bb, sb_, hb)):
            L.check(lib.hdrsky_in_affine(part.data_ptr(), nparts, B, C, st.count, gm.data_ptr(), bt.data_ptr(), K.IN_EPS, s_.data_ptr(), h_.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "in_affine")
        assert torch.equal(sc[:B], sa) and torch.equal(sc[B:], sb_) and torch.equal(sh[:B], ha) and torch.equal(sh[B:], hb)
        xf = K.InXf(mode=L.IN_AFFINE, slope=0.1, scale=sc, shift=sh)
        xfa = K.InXf(mode=L.IN_AFFINE, slope=0.1, scale=sa, shift=ha)
        xfb = K.InXf(mode=L.IN_AFFINE, slope=0.1, scale=sb_, shift=hb)
    w1, w2 = mk(7, 7, C, F) / 40, mk(7, 7, C, F) / 40
    pw1, pw2 = K.PackedConv(w1, precise=False), K.PackedConv(w2, precise=False)
    b1, b2, r1, r2 = mk(F), mk(F), mk(B, H, W, F), mk(B, H, W, F)
    kw = dict(compute=K.BF16, out_slope=0.1, final_relu=True)
    ya, _ = K.conv2d(x[:B], pw1, b1, xf=xfa, residual=r1, **kw)
    yb, _ = K.conv2d(x[B:], pw2, b2, xf=xfb, residual=r2, **kw)
    y2, _ = K.conv2d(x, pw1, b1, xf=xf, residual=r1, pair=K.ConvPair(pw2, b2, r2), **kw)
    assert torch.equal(y2[:B], ya) and torch.equal(y2[B:], yb)
    # in_xf(pair=) hands out the same transform (partials below 64 tiles per sample, tables from there on)
    got = K.in_xf(st, ga, ba, 0.1, pair=(gb, bb))
    assert got.mode == (L.IN_PARTIALS if st.nparts < pkg("hooks").H.inxf_affine_min else L.IN_AFFINE)


@pytest.mark.parametrize("shape,pooled", [((3, 32, 128, 32), False), ((3, 16, 64, 64), False), ((3, 32, 128, 32), True), ((3, 64, 256, 32), False)])
def test_paired_pointwise_kernels_equal_the_two_launches(dev, shape, pooled):
    """hdrsky_norm_act_bwd_pair (one-launch and - the 64x256 map - sliced form), hdrsky_up2x_xf_bf16_pair and the pair-sum form of
    hdrsky_up2x_bwd against two calls on the halves: bit for bit."""
    K, L = pkg("kernels"), pkg("_lib")
    B, H, W, C = shape
    g = _g(dev, H + C)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = (mk(2 * B, H, W, C) * 1.3 + 0.2).contiguous()
    eye = torch.eye(C, device=dev).reshape(1, 1, C, C).contiguous()
    _, st = K.conv2d(x, K.PackedConv(eye, precise=False), None, want_stats=True, compute=K.BF16)
    half = lambda lo, hi: K.Stats(st.part[lo:hi], st.nparts, st.count)
    ga, ba, gb, bb = torch.rand(C, device=dev, generator=g) + 0.5, mk(C) * 0.3, torch.rand(C, device=dev, generator=g) + 0.5, mk(C) * 0.3
    dshape = (2 * B, H // 2, W // 2, C) if pooled else (2 * B, H, W, C)
    for dy in (mk(*dshape), mk(*dshape).to(torch.bfloat16)):
        for ob in (False, True):
            one_launch = ob and L.load().hdrsky_norm_act_bwd_one_launch(H, W, int(pooled), 0)      # (bf16-output calls only)
            d2, s2 = K.norm_act_bwd(x, st, ga, ba, 0.1, dy, pooled, want_sums=True, out_bf16=ob, pair=(gb, bb))
            da, sa = K.norm_act_bwd(x[:B], half(0, B), ga, ba, 0.1, dy[:B], pooled, want_sums=True, out_bf16=ob)
            db, sb = K.norm_act_bwd(x[B:], half(B, 2 * B), gb, bb, 0.1, dy[B:], pooled, want_sums=True, out_bf16=ob)
            # (one-launch form or sliced form - a paired tensor is sliced like one of its layers' launches: bit for bit either way)
            assert torch.equal(d2[:B], da) and torch.equal(d2[B:], db) and torch.equal(s2[:B], sa) and torch.equal(s2[B:], sb), (one_launch, ob)
    if not pooled and H <= 32:
        xf = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=ga, beta=ba, gamma2=gb, beta2=bb)
        u2 = K.up2x_act_bf16(x, xf)
        ua = K.up2x_act_bf16(x[:B], K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=half(0, B), gamma=ga, beta=ba))
        ub = K.up2x_act_bf16(x[B:], K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=half(B, 2 * B), gamma=gb, beta=bb))
        assert torch.equal(u2[:B], ua) and torch.equal(u2[B:], ub)
        for dyu in (mk(2 * B, H, W, C), mk(2 * B, H, W, C).to(torch.bfloat16)):
            seq = torch.zeros(B, H // 2, W // 2, C, device=dev)
            K.up2x_bwd(dyu[:B], 1.0, out=seq); K.up2x_bwd(dyu[B:], 1.0, out=seq)
            par = torch.zeros(B, H // 2, W // 2, C, device=dev)
            K.up2x_bwd(dyu, 1.0, out=par, pair_sum=True)
            assert torch.equal(par, seq)
            fresh = K.up2x_bwd(dyu, 1.0, pair_sum=True)
            assert torch.equal(fresh, K.up2x_bwd(dyu[:B], 1.0) + K.up2x_bwd(dyu[B:], 1.0))


def test_training_step_with_paired_decoders_is_bit_identical(dev, monkeypatch):
    """One bench-mode training step (B = 4) with the decoders' launches paired (default) and unpaired (HDRSKY_DEC_PAIR=0): every
    gradient of both optimizers, the prediction and the BatchNorm statistics bit for bit; loss terms (atomic sums) to rounding;
    and the paired plan has 12 launches fewer on the main chain."""
    params, synth, trainer, K, HK = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels"), pkg("hooks")
    nets = [params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
            params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3)]
    b = synth.make_batch(4, seed=99)
    ldr, hdr, gt = (torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    res = {}
    for pair in ("1", "0"):
        monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1"); monkeypatch.setenv("HDRSKY_DEC_PAIR", pair); HK.reload()
        tr = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16)
        K.TRACE = []
        try:
            out = tr.step(ldr, hdr, gt, update=False)
            torch.cuda.synchronize()
            ntraced = len(K.TRACE)
        finally:
            K.TRACE = None
        res[pair] = (tr.gs.grad.clone(), tr.ds.grad.clone(), out["y_final_lin"].clone(), tr.losses.clone(), ntraced)
        del tr
    monkeypatch.delenv("HDRSKY_DEC_PAIR"); monkeypatch.delenv("HDRSKY_EXPERIMENTS"); HK.reload()
    a, c = res["1"], res["0"]
    assert torch.equal(a[2], c[2]), "prediction"
    assert torch.equal(a[0], c[0]), "generator / sun-pose gradients: %d elements differ" % int((a[0] != c[0]).sum())
    assert torch.equal(a[1], c[1]), "discriminator gradients"
    assert torch.allclose(a[3], c[3], rtol=1e-5, atol=0)
    assert c[4] - a[4] == 5, (a[4], c[4])      # traced matrix-core launches: 2 forward + 3 data-gradient launches fewer
