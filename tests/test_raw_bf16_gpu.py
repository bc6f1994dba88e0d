"""Raw conv outputs stored as bf16 in front of a norm layer (include/hdrsky.h, "RAW CONV OUTPUTS AS bf16", ABI 4): every
reader widens while loading and then does what it does for fp32 storage - so each reader, given the bf16 tensor, must return
BIT-IDENTICAL results to the same launch on the widened (fp32) copy of that tensor; and the producing conv must store the
round-to-nearest-even of what it stores in fp32 with the statistics of the fp32 accumulators.  (The end-to-end effect of the
storage format on the training step is bounded by tests/test_fullsize_gpu.py against the oracle.)"""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _g(dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return g


def _raw_pair(dev, shape, seed, scale=1.5):
    x16 = (torch.randn(*shape, device=dev, generator=_g(dev, seed)) * scale + 0.3).to(torch.bfloat16).contiguous()
    return x16, x16.float().contiguous()


def _stats_of(K, x32):
    """Stats partials as a conv epilogue would write them: one tile per sample."""
    B, H, W, C = x32.shape
    part = torch.stack([x32.sum((1, 2)), (x32 * x32).sum((1, 2))], 1).reshape(B, 1, 2, C).contiguous()
    return K.Stats(part, 1, H * W)


@pytest.mark.parametrize("shape,k,stride,cout", [((3, 16, 64, 64), 3, 2, 128), ((2, 32, 128, 32), 3, 2, 64),
                                                 ((2, 32, 128, 32), 7, 1, 3), ((2, 8, 32, 128), 4, 2, 256),
                                                 ((2, 9, 21, 64), 3, 1, 64)])
@pytest.mark.parametrize("mode", ["partials", "affine"])
def test_conv_reads_a_bf16_raw_operand_like_its_widened_copy(dev, shape, k, stride, cout, mode):
    K, L = pkg("kernels"), pkg("_lib")
    B, H, W, C = shape
    x16, x32 = _raw_pair(dev, shape, 1)
    g = _g(dev, 2)
    w = torch.randn(k, k, C, cout, device=dev, generator=g) / (k * C ** 0.5)
    b = torch.randn(cout, device=dev, generator=g)
    gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.2
    st = _stats_of(K, x32)
    if mode == "partials":
        xf = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=gamma, beta=beta, eps=K.IN_EPS)
    else:
        sc = torch.rand(B, C, device=dev, generator=g) + 0.5
        sh = torch.randn(B, C, device=dev, generator=g) * 0.3
        xf = K.InXf(mode=L.IN_AFFINE, slope=0.3, scale=sc, shift=sh)
    pw = K.PackedConv(w, precise=False)
    for kw in (dict(), dict(want_stats=True, out_bf16=(cout % 4 == 0))):
        y16, s16 = K.conv2d(x16, pw, b, stride=stride, xf=xf, compute=K.BF16, **kw)
        y32, s32 = K.conv2d(x32, pw, b, stride=stride, xf=xf, compute=K.BF16, **kw)
        assert torch.equal(y16, y32), (shape, k, mode, float((y16.float() - y32.float()).abs().max()))
        if s16 is not None:
            assert torch.equal(s16.part, s32.part)
    # the single-product mode only: the split-product kernels keep fp32 operands
    with pytest.raises(L.HdrSkyError):
        K.conv2d(x16, K.PackedConv(w, precise=True), b, stride=stride, xf=xf, compute=K.BF16X3)


def test_conv_stores_rounded_outputs_with_the_statistics_of_its_accumulators(dev):
    K = pkg("kernels")
    g = _g(dev, 3)
    for (B, H, W, C, cout, k, s) in ((4, 32, 128, 3, 32, 7, 1), (3, 32, 128, 32, 64, 3, 2), (2, 16, 64, 128, 64, 3, 1)):
        x = torch.randn(B, H, W, C, device=dev, generator=g)
        w = torch.randn(k, k, C, cout, device=dev, generator=g) / (k * C ** 0.5)
        b = torch.randn(cout, device=dev, generator=g)
        pw = K.PackedConv(w, precise=False)
        y32, s32 = K.conv2d(x, pw, b, stride=s, compute=K.BF16, want_stats=True)
        y16, s16 = K.conv2d(x, pw, b, stride=s, compute=K.BF16, want_stats=True, out_bf16=True)
        assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16))
        assert torch.equal(s16.part, s32.part)


@pytest.mark.parametrize("shape", [(3, 16, 64, 64), (2, 8, 32, 128), (2, 32, 128, 32), (2, 10, 14, 32)])
def test_norm_forward_and_backward_read_bf16_raw_like_the_widened_copy(dev, shape):
    K = pkg("kernels")
    B, H, W, C = shape
    x16, x32 = _raw_pair(dev, shape, 4)
    g = _g(dev, 5)
    gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.2
    st = _stats_of(K, x32)
    res = torch.randn(*shape, device=dev, generator=g)
    for kw in (dict(slope=0.1), dict(slope=1.0, residual=res), dict(slope=0.0, pool=True)):
        a, b_ = K.norm_apply(x16, st, gamma, beta, **kw), K.norm_apply(x32, st, gamma, beta, **kw)
        for u, v in zip(a if isinstance(a, tuple) else (a,), b_ if isinstance(b_, tuple) else (b_,)):
            assert torch.equal(u, v), kw
    for pooled in (False, True):
        dshape = (B, H // 2, W // 2, C) if pooled else shape
        dy = torch.randn(*dshape, device=dev, generator=g)
        for dyv in (dy, dy.to(torch.bfloat16)):
            for ob in (False, True):
                o16 = K.norm_act_bwd(x16, st, gamma, beta, 0.0 if pooled else 0.1, dyv, pooled, want_sums=True, out_bf16=ob)
                o32 = K.norm_act_bwd(x32, st, gamma, beta, 0.0 if pooled else 0.1, dyv, pooled, want_sums=True, out_bf16=ob)
                for u, v in zip(o16, o32):
                    assert torch.equal(u, v), (pooled, dyv.dtype, ob)


def test_batchnorm_backward_and_affine_backward_read_bf16_raw_like_the_widened_copy(dev):
    K = pkg("kernels")
    for shape in ((4, 8, 32, 128), (6, 4, 16, 256), (2, 16, 64, 64)):
        B, H, W, C = shape
        x16, x32 = _raw_pair(dev, shape, 6)
        g = _g(dev, 7)
        gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.2
        mean = x32.mean((0, 1, 2)).contiguous()
        rstd = (x32.var((0, 1, 2), unbiased=False) + 1e-3).rsqrt().contiguous()
        dy = torch.randn(*shape, device=dev, generator=g)
        for dyv in (dy, dy.to(torch.bfloat16)):
            for ob in (False, True):
                outs = []
                for x in (x16, x32):
                    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
                    dx = K.bn_act_bwd(x, dyv, mean, rstd, gamma, beta, 0.3, dg, db, out_bf16=ob)
                    outs.append((dx, dg, db))
                for u, v in zip(*outs):
                    assert torch.equal(u, v), (shape, dyv.dtype, ob)
        sc, sh = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.2
        for ob in (False, True):
            assert torch.equal(K.affine_act_bwd(x16, dy, sc, sh, 0.3, out_bf16=ob), K.affine_act_bwd(x32, dy, sc, sh, 0.3, out_bf16=ob))


def test_resize_operand_and_weight_gradient_operand_from_bf16_raw(dev, monkeypatch):
    K, L = pkg("kernels"), pkg("_lib")
    shape = (3, 8, 32, 64)
    B, H, W, C = shape
    x16, x32 = _raw_pair(dev, shape, 8)
    g = _g(dev, 9)
    gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.2
    st = _stats_of(K, x32)
    xf = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=gamma, beta=beta, eps=K.IN_EPS)
    assert torch.equal(K.up2x_act_bf16(x16, xf), K.up2x_act_bf16(x32, xf))
    # weight gradients: the LDS-DMA kernel behind hdrsky_act_bf16, the register-staged kernel (HDRSKY_WGRAD2=0) and the
    # narrow-output kernel (3 output channels) each on the bf16 raw tensor and on its widened copy
    for cout, k, dy_dt, env in ((128, 3, torch.bfloat16, None), (128, 3, torch.bfloat16, "0"), (64, 3, torch.float32, None),
                                (3, 7, torch.float32, None)):
        if env is not None:
            monkeypatch.setenv("HDRSKY_WGRAD2", env)
        dy = (torch.randn(B, H, W, cout, device=dev, generator=g) * 0.1).to(dy_dt)
        outs = []
        for x in (x16, x32):
            dw, db = torch.zeros(k, k, C, cout, device=dev), torch.zeros(cout, device=dev)
            K.conv2d_wgrad_multi([K.wgrad_job(x, dy, k, k, dw, db, xf=xf, compute=K.BF16)])
            outs.append((dw, db))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (cout, k, dy_dt, env)
        assert float(outs[0][0].abs().max()) > 0
        if env is not None:
            monkeypatch.delenv("HDRSKY_WGRAD2")


def test_one_channel_output_conv_reads_bf16(dev):
    """The discriminator's 4x4 VALID 512 -> 1 conv (conv_dot1_kernel) on a bf16 raw operand with its BatchNorm affine."""
    K, L = pkg("kernels"), pkg("_lib")
    shape = (4, 4, 16, 512)
    x16, x32 = _raw_pair(dev, shape, 10)
    g = _g(dev, 11)
    w = torch.randn(4, 4, 512, 1, device=dev, generator=g) * 0.02
    b = torch.randn(1, device=dev, generator=g)
    sc, sh = torch.rand(512, device=dev, generator=g) + 0.5, torch.randn(512, device=dev, generator=g) * 0.2
    xf = K.InXf(mode=L.IN_AFFINE, slope=0.3, scale=sc, shift=sh)
    pw = K.PackedConv(w, precise=False)
    y16, _ = K.conv2d(x16, pw, b, same=False, xf=xf, compute=K.BF16)
    y32, _ = K.conv2d(x32, pw, b, same=False, xf=xf, compute=K.BF16)
    assert torch.equal(y16, y32)


@pytest.mark.parametrize("shape,k,stride,cout", [((3, 16, 64, 64), 3, 2, 128), ((2, 32, 128, 32), 3, 2, 64), ((2, 8, 32, 128), 4, 2, 256),
                                                 ((2, 32, 128, 32), 7, 1, 32), ((2, 9, 21, 64), 3, 1, 64), ((3, 9, 37, 32), 3, 2, 64),
                                                 ((2, 4, 16, 256), 4, 1, 512)])
@pytest.mark.parametrize("mode", ["partials", "affine"])
def test_forward_conv_writes_the_operand_of_its_weight_gradient(dev, shape, k, stride, cout, mode):
    """hdrsky_conv2d_fwd_emit (conv_igemm_kernel<..., EMIT>): the launch's result and statistics are those of hdrsky_conv2d_fwd, and
    the bf16 tensor it writes beside them - act(norm(x)), every input pixel by the tile that owns it - equals hdrsky_act_bf16's
    bit for bit (incl. stride 2, odd sizes, several channel groups); the weight gradient taken on it is the one taken through the
    materialising launch."""
    K, L = pkg("kernels"), pkg("_lib")
    B, H, W, C = shape
    g = _g(dev, 21)
    x = (torch.randn(*shape, device=dev, generator=g) * 1.5 + 0.3).contiguous()
    w = torch.randn(k, k, C, cout, device=dev, generator=g) / (k * C ** 0.5)
    b = torch.randn(cout, device=dev, generator=g)
    gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.2
    if mode == "partials":
        xf = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=_stats_of(K, x), gamma=gamma, beta=beta, eps=K.IN_EPS)
    else:
        xf = K.InXf(mode=L.IN_AFFINE, slope=0.3, scale=torch.rand(B, C, device=dev, generator=g) + 0.5,
                    shift=torch.randn(B, C, device=dev, generator=g) * 0.3)
    pw = K.PackedConv(w, precise=False)
    y0, s0 = K.conv2d(x, pw, b, stride=stride, xf=xf, compute=K.BF16, want_stats=True)
    op = K.Operand()
    y1, s1 = K.conv2d(x, pw, b, stride=stride, xf=xf, compute=K.BF16, want_stats=True, emit_xb=op)
    assert torch.equal(y0, y1) and torch.equal(s0.part, s1.part)
    assert not hasattr(x, "_xb")          # the operand travels in the caller's handle, not on the tensor object
    kept = (op.key, op.tensor)
    assert kept[1] is not None and kept[0] is xf and kept[1].dtype == torch.bfloat16 and kept[1].shape == x.shape
    # the materialising launch on the same tensor and transform
    d = K.conv_desc(B, H, W, C, cout, k, k, stride, True, 1)
    d.compute = K.BF16
    tabs = K._xf_args(d, xf, B, H, W, C)
    ref = torch.empty_like(kept[1])
    L.check(L.load().hdrsky_act_bf16(K._p(x), 0, B, H * W, C, d.in_mode, K._p(tabs[0]), K._p(tabs[1]), d.ss_bstride, K._p(tabs[2]),
                                     d.in_nparts, K._p(tabs[3]), K._p(tabs[4]), d.in_eps, d.in_slope, K._p(ref), K._stream()), "act_bf16")
    assert torch.equal(kept[1], ref), int((kept[1] != ref).sum())
    # and the weight gradient: through the kept operand (wgrad_job finds it) == through the materialising path
    dy = (torch.randn(*y0.shape, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    outs = []
    for use_kept in (True, False):
        dw, db = torch.zeros(k, k, C, cout, device=dev), torch.zeros(cout, device=dev)
        K.conv2d_wgrad_multi([K.wgrad_job(x, dy, k, k, dw, db, stride=stride, xf=xf, compute=K.BF16, operand=op if use_kept else None)])
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # a VALID-geometry layer is not taken (its output blocks do not cover the image)
    dv = K.conv_desc(B, H, W, C, cout, k, k, 1, False, 1); dv.compute = K.BF16; dv.in_mode, dv.in_slope = d.in_mode, d.in_slope
    assert not L.load().hdrsky_conv2d_emit_supported(dv) and L.load().hdrsky_conv2d_emit_supported(d)
