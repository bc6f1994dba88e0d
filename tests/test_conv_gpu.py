"""GPU parity: hdrsky_conv2d_fwd (through the C ABI) vs the CPU oracle, every layer shape of the
generator / sun-pose / sun-radiance / discriminator / VGG stacks at small batch, both compute modes."""
import zlib

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tfsem as T
from util import TOL_X3, assert_close, assert_close_bf16

pytestmark = pytest.mark.gpu

# (name, H, W, Cin, Cout, k, stride, same, upsample)
LAYERS = [
    ("g.conv1_d 7x7 3->32", 32, 128, 3, 32, 7, 1, True, 1),
    ("g.conv2_d 3x3 s2 32->64", 32, 128, 32, 64, 3, 2, True, 1),
    ("g.conv3_d 3x3 s2 64->128", 16, 64, 64, 128, 3, 2, True, 1),
    ("g.res 3x3 128->128", 8, 32, 128, 128, 3, 1, True, 1),
    ("g.conv3_f up+3x3 128->64", 8, 32, 128, 64, 3, 1, True, 2),
    ("g.conv2_f up+3x3 64->32", 16, 64, 64, 32, 3, 1, True, 2),
    ("g.conv1_f 7x7 32->3", 32, 128, 32, 3, 7, 1, True, 1),
    ("s.l1.conv2 7x7 32->32", 32, 128, 32, 32, 7, 1, True, 1),
    ("s.l2.conv2 3x3 64->64", 16, 64, 64, 64, 3, 1, True, 1),
    ("d1 4x4 s2 6->64", 32, 128, 6, 64, 4, 2, True, 1),
    ("d2 4x4 s2 64->128", 16, 64, 64, 128, 4, 2, True, 1),
    ("d3 4x4 s2 128->256", 8, 32, 128, 256, 4, 2, True, 1),
    ("d4 4x4 s1 256->512", 4, 16, 256, 512, 4, 1, True, 1),
    ("dis.out 4x4 VALID 512->1", 4, 16, 512, 1, 4, 1, False, 1),
    ("vgg.conv1_1 3x3 3->64", 32, 128, 3, 64, 3, 1, True, 1),
    ("vgg.conv3_2 3x3 256->256", 8, 32, 256, 256, 3, 1, True, 1),
    ("odd 3x3 s1 32->32 on 10x20", 10, 20, 32, 32, 3, 1, True, 1),
    ("odd 3x3 s2 32->64 on 9x37", 9, 37, 32, 64, 3, 2, True, 1),
]


def _ref(x, w, b, stride, same, upsample):
    xt = torch.from_numpy(x)
    if upsample == 2:
        xt = T.resize_bilinear(xt, 2 * x.shape[1], 2 * x.shape[2])
    return T.conv2d(xt, torch.from_numpy(w), torch.from_numpy(b), stride, "SAME" if same else "VALID").numpy()


@pytest.mark.parametrize("case", LAYERS, ids=[c[0] for c in LAYERS])
@pytest.mark.parametrize("B", [2, 5])
def test_conv_layers(dev, case, B):
    K = pkg("kernels")
    name, H, W, Cin, Cout, k, stride, same, up = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = _ref(x, w, b, stride, same, up)
    xd, wd, bd = (torch.from_numpy(a).to(dev) for a in (x, w, b))
    pw = K.PackedConv(wd, precise=True)
    y, _ = K.conv2d(xd, pw, bd, stride=stride, same=same, upsample=up, compute=K.BF16X3)
    assert_close(y, ref, TOL_X3, name + " x3")
    y, _ = K.conv2d(xd, pw, bd, stride=stride, same=same, upsample=up, compute=K.BF16)
    assert_close_bf16(y, ref, name + " bf16")


def test_conv_fused_in_stats_residual(dev):
    """producer partials -> consumer IN + lrelu fused into the operand load; epilogue act + residual + relu."""
    K = pkg("kernels"); L = pkg("_lib")
    rng = np.random.default_rng(7)
    B, H, W = 3, 16, 64
    x = rng.standard_normal((B, H, W, 32)).astype(np.float32) * 2 + 0.5
    w1 = (rng.standard_normal((3, 3, 32, 64)) / 17).astype(np.float32); b1 = rng.standard_normal(64).astype(np.float32)
    w2 = (rng.standard_normal((3, 3, 64, 32)) / 24).astype(np.float32); b2 = rng.standard_normal(32).astype(np.float32)
    gam = rng.uniform(0.5, 1.5, 64).astype(np.float32); bet = rng.standard_normal(64).astype(np.float32)
    res = rng.standard_normal((B, H, W, 32)).astype(np.float32)
    xt = torch.from_numpy(x)
    c1 = T.conv2d(xt, torch.from_numpy(w1), torch.from_numpy(b1))
    a1 = T.leaky_relu(T.instance_norm(c1, torch.from_numpy(gam), torch.from_numpy(bet)), 0.1)
    c2 = T.conv2d(a1, torch.from_numpy(w2), torch.from_numpy(b2))
    ref = torch.relu(T.leaky_relu(c2, 0.1) + torch.from_numpy(res))
    d = lambda a: torch.from_numpy(a).to(dev)
    p1, p2 = K.PackedConv(d(w1)), K.PackedConv(d(w2))
    r1, st = K.conv2d(d(x), p1, d(b1), want_stats=True, compute=K.BF16X3)
    assert_close(r1, c1, TOL_X3, "producer")
    # statistics partials reproduce the moments
    mean, rstd, _, _ = K.in_finalize(st, d(gam), d(bet), B, 64)
    assert_close(mean, c1.mean(dim=(1, 2)), 1e-4, "IN mean")
    var = ((c1 - c1.mean(dim=(1, 2), keepdim=True)) ** 2).mean(dim=(1, 2))
    assert_close(rstd, torch.rsqrt(var + 1e-3), 1e-4, "IN rstd")
    xf = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=d(gam), beta=d(bet))
    y, _ = K.conv2d(r1, p2, d(b2), xf=xf, out_slope=0.1, residual=d(res), final_relu=True, compute=K.BF16X3)
    assert_close(y, ref, 3e-4, "fused consumer")
    # the tables computed once per tensor (hdrsky_in_affine, what kernels.in_xf switches to from hooks.H.inxf_affine_min = 64 tiles per
    # sample on) are what every workgroup derives from the partials, to the last ulp or two (same formula; the compiler
    # contracts the multiply-adds of each kernel its own way): forward and weight gradient agree to fp32 round-off
    H = pkg("hooks").H
    saved = H.inxf_affine_min
    try:
        H.inxf_affine_min = 1
        xa = K.in_xf(st, d(gam), d(bet), 0.1)
        H.inxf_affine_min = 1 << 30
        xp = K.in_xf(st, d(gam), d(bet), 0.1)
    finally:
        H.inxf_affine_min = saved
    assert xa.mode == L.IN_AFFINE and xp.mode == L.IN_PARTIALS
    for cp in (K.BF16X3, K.BF16):
        ya, _ = K.conv2d(r1, p2, d(b2), xf=xa, out_slope=0.1, residual=d(res), final_relu=True, compute=cp)
        yp, _ = K.conv2d(r1, p2, d(b2), xf=xp, out_slope=0.1, residual=d(res), final_relu=True, compute=cp)
        tol = 2e-6 if cp == K.BF16X3 else 4e-3      # (bf16 operands: an ulp of the table can move a rounding boundary)
        assert_close(ya, yp, tol, "affine tables vs partials, forward")
        dy = torch.randn_like(ya)
        ga = K.conv2d_wgrad(r1, dy, 3, 3, xf=xa, compute=cp); gp = K.conv2d_wgrad(r1, dy, 3, 3, xf=xp, compute=cp)
        assert_close(ga[0], gp[0], tol, "affine tables vs partials, weight gradient"); assert torch.equal(ga[1], gp[1])


def test_conv_dgrad_via_flipped_filter(dev):
    """data gradient of a stride-1 SAME conv == conv with the transposed/flipped packed filter."""
    K = pkg("kernels")
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal((2, 8, 32, 64)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((3, 3, 64, 128)) / 24).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((2, 8, 32, 128)).astype(np.float32))
    (gx,) = torch.autograd.grad(T.conv2d(x, w, None), x, dy)
    pT = K.PackedConv(w.to(dev), transpose_flip=True)
    got, _ = K.conv2d(dy.to(dev), pT, None, compute=K.BF16X3)
    assert_close(got, gx, TOL_X3, "dgrad")


ALL_TILES = ["2,2,4,2,32", "2,2,2,2,32", "2,2,2,2,16", "4,1,4,2,32", "4,1,2,2,32", "2,2,2,1,32", "2,2,2,1,16",
             "4,1,4,1,32", "4,1,2,1,32", "4,1,1,1,16", "2,4,2,1,32", "4,2,2,2,32", "4,2,2,1,32", "8,1,2,2,32",
             "8,1,4,2,32", "2,4,2,1,16", "8,1,2,1,32",
             # direct-B (barrier-free) variants
             "1,4,4,1,32,1", "2,4,4,1,32,1", "1,8,4,1,32,1", "2,2,4,1,32,1", "4,2,4,1,32,1", "2,4,2,1,32,1",
             "4,1,4,1,32,1", "8,1,4,1,32,1", "1,4,4,1,16,1", "1,4,2,1,16,1", "2,2,4,2,32,1", "2,4,4,2,32,1"]


@pytest.mark.parametrize("tile", ALL_TILES)
def test_every_tile_instantiation(dev, tile, monkeypatch):
    """Each (WM,WN,MI,NI,TW[,direct-B]) instantiation (4- and 8-wave workgroups) on a wide, a narrow and a
    strided layer, with statistics output, in both compute modes."""
    K = pkg("kernels")
    monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1")         # HDRSKY_TILE is a tuning hook: behind the gate
    monkeypatch.setenv("HDRSKY_TILE", tile)
    rng = np.random.default_rng(zlib.crc32(tile.encode()))
    B = 2
    for (H, W, Cin, Cout, k, stride) in [(16, 64, 64, 128, 3, 1), (32, 64, 3, 64, 7, 1), (16, 64, 32, 64, 3, 2)]:
        x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
        w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        ref = _ref(x, w, b, stride, True, 1)
        xd, wd, bd = (torch.from_numpy(a).to(dev) for a in (x, w, b))
        pw = K.PackedConv(wd, precise=True)
        try:
            y, st = K.conv2d(xd, pw, bd, stride=stride, compute=K.BF16X3, want_stats=True)
        except pkg("_lib").HdrSkyError as e:
            # a hook-forced tile may not fit a layer's halo in LDS (the built-in heuristic never picks such a pair)
            assert "EUNSUPPORTED" in str(e)
            continue
        assert_close(y, ref, TOL_X3, "%s %s" % (tile, (H, W, Cin, Cout, k, stride)))
        mean, _, _, _ = K.in_finalize(st, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev), B, Cout)
        assert_close(mean, torch.from_numpy(ref).mean(dim=(1, 2)), 2e-4, "stats " + tile)
        y16, _ = K.conv2d(xd, pw, bd, stride=stride, compute=K.BF16)
        assert_close_bf16(y16, ref, tile + " bf16")


def test_bf16_activation_storage_is_bit_neutral_for_relu_chains(dev):
    """HDRSKY_BF16 mode, bf16 activation storage (hdrsky_conv_desc.x_bf16 / y_bf16; the VGG16 chain): a conv writing
    bf16 = the fp32 result rounded to nearest even; a conv reading a bf16 activation = the same conv on its fp32 widening,
    bit for bit (the staging rounds its operand to bf16 anyway); pool / ReLU backward variants on bf16 maps likewise."""
    K = pkg("kernels")
    rng = np.random.default_rng(33)
    d = lambda a: torch.from_numpy(a).to(dev)
    for (B, H, W, Cin, Cout, k) in ((2, 32, 128, 64, 64, 3), (3, 16, 64, 64, 128, 3), (2, 8, 32, 256, 256, 3), (2, 32, 128, 3, 64, 3)):
        x = d(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
        w = d((rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32))
        b = d(rng.standard_normal(Cout).astype(np.float32))
        pw = K.PackedConv(w)
        y32, _ = K.conv2d(x, pw, b, out_slope=0.0, compute=K.BF16)
        y16, st = K.conv2d(x, pw, b, out_slope=0.0, compute=K.BF16, out_bf16=True, want_stats=True)
        assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16))
        _, st32 = K.conv2d(x, pw, b, out_slope=0.0, compute=K.BF16, want_stats=True)
        assert torch.equal(st.part, st32.part)                       # statistics come from the fp32 values
        if Cin % 32 == 0:
            xb = x.to(torch.bfloat16)
            ya, _ = K.conv2d(xb, pw, b, out_slope=0.0, compute=K.BF16)
            yb, _ = K.conv2d(xb.float(), pw, b, out_slope=0.0, compute=K.BF16)
            assert torch.equal(ya, yb)
            with pytest.raises(Exception):
                K.conv2d(xb, pw, b, compute=K.BF16X3)
    # the activation backward in the epilogue of a data-gradient conv (res_mode 1) + bf16 output
    x = d(rng.standard_normal((2, 16, 64, 128)).astype(np.float32)).to(torch.bfloat16)
    w = d((rng.standard_normal((3, 3, 64, 128)) / 24).astype(np.float32))            # forward filter 64 -> 128
    act = d(rng.standard_normal((2, 16, 64, 64)).astype(np.float32)).to(torch.bfloat16)
    pT = K.PackedConv(w, transpose_flip=True)
    fd = K.conv_desc(2, 16, 64, 64, 128, 3, 3, 1, True, 1)
    g_plain, _ = K.conv2d_dgrad(x, pT, fd, compute=K.BF16)
    g_mask, _ = K.conv2d_dgrad(x, pT, fd, compute=K.BF16, mask_bf16=act, mask_slope=0.0, out_bf16=True)
    assert torch.equal(g_mask, (g_plain * (act.float() > 0)).to(torch.bfloat16))
    assert torch.equal(g_mask.float(), K.affine_act_bwd(act, g_plain, None, None, 0.0).to(torch.bfloat16).float())
    y = d(rng.standard_normal((3, 16, 64, 128)).astype(np.float32)).to(torch.bfloat16)
    p32, p16 = K.maxpool(y, want_bf16=True)
    assert torch.equal(p32, K.maxpool(y.float())) and torch.equal(p16, p32.to(torch.bfloat16))
    dp = d(rng.standard_normal((3, 8, 32, 128)).astype(np.float32))
    assert torch.equal(K.maxpool_relu_bwd(y, dp), K.maxpool_relu_bwd(y.float(), dp))
    assert torch.equal(K.maxpool_relu_bwd(y, dp, out_bf16=True), K.maxpool_relu_bwd(y.float(), dp).to(torch.bfloat16))
    g = d(rng.standard_normal((3, 16, 64, 128)).astype(np.float32))
    assert torch.equal(K.affine_act_bwd(y, g, None, None, 0.0), K.affine_act_bwd(y.float(), g, None, None, 0.0))


def test_materialised_deconv_operand_equals_the_fused_upsample(dev):
    """hdrsky_up2x_xf_bf16: the resize-deconvolution's operand bf16(resize2x(leaky(IN(x)))) written once; a plain conv on
    it (x_bf16) against the conv that resizes while staging (upsample = 2), and the plain weight gradient likewise."""
    K = pkg("kernels"); L = pkg("_lib")
    rng = np.random.default_rng(41)
    d = lambda a: torch.from_numpy(a).to(dev)
    for (B, H, W, Cin, Cout) in ((2, 8, 32, 128, 64), (3, 16, 64, 64, 32)):
        x0 = d(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
        w0 = d((rng.standard_normal((3, 3, Cin, Cin)) / np.sqrt(9 * Cin)).astype(np.float32))
        raw, st = K.conv2d(x0, K.PackedConv(w0), None, want_stats=True, compute=K.BF16)
        gam = d(rng.uniform(0.5, 1.5, Cin).astype(np.float32)); bet = d(rng.standard_normal(Cin).astype(np.float32))
        w = d((rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32))
        b = d(rng.standard_normal(Cout).astype(np.float32))
        pw = K.PackedConv(w)
        for xf in (None, K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=gam, beta=bet)):
            y_fused, s_fused = K.conv2d(raw, pw, b, upsample=2, xf=xf, want_stats=True, compute=K.BF16)
            u = K.up2x_act_bf16(raw, xf)
            assert u.shape == (B, 2 * H, 2 * W, Cin) and u.dtype == torch.bfloat16
            y_plain, s_plain = K.conv2d(u, pw, b, want_stats=True, compute=K.BF16)
            assert_close(y_plain, y_fused, 2e-6, "plain conv on the materialised operand vs fused upsample")
            dy = d(rng.standard_normal((B, 2 * H, 2 * W, Cout)).astype(np.float32))
            dw_f, db_f = K.conv2d_wgrad(raw, dy, 3, 3, upsample=2, xf=xf, compute=K.BF16)
            dw_p = torch.zeros_like(dw_f); db_p = torch.zeros_like(db_f)
            K.conv2d_wgrad_multi([K.wgrad_job(u, dy, 3, 3, dw_p, db_p, compute=K.BF16)])
            assert_close(dw_p, dw_f, 2e-5, "plain weight gradient on the materialised operand vs fused upsample")
            assert_close(db_p, db_f, 1e-6, "bias gradient")
