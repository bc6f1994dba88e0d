"""GPU parity: hdrsky_conv2d_wgrad vs torch autograd of the oracle conv (all trainable layer shapes)."""
import zlib

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tfsem as T
from util import assert_close, assert_close_bf16

pytestmark = pytest.mark.gpu

# (name, H, W, Cin, Cout, k, stride, same, upsample)
LAYERS = [
    ("conv1_d 7x7 3->32", 32, 128, 3, 32, 7, 1, True, 1),
    ("conv2_d 3x3 s2 32->64", 32, 128, 32, 64, 3, 2, True, 1),
    ("conv3_d 3x3 s2 64->128", 16, 64, 64, 128, 3, 2, True, 1),
    ("res 3x3 128->128", 8, 32, 128, 128, 3, 1, True, 1),
    ("conv3_f up+3x3 128->64", 8, 32, 128, 64, 3, 1, True, 2),
    ("conv2_f up+3x3 64->32", 16, 64, 64, 32, 3, 1, True, 2),
    ("conv1_f 7x7 32->3", 32, 128, 32, 3, 7, 1, True, 1),
    ("s.l1.conv2 7x7 32->32", 32, 128, 32, 32, 7, 1, True, 1),
    ("d1 4x4 s2 6->64", 32, 128, 6, 64, 4, 2, True, 1),
    ("d2 4x4 s2 64->128", 16, 64, 64, 128, 4, 2, True, 1),
    ("d3 4x4 s2 128->256", 8, 32, 128, 256, 4, 2, True, 1),
    ("d4 4x4 s1 256->512", 4, 16, 256, 512, 4, 1, True, 1),
    ("dis.out 4x4 VALID 512->1", 4, 16, 512, 1, 4, 1, False, 1),
]


@pytest.mark.parametrize("case", LAYERS, ids=[c[0] for c in LAYERS])
def test_wgrad_layers(dev, case):
    K = pkg("kernels")
    name, H, W, Cin, Cout, k, stride, same, up = case
    B = 3
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    xin = T.resize_bilinear(x, 2 * H, 2 * W) if up == 2 else x
    y = T.conv2d(xin, w, b, stride, "SAME" if same else "VALID")
    dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    gw, gb = torch.autograd.grad(y, (w, b), dy)
    dw, db = K.conv2d_wgrad(x.to(dev), dy.to(dev), k, k, stride=stride, same=same, upsample=up, compute=K.BF16X3)
    assert_close(dw, gw, 3e-4, name + " dw x3")
    assert_close(db, gb, 1e-4, name + " db")
    dw16, _ = K.conv2d_wgrad(x.to(dev), dy.to(dev), k, k, stride=stride, same=same, upsample=up, compute=K.BF16)
    assert_close_bf16(dw16, gw, name + " dw bf16")


def test_wgrad_with_fused_in_transform(dev):
    """wgrad consumes the same fused operand (IN + lrelu from the producer's partials) as the forward conv."""
    K = pkg("kernels"); L = pkg("_lib")
    rng = np.random.default_rng(3)
    B, H, W = 2, 16, 64
    x = rng.standard_normal((B, H, W, 32)).astype(np.float32) * 2 + 0.3
    w1 = (rng.standard_normal((3, 3, 32, 64)) / 17).astype(np.float32)
    gam = rng.uniform(0.5, 1.5, 64).astype(np.float32); bet = rng.standard_normal(64).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(dev)
    r1, st = K.conv2d(d(x), K.PackedConv(d(w1)), None, want_stats=True, compute=K.BF16X3)
    a1 = T.leaky_relu(T.instance_norm(r1.cpu(), torch.from_numpy(gam), torch.from_numpy(bet)), 0.1)
    w2 = torch.from_numpy((rng.standard_normal((3, 3, 64, 32)) / 24).astype(np.float32)).requires_grad_(True)
    y = T.conv2d(a1, w2, None)
    dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    (gw,) = torch.autograd.grad(y, w2, dy)
    xf = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=d(gam), beta=d(bet))
    dw, _ = K.conv2d_wgrad(r1, dy.to(dev), 3, 3, xf=xf, compute=K.BF16X3)
    assert_close(dw, gw, 4e-4, "fused-operand wgrad")


def test_wgrad_multi_matches_autograd(dev):
    """hdrsky_conv2d_wgrad_multi: a mixed list of layers (>= 3 wide stride-1 layers -> the large-block grouped geometry,
    plus narrow / 7x7 / strided / resize-deconv layers in the same call) against torch autograd of the oracle conv."""
    K = pkg("kernels")
    B = 3
    shapes = [("res a", 8, 32, 128, 128, 3, 1, 1), ("res b", 8, 32, 128, 128, 3, 1, 1), ("l3a", 8, 32, 64, 128, 3, 1, 1),
              ("l2b", 16, 64, 64, 64, 3, 1, 1), ("d4", 4, 16, 256, 512, 4, 1, 1), ("up", 8, 32, 128, 64, 3, 1, 2),
              ("conv1", 32, 128, 3, 32, 7, 1, 1), ("l1b", 16, 64, 32, 32, 7, 1, 1), ("d2", 16, 64, 64, 128, 4, 2, 1)]
    for compute, tol in ((K.BF16X3, 3e-4), (K.BF16, None)):
        jobs, refs = [], []
        for name, H, W, Cin, Cout, k, stride, up in shapes:
            rng = np.random.default_rng(zlib.crc32(name.encode()))
            x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
            w = torch.zeros(k, k, Cin, Cout, requires_grad=True)
            b = torch.zeros(Cout, requires_grad=True)
            xin = T.resize_bilinear(x, 2 * H, 2 * W) if up == 2 else x
            y = T.conv2d(xin, w, b, stride, "SAME")
            dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
            refs.append(torch.autograd.grad(y, (w, b), dy))
            dw = torch.zeros(k, k, Cin, Cout, device=dev)
            db = torch.zeros(Cout, device=dev)
            jobs.append(K.wgrad_job(x.to(dev), dy.to(dev), k, k, dw, db, stride=stride, upsample=up, compute=compute))
        K.conv2d_wgrad_multi(jobs)
        for (name, *_), job, (gw, gb) in zip(shapes, jobs, refs):
            if tol is None:
                assert_close_bf16(job[4], gw, name + " dw bf16 (multi)")
            else:
                assert_close(job[4], gw, tol, name + " dw (multi)")
            assert_close(job[5], gb, 1e-4, name + " db (multi)")


# (name, H, W, Cin, Cout, k, stride): every layer geometry the LDS-DMA weight-gradient kernel (conv_wgrad2_kernel) takes in
# the training step, both strides, all four dW block shapes, odd map sizes, the 128x512 configuration's res maps
V2_LAYERS = [
    ("res 3x3 128->128 @8x32", 8, 32, 128, 128, 3, 1), ("l3a 3x3 64->128 @8x32", 8, 32, 64, 128, 3, 1),
    ("l2b 3x3 64->64 @16x64", 16, 64, 64, 64, 3, 1), ("l2a 3x3 32->64 @16x64", 16, 64, 32, 64, 3, 1),
    ("dec2 3x3 64->32 @32x128", 32, 128, 64, 32, 3, 1), ("l1b 7x7 32->32 @32x128", 32, 128, 32, 32, 7, 1),
    ("d4 4x4 256->512 @4x16", 4, 16, 256, 512, 4, 1), ("d2 4x4 s2 64->128 @16x64", 16, 64, 64, 128, 4, 2),
    ("d3 4x4 s2 128->256 @8x32", 8, 32, 128, 256, 4, 2), ("conv2_d 3x3 s2 32->64 @32x128", 32, 128, 32, 64, 3, 2),
    ("conv3_d 3x3 s2 64->128 @16x64", 16, 64, 64, 128, 3, 2), ("odd 3x3 64->64 @10x36", 10, 36, 64, 64, 3, 1),
    ("odd 5x5 32->32 @7x19", 7, 19, 32, 32, 5, 1), ("hires res 3x3 128->128 @32x128", 32, 128, 128, 128, 3, 1),
]


@pytest.mark.parametrize("case", V2_LAYERS, ids=[c[0] for c in V2_LAYERS])
@pytest.mark.parametrize("B", [1, 5])
def test_wgrad_dma_kernel_layers(dev, case, B, monkeypatch):
    """Both operands final bf16 tensors -> conv_wgrad2_kernel (LDS-DMA ring, tap groups x dW blocks x pixel chunks, shared
    fixed-order reduce).  On the SAME bf16 operands: (a) torch autograd of the oracle conv in float64 - only the fp32
    accumulation order differs, (b) the register-staged kernel (HDRSKY_WGRAD2=0), (c) bit-reproducible on a second call."""
    K = pkg("kernels")
    name, H, W, Cin, Cout, k, stride = case
    rng = np.random.default_rng(zlib.crc32((name + str(B)).encode()))
    xb = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).to(torch.bfloat16)
    w = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = T.conv2d(xb.double(), w, b, stride, "SAME")
    dyb = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).to(torch.bfloat16)
    gw, gb = torch.autograd.grad(y, (w, b), dyb.double())

    def run():
        dw, db = torch.zeros(k, k, Cin, Cout, device=dev), torch.zeros(Cout, device=dev)
        K.conv2d_wgrad_multi([K.wgrad_job(xb.to(dev), dyb.to(dev), k, k, dw, db, stride=stride, compute=K.BF16)])
        return dw, db
    dw, db = run()
    assert_close(dw, gw, 2e-5, name + " dw (LDS-DMA kernel)")
    assert_close(db, gb, 2e-5, name + " db (LDS-DMA kernel)")
    dw2, db2 = run()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    monkeypatch.setenv("HDRSKY_WGRAD2", "0")
    dw1, db1 = run()
    assert_close(dw1, dw, 2e-5, name + " register-staged vs LDS-DMA kernel")
    assert_close(db1, db, 2e-5, name + " db register-staged vs LDS-DMA kernel")


def test_wgrad_dma_kernel_on_materialised_operands(dev, monkeypatch):
    """fp32 inputs that still need their operand transform - InstanceNorm from the producer's partials + leaky, a per-sample
    BatchNorm affine + LeakyReLU(0.3), a plain fp32 activation - are materialised once as bf16 (hdrsky_act_bf16, the staging's
    own arithmetic) and go through the LDS-DMA kernel: same result as the register-staged kernel that transforms while
    staging (bf16 operands in both: only the summation order differs), several layers in one call."""
    K = pkg("kernels"); L = pkg("_lib")
    rng = np.random.default_rng(12)
    B = 4
    d = lambda a: torch.from_numpy(a).to(dev)

    def jobs():
        out = []
        x1 = d(rng.standard_normal((B, 16, 64, 32)).astype(np.float32) * 2 + 0.3)
        r1, st = K.conv2d(x1, K.PackedConv(d((rng.standard_normal((3, 3, 32, 64)) / 17).astype(np.float32))), None, want_stats=True,
                          compute=K.BF16)
        xf1 = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=d(rng.uniform(0.5, 1.5, 64).astype(np.float32)),
                     beta=d(rng.standard_normal(64).astype(np.float32)))
        dy1 = d(rng.standard_normal((B, 16, 64, 64)).astype(np.float32)).to(torch.bfloat16)
        out.append(K.wgrad_job(r1, dy1, 3, 3, torch.zeros(3, 3, 64, 64, device=dev), torch.zeros(64, device=dev), xf=xf1, compute=K.BF16))
        x2 = d(rng.standard_normal((B, 8, 32, 128)).astype(np.float32))
        xf2 = K.InXf(mode=L.IN_AFFINE, slope=0.3, scale=d(rng.uniform(0.5, 1.5, (B, 128)).astype(np.float32)),
                     shift=d(rng.standard_normal((B, 128)).astype(np.float32)))
        dy2 = d(rng.standard_normal((B, 4, 16, 256)).astype(np.float32)).to(torch.bfloat16)
        out.append(K.wgrad_job(x2, dy2, 4, 4, torch.zeros(4, 4, 128, 256, device=dev), None, stride=2, xf=xf2, compute=K.BF16))
        x3 = d(np.maximum(rng.standard_normal((B, 16, 64, 32)), 0).astype(np.float32))
        dy3 = d(rng.standard_normal((B, 16, 64, 64)).astype(np.float32)).to(torch.bfloat16)
        out.append(K.wgrad_job(x3, dy3, 3, 3, torch.zeros(3, 3, 32, 64, device=dev), torch.zeros(64, device=dev), compute=K.BF16))
        return out
    state = rng.bit_generator.state
    a = jobs()
    K.conv2d_wgrad_multi(a)
    rng.bit_generator.state = state
    monkeypatch.setenv("HDRSKY_WGRAD2", "0")
    b = jobs()
    K.conv2d_wgrad_multi(b)
    for i, (ja, jb) in enumerate(zip(a, b)):
        assert float(ja[4].abs().max()) > 0
        assert_close(ja[4], jb[4], 3e-5, "layer %d dw: materialised bf16 operand + LDS-DMA kernel vs transform while staging" % i)
        if ja[5] is not None:
            assert_close(ja[5], jb[5], 3e-5, "layer %d db" % i)


# (name, H, W, Cin, Cout, k, stride, same): the layers with a narrow side (conv_wgrad3_kernel): <= 8 input channels or <= 4
# output channels, both block widths, both pixel paddings (4 / 8 channels), stride 2, VALID, odd map sizes, the 128x512 stem
V3_LAYERS = [
    ("stem 7x7 3->32 @32x128", 32, 128, 3, 32, 7, 1, True), ("tail 7x7 32->3 @32x128", 32, 128, 32, 3, 7, 1, True),
    ("d1 4x4 s2 6->64 @32x128", 32, 128, 6, 64, 4, 2, True), ("dis.out 4x4 VALID 512->1 @4x16", 4, 16, 512, 1, 4, 1, False),
    ("odd 5x5 4->16 @9x37", 9, 37, 4, 16, 5, 1, True), ("odd 3x3 s2 8->48 @11x21", 11, 21, 8, 48, 3, 2, True),
    ("odd 3x3 48->2 @7x19", 7, 19, 48, 2, 3, 1, True), ("odd 4x4 64->4 @6x10", 6, 10, 64, 4, 4, 1, True),
    ("valid 3x3 1->16 @8x40", 8, 40, 1, 16, 3, 1, False), ("hires stem 7x7 3->32 @128x512", 128, 512, 3, 32, 7, 1, True),
]


def _bf16_values(rng, shape):
    """fp32 tensor whose values are bf16-representable: both kernels (and float64 autograd) then multiply the same numbers."""
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(torch.bfloat16).float()


@pytest.mark.parametrize("case", V3_LAYERS, ids=[c[0] for c in V3_LAYERS])
@pytest.mark.parametrize("B", [1, 5])
def test_wgrad_narrow_kernel_layers(dev, case, B, monkeypatch):
    """Layers with a narrow side -> conv_wgrad3_kernel (the narrow tensor as [row][column][4 or 8 channels] in LDS, M fragments =
    neighbouring columns x channels of a filter row; narrow OUTPUT: the same with the roles of x and dy swapped, taps
    flipped).  On bf16-representable operands: (a) float64 autograd of the oracle conv, (b) the register-staged kernel
    (HDRSKY_WGRAD3=0), (c) bit-reproducible, (d) the wide operand handed over as a bf16 tensor gives the same bits."""
    K = pkg("kernels")
    name, H, W, Cin, Cout, k, stride, same = case
    if H * W * B > 200000:
        B = min(B, 2)
    rng = np.random.default_rng(zlib.crc32((name + str(B)).encode()))
    x = _bf16_values(rng, (B, H, W, Cin))
    if Cout == 1:    # (float64 autograd of a one-filter conv trips torch's slow_conv2d contiguity check: the sum written out)
        pt, pl = (T.same_pad(H, k, stride)[0], T.same_pad(W, k, stride)[0]) if same else (0, 0)
        Ho, Wo = (-(-H // stride), -(-W // stride)) if same else ((H - k) // stride + 1, (W - k) // stride + 1)
        dy = _bf16_values(rng, (B, Ho, Wo, Cout))
        xp = torch.zeros(B, H + 2 * k, W + 2 * k, Cin, dtype=torch.float64)
        xp[:, pt:pt + H, pl:pl + W] = x.double()
        gw = torch.zeros(k, k, Cin, Cout, dtype=torch.float64)
        for ky in range(k):
            for kx in range(k):
                win = xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride]
                gw[ky, kx] = torch.einsum("bhwc,bhwo->co", win, dy.double())
        gb = dy.double().sum((0, 1, 2))
    else:
        w = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
        b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
        y = T.conv2d(x.double(), w, b, stride, "SAME" if same else "VALID")
        dy = _bf16_values(rng, tuple(y.shape))
        gw, gb = torch.autograd.grad(y, (w, b), dy.double())

    def run(xd, dyd):
        dw, db = torch.zeros(k, k, Cin, Cout, device=dev), torch.zeros(Cout, device=dev)
        K.conv2d_wgrad_multi([K.wgrad_job(xd, dyd, k, k, dw, db, stride=stride, same=same, compute=K.BF16)])
        return dw, db
    dw, db = run(x.to(dev), dy.to(dev))
    assert_close(dw, gw, 2e-5, name + " dw (narrow kernel)")
    assert_close(db, gb, 2e-5, name + " db (narrow kernel)")
    dw2, db2 = run(x.to(dev), dy.to(dev))
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    if Cin > 8:
        dwb, dbb = run(x.to(dev).to(torch.bfloat16), dy.to(dev))
        assert torch.equal(dw, dwb) and torch.equal(db, dbb)
    elif Cout % 8 == 0:
        dwb, dbb = run(x.to(dev), dy.to(dev).to(torch.bfloat16))
        assert torch.equal(dw, dwb) and torch.equal(db, dbb)
    if Cin > 8 and Cin % 32:
        return      # (the register-staged kernel takes 32-channel multiples only)
    monkeypatch.setenv("HDRSKY_WGRAD3", "0")
    dw1, db1 = run(x.to(dev), dy.to(dev))
    assert_close(dw1, dw, 2e-5, name + " register-staged vs narrow kernel")
    assert_close(db1, db, 2e-5, name + " db register-staged vs narrow kernel")


def test_wgrad_narrow_kernel_with_operand_transform(dev, monkeypatch):
    """The decoder tails (7x7 32->3) read a raw conv output through InstanceNorm-from-partials + leaky, the discriminator head
    a BatchNorm affine per sample + LeakyReLU(0.3): conv_wgrad3_kernel applies the wide operand's transform while staging,
    like the register-staged kernel - same bf16 operands, only the summation order differs.  Both in one call, together with
    a stem layer and a wide layer that goes to the LDS-DMA kernel."""
    K = pkg("kernels"); L = pkg("_lib")
    B = 4
    d = lambda a: torch.from_numpy(a).to(dev)

    def jobs():
        rng = np.random.default_rng(21)
        out = []
        x1 = d(rng.standard_normal((B, 16, 64, 64)).astype(np.float32) * 2 + 0.3)
        r1, st = K.conv2d(x1, K.PackedConv(d((rng.standard_normal((3, 3, 64, 32)) / 24).astype(np.float32))), None, want_stats=True,
                          compute=K.BF16)
        xf1 = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=d(rng.uniform(0.5, 1.5, 32).astype(np.float32)),
                     beta=d(rng.standard_normal(32).astype(np.float32)))
        dy1 = d(rng.standard_normal((B, 16, 64, 3)).astype(np.float32))
        out.append(K.wgrad_job(r1, dy1, 7, 7, torch.zeros(7, 7, 32, 3, device=dev), torch.zeros(3, device=dev), xf=xf1, compute=K.BF16))
        x2 = d(rng.standard_normal((B, 4, 16, 512)).astype(np.float32))
        xf2 = K.InXf(mode=L.IN_AFFINE, slope=0.3, scale=d(rng.uniform(0.5, 1.5, (B, 512)).astype(np.float32)),
                     shift=d(rng.standard_normal((B, 512)).astype(np.float32)))
        dy2 = d(rng.standard_normal((B, 1, 13, 1)).astype(np.float32))
        out.append(K.wgrad_job(x2, dy2, 4, 4, torch.zeros(4, 4, 512, 1, device=dev), torch.zeros(1, device=dev), same=False, xf=xf2,
                               compute=K.BF16))
        x3 = d(rng.standard_normal((B, 32, 128, 3)).astype(np.float32))
        dy3 = d(rng.standard_normal((B, 32, 128, 32)).astype(np.float32)).to(torch.bfloat16)
        out.append(K.wgrad_job(x3, dy3, 7, 7, torch.zeros(7, 7, 3, 32, device=dev), torch.zeros(32, device=dev), compute=K.BF16))
        x4 = d(rng.standard_normal((B, 8, 32, 128)).astype(np.float32)).to(torch.bfloat16)
        dy4 = d(rng.standard_normal((B, 8, 32, 128)).astype(np.float32)).to(torch.bfloat16)
        out.append(K.wgrad_job(x4, dy4, 3, 3, torch.zeros(3, 3, 128, 128, device=dev), torch.zeros(128, device=dev), compute=K.BF16))
        return out
    a = jobs()
    K.conv2d_wgrad_multi(a)
    monkeypatch.setenv("HDRSKY_WGRAD3", "0")
    b = jobs()
    K.conv2d_wgrad_multi(b)
    for i, (ja, jb) in enumerate(zip(a, b)):
        assert float(ja[4].abs().max()) > 0
        assert_close(ja[4], jb[4], 3e-5, "layer %d dw: narrow kernel vs register-staged kernel" % i)
        assert_close(ja[5], jb[5], 3e-5, "layer %d db" % i)
