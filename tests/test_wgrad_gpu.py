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
