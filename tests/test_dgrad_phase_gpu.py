"""GPU parity: the data gradient of the stride-2 convolutions (discriminator.py:11-13, sunrad_net.py:12-14 - Conv2D(4, strides=2);
generator.py:95-96 - ops.conv2d(k_h=3, stride=2)) by OUTPUT PHASES on the un-stuffed gradient (conv_igemm_kernel<..., PH = true>)
against (a) the zero-stuffed form it replaces - the same non-zero products in the same order, so bit for bit - and (b) the
autograd gradient of the oracle's conv (oracle/tfsem.py::conv2d, TF SAME padding incl. the asymmetric stride-2 pad)."""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tfsem as T
from util import TOL_X3, assert_close, assert_close_bf16

pytestmark = pytest.mark.gpu

# (name, H, W, Cin, Cout, k) of the FORWARD stride-2 SAME conv
CASES = [
    ("dis.d1 4x4 6->64 @32x128", 32, 128, 6, 64, 4),
    ("dis.d2 4x4 64->128 @16x64", 16, 64, 64, 128, 4),
    ("dis.d3 4x4 128->256 @8x32", 8, 32, 128, 256, 4),
    ("gen.conv2_d 3x3 32->64 @32x128", 32, 128, 32, 64, 3),
    ("gen.conv3_d 3x3 64->128 @16x64", 16, 64, 64, 128, 3),
    ("wide 4x4 256->512 @8x32 (several channel groups)", 8, 32, 256, 512, 4),
    ("odd 3x3 32->64 on 9x37", 9, 37, 32, 64, 3),
    ("odd 4x4 32->64 on 7x13", 7, 13, 32, 64, 4),
    ("even 4x4 32->32 on 10x22", 10, 22, 32, 32, 4),
    ("5x5 32->64 on 12x20", 12, 20, 32, 64, 5),
]


def _no_phase(flag):
    if flag:
        os.environ["HDRSKY_NO_PHASE"] = "1"
    else:
        os.environ.pop("HDRSKY_NO_PHASE", None)
    pkg("hooks").reload()          # the switches are read once (hooks.py, csrc/hooks.h)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("B", [2, 3])
def test_stride2_dgrad_by_phases(dev, case, B):
    K = pkg("kernels")
    name, H, W, Cin, Cout, k = case
    rng = np.random.default_rng(zlib.crc32(name.encode()) + B)
    w = torch.from_numpy((rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32))
    x = torch.zeros((B, H, W, Cin), dtype=torch.float32, requires_grad=True)
    y = T.conv2d(x, w, None, 2, "SAME")
    dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    (gx,) = torch.autograd.grad(y, x, dy)
    fd = K.conv_desc(B, H, W, Cin, Cout, k, k, 2, True, 1)
    assert (fd.Ho, fd.Wo) == tuple(y.shape[1:3])
    pT = K.PackedConv(w.to(dev), precise=True, transpose_flip=True)
    dyd = dy.to(dev)
    try:
        _no_phase(False)
        assert K.conv_kernel_name(K.conv_dgrad_desc(fd)).endswith("true, false>"), "the phase form should be selected"
        g_ph, _ = K.conv2d_dgrad(dyd, pT, fd, compute=K.BF16)
        b16 = Cin % 4 == 0                                            # (bf16 outputs are stored in groups of four channels)
        g_ph16, _ = K.conv2d_dgrad(dyd.to(torch.bfloat16), pT, fd, compute=K.BF16, out_bf16=b16)
        res = torch.from_numpy(rng.standard_normal(tuple(gx.shape)).astype(np.float32)).to(dev)
        g_ph_res, _ = K.conv2d_dgrad(dyd, pT, fd, residual=res, compute=K.BF16)
        _no_phase(True)
        assert K.conv_kernel_name(K.conv_dgrad_desc(fd)).endswith("false, false>")
        g_st, _ = K.conv2d_dgrad(dyd, pT, fd, compute=K.BF16)
        g_st16, _ = K.conv2d_dgrad(dyd.to(torch.bfloat16), pT, fd, compute=K.BF16, out_bf16=b16)
        g_st_res, _ = K.conv2d_dgrad(dyd, pT, fd, residual=res, compute=K.BF16)
        g_x3, _ = K.conv2d_dgrad(dyd, pT, fd, compute=K.BF16X3)       # (two-plane mode: always the zero-stuffed form)
    finally:
        _no_phase(False)
    assert tuple(g_ph.shape) == (B, H, W, Cin)
    if Cout <= 256:
        assert torch.equal(g_ph, g_st), "%s: phases vs zero-stuffed differ by %g" % (name, float((g_ph - g_st).abs().max()))
        assert torch.equal(g_ph16, g_st16) and torch.equal(g_ph_res, g_st_res)
    else:
        # the zero-stuffed halo planes of a 512-channel gradient need several channel groups in LDS, the un-stuffed ones do
        # not: the same products, summed group-major there and tap-major here (fp32 rounding apart)
        assert_close(g_ph, g_st, 4e-6, name + " phases vs zero-stuffed")
        assert_close(g_ph_res, g_st_res, 4e-6, name + " phases vs zero-stuffed (+ residual)")
    assert_close(g_x3, gx, TOL_X3, name + " x3")
    assert_close_bf16(g_ph, gx, name + " phases vs oracle autograd")
