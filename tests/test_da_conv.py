"""Distortion-aware conv: host offset table (CPU) and the fused gather+MFMA kernel (GPU) vs the numpy oracle."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import da_ops
from util import TOL_X3, assert_close, assert_close_bf16


@pytest.mark.parametrize("h,w,k,dil", [(8, 32, 3, 1), (32, 128, 3, 1), (16, 64, 5, 1), (8, 32, 3, 2), (32, 128, 7, 1)])
def test_offset_table_matches_oracle(h, w, k, dil):
    """hdrsky_da_offsets (C, libm float32) vs the numpy float32 restatement, incl. the row-0 branch flip."""
    K = pkg("kernels")
    got = K.da_offsets(h, w, k, dil, True)
    ref = da_ops.distortion(h, w, k, dil, True)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-4, np.abs(got - ref).max()   # float32 libm vs numpy: a few ulp of ~64
    # row 0: cos(float32(pi)/2) < 0 sends the gx=-1 taps ~2w to the left
    if k == 3 and dil == 1:
        assert got[0, 2, 1] < -1.5 * w and got[0, 5, 1] < -1.5 * w and got[0, 8, 1] < -1.5 * w
    got_flat = K.da_offsets(h, w, k, dil, False)
    assert np.abs(got_flat - da_ops.distortion(h, w, k, dil, False)).max() <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 8, 32, 128, 128), (2, 16, 64, 64, 64), (1, 32, 128, 32, 32), (3, 8, 32, 32, 96)])
def test_da_conv_matches_oracle(dev, shape):
    K = pkg("kernels")
    B, H, W, C, F = shape
    rng = np.random.default_rng(B * 131 + C)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    kern = (rng.standard_normal((9 * C, F)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(F).astype(np.float32)
    offs = K.da_offsets(H, W, 3, 1, True)
    ref = da_ops.da_conv2d(x, kern, bias, da_ops.distortion(H, W))
    d = lambda a: torch.from_numpy(a).to(dev)
    pw = K.PackedConv(d(kern).view(3, 3, C, F))
    y = K.da_conv2d(d(x), pw, d(bias), d(offs), K.BF16X3)
    assert_close(y, ref, 3e-4, "da conv x3")
    y16 = K.da_conv2d(d(x), pw, d(bias), d(offs), K.BF16)
    assert_close_bf16(y16, ref, "da conv bf16")
    # zero offsets == SAME stride-1 conv (known-answer reduction)
    y0 = K.da_conv2d(d(x), pw, d(bias), torch.zeros_like(d(offs)), K.BF16X3)
    c0, _ = K.conv2d(d(x), pw, d(bias), compute=K.BF16X3)
    assert_close(y0, c0, 1e-5, "zero-offset DA conv == SAME conv")


@pytest.mark.gpu
def test_da_layer_api(dev):
    """Reference-style layer objects: conv2d as a res-block drop-in and the resize-deconv."""
    da = pkg("distortion_aware_ops"); K = pkg("kernels")
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 8, 32, 128)).astype(np.float32)).to(dev)
    layer = da.conv2d(128, kernel_size=3, strides=1, dilation_rate=1, compute=K.BF16X3)
    y = layer(x)
    assert tuple(layer.kernel.shape) == (9 * 128, 128) and layer.offset.shape == (1, 8, 32, 9, 2)
    ref = da_ops.da_conv2d(x.cpu().numpy(), layer.kernel.cpu().numpy(), layer.bias.cpu().numpy(), da_ops.distortion(8, 32))
    assert_close(y, ref, 3e-4, "layer conv2d")
    dl = da.deconv2d(64, kernel_size=3, output_imshape=[16, 64], compute=K.BF16X3)
    z = dl(x)
    refz = da_ops.da_deconv2d(x.cpu().numpy(), dl.kernel.cpu().numpy(), dl.bias.cpu().numpy(), da_ops.distortion(16, 64), 16, 64)
    assert_close(z, refz, 3e-4, "layer deconv2d")
