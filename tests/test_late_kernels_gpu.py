"""Unit parity of the four fused kernels that entered at the end of round 4 and were covered only through whole-step
comparisons (missing 4): hdrsky_head_bwd, hdrsky_maxpool_relu_l1_bwd_bf16, hdrsky_rmsprop2 and
hdrsky_rmsprop_fc_fused_bias - each against the oracle's arithmetic (oracle/tfsem.py) or a torch fp32 autograd restatement of
the reference lines it replaces."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tfsem as T
from util import assert_close, to_np

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_adv,with_dyg", [(True, True), (False, True), (True, False)])
def test_head_bwd_against_autograd_of_the_blend_and_decoder_tails(dev, with_adv, with_dyg):
    """hdrsky_head_bwd = backward of train.py:293-299 (alpha mask as a constant, blend, log decompression) and of both decoder
    tails y = relu(res + lrelu(c, 0.1)) (generator.py:121-125,152-156) + the slice [..., 3:6] of the adversarial term's input
    gradient, in one launch.  Checked against torch autograd over oracle/tfsem on random inputs whose sky radiance covers the
    three regimes of the mask: alpha = 0 (max_c sky_lin <= 0.88), the ramp, alpha = 1 (>= 1)."""
    K = pkg("kernels")
    rng = np.random.default_rng(5)
    B, H, W = 3, 32, 128
    shp = (B, H, W, 3)
    f = lambda lo, hi: torch.from_numpy(rng.uniform(lo, hi, shp).astype(np.float32))
    c_f, c_u = f(-0.6, 0.8), f(-0.6, 0.8)
    res_f, res_u = f(0.0, 0.75), f(-0.2, 0.8)
    dyg, dyl, din6 = f(-1, 1), f(-1, 1), torch.from_numpy(rng.standard_normal((B, H, W, 6)).astype(np.float32))
    cf, cu, ru = (t.clone().requires_grad_(True) for t in (c_f, c_u, res_u))
    y_f = torch.relu(res_f + T.leaky_relu(cf, 0.1))
    y_u = torch.relu(ru + T.leaky_relu(cu, 0.1))
    with torch.no_grad():        # gen_tape.stop_recording (train.py:257-261): the mask is a constant of the gradient
        sky_lin = T.hdr_log_decompression(y_f)
        alpha = torch.clamp((sky_lin.max(dim=-1, keepdim=True).values - 1.0 + 0.12) / 0.12, 0.0, 1.0).expand(shp).contiguous()
    frac = [float((alpha == 0).float().mean()), float(((alpha > 0) & (alpha < 1)).float().mean()), float((alpha == 1).float().mean())]
    assert min(frac) > 0.02, "the inputs must cover alpha = 0 / ramp / 1: %s" % frac
    y_gamma = (1.0 - alpha) * y_f + alpha * y_u
    y_lin = T.hdr_log_decompression(y_gamma)
    gl = dyl + (din6[..., 3:6] if with_adv else 0)
    loss = (y_lin * gl).sum() + ((y_gamma * dyg).sum() if with_dyg else 0.0)
    loss.backward()
    d = lambda t: None if t is None else t.detach().to(dev).contiguous()
    dc_f, dc_u, dres_u = K.head_bwd(d(y_gamma), d(alpha), d(dyg) if with_dyg else None, d(dyl), d(din6) if with_adv else None,
                                    d(y_f), d(res_f), d(y_u), d(ru))
    torch.cuda.synchronize()
    # fp32 pointwise arithmetic on both sides: expf against torch.exp, one rounding apart
    for name, got, ref in (("d c_f", dc_f, cf.grad), ("d c_u", dc_u, cu.grad), ("d res_u", dres_u, ru.grad)):
        assert_close(got, ref, 2e-6, "head_bwd %s" % name)
    # and the three launches it replaced
    ds, du = K.blend_bwd(d(y_gamma), d(alpha), d(dyg) if with_dyg else None, d((dyl + din6[..., 3:6]) if with_adv else dyl))
    a, _ = K.decoder_tail_bwd(d(y_f), d(res_f), ds, False)
    b, c = K.decoder_tail_bwd(d(y_u), d(ru), du, True)
    assert_close(dc_f, a, 1e-6, "head_bwd vs blend_bwd + decoder_tail_bwd (sky)")
    assert_close(dc_u, b, 1e-6, "head_bwd vs blend_bwd + decoder_tail_bwd (sun)")
    assert_close(dres_u, c, 1e-6, "head_bwd vs blend_bwd + decoder_tail_bwd (residual)")


@pytest.mark.parametrize("with_dp,out_bf16", [(False, True), (True, True), (True, False)])
def test_maxpool_relu_l1_bwd_against_torch_with_ties(dev, with_dp, out_bf16):
    """hdrsky_maxpool_relu_l1_bwd_bf16: the perceptual L1 term of one VGG16 block (train.py:311-313: mean|pool_i(pred) -
    pool_i(target)|, weight 0.01) folded into the backward of the block's 2x2 max-pool + ReLU (vgg16.py:38-41).  Against torch
    fp32: F.max_pool2d backward routes a tie to the FIRST maximum of the window in row-major order (so does TF's MaxPoolGrad
    and the kernel); a bf16 ReLU map holds many ties (zeros, and equal positive values on the coarse bf16 lattice) - the
    test map is quantised to multiples of 1/8 to make them frequent."""
    K = pkg("kernels")
    rng = np.random.default_rng(9)
    B, H, W, C = 4, 16, 64, 64
    y32 = np.maximum(np.round(rng.standard_normal((B, H, W, C)) * 4) / 8, 0).astype(np.float32)   # ReLU output, exact in bf16
    target = (rng.standard_normal((B, H // 2, W // 2, C)) * 0.4).astype(np.float32)
    dp = rng.standard_normal((B, H // 2, W // 2, C)).astype(np.float32) * 1e-3 if with_dp else None
    wl, wg = 0.37, 0.01 * 0.37
    yt = torch.from_numpy(y32).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    pool = torch.nn.functional.max_pool2d(torch.relu(yt), 2, 2)
    ties = float(((torch.nn.functional.unfold(torch.from_numpy(y32).permute(0, 3, 1, 2).reshape(B * C, 1, H, W), 2, stride=2)
                   == pool.detach().reshape(B * C, 1, -1)).sum(1) > 1).float().mean())
    assert ties > 0.1, "the map must hold ties (%.2f of the windows)" % ties
    tgt = torch.from_numpy(target).permute(0, 3, 1, 2)
    l1 = (pool - tgt).abs().mean()
    obj = wg * l1 + ((pool * torch.from_numpy(dp).permute(0, 3, 1, 2)).sum() if with_dp else 0.0)
    obj.backward()
    ref = yt.grad.permute(0, 2, 3, 1).contiguous()
    yb = torch.from_numpy(y32).to(dev).to(torch.bfloat16)
    p32, _ = K.maxpool(yb)
    assert torch.equal(p32.cpu(), pool.detach().permute(0, 2, 3, 1).contiguous())
    slot = torch.zeros(1, device=dev)
    got = K.maxpool_relu_l1_bwd(yb, p32, torch.from_numpy(target).to(dev), None if dp is None else torch.from_numpy(dp).to(dev),
                                wl, wg, slot, out_bf16=out_bf16)
    torch.cuda.synchronize()
    assert got.dtype == (torch.bfloat16 if out_bf16 else torch.float32)
    if out_bf16:      # the stored gradient is the fp32 value rounded to bf16 (round to nearest even)
        assert torch.equal(got.cpu(), ref.to(torch.bfloat16)) or \
            float((got.float().cpu() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max())
        assert float(((got.float().cpu() != 0) != (ref != 0)).float().mean()) == 0.0, "routing differs (ties / ReLU mask)"
    else:
        assert_close(got, ref, 1e-6, "maxpool_relu_l1_bwd gradient")
    assert abs(float(slot.item()) - wl * float(l1)) <= 1e-5 * wl * float(l1) + 1e-9, "L1 term value"     # (fp32 block sums + atomics over 131 k terms)
    # equal to the two launches it replaced: hdrsky_l1 (gradient into dp) + hdrsky_maxpool_relu_bwd_bf16
    dpl = torch.zeros_like(p32) if dp is None else torch.from_numpy(dp).to(dev)
    slot2 = torch.zeros(1, device=dev)
    K.l1(p32, torch.from_numpy(target).to(dev), wl, wg, slot2, da=dpl, accumulate=True)
    two = K.maxpool_relu_bwd(yb, dpl, out_bf16=out_bf16)
    assert torch.equal(two, got), "fused launch != hdrsky_l1 + hdrsky_maxpool_relu_bwd_bf16"


def test_rmsprop2_against_the_oracle_update_per_element(dev):
    """hdrsky_rmsprop2 (both optimizers' conv-side parameters in one launch, train.py:402-406) against
    oracle/tfsem.rmsprop_update (Keras-2 OptimizerV2: ms <- rho ms + (1 - rho) g^2; w <- w - lr g / (sqrt(ms) + 1e-7)) per
    element, 1e-6, with a gradient scale (the data-parallel 1/world), ragged lengths (multiples of 4: the flat buffers'
    padding) and gradients spanning zero / tiny / large magnitudes; and bit-identical to two hdrsky_rmsprop launches."""
    K = pkg("kernels")
    rng = np.random.default_rng(3)
    n1, n2 = 4 * 12345, 4 * 777
    mk = lambda n: (rng.standard_normal(n).astype(np.float32), (rng.standard_normal(n) * 10.0 ** rng.integers(-8, 2, n)).astype(np.float32),
                    rng.uniform(0, 1e-2, n).astype(np.float32))
    (w1, g1, m1), (w2, g2, m2) = mk(n1), mk(n2)
    g1[:100] = 0.0; m1[:50] = 0.0
    d = lambda a: torch.from_numpy(a.copy()).to(dev)
    lr, gs = 1e-4, 0.125
    W1, G1, M1, W2, G2, M2 = d(w1), d(g1), d(m1), d(w2), d(g2), d(m2)
    K.rmsprop2(W1, G1, M1, W2, G2, M2, lr, gscale=gs)
    torch.cuda.synchronize()
    for w, g, m, Wd, Md, tag in ((w1, g1, m1, W1, M1, "generator"), (w2, g2, m2, W2, M2, "discriminator")):
        wr, mr = T.rmsprop_update(torch.from_numpy(w), torch.from_numpy(g) * gs, torch.from_numpy(m), lr)
        assert_close(Md, mr, 1e-6, "rmsprop2 %s slots" % tag)
        assert float((Wd.cpu() - wr).abs().max()) <= 1e-6 * float(wr.abs().max()), "rmsprop2 %s weights" % tag
        Ws, Ms = d(w), d(m)
        K.rmsprop(Ws, d(g), Ms, lr, gscale=gs)
        assert torch.equal(Ws, Wd) and torch.equal(Ms, Md), "rmsprop2 != hdrsky_rmsprop (%s)" % tag


@pytest.mark.parametrize("M,Kd,N", [(32, 512, 1024), (32, 8192, 256)])
def test_rmsprop_fc_fused_bias_updates_the_bias_like_the_oracle(dev, M, Kd, N):
    """hdrsky_rmsprop_fc_fused_bias: the Dense kernel's fused update (covered by test_train_gpu) + the bias vector's own
    RMSprop step in the same call (train.py:402-403; sunpose_net.py:48-51): db = gscale-free column sum of dy as before, bias
    and its slot against oracle/tfsem.rmsprop_update on gscale * db per element (1e-6); kernel, slots and bf16 images bit-
    identical to the call without the bias."""
    K = pkg("kernels")
    rng = np.random.default_rng(21)
    x = torch.from_numpy(rng.standard_normal((M, Kd)).astype(np.float32)).to(dev)
    dy = torch.from_numpy((rng.standard_normal((M, N)) * 0.05).astype(np.float32)).to(dev)
    w0 = torch.from_numpy(rng.standard_normal((Kd, N)).astype(np.float32)).to(dev)
    ms0 = torch.from_numpy(rng.uniform(0, 1e-2, (Kd, N)).astype(np.float32)).to(dev)
    b0 = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    bms0 = torch.from_numpy(rng.uniform(0, 1e-2, N).astype(np.float32)).to(dev)
    lr, gs = 1e-3, 0.25
    wa, msa, pfa, dba = w0.clone(), ms0.clone(), K.PackedFC(w0, precise=False), torch.empty(N, device=dev)
    K.rmsprop_fc_fused(wa, msa, x, dy, pfa, lr, db=dba, gscale=gs)
    wb, msb, pfb, dbb, bias, bms = w0.clone(), ms0.clone(), K.PackedFC(w0, precise=False), torch.empty(N, device=dev), b0.clone(), bms0.clone()
    K.rmsprop_fc_fused(wb, msb, x, dy, pfb, lr, db=dbb, gscale=gs, bias=bias, bias_ms=bms)
    torch.cuda.synchronize()
    assert torch.equal(wa, wb) and torch.equal(msa, msb) and torch.equal(dba, dbb)
    assert torch.equal(pfa.pk_hi.view(torch.int16), pfb.pk_hi.view(torch.int16))
    assert torch.equal(pfa.nat_hi.view(torch.int16), pfb.nat_hi.view(torch.int16))
    assert_close(dbb, dy.double().sum(0), 2e-6, "bias gradient")
    br, mr = T.rmsprop_update(b0.cpu(), dbb.cpu() * gs, bms0.cpu(), lr)
    assert_close(bms, mr, 1e-6, "bias slots")
    assert float((bias.cpu() - br).abs().max()) <= 1e-6 * float(br.abs().max()), "bias"
