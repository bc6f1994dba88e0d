"""GPU parity of the non-conv kernels (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import networks as N
from oracle import tfsem as T
from util import TOL_F32, TOL_X3, assert_close, assert_close_bf16

pytestmark = pytest.mark.gpu


def _stats_for(K, x_dev, C):
    """(sum,sumsq) partials of x itself, produced by a 1x1 identity conv through the real conv kernel."""
    eye = torch.eye(C, device=x_dev.device).reshape(1, 1, C, C).contiguous()
    pw = K.PackedConv(eye)
    y, st = K.conv2d(x_dev, pw, None, want_stats=True, compute=K.BF16X3)
    return y, st


@pytest.mark.parametrize("shape", [(2, 32, 128, 32), (3, 16, 64, 64), (2, 8, 32, 128)])
@pytest.mark.parametrize("pool", [False, True])
def test_norm_apply(dev, shape, pool):
    K = pkg("kernels")
    rng = np.random.default_rng(3)
    B, H, W, C = shape
    x = (rng.standard_normal(shape) * 1.7 + 0.3).astype(np.float32)
    gam = rng.uniform(0.5, 1.5, C).astype(np.float32); bet = rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal(shape).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(dev)
    xr, st = _stats_for(K, d(x), C)   # xr == x up to the split-bf16 rounding of the identity conv
    ref = T.leaky_relu(T.instance_norm(xr.cpu(), torch.from_numpy(gam), torch.from_numpy(bet)), 0.1) + torch.from_numpy(res)
    out = K.norm_apply(xr, st, d(gam), d(bet), slope=0.1, residual=d(res), pool=pool)
    if pool:
        y, yp = out
        assert_close(yp, T.maxpool2x2(ref), 1e-4, "pooled")
    else:
        y = out
    assert_close(y, ref, 1e-4, "norm_apply")


@pytest.mark.parametrize("shape", [(2, 16, 64, 64), (2, 8, 32, 128), (3, 32, 128, 32)])
@pytest.mark.parametrize("pooled", [False, True])
def test_norm_act_bwd(dev, shape, pooled):
    K = pkg("kernels")
    rng = np.random.default_rng(5)
    B, H, W, C = shape
    x = (rng.standard_normal(shape) * 1.3 + 0.2).astype(np.float32)
    gam = rng.uniform(0.5, 1.5, C).astype(np.float32); bet = (rng.standard_normal(C) * 0.5).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(dev)
    xr, st = _stats_for(K, d(x), C)
    xt = xr.cpu().clone().requires_grad_(True)
    y = torch.relu(T.instance_norm(xt, torch.from_numpy(gam), torch.from_numpy(bet)))
    if pooled:
        y = T.maxpool2x2(y)
    dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    (gx,) = torch.autograd.grad(y, xt, torch.from_numpy(dy))
    got = K.norm_act_bwd(xr, st, d(gam), d(bet), 0.0, d(dy), pooled)
    assert_close(got, gx, 2e-4, "norm_act_bwd")
    # the incoming gradient handed over as a bf16 tensor (a data-gradient conv's bf16 output): the same bits as the fp32
    # tensor holding those values, on one-launch and two-launch (reduce + apply) sizes, fp32 and bf16 output
    dyb = d(dy).to(torch.bfloat16)
    for out_bf16 in (False, True):
        a = K.norm_act_bwd(xr, st, d(gam), d(bet), 0.0, dyb, pooled, out_bf16=out_bf16)
        b = K.norm_act_bwd(xr, st, d(gam), d(bet), 0.0, dyb.float(), pooled, out_bf16=out_bf16)
        assert a.dtype == b.dtype and torch.equal(a, b)
        if out_bf16:      # (the calls that store dx as bf16 run the one-launch register-resident form on these shapes: against the oracle
                          # at the precision of a bf16 gradient in, a bf16 gradient out)
            assert_close(a.float(), gx, 1.2e-2, "norm_act_bwd, bf16 gradient in / out")


def test_norm_act_bwd_pool_routing_tight_off_the_tie_channels(dev):
    """Grad-CAM conditioning, encoded: the backward of max-pool(relu(InstanceNorm(x))) routes each
    pooled gradient to the window's arg-max.  Where two activations of a window tie to within rounding, the winner is
    decided by last-bit noise - in torch autograd as much as here - and that (sample, channel) slice may legitimately
    differ; everywhere else the kernel must match autograd to fp32 accuracy.  The input has EXACT ties planted in a few
    channels (bitwise-equal values in a window, as flat image regions produce them) plus whatever near-ties random data
    holds; the tie mask is computed on the oracle side from the activations (top two of every window within 1e-5 of
    the slice's scale).  A 3 % routing bug would fail the tight bound on the ~95 % of slices that are tie-free."""
    K = pkg("kernels")
    rng = np.random.default_rng(17)
    B, H, W, C = 2, 8, 32, 128
    x = (rng.standard_normal((B, H, W, C)) * 1.3 + 0.2).astype(np.float32)
    planted = [(0, 3), (0, 77), (1, 5), (1, 126)]
    for b, c in planted:                       # a flat 2x2 window with a large value -> a four-way tie above zero
        x[b, 2:4, 6:8, c] = 2.5
    gam = rng.uniform(0.5, 1.5, C).astype(np.float32); bet = (rng.standard_normal(C) * 0.5).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(dev)
    xr, st = _stats_for(K, d(x), C)
    xt = xr.cpu().clone().requires_grad_(True)
    act = torch.relu(T.instance_norm(xt, torch.from_numpy(gam), torch.from_numpy(bet)))
    y = T.maxpool2x2(act)
    dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    (gx,) = torch.autograd.grad(y, xt, torch.from_numpy(dy))
    got = K.norm_act_bwd(xr, st, d(gam), d(bet), 0.0, d(dy), True).cpu()
    # tie mask per (sample, channel): top two activations of some window closer than 1e-5 of the slice's maximum, top > 0
    a = act.detach().reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, C, 4)
    top2 = a.topk(2, dim=-1).values
    scale = act.detach().amax(dim=(1, 2)).clamp_min(1e-12)                     # [B, C]
    tie = (((top2[..., 0] - top2[..., 1]) <= 1e-5 * scale[:, None, None, :]) & (top2[..., 0] > 0)).any(dim=1).any(dim=1)
    assert all(bool(tie[b, c]) for b, c in planted)
    assert 4 <= int(tie.sum()) <= B * C // 8, int(tie.sum())                   # the planted ones, a few natural ones at most
    err = (got - gx).abs().amax(dim=(1, 2)) / gx.abs().amax(dim=(1, 2)).clamp_min(1e-30)   # [B, C] relative max error
    clean, tied = err[~tie], err[tie]
    print("tie-free slices: %d, worst %.2e | tie slices: %d, worst %.2e" % (clean.numel(), float(clean.max()), tied.numel(), float(tied.max())))
    assert float(clean.max()) <= 1e-4, float(clean.max())
    assert float(tied.max()) <= 2.0            # a re-routed gradient: wrong pixel, same magnitude - bounded, not tight


@pytest.mark.parametrize("M,Kd,Nd", [(1, 1024, 512), (4, 1024, 512), (32, 1024, 512), (3, 5760, 2880), (2, 2880, 2880), (5, 72, 8)])
def test_fc_fwd_dgrad_softmax(dev, M, Kd, Nd):
    """Dense forward / data gradient / soft-max head; the ragged cases are the Dense shapes of a 40x72 image (5*9*128 ->
    2880 -> 2880: reduction lengths that are not multiples of the kernel's 256-element chunk) and a sub-chunk layer."""
    K = pkg("kernels")
    rng = np.random.default_rng(9)
    x = rng.standard_normal((M, Kd)).astype(np.float32)
    w = (rng.standard_normal((Kd, Nd)) / 32).astype(np.float32)
    b = rng.standard_normal(Nd).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(dev)
    pf = K.PackedFC(d(w))
    ref = torch.from_numpy(x) @ torch.from_numpy(w) + torch.from_numpy(b)
    y = K.fc_finalize(K.fc_fwd(d(x), pf, K.BF16X3), d(b))
    assert_close(y, ref, TOL_X3, "fc x3")
    y16 = K.fc_finalize(K.fc_fwd(d(x), pf, K.BF16), d(b))
    assert_close_bf16(y16, ref, "fc bf16")
    dy = rng.standard_normal((M, Nd)).astype(np.float32)
    gref = torch.from_numpy(dy) @ torch.from_numpy(w).T
    g = K.fc_finalize(K.fc_dgrad(d(dy), pf, K.BF16X3))
    assert_close(g, gref, TOL_X3, "fc dgrad x3")
    # soft-max head + picked-probability backward
    gmax = torch.zeros(1, dtype=torch.int32, device=dev)
    z, cmf = K.softmax_head(K.fc_fwd(d(x), pf, K.BF16X3), d(b), gmax)
    zr = torch.relu(ref).requires_grad_(True)
    cr = torch.softmax(zr, dim=-1)
    assert_close(cmf, cr, 5e-4, "softmax")
    assert abs(float(gmax.view(torch.float32).item()) - float(cr.max())) <= 5e-4 * float(cr.max())
    assert K.global_max(cmf).view(torch.float32).item() == float(cmf.max())      # hdrsky_global_max: the same word, exactly
    yc = cr.max(dim=1).values.sum()
    (gz,) = torch.autograd.grad(yc, zr)
    dz, idx = K.softmax_pick_bwd(cmf, z, cmf)
    assert (idx.cpu().numpy() == cr.argmax(dim=1).numpy()).all()
    assert_close(dz, gz * (zr > 0), 1e-3, "softmax pick bwd")


def test_cam_plz_heads_rad_blend(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(13)
    B, H, W = 3, 32, 128
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    A = rng.standard_normal((B, 16, 64, 64)).astype(np.float32)
    g = rng.standard_normal((B, 8, 32, 64)).astype(np.float32)
    w = K.spatial_sum(d(g), 1.0 / (16 * 64))
    assert_close(w, g.sum(axis=(1, 2)) / (16 * 64), TOL_F32 * 10, "spatial_sum")
    cam = K.grad_cam_map(d(A), w)
    ref = np.maximum(np.einsum("bc,bhwc->bhw", w.cpu().numpy(), A), 0)[..., None]
    assert_close(cam, ref, 1e-5, "cam")
    # same weights taken from the statistics partials of the conv that produced the gradient
    _, st = _stats_for(K, d(g), 64)
    cam2 = K.grad_cam_map(d(A), st, 1.0 / (16 * 64))
    assert_close(cam2, ref, 1e-4, "cam from partials")
    ldr = rng.uniform(0, 1, (B, H, W, 3)); c1 = rng.uniform(0, 1, (B, H, W, 1))
    c2 = rng.uniform(0, 1, (B, H // 2, W // 2, 1)); c3 = rng.uniform(0, 1, (B, H // 4, W // 4, 1))
    plz = K.plz_build(d(ldr), d(c1), d(c2), d(c3))
    t = lambda a: torch.from_numpy(a.astype(np.float32))
    ref = torch.cat([t(ldr), t(c1), T.resize_bilinear(t(c2), H, W), T.resize_bilinear(t(c3), H, W)], dim=-1)
    assert_close(plz, ref, 1e-6, "plz")
    # dense heads + delta function
    x = rng.standard_normal((B, 4, 16, 512)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 512).astype(np.float32); sh = rng.standard_normal(512).astype(np.float32)
    kg = (rng.standard_normal((32768, 1)) / 180).astype(np.float32); kb = (rng.standard_normal((32768, 1)) / 180).astype(np.float32)
    bg = np.array([0.1], np.float32); bb = np.array([-0.2], np.float32)
    part = K.dense_heads(d(x), d(sc), d(sh), 0.3, d(kg), d(kb))
    flat = T.leaky_relu(t(x) * t(sc) + t(sh), 0.3).reshape(B, -1)
    gr = torch.sigmoid(flat @ t(kg) + t(bg)); br = torch.sigmoid(flat @ t(kb) + t(bb))
    cmf = torch.softmax(t(rng.standard_normal((B, H * W)) * 3), dim=-1)
    gmax = torch.tensor([float(cmf.max())], dtype=torch.float32).view(torch.int32).to(dev)
    lin, gm, gam, bet = K.sun_rad(cmf.to(dev), gmax, part, d(bg), d(bb), H, W)
    assert_close(gam.reshape(B, 1), gr, 1e-4, "gamma head"); assert_close(bet.reshape(B, 1), br, 1e-4, "beta head")
    p = {"x": None}
    xn = (cmf / cmf.max()).reshape(B, H, W, 1)
    y = torch.exp(-torch.pow(1.0 - xn, 2.0) / (bet.cpu() + 1e-5)) * gam.cpu() / (bet.cpu() * 1.7724539 + 1e-5)
    y = torch.where(y > 30000.0, torch.full_like(y, 30000.0), y).repeat(1, 1, 1, 3)
    assert_close(lin, y, 1e-5, "sun_rad lin"); assert_close(gm, T.hdr_log_compression(y), 1e-5, "sun_rad gamma")
    # blend
    sky = rng.uniform(0, 1.4, (B, H, W, 3)).astype(np.float32); sun = rng.uniform(0, 3.0, (B, H, W, 3)).astype(np.float32)
    yg, yl, al, sl, ul = K.blend(d(sky), d(sun))
    a = T.hdr_log_decompression(t(sky)).max(dim=3).values
    a = torch.minimum(torch.ones_like(a), torch.clamp(a - 1.0 + 0.12, min=0.0) / 0.12).unsqueeze(-1).repeat(1, 1, 1, 3)
    s, u = (1 - a) * t(sky), a * t(sun)
    assert_close(al, a, 2e-5, "alpha"); assert_close(yg, s + u, 1e-6, "y_gamma")
    assert_close(yl, T.hdr_log_decompression(s + u), 1e-5, "y_lin")
    assert_close(sl, T.hdr_log_decompression(s), 1e-5, "sky_lin"); assert_close(ul, T.hdr_log_decompression(u), 1e-5, "sun_lin")
    # tone map round trip
    v = d(rng.uniform(0, 50, (1000,)))
    assert_close(K.tonemap(K.tonemap(v, False), True), v, 1e-5, "tonemap round trip")
    assert_close(K.tonemap(v, False), T.hdr_log_compression(v.cpu()), 1e-6, "log compression")


def test_device_input_synthesis(dev):
    """hdrsky_ldr_synth / hdrsky_vmf_target vs the numpy restatement of train.py:42-94 (JPEG excluded), and the
    device batch generator built on them."""
    from oracle import preproc
    K, synth = pkg("kernels"), pkg("synth")
    rng = np.random.default_rng(31)
    B, H, W, Kc = 4, 32, 128, 1024
    hdr = (rng.random((B, H, W, 3)) ** 6 * 40).astype(np.float32)
    t = (2.0 ** rng.uniform(-3, 3, B)).astype(np.float32)
    ss = (0.08 / 6 * rng.random((B, 3))).astype(np.float32); sc = (0.005 * rng.random((B, 3))).astype(np.float32)
    ns, nc = rng.standard_normal((2, B, H, W, 3)).astype(np.float32)
    xs = np.linspace(0, 1, Kc)
    crf = np.stack([xs ** (1 / g) for g in (2.2, 1.8, 2.6, 1.0)]).astype(np.float32)
    ref_t, ref_l = preproc.preprocessing(hdr, t, ss, sc, ns, nc, crf)
    d = lambda a: torch.from_numpy(a).to(dev)
    got_t, got_l = K.ldr_synth(d(hdr), d(t), d(ss), d(sc), d(ns), d(nc), d(crf))
    assert_close(got_t, ref_t, 2e-6, "hdr_t")
    lat = np.round(got_l.cpu().numpy() * 255.0)
    assert np.abs(lat / 255.0 - got_l.cpu().numpy()).max() < 1e-6                     # on the k/255 lattice
    diff = np.abs(lat - np.round(ref_l * 255.0))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())   # ties of the 8-bit rounding only
    elev = np.array([0.0, 5.0, 17.0, 31.0], np.float32)
    pm = K.vmf_target(d(elev), W * 0.5 - 1.0, H, W)
    for b in range(B):
        assert_close(pm[b], preproc.vmf(W * 0.5 - 1.0, float(elev[b]), H, W), 2e-4, "vMF target %d" % b)
    assert torch.allclose(pm.sum(dim=1), torch.ones(B, device=dev), atol=1e-5)
    bt = synth.make_batch_device(8, H, W, seed=5, device=dev)
    assert tuple(bt["hdr_t"].shape) == (8, H, W, 3) and tuple(bt["sunpose_gt"].shape) == (8, H * W)
    assert float(bt["ldr"].min()) >= 0.0 and float(bt["ldr"].max()) <= 1.0 and torch.isfinite(bt["hdr_t"]).all()
    bt2 = synth.make_batch_device(8, H, W, seed=5, device=dev)
    assert torch.equal(bt["hdr_t"], bt2["hdr_t"]) and torch.equal(bt["sunpose_gt"], bt2["sunpose_gt"])   # seeded


def test_pad_channels_zero_word_and_cam_from_the_gradient_map(dev):
    """hdrsky_pad_channels; hdrsky_fc_finalize clearing a word for a later hdrsky_softmax_head; hdrsky_grad_cam summing
    the activation-gradient map itself (w_nparts < 0) = hdrsky_spatial_sum + the table form."""
    K = pkg("kernels")
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.standard_normal((2, 5, 7, 3)).astype(np.float32)).to(dev)
    y = K.pad_channels(x, 32)
    assert y.shape == (2, 5, 7, 32) and torch.equal(y[..., :3], x) and float(y[..., 3:].abs().max()) == 0.0
    part = torch.from_numpy(rng.standard_normal((4, 3, 64)).astype(np.float32)).to(dev)
    word = torch.full((1,), 123456, dtype=torch.int32, device=dev)
    out = K.fc_finalize(part, None, relu=True, zero_word=word)
    assert int(word) == 0 and torch.allclose(out, torch.relu(part.sum(0)), atol=1e-6)
    A = torch.from_numpy(rng.standard_normal((3, 8, 32, 128)).astype(np.float32)).to(dev)
    dP = torch.from_numpy(rng.standard_normal((3, 4, 16, 128)).astype(np.float32)).to(dev)
    scale = 1.0 / 256.0
    ref = K.grad_cam_map(A, K.spatial_sum(dP, scale))
    got = K.grad_cam_map(A, dP, scale)
    assert_close(got, ref, 1e-5, "cam from the gradient map")
    expect = torch.relu(torch.einsum("bc,bhwc->bhw", dP.sum(dim=(1, 2)) * scale, A)).unsqueeze(-1)
    assert_close(got, expect, 1e-4, "cam formula")


@pytest.mark.parametrize("ties", [1, 3])
def test_sun_rad_bwd_shares_the_maximum_gradient_among_ties(dev, ties):
    """generator.py:160 divides the cmf by tf.reduce_max over the whole batch tensor; TF's gradient of reduce_max
    (_MinOrMaxGrad: indicators / num_selected) gives every element equal to the maximum an equal share - torch's full-reduce
    max does the same, so autograd through the formula of sunrad_net.py:55-69 + the log compression is the reference."""
    import math
    K = pkg("kernels")
    rng = np.random.default_rng(31)
    B, H, W = 3, 8, 16
    P = H * W
    cmf = rng.uniform(0.0, 0.01, (B, P)).astype(np.float32)
    top = np.float32(0.0371)
    for b, p in [(1, 9), (0, 5), (2, 100)][:ties]:
        cmf[b, p] = top
    pre = rng.standard_normal((B, 2)).astype(np.float32)
    drg3 = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    # reference: torch autograd
    c = torch.from_numpy(cmf).double().requires_grad_(True)
    pr = torch.from_numpy(pre).double().requires_grad_(True)
    gam, bet = torch.sigmoid(pr[:, 0]).view(B, 1), torch.sigmoid(pr[:, 1]).view(B, 1)
    x = c / c.max()
    const = float(torch.sqrt(torch.tensor(math.pi, dtype=torch.float32)))
    y = torch.exp(-(1.0 - x) ** 2 / (bet + 1e-5)) * gam / (bet * const + 1e-5)
    y = torch.where(y > 30000.0, torch.full_like(y, 30000.0), y)
    rg = torch.log(1.0 + 10.0 * y) / math.log(11.0)
    (rg.view(B, H, W, 1) * torch.from_numpy(drg3).double()).sum().backward()
    # kernel
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    gmax = d(np.array([top]).view(np.int32))
    g32 = torch.sigmoid(torch.from_numpy(pre[:, 0])).view(B, 1, 1, 1).to(dev)
    b32 = torch.sigmoid(torch.from_numpy(pre[:, 1])).view(B, 1, 1, 1).to(dev)
    runs = []
    for _ in range(2):
        dcmf = torch.zeros((B, P), dtype=torch.float32, device=dev)
        dpre = K.sun_rad_bwd(d(cmf), gmax, g32, b32, d(drg3), dcmf)
        runs.append((dcmf, dpre))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    ref = c.grad
    scale = float(ref.abs().max())
    assert float((runs[0][0].double().cpu() - ref).abs().max()) <= 1e-4 * scale
    sel = torch.from_numpy(cmf == top)
    assert int(sel.sum()) == ties
    assert float((runs[0][0].cpu()[sel].double() - ref[sel]).abs().max()) <= 1e-4 * scale      # the shares themselves
    assert_close(runs[0][1], pr.grad.float(), 1e-4, "dpre")


def test_three_grad_cam_maps_in_one_launch_equal_three_launches(dev):
    """grad_cam.py:52-60: the maps of the three layers; the sweep issues them as one launch (hdrsky_grad_cam3)."""
    K = pkg("kernels")
    rng = np.random.default_rng(21)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)
    B = 3
    A1, A2, A3 = t(B, 32, 128, 32), t(B, 16, 64, 64), t(B, 8, 32, 128)
    st1 = K.Stats(t(B, 64, 2, 32), 64, 32 * 128)
    for jobs in ([(A1, st1, 1.0 / 4096), (A2, t(B, 64), 1.0), (A3, t(B, 4, 16, 128), 1.0 / 512)],
                 [(A1, t(B, 32), 1.0), (A2, K.Stats(t(B, 16, 2, 64), 16, 1024), 0.5), (A3, t(B, 128), 2.0)]):
        got = K.grad_cam_maps(jobs)
        for g, (A, w, sc) in zip(got, jobs):
            assert g.shape == A.shape[:3] + (1,)
            assert torch.equal(g, K.grad_cam_map(A, w, sc))
    with pytest.raises(ValueError):
        K.grad_cam_maps([(A1, t(B, 32), 1.0)] * 2)


@pytest.mark.parametrize("N,self_pick", [(4096, True), (4096, False), (16384, True), (1024, False)])
def test_softmax_head_pick_equals_the_two_launches(dev, N, self_pick):
    """hdrsky_softmax_head_pick = hdrsky_softmax_head + hdrsky_softmax_pick_bwd, bit for bit (z, cmf, the global maximum and
    the Grad-CAM seed), including rows with tied maxima in the picking tensor (first index wins)."""
    K = pkg("kernels")
    rng = np.random.default_rng(N + self_pick)
    M, ns = 7, 4
    part = torch.from_numpy(rng.standard_normal((ns, M, N)).astype(np.float32)).to(dev)
    bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    pick = rng.random((M, N)).astype(np.float32)
    pick[2, 100] = pick[2, 900] = 2.0                     # a tie: element 100 wins
    pick = torch.from_numpy(pick).to(dev)
    g0 = torch.zeros(1, dtype=torch.int32, device=dev); g1 = torch.zeros(1, dtype=torch.int32, device=dev)
    z0, c0 = K.softmax_head(part, bias, g0)
    dz0, idx0 = K.softmax_pick_bwd(c0, z0, c0 if self_pick else pick)
    z1, c1, dz1 = K.softmax_head_pick(part, bias, g1, None if self_pick else pick)
    assert torch.equal(z0, z1) and torch.equal(c0, c1) and torch.equal(g0, g1) and torch.equal(dz0, dz1)
    if not self_pick:
        assert int(idx0[2]) == 100


def test_zero_fill_kernel_sizes_and_alignments_also_under_graph_replay(dev):
    """hdrsky_zero (a kernel, not a memset node): every size / alignment class, guard elements on both sides untouched,
    and the fill repeats in every replay of a captured graph (the property the memset nodes did not have)."""
    K = pkg("kernels")
    for n, off in ((1, 0), (3, 1), (4, 0), (5, 3), (64, 2), (1000, 1), (4099, 5), (1 << 20, 0)):
        buf = torch.full((n + 16,), 7.0, device=dev)
        K.zero_(buf[off:off + n])
        assert float(buf[off:off + n].abs().max()) == 0.0 and float(buf[:off].sum()) == 7.0 * off
        assert float(buf[off + n:].sum()) == 7.0 * (16 - off)
    b8 = torch.full((1031,), 9, dtype=torch.uint8, device=dev)
    K.zero_(b8[1:1030])
    assert int(b8[1:1030].max()) == 0 and int(b8[0]) == 9 and int(b8[1030]) == 9
    x = torch.ones(4096, device=dev)
    g = torch.cuda.CUDAGraph()
    with pkg("kernels").no_gc(), torch.cuda.graph(g):
        K.zero_(x)
        x.add_(1.0)
    for it in range(4):
        g.replay()
        torch.cuda.synchronize()
        assert float(x.min()) == 1.0 and float(x.max()) == 1.0, it


@pytest.mark.parametrize("shape,pooled", [((4, 32, 128, 32), True), ((3, 16, 64, 64), False), ((3, 16, 64, 64), True), ((5, 8, 32, 128), False),
                                          ((2, 16, 64, 128), False)])
def test_norm_act_bwd_one_launch_equals_the_sliced_form(dev, shape, pooled, monkeypatch):
    """Round 5: the InstanceNorm (+ activation, + max-pool) backward of the 32x128 network's maps runs as ONE launch that reads
    x and dy once (norm_act_bwd1_kernel: the (sample, channel group) slab in the registers of one workgroup) instead of the
    sliced reduce + apply pair.  Same arithmetic per element; the two per-(sample, channel) sums are added in another fixed
    order: dx and the (d beta, d gamma) terms agree to fp32 rounding of those sums (1e-5 of the tensor's scale), for fp32 / bf16
    incoming gradients, fp32 / bf16 outputs and a bf16-stored x; the launch is bit-reproducible; HDRSKY_NAB_ONE=0 is the switch."""
    K, HK, L = pkg("kernels"), pkg("hooks"), pkg("_lib")
    B, H, W, C = shape
    g = torch.Generator(device=dev); g.manual_seed(B + H + C)
    x = (torch.randn(*shape, device=dev, generator=g) * 1.3 + 0.2).contiguous()
    gam, bet = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.5
    xr, st = _stats_for(K, x, C)
    dshape = (B, H // 2, W // 2, C) if pooled else shape
    dy = torch.randn(*dshape, device=dev, generator=g)
    assert L.load().hdrsky_norm_act_bwd_one_launch(H, W, int(pooled), 0) == 1
    # (the one-launch form takes the calls that store dx as bf16 - the single-product mode's; fp32 outputs stay on the sliced form)
    cases = [(xr, dy, True), (xr, dy.to(torch.bfloat16), True), (xr.to(torch.bfloat16), dy.to(torch.bfloat16), True)]
    for xin, dyin, ob in cases:
        run = lambda: K.norm_act_bwd(xin, st, gam, bet, 0.0 if pooled else 0.1, dyin, pooled, want_sums=True, out_bf16=ob)
        one, s_one = run()
        again, s_again = run()
        assert torch.equal(one, again) and torch.equal(s_one, s_again), "one-launch form is not bit-reproducible"
        monkeypatch.setenv("HDRSKY_NAB_ONE", "0"); HK.reload()
        assert L.load().hdrsky_norm_act_bwd_one_launch(H, W, int(pooled), 0) == 0
        two, s_two = run()
        monkeypatch.delenv("HDRSKY_NAB_ONE"); HK.reload()
        assert one.dtype == two.dtype
        scale = float(two.float().abs().max())
        tol = (2 ** -8 if ob else 1e-5) * scale          # bf16 output: one bf16 ulp where the fp32 values straddle a rounding boundary
        assert float((one.float() - two.float()).abs().max()) <= tol, (str(xin.dtype), str(dyin.dtype), ob)
        if ob:
            assert float((one != two).float().mean()) < 1e-3
        assert_close(s_one, s_two, 1e-5, "per-sample (d beta, d gamma) terms")
