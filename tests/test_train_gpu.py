"""GPU parity of the training step (train.py:382-415) vs the oracle's autograd restatement, plus its building blocks."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import step as ostep
from oracle import tfsem as T
from util import assert_close, rel_max, rel_rms

pytestmark = pytest.mark.gpu


def test_resize_adjoint_blur_adjoint_dog(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((2, 6, 10, 3)).astype(np.float32)).requires_grad_(True)
    up = T.resize_bilinear(x, 12, 20)
    dy = torch.from_numpy(rng.standard_normal((2, 12, 20, 3)).astype(np.float32))
    (gx,) = torch.autograd.grad(up, x, dy)
    assert_close(K.up2x(x.detach().to(dev)), up.detach(), 1e-6, "up2x")
    assert_close(K.up2x_bwd(dy.to(dev)), gx, 1e-5, "up2x adjoint")
    # the gradient handed over as bf16 (a data-gradient conv's bf16 output): the same bits as the fp32 tensor of those values,
    # for the 16-byte (C % 4 == 0) and the scalar channel paths, plain and accumulating
    for shp in ((2, 12, 20, 3), (2, 12, 20, 8)):
        dyb = torch.from_numpy(rng.standard_normal(shp).astype(np.float32)).to(dev).to(torch.bfloat16)
        assert torch.equal(K.up2x_bwd(dyb), K.up2x_bwd(dyb.float()))
        acc = torch.ones((shp[0], shp[1] // 2, shp[2] // 2, shp[3]), device=dev)
        assert torch.equal(K.up2x_bwd(dyb, 0.5, out=acc.clone()), K.up2x_bwd(dyb.float(), 0.5, out=acc.clone()))
    y = torch.from_numpy(rng.standard_normal((2, 8, 12, 3)).astype(np.float32)).requires_grad_(True)
    bl = T.gaussian_filter2d_3x3(y, 1.5450078)
    dz = torch.from_numpy(rng.standard_normal((2, 8, 12, 3)).astype(np.float32))
    (gy,) = torch.autograd.grad(bl, y, dz)
    assert_close(K.blur3(y.detach().to(dev), 1.5450078), bl.detach(), 1e-5, "blur")
    assert_close(K.blur3(dz.to(dev), 1.5450078, transpose=True), gy, 1e-5, "blur adjoint")
    # full DoG loss + gradient
    a = torch.from_numpy(rng.uniform(0, 3, (2, 8, 16, 3)).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rng.uniform(0, 3, (2, 8, 16, 3)).astype(np.float32))
    loss = sum((p - q).abs().mean() for p, q in zip(T.dog(a), T.dog(b)))
    (ga,) = torch.autograd.grad(1000.0 * loss, a)
    slot = torch.zeros(1, device=dev); dyo = torch.zeros((2, 8, 16, 3), device=dev)
    K.dog_loss(a.detach().to(dev), b.to(dev), 1000.0, slot, dyo)
    assert abs(float(slot) - float(loss)) <= 1e-4 * float(loss)
    assert rel_rms(dyo, ga) < 2e-2   # sign() flips at |d| ~ 0 make a few entries differ
    assert_close(dyo, ga, 0.5, "dog grad (max)")


@pytest.mark.parametrize("shape", [(2, 8, 16, 3), (3, 32, 128, 3), (2, 5, 7, 3), (1, 4, 4, 1), (2, 9, 33, 2), (2, 8, 200, 3),
                                   (1, 11, 347, 3), (1, 128, 512, 3), (1, 6, 300, 5)])
def test_dog_loss_one_launch_equals_the_staged_path(dev, shape, monkeypatch):
    """hdrsky_dog_loss (the chain through LDS bands, one launch) against the seven staged launches it replaces (which
    test_custom_ops_match_autograd pins to the oracle): the same operators - on the training size, on sizes whose bands are
    clipped at both borders, one channel, rows that are split into column strips (more than 1 024 floats at 2x resolution:
    two strips, ragged strips, the 128x512 maps), and accumulating into a non-zero gradient."""
    K = pkg("kernels")
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.uniform(0, 3, shape).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(0, 3, shape).astype(np.float32)).to(dev)
    g0 = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dev)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("HDRSKY_DOG_FUSED", fused)
        slot, dy = torch.zeros(1, device=dev), g0.clone()
        K.dog_loss(a, b, 1000.0, slot, dy)
        out[fused] = (float(slot), dy)
    assert abs(out["1"][0] - out["0"][0]) <= 1e-5 * abs(out["0"][0])
    ga, gb = out["1"][1] - g0, out["0"][1] - g0
    assert float(gb.abs().max()) > 0
    # the same operators in another fp32 summation order: round-off everywhere, and a sign(d) that flips where |d| ~ 0
    # moves the few entries under that pixel's stencil by a visible amount
    assert rel_rms(ga, gb) < 2e-2
    assert_close(ga, gb, 0.5, "dog gradient, one launch vs staged (max)")


def test_losses_softmax_bn_pool(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    # KL + softmax backward
    z = torch.relu(torch.from_numpy(rng.standard_normal((3, 512)).astype(np.float32))).requires_grad_(True)
    gt = torch.softmax(torch.from_numpy(rng.standard_normal((3, 512)).astype(np.float32) * 3), -1)
    cmf = torch.softmax(z, -1)
    kl = T.kl_divergence(gt, cmf)
    (gz,) = torch.autograd.grad(kl, z)
    slot = torch.zeros(1, device=dev)
    dcmf = K.kl(gt.to(dev), cmf.detach().to(dev), slot)
    assert abs(float(slot) - float(kl)) < 1e-5 * abs(float(kl))
    dz = K.softmax_bwd(cmf.detach().to(dev), dcmf, z.detach().to(dev))
    assert_close(dz, gz * (z > 0), 1e-4, "kl->softmax->relu backward")
    # LSGAN
    x = torch.from_numpy(rng.standard_normal((3, 1, 13, 1)).astype(np.float32)).requires_grad_(True)
    l = ((x - 1.0) ** 2).mean()
    (gx,) = torch.autograd.grad(0.5 * l, x)
    slot.zero_()
    dx = K.mse(x.detach().to(dev), 1.0, 1.0, 0.5, slot)
    assert abs(float(slot) - float(l)) < 1e-5 and rel_max(dx, gx) < 1e-5
    # BatchNorm train: stats from conv partials, backward with dgamma/dbeta
    B, H, W, C = 3, 8, 32, 128
    xr = (rng.standard_normal((B, H, W, C)) * 1.5 + 0.4).astype(np.float32)
    eye = torch.eye(C, device=dev).reshape(1, 1, C, C).contiguous()
    xd, st = K.conv2d(d(xr), K.PackedConv(eye), None, want_stats=True, compute=K.BF16X3)
    gam = rng.uniform(0.5, 1.5, C).astype(np.float32); bet = rng.standard_normal(C).astype(np.float32)
    mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
    mean, rstd, sc, sh = K.bn_train_finalize(st, d(gam), d(bet), B, C, mm, mv)
    xt = xd.cpu().clone().requires_grad_(True)
    gt_, bt_ = torch.from_numpy(gam).requires_grad_(True), torch.from_numpy(bet).requires_grad_(True)
    yb, nmm, nmv = T.batch_norm(xt, gt_, bt_, torch.zeros(C), torch.ones(C), True)
    ya = T.leaky_relu(yb, 0.3)
    dy = rng.standard_normal((B, H, W, C)).astype(np.float32)
    gx, gg, gb = torch.autograd.grad(ya, (xt, gt_, bt_), torch.from_numpy(dy))
    assert_close(mm, nmm, 1e-4, "moving mean"); assert_close(mv, nmv, 1e-4, "moving var")
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx = K.bn_act_bwd(xd, d(dy), mean, rstd, d(gam), d(bet), 0.3, dg, db)
    assert_close(dx, gx, 2e-4, "bn dx"); assert_close(dg, gg, 2e-4, "bn dgamma"); assert_close(db, gb, 2e-4, "bn dbeta")
    # maxpool + relu backward
    yv = torch.relu(torch.from_numpy(rng.standard_normal((2, 8, 16, 64)).astype(np.float32)))
    pre = yv.clone().requires_grad_(True)
    pl = T.maxpool2x2(torch.relu(pre))
    dp = rng.standard_normal(tuple(pl.shape)).astype(np.float32)
    (gp,) = torch.autograd.grad(pl, pre, torch.from_numpy(dp))
    assert_close(K.maxpool(yv.to(dev)), pl.detach(), 1e-6, "maxpool")
    got = K.maxpool_relu_bwd(yv.to(dev), d(dp))
    mism = ((got.cpu() != 0) != (gp != 0)).float().mean()   # ties among zeros route differently but carry no gradient
    assert float(mism) == 0.0 and rel_max(got, gp) < 1e-6
    # Dense wgrad + RMSprop
    xm = rng.standard_normal((32, 256)).astype(np.float32); dym = rng.standard_normal((32, 128)).astype(np.float32)
    dw = torch.zeros((256, 128), device=dev); dbv = torch.zeros(128, device=dev)
    K.fc_wgrad(d(xm), d(dym), dw, dbv)
    assert_close(dw, xm.T @ dym, 1e-5, "fc wgrad"); assert_close(dbv, dym.sum(0), 1e-5, "fc bias grad")
    w0 = rng.standard_normal(1024).astype(np.float32); g0 = rng.standard_normal(1024).astype(np.float32)
    ms0 = rng.uniform(0, 1, 1024).astype(np.float32)
    wd, msd = d(w0), d(ms0)
    K.rmsprop(wd, d(g0), msd, 1e-4)
    wr, mr = T.rmsprop_update(torch.from_numpy(w0), torch.from_numpy(g0), torch.from_numpy(ms0), 1e-4)
    assert_close(wd, wr, 1e-6, "rmsprop w"); assert_close(msd, mr, 1e-6, "rmsprop ms")


def _mk(dev, B, compute_name="BF16X3"):
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
    dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
    batch = synth.make_batch(B, seed=1234)
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=(compute_name == "BF16X3"), compute=getattr(K, compute_name))
    return tr, (gen, sun, dis, vgg), batch


def test_train_step_gradients_match_oracle(dev):
    tr, (gen, sun, dis, vgg), batch = _mk(dev, 2)
    tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(tt(gen), tt(sun), tt(dis), tt(vgg), ldr, hdr, gt)
    out = tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    got = tr.loss_dict()
    print({k: (got[k], losses.get(k)) for k in got})
    ref_names = {"kl": "kl", "perceptual": "perceptual", "dog": "dog", "l1": "l1", "adv": "adv",
                 "disc_generated": "generated", "disc_real": "real", "total_gen_loss": "total_gen_loss",
                 "total_disc_loss": "total_disc_loss"}
    for k, rk in ref_names.items():
        assert abs(got[k] - losses[rk]) <= 2e-3 * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    assert_close(out["y_final_gamma"], outs["y_final_gamma"], 1e-3, "y_final_gamma (training mode)")
    # ---- generator-step gradients (gen + sun variables) vs the oracle's autograd --------------------------------
    # Tolerance relative to each tensor's max |gradient|.  The contraction error is fp32-class (BF16X3); what is left
    # are discrete effects (sign() of the L1/DoG terms, ReLU / max-pool masks, the 1%-level Grad-CAM map differences
    # explained in test_forward_gpu) - hence 5e-2 on the worst tensor and 3e-3 on the median tensor.
    worst = []
    for prefix, ref, flat in (("gen.", gg, tr.gs), ("sun.", gs, tr.gs)):
        for k, v in ref.items():
            g = flat.g[prefix + k]
            is_bias = k.endswith(".b") or k.endswith("bias_deconv2d")
            if is_bias and not k.startswith("conv1_f") and not k.startswith("conv1_u"):
                # a conv bias in front of InstanceNorm has an exactly-zero gradient (IN removes the mean): both sides
                # hold rounding noise only - require it to be negligible against the layer's weight gradient
                wk = k[:-2] + ".w" if k.endswith(".b") else k.replace("bias_deconv2d", "kernel_deconv2d")
                scale = float(ref[wk].abs().max()) * float(np.prod(ref[wk].shape[:3]))
                assert float(g.abs().max()) <= 1e-4 * scale, (prefix + k, float(g.abs().max()), scale)
                continue
            worst.append((rel_max(g, v), prefix + k))
    worst.sort(reverse=True)
    print("worst generator-step gradient errors:", worst[:8])
    assert worst[0][0] < 5e-2, worst[:5]
    assert np.median([e for e, _ in worst]) < 3e-3
    for k, v in sg.items():   # sunRadNet BN moving statistics after one train-mode call
        assert_close(tr.gs.w["gen." + k], v, 1e-4, "gen " + k)

    # ---- discriminator-step gradients --------------------------------------------------------------------------
    # The generated-image pass is chaotic w.r.t. its input: with random weights and B=2 a 1e-4 relative perturbation
    # of y_final_gamma (HDR peaks ~2e4 through lrelu + batch-stat BN) moves these gradients by 2-8 %% in the ORACLE
    # itself (measured).  So the discriminator step is checked on identical inputs: the oracle's discriminator step is
    # evaluated at the y_final_lin this implementation produced.
    dr = {k: torch.from_numpy(v).clone().requires_grad_("moving" not in k) for k, v in dis.items()}
    stats = {}
    dl = ostep.discriminator_losses(dr, ldr, hdr, out["y_final_lin"].cpu(), training=True, new_stats=stats)
    names = [k for k in dr if "moving" not in k]
    refs = torch.autograd.grad(dl["total_disc_loss"], [dr[k] for k in names])
    for k, v in zip(names, refs):
        assert_close(tr.ds.g["dis." + k], v, 3e-4, "dis grad " + k)
    for k, v in stats.items():
        assert_close(tr.ds.w["dis." + k], v, 1e-4, "dis " + k)


@pytest.mark.parametrize("B", [1, 3])
def test_train_step_edge_batch_sizes(dev, B):
    """Batch 1 (BatchNorm over one sample, the reduce_max over a single map) and an odd batch (ragged against every
    power-of-two split inside the kernels): losses and the gradient direction against the oracle."""
    tr, (gen, sun, dis, vgg), batch = _mk(dev, B)
    tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(tt(gen), tt(sun), tt(dis), tt(vgg), ldr, hdr, gt)
    out = tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    got = tr.loss_dict()
    for k, rk in (("kl", "kl"), ("perceptual", "perceptual"), ("dog", "dog"), ("l1", "l1"), ("adv", "adv"),
                  ("disc_generated", "generated"), ("disc_real", "real")):
        assert abs(got[k] - losses[rk]) <= 2e-3 * abs(losses[rk]) + 1e-6, (B, k, got[k], losses[rk])
    assert_close(out["y_final_gamma"], outs["y_final_gamma"], 1e-3, "y_final_gamma")
    dot = na = nb = 0.0
    for prefix, ref in (("gen.", gg), ("sun.", gs)):
        for k, v in ref.items():
            g = tr.gs.g[prefix + k].cpu().double()
            dot += float((g * v.double()).sum()); na += float((g * g).sum()); nb += float((v.double() ** 2).sum())
    cos = dot / (na * nb) ** 0.5
    print("B=%d: gradient cosine %.6f, norm ratio %.5f" % (B, cos, (na / nb) ** 0.5))
    assert cos > 0.999 and abs((na / nb) ** 0.5 - 1.0) < 1e-2, (B, cos, (na / nb) ** 0.5)
    tr.apply_gradients()
    assert torch.isfinite(tr.gs.flat).all() and torch.isfinite(tr.ds.flat).all()


def test_validation_step_matches_oracle(dev):
    """Trainer.test_step = train.py:417-442: BatchNorm in inference mode everywhere, all nine loss terms, no update and
    no change of the moving statistics; a captured training step keeps working afterwards."""
    tr, (gen, sun, dis, vgg), batch = _mk(dev, 2)
    tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    # make the moving statistics differ from their initial values so that inference-mode BN is really exercised
    tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=True)
    torch.cuda.synchronize()
    gen2 = {k: tr.gs.w["gen." + k].cpu() for k in gen}; sun2 = {k: tr.gs.w["sun." + k].cpu() for k in sun}
    dis2 = {k: tr.ds.w["dis." + k].cpu() for k in dis}
    losses, outs = ostep.test_step(gen2, sun2, dis2, tt(vgg), ldr, hdr, gt)
    g0, d0 = tr.gs.flat.clone(), tr.ds.flat.clone()
    out = tr.test_step(ldr.to(dev), hdr.to(dev), gt.to(dev))
    got = tr.loss_dict()
    for k, rk in (("kl", "kl"), ("perceptual", "perceptual"), ("dog", "dog"), ("l1", "l1"), ("adv", "adv"),
                  ("disc_generated", "generated"), ("disc_real", "real"), ("total_gen_loss", "total_gen_loss"),
                  ("total_disc_loss", "total_disc_loss")):
        assert abs(got[k] - losses[rk]) <= 2e-3 * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    assert_close(out["y_final_gamma"], outs["y_final_gamma"], 1e-3, "y_final_gamma (validation)")
    assert_close(out["gamma"], outs["gamma"], 1e-3, "gamma"); assert_close(out["beta"], outs["beta"], 1e-3, "beta")
    assert torch.equal(tr.gs.flat, g0) and torch.equal(tr.ds.flat, d0)          # nothing trained, no moving-average update
    # training-mode BN would give a different sun radiance (the batch statistics of 2 samples): the flag matters
    out_t = tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    assert float((out_t["gamma"] - out["gamma"]).abs().max()) > 1e-6


def test_train_step_updates_weights_like_rmsprop(dev):
    tr, _, batch = _mk(dev, 2)
    ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    w0 = tr.gs.flat[:tr.gs.ntrain].clone()
    tr.step(ldr, hdr, gt, update=True)
    g = tr.gs.grad
    ms = 0.1 * g * g
    ref = w0 - 1e-4 * g / (ms.sqrt() + 1e-7)
    assert_close(tr.gs.flat[:tr.gs.ntrain], ref, 1e-6, "first RMSprop step")
    # second step runs on the repacked weights and stays finite
    tr.step(ldr, hdr, gt, update=True)
    assert torch.isfinite(tr.gs.flat).all() and torch.isfinite(tr.ds.flat).all()
    v = tr.loss_dict()
    assert all(np.isfinite(x) for x in v.values())


def test_train_step_in_bench_mode_bf16(dev):
    """The configuration bench.py times (single bf16 MFMA product, fp32 accumulate) against the fp32 oracle: losses
    within 2 %, prediction PSNR above 40 dB, and the generator / sun-pose gradient as a whole pointing the same way
    (cosine > 0.98; measured 0.990) - the bf16 operand rounding (2^-9 relative per product term) is the only difference to the
    BF16X3 step that the tighter tests above pin."""
    tr, (gen, sun, dis, vgg), batch = _mk(dev, 2, "BF16")
    tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(tt(gen), tt(sun), tt(dis), tt(vgg), ldr, hdr, gt)
    out = tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    got = tr.loss_dict()
    for k, rk in (("kl", "kl"), ("perceptual", "perceptual"), ("dog", "dog"), ("l1", "l1"), ("adv", "adv"),
                  ("total_gen_loss", "total_gen_loss"), ("total_disc_loss", "total_disc_loss")):
        assert abs(got[k] - losses[rk]) <= 2e-2 * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    a, b = out["y_final_gamma"].cpu().double(), outs["y_final_gamma"].double()
    psnr = 10 * torch.log10(b.abs().max() ** 2 / ((a - b) ** 2).mean())
    assert float(psnr) > 40.0, float(psnr)
    dot = na = nb = 0.0
    for prefix, ref in (("gen.", gg), ("sun.", gs)):
        for k, v in ref.items():
            g = tr.gs.g[prefix + k].cpu().double()
            dot += float((g * v.double()).sum()); na += float((g * g).sum()); nb += float((v.double() ** 2).sum())
    cos = dot / (na * nb) ** 0.5
    print("bf16 step: psnr %.1f dB, gradient cosine %.5f" % (float(psnr), cos))
    assert cos > 0.98, cos


def test_train_step_with_raw_conv_outputs_stored_as_bf16(dev, monkeypatch):
    """HDRSKY_RAW_BF16=1 (tuning hook, off by default: profiles/r04_raw_bf16_ab.txt): the bench-mode step with every raw conv
    output in front of an InstanceNorm / BatchNorm layer stored as bf16.  Every reader is bit-exact on the widened tensor
    (tests/test_raw_bf16_gpu.py); what this pins is the end-to-end cost of the extra rounding - losses within 5 % (adversarial: 8 %) of the fp32
    oracle (fp32 storage: 2 %), prediction PSNR above 40 dB, gradient cosine above 0.97 - and that the step really ran on
    bf16 tensors."""
    monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1")
    monkeypatch.setenv("HDRSKY_RAW_BF16", "1")
    tr, (gen, sun, dis, vgg), batch = _mk(dev, 2, "BF16")
    assert tr._raw_bf16()
    tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(tt(gen), tt(sun), tt(dis), tt(vgg), ldr, hdr, gt)
    seen = []
    K = pkg("kernels")
    conv2d = K.conv2d
    def spy(x, *a, **kw):
        y, st = conv2d(x, *a, **kw)
        seen.append((x.dtype, kw.get("xf") is not None and kw["xf"].mode != 0, y.dtype, st is not None))
        return y, st
    monkeypatch.setattr(K, "conv2d", spy)
    out = tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    assert sum(1 for xd, xf, yd, st in seen if yd == torch.bfloat16 and st) >= 10      # producers in front of a norm layer
    assert sum(1 for xd, xf, yd, st in seen if xd == torch.bfloat16 and xf) >= 8        # their consuming convs
    got = tr.loss_dict()
    # The adversarial term (and through it the generator total) gets 8 %, the rest 5 %.  At batch 2 with random weights this
    # network amplifies ANY fp32 rounding difference by 1e4-1e5 over its ten normalisation layers (the one-pass variance
    # E[x^2] - mean^2 of channels whose mean dwarfs their spread): two equally valid summation orders of the conv epilogue's
    # statistics partials - bit-identical conv outputs, partials equal to 1e-7 - put this term at 873.1 and 886.5 with fp32
    # storage (oracle 884.7) and at 913.4 and 943.0 with bf16 storage (profiles/r05_epi_direct_calls.txt: the call-by-call
    # divergence).  The bound therefore protects "bf16 storage costs a few percent", not a digit of this value.
    for k, rk in (("kl", "kl"), ("perceptual", "perceptual"), ("dog", "dog"), ("l1", "l1"), ("adv", "adv"),
                  ("total_gen_loss", "total_gen_loss"), ("total_disc_loss", "total_disc_loss")):
        tol = 8e-2 if k in ("adv", "total_gen_loss") else 5e-2
        assert abs(got[k] - losses[rk]) <= tol * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    a, b = out["y_final_gamma"].cpu().double(), outs["y_final_gamma"].double()
    psnr = 10 * torch.log10(b.abs().max() ** 2 / ((a - b) ** 2).mean())
    assert float(psnr) > 40.0, float(psnr)
    dot = na = nb = 0.0
    for prefix, ref in (("gen.", gg), ("sun.", gs)):
        for k, v in ref.items():
            g = tr.gs.g[prefix + k].cpu().double()
            dot += float((g * v.double()).sum()); na += float((g * g).sum()); nb += float((v.double() ** 2).sum())
    cos = dot / (na * nb) ** 0.5
    print("raw bf16 step: psnr %.1f dB, gradient cosine %.5f" % (float(psnr), cos))
    assert cos > 0.97, cos


def test_fc_wgrad_bf16_and_fused_update(dev):
    """hdrsky_fc_wgrad_bf16: x^T dy with bf16-rounded operands and fp32 accumulation, any row count, row-strided operands;
    hdrsky_rmsprop_fc_fused: the same update hdrsky_rmsprop_fc applies to that gradient, without writing it."""
    K = pkg("kernels")
    rng = np.random.default_rng(5)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float64)
    Kd, N = 160, 512
    for M in (1, 2, 31, 32, 33, 70):
        wide = torch.from_numpy(rng.standard_normal((M, Kd + N + 8)).astype(np.float32)).to(dev)
        x, dy = wide[:, 4:4 + Kd], wide[:, 4 + Kd:4 + Kd + N]          # column blocks of one buffer: strided rows
        dw = torch.full((Kd, N), 7.0, device=dev); db = torch.full((N,), 7.0, device=dev)
        K.fc_wgrad_bf16(x, dy, dw, db)
        ref = bf(x).T @ bf(dy)
        assert_close(dw, ref, 2e-6, "fc_wgrad_bf16 M=%d" % M)
        assert_close(db, dy.double().sum(0), 2e-6, "fc_wgrad_bf16 bias M=%d" % M)
        K.fc_wgrad_bf16(x, dy, dw, db, accumulate=True)
        assert_close(dw, 2 * ref, 2e-6, "fc_wgrad_bf16 accumulate")
        assert_close(db, 2 * dy.double().sum(0), 2e-6, "fc_wgrad_bf16 bias accumulate")
    with pytest.raises(ValueError):
        K.fc_wgrad_bf16(wide[:, 1:1 + Kd], dy, dw, db)                  # misaligned base
    # fused update == materialise + hdrsky_rmsprop_fc
    M = 48
    x = torch.from_numpy(rng.standard_normal((M, Kd)).astype(np.float32)).to(dev)
    dy = torch.from_numpy((rng.standard_normal((M, N)) * 0.1).astype(np.float32)).to(dev)
    w0 = torch.from_numpy(rng.standard_normal((Kd, N)).astype(np.float32)).to(dev)
    ms0 = torch.from_numpy(rng.uniform(0, 1e-2, (Kd, N)).astype(np.float32)).to(dev)
    g = torch.empty_like(w0); gb = torch.empty(N, device=dev)
    K.fc_wgrad_bf16(x, dy, g, gb)
    wa, msa, pfa = w0.clone(), ms0.clone(), K.PackedFC(w0, precise=False)
    K.rmsprop_fc(wa, g, msa, pfa, 1e-3, gscale=0.5)
    wb, msb, pfb = w0.clone(), ms0.clone(), K.PackedFC(w0, precise=False)
    gb2 = torch.empty(N, device=dev)
    K.rmsprop_fc_fused(wb, msb, x, dy, pfb, 1e-3, db=gb2, gscale=0.5)
    assert torch.equal(gb, gb2)
    assert_close(wb, wa, 1e-7, "fused RMSprop: weights"); assert_close(msb, msa, 1e-6, "fused RMSprop: slots")
    fresh = K.PackedFC(wb, precise=False)
    assert torch.equal(fresh.pk_hi.view(torch.int16), pfb.pk_hi.view(torch.int16))
    assert torch.equal(fresh.nat_hi.view(torch.int16), pfb.nat_hi.view(torch.int16))


@pytest.mark.parametrize("M,Kd,N", [(256, 512, 1024), (256, 8192, 256), (255, 160, 512), (64, 256, 256)])
def test_fused_dense_update_at_the_gathered_row_count(dev, M, Kd, N):
    """hdrsky_rmsprop_fc_fused with M = 256 operand rows - what the `gather_dense` exchange hands it at 8 replicas x batch 32
    (parallel.py) - against the Keras-2 RMSprop formula in float64 on the bf16-rounded operands
    (train.py:201-202; sunpose_net.py:48-51): g = gscale * bf16(x)^T bf16(dy) with fp32 accumulation, ms <- 0.9 ms + 0.1 g^2,
    w <- w - lr g / (sqrt(ms) + 1e-7); row-strided operand views (columns of one gathered buffer), a reduction length of fc1's
    size, a ragged row count; the bf16 images written by the update equal a fresh pack of the new weights; the bias gradient
    is the fp32 column sum of dy."""
    K = pkg("kernels")
    rng = np.random.default_rng(11)
    wide = torch.from_numpy(rng.standard_normal((M, Kd + N + 16)).astype(np.float32)).to(dev)
    x, dy = wide[:, 8:8 + Kd], wide[:, 8 + Kd:8 + Kd + N]
    dy.mul_(0.05)
    w0 = torch.from_numpy(rng.standard_normal((Kd, N)).astype(np.float32)).to(dev)
    ms0 = torch.from_numpy(rng.uniform(0, 1e-2, (Kd, N)).astype(np.float32)).to(dev)
    lr, gscale = 1e-3, 1.0 / 8
    w, ms, pf = w0.clone(), ms0.clone(), K.PackedFC(w0, precise=False)
    db = torch.empty(N, device=dev)
    K.rmsprop_fc_fused(w, ms, x, dy, pf, lr, db=db, gscale=gscale)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float64)
    g = gscale * (bf(x).T @ bf(dy))
    ms_ref = 0.9 * ms0.double() + 0.1 * g * g
    w_ref = w0.double() - lr * g / (ms_ref.sqrt() + 1e-7)
    # fp32 accumulation over M <= 256 rows in MFMA order: 1e-6-class on g; the update divides by sqrt(ms) ~ 0.03-0.1
    assert_close(ms, ms_ref, 2e-6, "fused RMSprop M=%d: slots" % M)
    assert float((w.double() - w_ref).abs().max()) <= 2e-6 * float(w_ref.abs().max()) + 1e-7, "fused RMSprop M=%d: weights" % M
    assert_close(db, dy.double().sum(0), 2e-6, "fused RMSprop M=%d: bias gradient" % M)
    fresh = K.PackedFC(w, precise=False)
    assert torch.equal(fresh.pk_hi.view(torch.int16), pf.pk_hi.view(torch.int16))
    assert torch.equal(fresh.nat_hi.view(torch.int16), pf.nat_hi.view(torch.int16))
    # and against the materialised path (hdrsky_fc_wgrad_bf16 + hdrsky_rmsprop_fc) on the same operands
    g2 = torch.empty_like(w0); gb = torch.empty(N, device=dev)
    K.fc_wgrad_bf16(x, dy, g2, gb)
    wa, msa, pfa = w0.clone(), ms0.clone(), K.PackedFC(w0, precise=False)
    K.rmsprop_fc(wa, g2, msa, pfa, lr, gscale=gscale)
    assert_close(w, wa, 1e-6, "fused vs materialised: weights"); assert_close(ms, msa, 2e-6, "fused vs materialised: slots")


def test_fused_dense_step_equals_materialised_step(dev):
    """Trainer(fused_dense=True) (the HDRSKY_BF16 default): two updating steps leave the same weights and RMSprop slots
    as the trainer that writes the Dense gradients out and reads them back; the Dense kernel gradient buffers of the
    fused trainer are untouched by an updating step and filled by step(update=False)."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    nets = [params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
            params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3)]
    batch = synth.make_batch(3, seed=77)
    ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    res = []
    for fused in (True, False):
        tr = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16, fused_dense=fused)
        assert tr.fused_dense == fused
        for _ in range(2):
            tr.step(ldr, hdr, gt, update=True)
        gk = tr.gs.g["sun.fc1.kernel"]
        if fused:
            assert float(gk.abs().max()) == 0.0                         # never written on an updating step
        res.append((tr.gs.flat.clone(), tr.gs.ms.clone(), [p.pk_hi.clone() for p in (tr.fc1, tr.fc2)]))
        if fused:
            tr.step(ldr, hdr, gt, update=False)
            assert float(gk.abs().max()) > 0.0                          # a gradient-only step materialises it
    (wa, ma, pa), (wb, mb, pb) = res
    assert_close(wa, wb, 1e-6, "fused vs materialised Dense update: weights")
    assert_close(ma, mb, 1e-5, "fused vs materialised Dense update: RMSprop slots")
    for u, v in zip(pa, pb):
        assert float((u.view(torch.int16) != v.view(torch.int16)).float().mean()) < 1e-4


def test_bf16_step_fused_dense_optimizer(dev):
    """BF16 trainer: the Dense kernels are updated by hdrsky_rmsprop_fc, which also refreshes their bf16 MFMA images -
    same weights as the closed form and bit-identical images to a fresh hdrsky_fc_pack_weights of the result."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
    dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
    batch = synth.make_batch(2, seed=1234)
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16, fused_dense=False)
    ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    w0 = tr.gs.flat[:tr.gs.ntrain].clone()
    tr.step(ldr, hdr, gt, update=True)
    g = tr.gs.grad
    ref = w0 - 1e-4 * g / ((0.1 * g * g).sqrt() + 1e-7)
    assert_close(tr.gs.flat[:tr.gs.ntrain], ref, 1e-6, "first RMSprop step (fused Dense path)")
    for name, pf in (("sun.fc1.kernel", tr.fc1), ("sun.fc2.kernel", tr.fc2)):
        fresh = K.PackedFC(tr.gs.w[name], precise=False)
        assert torch.equal(fresh.pk_hi.view(torch.int16), pf.pk_hi.view(torch.int16)), name
        assert torch.equal(fresh.nat_hi.view(torch.int16), pf.nat_hi.view(torch.int16)), name


def test_cli_train_and_inference_smoke(dev, tmp_path, capsys):
    """train CLI (2 synthetic epochs at batch 2) writes nothing before epoch 10; inference CLI turns a jpg into a .hdr."""
    train = pkg("train"); inference = pkg("inference"); hdr_io = pkg("hdr_io")
    sky, sun = str(tmp_path / "SKY"), str(tmp_path / "SUN")
    train.main(["--batchsize", "2", "--epochs", "2", "--steps-per-epoch", "1", "--sky", sky, "--sun", sun,
                "--logdir", str(tmp_path), "--val-steps", "1"])
    out = capsys.readouterr().out
    assert "gen_total_loss=" in out and "disc_real_loss=" in out and "[epoch 2][val] gen_total_loss=" in out
    import glob
    (evf,) = glob.glob(str(tmp_path / "tensorboard" / "SKY" / "*" / "train" / "events.out.tfevents.*"))
    ev = pkg("tb_logging").read_events(evf)
    tags = set().union(*[set(e["scalars"]) for e in ev])
    assert {"gen_total_loss", "gen_l1_loss", "gen_perceptual_loss", "gen_DoG_loss", "gen_adv_loss", "gen_kl_div",
            "disc_total_loss", "disc_generated_loss", "disc_real_loss", "g_out", "b_out"} == tags
    assert sorted({e["step"] for e in ev[1:]}) == [1, 2]
    (evv,) = glob.glob(str(tmp_path / "tensorboard" / "SKY" / "*" / "val" / "events.out.tfevents.*"))
    val = pkg("tb_logging").read_events(evv)
    assert [e["step"] for e in val[1:]] == [1, 2] and "gen_kl_div" in val[1]["scalars"]
    from PIL import Image
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    img = (np.random.default_rng(0).uniform(0, 255, (32, 128, 3))).astype(np.uint8)
    Image.fromarray(img).save(str(indir / "sky1.jpg"), quality=95)
    inference.main(["--indir", str(indir), "--outdir", str(outdir), "--sky", sky, "--sun", sun])
    back = hdr_io.read_hdr(str(outdir / "sky1.hdr"))
    assert back.shape == (32, 128, 3) and np.isfinite(back).all() and back.max() > 0


def test_cli_train_sun_smoke(dev, tmp_path, capsys):
    """sun-pose pre-training CLI: 10 tiny synthetic epochs -> one SUN checkpoint (every 10th epoch, max_to_keep=5) that
    the next invocation restores (weights, Adam slots and step count) and that train.py / inference.py can read."""
    train_sun, ckpt = pkg("train_sun"), pkg("checkpoint")
    sun = str(tmp_path / "SUN")
    train_sun.main(["--batchsize", "2", "--epochs", "10", "--steps-per-epoch", "1", "--sun", sun])
    out = capsys.readouterr().out
    assert "train_loss_SUN" in out and "Saved checkpoint for epoch 10" in out
    tensors, epoch = ckpt.CheckpointManager(sun).restore()
    assert epoch == 10 and int(tensors["optimizer/iter"]) == 10
    assert tensors["lin/fc1/kernel"].shape == (8192, 4096) and np.isfinite(tensors["lin/fc1/kernel"]).all()
    train_sun.main(["--batchsize", "2", "--epochs", "11", "--steps-per-epoch", "1", "--sun", sun])
    out = capsys.readouterr().out
    assert "Latest checkpoint has restored!!" in out and "[epoch 11]" in out and "[epoch 10]" not in out
    train_sun.main(["--train", "false", "--batchsize", "2", "--sun", sun])
    assert "inference: cmf max" in capsys.readouterr().out


@pytest.mark.parametrize("mode", ["BF16X3", "BF16"])
def test_step_is_repeatable_under_stream_concurrency(dev, mode):
    """The step runs as segments on three streams.  Repeating it from the same state must reproduce EVERY gradient of both
    optimizers bit for bit: the conv weight gradients' split-K partials are summed in a fixed order
    (hdrsky_conv2d_wgrad_multi_det), the norm layers' (d gamma, d beta) by hdrsky_dgb_reduce, nothing is accumulated by
    floating-point atomics.  The Dense gradients must also equal flat^T @ df1 of the step's own tensors - this caught a
    kernel whose operands were disturbed by a concurrently running segment."""
    tr, _, batch = _mk(dev, 2, mode)
    ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    w0g, w0d = tr.gs.flat.clone(), tr.ds.flat.clone()
    first = None
    for it in range(6):
        tr.gs.flat.copy_(w0g); tr.ds.flat.copy_(w0d)
        tr.step(ldr, hdr, gt, update=False)
        torch.cuda.synchronize()
        T = tr._T
        opd = (lambda t: t.to(torch.bfloat16).double()) if mode == "BF16" else (lambda t: t.double())   # hdrsky_fc_wgrad_bf16
        ref = opd(T["t"]["flat"]).t() @ opd(T["df1"])
        got = tr.gs.g["sun.fc1.kernel"]
        assert float((got.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), it
        snap = {"gs": tr.gs.grad.clone(), "ds": tr.ds.grad.clone(), "y": T["y_lin"].clone(), "losses": tr.losses.clone()}
        if first is None:
            first = snap
            assert float(snap["gs"].abs().max()) > 0 and float(snap["ds"].abs().max()) > 0
        else:
            for k, fp in (("gs", tr.gs), ("ds", tr.ds)):
                if not torch.equal(first[k], snap[k]):
                    bad = (first[k] != snap[k]).nonzero().flatten()
                    names = sorted({n for n, (o, cnt, _) in fp.offsets.items() if o < fp.ntrain and
                                    bool(((bad >= o) & (bad < o + cnt)).any())})
                    raise AssertionError("iteration %d: %d gradient elements of %s differ between runs: %s" %
                                         (it, bad.numel(), k, names[:8]))
            assert torch.equal(first["y"], snap["y"]), it
            assert float((first["losses"] - snap["losses"]).abs().max()) <= 1e-5 * float(first["losses"].abs().max())


def test_train_step_distortion_aware_res_stack(dev):
    """Trainer(distortion_aware=True): the res blocks' convolutions are distortion_aware_ops.conv2d (generator.py:14,18's
    commented-out variant) in the forward AND the backward pass; one step at B=2 against the oracle's autograd through the
    numpy restatement of the layer and its adjoint (oracle/networks._DAConv)."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    nets = [params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
            params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3)]
    batch = synth.make_batch(2, seed=1234)
    tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(*[tt(n) for n in nets], ldr, hdr, gt, distortion_aware=True)
    plain = ostep.generator_graph(tt(nets[0]), {k: v.clone().requires_grad_(True) for k, v in tt(nets[1]).items()}, ldr,
                                  y_index=gt.argmax(dim=1), training=True, new_stats={})
    assert float((plain["res_out"].detach() - outs["res_out"]).abs().max()) > 1e-3      # the variant really differs
    tr = trainer.Trainer(*nets, device=dev, precise=True, compute=K.BF16X3, distortion_aware=True)
    out = tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    got = tr.loss_dict()
    for k, rk in (("kl", "kl"), ("perceptual", "perceptual"), ("dog", "dog"), ("l1", "l1"), ("adv", "adv")):
        assert abs(got[k] - losses[rk]) <= 2e-3 * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    assert_close(out["y_final_gamma"], outs["y_final_gamma"], 1e-3, "y_final_gamma (distortion-aware res stack)")
    worst = []
    for i in range(6):
        for j in (1, 2):
            for leaf in ("conv%d.w" % j, "norm%d.gamma" % j, "norm%d.beta" % j):
                k = "res.%d.%s" % (i, leaf)
                worst.append((rel_max(tr.gs.g["gen." + k], gg[k]), k))
    worst.sort(reverse=True)
    print("worst distortion-aware res-stack gradient errors:", worst[:4])
    assert worst[0][0] < 5e-2 and np.median([e for e, _ in worst]) < 3e-3, worst[:4]
    for k in ("conv3_d.w", "conv1_d.w"):          # and what lies behind the stack in the backward chain
        assert rel_max(tr.gs.g["gen." + k], gg[k]) < 5e-2, k
    tr.apply_gradients()
    assert torch.isfinite(tr.gs.flat).all()
    # bench mode of the same variant runs and stays finite
    tr16 = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16, distortion_aware=True)
    tr16.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=True)
    assert all(np.isfinite(v) for v in tr16.loss_dict().values()) and torch.isfinite(tr16.gs.flat).all()


def test_adam_kernel(dev):
    """hdrsky_adam vs the oracle's Keras-OptimizerV2 Adam over three consecutive steps."""
    K = pkg("kernels")
    rng = np.random.default_rng(21)
    n = 4096 + 8
    w0 = rng.standard_normal(n).astype(np.float32)
    w, m, v = torch.from_numpy(w0).to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    wr, mr, vr = torch.from_numpy(w0), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):
        g = torch.from_numpy((rng.standard_normal(n) * 10.0 ** rng.uniform(-4, 1, n)).astype(np.float32))
        K.adam(w, g.to(dev), m, v, 1e-4, step)
        wr, mr, vr = T.adam_update(wr, g, mr, vr, 1e-4, step)
    assert_close(w, wr, 1e-6, "adam w"); assert_close(m, mr, 1e-6, "adam m"); assert_close(v, vr, 1e-6, "adam v")


def test_sunpose_pretraining_step_matches_oracle(dev):
    """train_sun.py:220-264: KL + DoG loss on the cmf image, gradients of every sun-pose variable, one Adam update.
    The DoG term is an L1 of differences that are ~1e-9 wherever both 1-channel images are flat (a random-init net's
    cmf is nearly uniform), so the sign() in its gradient is decided by rounding noise there - in the oracle as much as
    here.  The backward chain is therefore checked tightly with the DoG term switched off, and with the reference's
    weight 1 the losses are checked tightly and the gradients by direction (cosine)."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    from oracle import networks as onet
    sun = params.init_params(params.sunpose_spec(), 1)
    sun_t = {k: torch.from_numpy(v) for k, v in sun.items()}
    # A Dense unit whose pre-activation is within rounding of the ReLU kink (|o| ~ 1e-5 of the layer's maximum) is on in
    # one implementation and off in the other, and one such unit moves the fc1 gradients by >10 %: pick a batch whose
    # Dense pre-activations keep a margin from zero (with 2 x 4096 units per layer about every second batch has none).
    for seed in range(1234, 1264):
        batch = synth.make_batch(2, seed=seed)
        ldr, gt = torch.from_numpy(batch["ldr"]), torch.from_numpy(batch["sunpose_gt"])
        with torch.no_grad():
            a = ldr
            for l in (1, 2, 3):
                a = T.maxpool2x2(onet._sunpose_layer(sun_t, "sunlayer%d" % l, a))
            o1 = T.dense(T.flatten_nhwc(a), sun_t["fc1.kernel"], sun_t["fc1.bias"])
            o2 = T.dense(torch.relu(o1), sun_t["fc2.kernel"], sun_t["fc2.bias"])
        if min(float(o1.abs().min() / o1.abs().max()), float(o2.abs().min() / o2.abs().max())) > 3e-5:
            break
    else:
        pytest.skip("no batch with a ReLU margin found")
    print("batch seed", seed)
    tr = trainer.SunPoseTrainer(sun, device=dev, precise=True, compute=K.BF16X3)
    for dog_weight in (0.0, 1.0):
        losses, grads, outs = ostep.sun_train_step_grads(sun_t, ldr, gt, dog_weight)
        pred, gt_img, cams = tr.step(ldr.to(dev), gt.to(dev), update=False, dog_weight=dog_weight)
        got = tr.loss_dict()
        for k in (("kl",) if dog_weight == 0.0 else ("kl", "dog", "sun_loss")):
            assert abs(got[k] - losses[k]) <= 1e-3 * abs(losses[k]) + 1e-7, (k, got[k], losses[k])
        assert_close(pred.reshape(2, -1), outs["sunpose_cmf"], 2e-3, "cmf")
        worst, dots = [], [0.0, 0.0, 0.0]
        for k, ref in grads.items():
            g = tr.gs.g["sun." + k].cpu()
            scale = float(ref.abs().max())
            if (k.endswith(".b") and "conv" in k) or scale == 0.0:   # conv bias before an InstanceNorm: exactly zero gradient
                wk = "sun." + k[:-2] + ".w"
                assert float(g.abs().max()) <= 1e-4 * float(tr.gs.g[wk].abs().max()), k
                continue
            worst.append((float((g - ref).abs().max()) / scale, k))
            dots[0] += float((g.double() * ref.double()).sum()); dots[1] += float((g.double() ** 2).sum()); dots[2] += float((ref.double() ** 2).sum())
        worst.sort(reverse=True)
        cos = dots[0] / (dots[1] * dots[2]) ** 0.5
        print("dog_weight", dog_weight, "cosine", cos, worst[:4])
        if dog_weight == 0.0:   # same bounds as the full training step: ReLU / max-pool mask flips at the 1e-2 level
            assert worst[0][0] < 5e-2, worst[:5]
            assert np.median([e for e, _ in worst]) < 3e-3
        assert cos > (0.9999 if dog_weight == 0.0 else 0.98), cos
    # one Adam update of the flat buffer
    w0 = tr.gs.flat[:tr.gs.ntrain].clone(); g0 = tr.gs.grad.clone()
    tr.apply_gradients()
    wr, _, _ = T.adam_update(w0.cpu(), g0.cpu(), torch.zeros_like(w0.cpu()), torch.zeros_like(w0.cpu()), tr.lr, 1)
    assert_close(tr.gs.flat[:tr.gs.ntrain], wr, 1e-6, "adam step of the sun-pose net")


def test_three_optimizer_steps_follow_the_oracle(dev):
    """Three consecutive train_steps (train.py:382-415: both tapes on the pre-update weights, RMSprop x2, BatchNorm moving
    statistics carried over) from the same initial weights: the loss terms of every step against the oracle's loop.
    (From a random initialisation the first RMSprop step - every weight moves by ~3.2 lr - saturates the sun-pose softmax:
    the KL term jumps and then stays put, in the oracle exactly as here.)"""
    tr, (gen, sun, dis, vgg), batch = _mk(dev, 2)
    tt = lambda dd: {k: torch.from_numpy(v).clone() for k, v in dd.items()}
    g_, s_, d_, v_ = tt(gen), tt(sun), tt(dis), tt(vgg)
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    ms = {}
    names = ("kl", "perceptual", "dog", "l1", "adv", "total_gen_loss", "total_disc_loss")
    for it in range(3):
        losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(g_, s_, d_, v_, ldr, hdr, gt)
        tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=True)
        got = tr.loss_dict()
        print(it, {k: (round(got[k], 4), round(losses[k], 4)) for k in names})
        tol = 2e-3 if it == 0 else 1e-2      # later steps (measured <= 2e-3): RMSprop's sign-like first updates amplify rounding-level gradients
        for k in names:
            assert abs(got[k] - losses[k]) <= tol * abs(losses[k]) + 1e-5, (it, k, got[k], losses[k])
        for net, grads, pre in ((g_, gg, "g."), (s_, gs, "s."), (d_, gd, "d.")):
            for k, g in grads.items():
                net[k], ms[pre + k] = T.rmsprop_update(net[k], g, ms.get(pre + k, torch.zeros_like(g)), tr.lr)
        g_.update(sg); d_.update(sd)


@pytest.mark.parametrize("B,mode,da", [(4, "BF16", False), (32, "BF16", False), (3, "BF16X3", False), (4, "BF16", "all")])
def test_captured_replays_match_eager_steps(dev, B, mode, da):
    """The captured step (one hipGraph per segment, three streams) replayed several times against the same number of eager
    steps from the same state: weights of both optimizers, BatchNorm moving statistics and every loss term stay equal step
    after step - and a gradient-only replay repeated from a restored state reproduces its gradients bit for bit.  (hipGraph
    memset nodes did their job in the first replay only: from the second replay on accumulators cleared that way were stale
    and the captured step diverged from the eager one; hdrsky_zero is a kernel since.)"""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    mk = lambda: trainer.Trainer(params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
                                 params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3),
                                 device=dev, precise=(mode == "BF16X3"), compute=getattr(K, mode), distortion_aware=da)
    batch = synth.make_batch(B, seed=4321)
    ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    te, tc = mk(), mk()
    tc.capture(ldr, hdr, gt)
    for it in range(4):
        te.step(ldr, hdr, gt, update=True)
        tc.replay(update=True)
        torch.cuda.synchronize()
        spare = [torch.empty(1 << 20, device=dev).normal_() for _ in range(8)]     # (eager allocations between replays)
        for name, a, b in (("generator + sun-pose", te.gs.flat, tc.gs.flat), ("discriminator", te.ds.flat, tc.ds.flat)):
            assert torch.isfinite(b).all(), (it, name)
            err = float((a - b).abs().max())
            assert err <= 1e-6 * float(a.abs().max()), (it, name, err)
        le, lc = te.losses.double(), tc.losses.double()
        assert float((le - lc).abs().max()) <= 1e-4 * float(le.abs().max()), (it, te.losses.tolist(), tc.losses.tolist())
        del spare
    # gradient-only replays from a restored state: the same gradients every time, bit for bit.  (This caught the gradient
    # of the batch-global maximum, generator.py:160, going to whichever of several TIED maximal elements a thread claimed
    # first: hdrsky_sun_rad_bwd now counts the tied elements and gives each an equal share, tf.reduce_max's gradient.)
    w0g, w0d = tc.gs.flat.clone(), tc.ds.flat.clone()
    ref = None
    for it in range(3):
        tc.gs.flat.copy_(w0g); tc.ds.flat.copy_(w0d)
        tc.replay(update=False)
        torch.cuda.synchronize()
        snap = (tc.gs.grad.clone(), tc.ds.grad.clone())
        if ref is None:
            ref = snap
            continue
        assert torch.equal(ref[1], snap[1]), it
        for name, (o, cnt, _) in tc.gs.offsets.items():
            if o >= tc.gs.ntrain:
                continue
            a, b = ref[0][o:o + cnt], snap[0][o:o + cnt]
            assert torch.equal(a, b), (it, name)


def test_replay_after_an_eager_step_runs_the_captured_binding(dev):
    """An eager step() between two replays (another batch, another batch size) re-binds the trainer's plan; replay() goes
    back to the captured binding: same gradients as before the eager pass, bit for bit, and the captured output tensors."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    tr = trainer.Trainer(params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
                         params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3), device=dev)
    with pytest.raises(RuntimeError):
        tr.replay()
    b4, b2 = synth.make_batch(4, seed=77), synth.make_batch(2, seed=78)
    t = lambda b: tuple(torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    out = tr.capture(*t(b4))
    state = (tr.gs.flat.clone(), tr.ds.flat.clone())          # (BatchNorm moving statistics move even without an update)
    tr.replay(update=False)
    torch.cuda.synchronize()
    ref = (tr.gs.grad.clone(), tr.ds.grad.clone(), out["y_final_gamma"].clone())
    eager = tr.step(*t(b2), update=False)
    assert eager["y_final_gamma"].shape[0] == 2
    tr.gs.flat.copy_(state[0]); tr.ds.flat.copy_(state[1])
    tr.replay(update=False)
    torch.cuda.synchronize()
    assert tr._outputs()["y_final_gamma"] is out["y_final_gamma"]
    assert torch.equal(tr.gs.grad, ref[0]) and torch.equal(tr.ds.grad, ref[1]) and torch.equal(out["y_final_gamma"], ref[2])


def test_train_step_at_a_size_outside_the_mfma_dense_tile(dev):
    """--imheight 40 --imwidth 72 (both multiples of 8, as the three stride-2 stages need): H*W = 2880 is not a multiple of
    the 256-column tile of the matrix-core Dense gradient (csrc/fc_update.hip), so the bench-mode trainer must fall back
    to hdrsky_fc_wgrad + hdrsky_rmsprop_fc instead of failing with EINVAL at the first step; losses and the Dense
    gradients against the oracle, then two updating steps."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    H, W, B = 40, 72, 2
    assert not K.fc_xtdy_supported(H * W, H * W) and K.fc_xtdy_supported(4096, 4096)
    gen = params.init_params(params.generator_spec(H, W), 0); sun = params.init_params(params.sunpose_spec(H, W), 1)
    dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
    batch = synth.make_batch(B, H, W, seed=77)
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16, im_height=H, im_width=W)
    assert not tr.dense_mfma and not tr.fused_dense
    tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(tt(gen), tt(sun), tt(dis), tt(vgg), ldr, hdr, gt)
    tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    got = tr.loss_dict()
    # (the adversarial term sees the prediction through the randomly initialised discriminator, which amplifies the bf16
    # rounding of y_final_lin - DESIGN.md section 2, "chaotic w.r.t. its input": 10 % there, 2 % on the direct terms)
    for k, rk, tol in (("kl", "kl", 2e-2), ("perceptual", "perceptual", 2e-2), ("dog", "dog", 2e-2), ("l1", "l1", 2e-2), ("adv", "adv", 1e-1)):
        assert abs(got[k] - losses[rk]) <= tol * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    for k in ("fc1.kernel", "fc2.kernel", "fc1.bias", "fc2.bias"):
        g, v = tr.gs.g["sun." + k].cpu().double(), gs[k].double()
        cos = float((g * v).sum() / (g.norm() * v.norm() + 1e-300))
        assert cos > 0.98, (k, cos)
    w0 = tr.gs.flat.clone()
    for _ in range(2):
        tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=True)
    assert torch.isfinite(tr.gs.flat).all() and torch.isfinite(tr.ds.flat).all() and not torch.equal(w0, tr.gs.flat)


def test_split_discriminator_passes_equal_the_paired_batch(dev, monkeypatch):
    """HDRSKY_DISC_SPLIT=1: the discriminator's pass over the real pairs as a segment of its own beside the forward pass
    (moving-statistics update deferred to its place in the program order), the generated pairs in disc_step - against the
    default (both halves as one batch of 2B): same losses, same discriminator gradients (two weight-gradient launches add
    in another order), same weights and BatchNorm moving statistics after two captured steps."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    mk = lambda: trainer.Trainer(params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
                                 params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3),
                                 device=dev, precise=False, compute=K.BF16)
    batch = synth.make_batch(4, seed=77)
    ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    res = {}
    monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1")
    for split in ("0", "1"):
        monkeypatch.setenv("HDRSKY_DISC_SPLIT", split)
        t = mk()
        t.step(ldr, hdr, gt, update=False)
        torch.cuda.synchronize()
        names = [n for n, *_ in t._segs]
        assert ("disc_real" in names) == (split == "1")
        g, losses = t.ds.grad.clone(), t.losses.clone()
        t.capture(ldr, hdr, gt)
        for _ in range(2):
            t.replay(update=True)
        torch.cuda.synchronize()
        res[split] = (g, losses, t.ds.flat.clone(), t.gs.flat.clone())
    (g0, l0, d0, w0), (g1, l1, d1, w1) = res["0"], res["1"]
    assert float((l0 - l1).abs().max()) <= 1e-5 * float(l0.abs().max()), (l0.tolist(), l1.tolist())
    assert rel_max(g1, g0) < 2e-4
    assert rel_max(d1, d0) < 1e-3 and rel_max(w1, w0) < 1e-3      # weights + moving statistics after two updates
