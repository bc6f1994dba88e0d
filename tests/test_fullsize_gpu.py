"""Parity at BASELINE.json's full size (batch 32, 32x128).  Two kinds of checks: (1) the training step against the CPU oracle
directly (`oracle/step.train_step_grads`: ~0.5 s per forward + a few seconds of autograd at this size on 16 threads) in
both contraction modes, and the 128x512 step at B = 2; (2) size-independent relations the arithmetic must satisfy by
itself: batch independence of the per-sample parts of the graph, linearity and adjointness of the convolution kernels, the
tone-map round trip, the equality of replayed and eagerly issued steps, and the sum rule that makes data parallelism
exact."""
import numpy as np
import pytest
import torch

from conftest import pkg
from util import assert_close, rel_max, rel_rms

pytestmark = pytest.mark.gpu
B, H, W = 32, 32, 128


def _nets(dev, precise):
    params, synth, engine = pkg("params"), pkg("synth"), pkg("engine")
    gen = params.init_params(params.generator_spec(), 0)
    sun = params.init_params(params.sunpose_spec(), 1)
    batch = synth.make_batch(B, seed=99)
    return engine, engine.Nets(gen, sun, device=dev, precise=precise), {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}


def test_forward_batch_independence(dev):
    """InstanceNorm couples nothing across samples: the encoder / sky decoder / sun-pose outputs of sample i in the
    batch of 32 equal those of the same sample in a batch of 4.  The tile shape - hence the summation order of the
    InstanceNorm partials - depends on the batch size, so the two runs differ by fp32 rounding; in BF16X3 that stays
    at 1e-4, in the single-product BF16 mode a last-bit difference can move an operand by one bf16 ulp, so that mode is
    checked at its own noise level."""
    K = pkg("kernels")
    # (the masked sky prediction is compared in the log-compressed domain and only in BF16X3: the alpha mask is a
    # ramp of width 0.12 on the decompressed value, so at saturated pixels a bf16-level difference of the decoder output
    # moves alpha - and the masked prediction - by tens of percent)
    for precise, compute, tols in ((True, K.BF16X3, dict(res_out=3e-4, sunpose_cmf=1e-3, sky_pred_gamma=3e-3)),
                                   (False, K.BF16, dict(res_out=3e-2, sunpose_cmf=3e-2))):
        engine, nets, bt = _nets(dev, precise)
        full = engine.generator_forward(nets, bt["ldr"], compute=compute)
        full["sky_pred_gamma"] = K.tonemap(full["sky_pred_lin"], False)
        for lo in (0, 12, 28):
            part = engine.generator_forward(nets, bt["ldr"][lo:lo + 4].contiguous(), compute=compute)
            part["sky_pred_gamma"] = K.tonemap(part["sky_pred_lin"], False)
            for k, tol in tols.items():
                assert_close(part[k], full[k][lo:lo + 4], tol, "%s[%d:%d]" % (k, lo, lo + 4))
    # the sun radiance divides by the maximum over the WHOLE batch tensor (generator.py:160): not batch independent
    assert float(full["sunpose_cmf"].max()) >= float(full["sunpose_cmf"][:4].max())


def test_conv_linearity_and_adjoint_fullsize(dev):
    """conv(a x1 + x2) = a conv(x1) + conv(x2) and <conv(x), y> = <x, dgrad(y)> for the res-block convolution on
    [32,8,32,128] and the stride-2 encoder convolution on [32,32,128,32] (bf16 products, fp32 accumulation: both sides
    of each relation round their operands identically only for the adjoint with exact bf16 inputs, so inputs are drawn
    on the bf16 grid)."""
    K = pkg("kernels")
    g = torch.Generator(device=dev); g.manual_seed(3)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    for (h, w, cin, cout, k, stride) in ((8, 32, 128, 128, 3, 1), (32, 128, 32, 64, 3, 2)):
        wgt = bf(torch.randn(k, k, cin, cout, device=dev, generator=g) / (k * k * cin) ** 0.5)
        pw, pwT = K.PackedConv(wgt, False), K.PackedConv(wgt, False, transpose_flip=True)
        x1, x2 = bf(torch.randn(B, h, w, cin, device=dev, generator=g)), bf(torch.randn(B, h, w, cin, device=dev, generator=g))
        y1, _ = K.conv2d(x1, pw, None, stride=stride); y2, _ = K.conv2d(x2, pw, None, stride=stride)
        y12, _ = K.conv2d(bf(2.0 * x1 + x2), pw, None, stride=stride)          # 2*x1+x2 stays on the bf16 grid or rounds: tolerance
        assert_close(y12, 2.0 * y1 + y2, 2e-2, "linearity %dx%d s%d" % (k, k, stride))
        dy = bf(torch.randn_like(y1))
        d = K.conv_desc(B, h, w, cin, cout, k, k, stride, True, 1)
        dx, _ = K.conv2d_dgrad(dy, pwT, d)
        lhs, rhs = float((y1.double() * dy.double()).sum()), float((x1.double() * dx.double()).sum())
        assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), float(y1.double().norm() * dy.double().norm()) * 1e-2), (lhs, rhs)
        # weight gradient: <conv(x; W'), dy> = <W', wgrad(x, dy)>
        dw, _ = K.conv2d_wgrad(x1, dy, k, k, stride=stride)
        lhs_w = float((wgt.double() * dw.double()).sum())
        assert abs(lhs_w - lhs) <= 2e-3 * abs(lhs) + 1e-3 * float(y1.double().norm() * dy.double().norm()), (lhs_w, lhs)


def test_tonemap_round_trip_and_blend_fullsize(dev):
    K = pkg("kernels")
    g = torch.Generator(device=dev); g.manual_seed(4)
    x = torch.rand(B, H, W, 3, device=dev, generator=g) ** 6 * 3.0e4          # up to the 30000 clamp of the sun radiance
    back = K.tonemap(K.tonemap(x, False), True)
    assert float(((back - x).abs() / (x + 1.0)).max()) < 2e-5
    sky, sun = torch.rand(B, H, W, 3, device=dev, generator=g) * 1.2, torch.rand(B, H, W, 3, device=dev, generator=g) * 2.0
    yg, yl, al, sl, ul = K.blend(sky, sun, 0.12)
    assert float(al.min()) >= 0.0 and float(al.max()) <= 1.0
    assert_close(yl, K.tonemap(yg, True), 1e-6, "y_lin = decompress(y_gamma)")
    assert_close(yg, (1 - al) * sky + al * sun, 1e-6, "alpha blend")


def test_replayed_step_equals_eager_step_fullsize(dev):
    """One training step at batch 32 issued eagerly and replayed from the per-segment hipGraphs: same losses and, for
    the atomics-free tensors, bit-identical results."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
    dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
    bt = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(B, seed=7).items()}
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16)
    w0g, w0d = tr.gs.flat.clone(), tr.ds.flat.clone()
    out = tr.step(bt["ldr"], bt["hdr_t"], bt["sunpose_gt"], update=False)
    torch.cuda.synchronize()
    eager = dict(y=out["y_final_lin"].clone(), fc1=tr.gs.g["sun.fc1.kernel"].clone(), losses=tr.losses.clone(),
                 gnorm=float(tr.gs.grad.double().norm()))
    tr.gs.flat.copy_(w0g); tr.ds.flat.copy_(w0d)
    out2 = tr.capture(bt["ldr"], bt["hdr_t"], bt["sunpose_gt"], warmup=0)
    tr.gs.flat.copy_(w0g); tr.ds.flat.copy_(w0d)       # the capture itself executes nothing, but keep the state explicit
    tr.replay(update=False)
    torch.cuda.synchronize()
    assert torch.equal(out2["y_final_lin"], eager["y"])
    assert torch.equal(tr.gs.g["sun.fc1.kernel"], eager["fc1"])
    assert float((tr.losses - eager["losses"]).abs().max()) <= 1e-5 * float(eager["losses"].abs().max())
    assert abs(float(tr.gs.grad.double().norm()) - eager["gnorm"]) <= 1e-4 * eager["gnorm"]
    assert all(np.isfinite(v) for v in tr.loss_dict().values())


def test_gradient_sum_rule_for_data_parallelism(dev):
    """Every loss is a batch mean, so for the batch-independent part of the model the gradient of a batch of 8 equals
    the mean of the gradients of its two halves - the identity behind 'all-reduce sum, scale by 1/world'.  Checked on
    the sun-pose pre-training step (InstanceNorm only: no batch coupling)."""
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    sun = params.init_params(params.sunpose_spec(), 1)
    bt = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(8, seed=11).items()}
    tr = trainer.SunPoseTrainer(sun, device=dev, precise=True, compute=K.BF16X3)
    tr.step(bt["ldr"], bt["sunpose_gt"], update=False, want_cams=False, dog_weight=0.0)
    whole = tr.gs.grad.clone()
    acc = torch.zeros_like(whole)
    for lo in (0, 4):
        tr.step(bt["ldr"][lo:lo + 4].contiguous(), bt["sunpose_gt"][lo:lo + 4].contiguous(), update=False, want_cams=False,
                dog_weight=0.0)
        acc += tr.gs.grad
    assert_close(0.5 * acc, whole, 2e-4, "mean of the shard gradients = gradient of the whole batch")


def test_batch_larger_than_32(dev):
    """The Dense kernels take 32 rows per launch; larger batches are sliced by the front end.  Forward of batch 48 ==
    the forwards of its samples in batches of 16 (BF16X3), and one training step at batch 40 runs and stays finite."""
    K, params, synth, engine, trainer = pkg("kernels"), pkg("params"), pkg("synth"), pkg("engine"), pkg("trainer")
    gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
    nets = engine.Nets(gen, sun, device=dev, precise=True)
    ldr = torch.from_numpy(synth.make_batch(48, seed=3)["ldr"]).to(dev)
    full = engine.generator_forward(nets, ldr, compute=K.BF16X3)
    for lo in (0, 16, 32):
        part = engine.generator_forward(nets, ldr[lo:lo + 16].contiguous(), compute=K.BF16X3)
        assert_close(part["sunpose_cmf"], full["sunpose_cmf"][lo:lo + 16], 1e-3, "cmf[%d:%d]" % (lo, lo + 16))
    dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
    bt = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(40, seed=4).items()}
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16)
    tr.step(bt["ldr"], bt["hdr_t"], bt["sunpose_gt"], update=True)
    assert torch.isfinite(tr.gs.flat).all() and torch.isfinite(tr.ds.flat).all()
    assert all(np.isfinite(v) for v in tr.loss_dict().values())


def test_other_image_size_64x256(dev):
    """--imheight 64 --imwidth 256: the sun-pose Dense layers grow to 32768x16384 / 16384x16384 and the soft-max row to
    64 KB of LDS.  Forward + one training step run, stay finite, and the soft-max rows sum to one.  (16x64 and smaller
    cannot train: the discriminator's VALID 4x4 output conv needs at least 4 rows after three stride-2 stages, as in
    the reference.)"""
    K, params, synth, engine, trainer = pkg("kernels"), pkg("params"), pkg("synth"), pkg("engine"), pkg("trainer")
    h, w = 64, 256
    gen = params.init_params(params.generator_spec(h, w), 0); sun = params.init_params(params.sunpose_spec(h, w), 1)
    bt = synth.make_batch_device(2, h, w, seed=1, device=dev)
    nets = engine.Nets(gen, sun, device=dev, precise=False, im_height=h, im_width=w)
    out = engine.generator_forward(nets, bt["ldr"], compute=K.BF16)
    assert tuple(out["y_final_lin"].shape) == (2, h, w, 3) and torch.isfinite(out["y_final_lin"]).all()
    assert torch.allclose(out["sunpose_cmf"].sum(dim=1), torch.ones(2, device=dev), atol=1e-4)
    del nets
    dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16, im_height=h, im_width=w)
    tr.step(bt["ldr"], bt["hdr_t"], bt["sunpose_gt"], update=True)
    assert all(np.isfinite(v) for v in tr.loss_dict().values()) and torch.isfinite(tr.gs.flat).all()


def test_bench_mode_psnr_at_trained_like_weights(dev):
    """north_star: 'output PSNR within 0.05 dB of the reference' - checked where it can fail.  At random initialisation
    PSNR(output, target) is 14 dB and any error 48 dB down moves it by 0.002 dB whatever the kernels do; here the weights
    come from bench.PARITY_FIT_STEPS steps of the product's own training step (fp32-class BF16X3 mode: the stand-in for a
    reference-trained model) on seeded synthetic data (no weight blob: <pkg>/train.py::fit_synthetic), after which the
    output must be a reconstruction (PSNR >= 30 dB against the log-compressed target).  bench.py's `parity` object at the
    bench size (B = 32): the benchmarked bf16 mode and the BF16X3 mode against the ORACLE's fp32 output on an 8-image
    subset and against the target; tolerance = the clause's
    0.05 dB, and the modes' mutual PSNR must leave room for it (q_max above the quality reached)."""
    import bench
    import importlib
    from conftest import PKG
    mods = {m: importlib.import_module(PKG + "." + m) for m in ("params", "synth", "engine", "trainer", "kernels", "train")}
    params = mods["params"]
    nets = (params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
            params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3))
    p = bench.parity_object(torch, mods, dev, nets, B, bench.oracle_outputs_fn(torch, 8))
    print(p)
    assert p["images"] == B and p["oracle_images"] == 8
    # trained-like, not noise (the 8-image subset scatters around the batch figure)
    assert p["psnr_x3_vs_target_db"] >= 30.0 and p["psnr_oracle_vs_target_db"] >= 27.0, p
    assert abs(p["delta_psnr_vs_target_db"]) <= 0.05 and abs(p["delta_psnr_vs_oracle_target_db"]) <= 0.05, p
    assert abs(p["delta_psnr_x3_vs_oracle_target_db"]) <= 0.01, p
    assert p["psnr_x3_vs_oracle_db"] >= 70.0 and p["psnr_bf16_vs_oracle_db"] >= 55.0, p
    assert p["q_max_db"] >= p["psnr_bf16_vs_target_db"], p       # the bound the clause needs at the quality reached
    # (the subset runs through the GPU graph as a batch of its own: generator.py:160's tf.reduce_max couples the images of a
    # batch, so rows of a 32-image run are NOT comparable with a subset run through the oracle - 47 dB apart when first tried)
    assert p["within_0p05_db"], p


# ---------------------------------------------------------------------------------------------------------------------
# The full training step (train.py:382-415) at the HEADLINE size against the oracle: the per-tensor
# comparisons tests/test_train_gpu.py makes at B = 2, at B = 32.
# ---------------------------------------------------------------------------------------------------------------------
def _step_vs_oracle(dev, mode, Bsz):
    from oracle import step as ostep
    params, synth, trainer, K = pkg("params"), pkg("synth"), pkg("trainer"), pkg("kernels")
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
    dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
    batch = synth.make_batch(Bsz, seed=1234)
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=(mode == "BF16X3"), compute=getattr(K, mode))
    tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(tt(gen), tt(sun), tt(dis), tt(vgg), ldr, hdr, gt)
    out = tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    return tr, tr.loss_dict(), losses, gg, gs, sg, out, outs, dis, (ldr, hdr)


LOSS_NAMES = (("kl", "kl"), ("perceptual", "perceptual"), ("dog", "dog"), ("l1", "l1"), ("adv", "adv"),
              ("disc_generated", "generated"), ("disc_real", "real"), ("total_gen_loss", "total_gen_loss"),
              ("total_disc_loss", "total_disc_loss"))


def _grad_errors(tr, gg, gs, zero_bias_tol=1e-4):
    """(rel max err, rel rms err, elements, name) of every generator-step gradient tensor that is not an exactly-zero bias.
    zero_bias_tol: a bias in front of an InstanceNorm has a zero gradient in exact arithmetic; what the kernels return is the sum
    over B * H * W pixels of the ROUNDED gradient with respect to the conv output - fp32 rounding in the fp32-class mode, the bf16
    storage rounding of that tensor (2^-9 relative per element, 131 072 elements for the full-resolution layers) in the bench mode:
    measured there up to 1.03e-4 of (largest kernel-gradient element x fan-in) on sun.sunlayer1.conv1 (round 5 build; 0.9e-4 in
    round 4's), asserted at 2e-4."""
    rows = []
    for prefix, ref in (("gen.", gg), ("sun.", gs)):
        for k, v in ref.items():
            g = tr.gs.g[prefix + k]
            is_bias = k.endswith(".b") or k.endswith("bias_deconv2d")
            if is_bias and not k.startswith("conv1_f") and not k.startswith("conv1_u"):
                wk = k[:-2] + ".w" if k.endswith(".b") else k.replace("bias_deconv2d", "kernel_deconv2d")
                scale = float(ref[wk].abs().max()) * float(np.prod(ref[wk].shape[:3]))
                assert float(g.abs().max()) <= zero_bias_tol * scale, (prefix + k, float(g.abs().max()), scale)
                continue
            rows.append((rel_max(g, v), rel_rms(g, v), int(v.numel()), prefix + k))
    return rows


def _cosine(tr, gg, gs):
    dot = na = nb = 0.0
    for prefix, ref in (("gen.", gg), ("sun.", gs)):
        for k, v in ref.items():
            g = tr.gs.g[prefix + k].cpu().double()
            dot += float((g * v.double()).sum()); na += float((g * g).sum()); nb += float((v.double() ** 2).sum())
    return dot / (na * nb) ** 0.5, (na / nb) ** 0.5


def test_train_step_b32_fp32_class_against_the_oracle(dev):
    """BF16X3 (the fp32-class mode), B = 32, 32x128: every loss term to 2e-3, the prediction to 1e-3 of its maximum, every
    generator / sun-pose gradient tensor to 5e-2 of its maximum (median tensor 3e-3: what is left are the discrete effects
    test_train_gpu.py lists - sign() of the L1 / DoG terms, ReLU / max-pool masks, Grad-CAM arg-max ties), BatchNorm moving
    statistics to 1e-4, and the discriminator step at OUR prediction (the generated pass is chaotic in its input)."""
    from oracle import step as ostep
    tr, got, losses, gg, gs, sg, out, outs, dis, (ldr, hdr) = _step_vs_oracle(dev, "BF16X3", B)
    for k, rk in LOSS_NAMES:
        assert abs(got[k] - losses[rk]) <= 2e-3 * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    # (1e-3 at B = 2; the maximum over 16x the pixels sits higher - measured 1.1e-3, on the alpha ramp of saturated pixels
    # (width 0.12 on the decompressed value, train.py:258-261) - while the rms error stays two orders below)
    print("B = 32 BF16X3 y_final_gamma: rel max %.3e, rel rms %.3e" % (rel_max(out["y_final_gamma"], outs["y_final_gamma"]),
                                                                  rel_rms(out["y_final_gamma"], outs["y_final_gamma"])))
    assert_close(out["y_final_gamma"], outs["y_final_gamma"], 2.5e-3, "y_final_gamma, B = 32 (training mode)")
    assert rel_rms(out["y_final_gamma"], outs["y_final_gamma"]) < 1e-4
    rows = sorted(_grad_errors(tr, gg, gs), reverse=True)
    print("B = 32 BF16X3 worst gradient tensors (rel max, rel rms, elements, name):", rows[:8])
    assert rows[0][0] < 5e-2, rows[:5]
    assert np.median([e for e, _, _, _ in rows]) < 3e-3
    cos, ratio = _cosine(tr, gg, gs)
    assert cos > 0.9999 and abs(ratio - 1.0) < 2e-3, (cos, ratio)
    for k, v in sg.items():
        assert_close(tr.gs.w["gen." + k], v, 1e-4, "gen " + k)
    dr = {k: torch.from_numpy(v).clone().requires_grad_("moving" not in k) for k, v in dis.items()}
    stats = {}
    dl = ostep.discriminator_losses(dr, ldr, hdr, out["y_final_lin"].cpu(), training=True, new_stats=stats)
    names = [k for k in dr if "moving" not in k]
    # (3e-4 at B = 2.  At B = 32 every tensor's relative rms error stays at 1e-4 ... 3e-4, but single output channels of d3 / d4
    # whose batch variance over the 64 samples is small amplify the fp32-class rounding of their inputs through rstd: 32 of
    # 524 288 elements of d3's kernel gradient and 323 of 2.1 M of d4's (one channel, the same whose d beta is worst) sit at
    # 1e-3 ... 5e-3 of the tensor's maximum - profiles/experiments/misc/dis_grad_b32.py; the launches are bit-reproducible)
    dref = list(zip(names, torch.autograd.grad(dl["total_disc_loss"], [dr[k] for k in names])))
    derr = sorted(((rel_max(tr.ds.g["dis." + k], v), rel_rms(tr.ds.g["dis." + k], v), k) for k, v in dref), reverse=True)
    print("B = 32 BF16X3 discriminator gradients (rel max, rel rms, name), worst:", derr[:4])
    assert derr[0][0] < 1e-2 and max(r for _, r, _ in derr) < 1e-3, derr[:4]
    for k, v in stats.items():
        assert_close(tr.ds.w["dis." + k], v, 1e-4, "dis " + k)


def test_train_step_b32_bench_mode_against_the_oracle(dev):
    """The mode bench.py times (one bf16 MFMA product, fp32 accumulation), B = 32, 32x128, against the fp32 oracle.
    Loss terms that do not pass through the discriminator: 2e-2 (measured: KL 5e-5, perceptual 1e-4, DoG 6e-3, L1 6e-3,
    real-pair term 6e-4).  The three terms that evaluate the randomly initialised discriminator AT THE PREDICTION (adv,
    disc_generated and the totals that contain them) are chaotic in that input - a 1e-4 perturbation of y inside the oracle
    moves its own discriminator gradients by 2-8 %, tests/test_train_gpu.py - so they are checked twice: against the oracle's
    step at 6e-2 (measured 3.3 % / 4.0 %), and against the oracle's discriminator evaluated at OUR prediction at 2e-2, which
    is the bf16 error of the discriminator itself.  Prediction PSNR > 40 dB, whole-gradient cosine > 0.99, norm ratio within
    7 %; the TEN LARGEST gradient tensors (99.6 % of the 55.6 M trainable scalars) each within TWICE the relative rms error this
    build measures for it (0.04 ... 0.13: the table below)."""
    from oracle import networks as N
    from oracle import step as ostep
    tr, got, losses, gg, gs, sg, out, outs, dis, (ldr, hdr) = _step_vs_oracle(dev, "BF16", B)
    print("B = 32 bf16 losses (got, oracle):", {k: (round(got[k], 5), round(float(losses[rk]), 5)) for k, rk in LOSS_NAMES})
    through_d = ("adv", "disc_generated", "total_gen_loss", "total_disc_loss")
    for k, rk in LOSS_NAMES:
        tol = 6e-2 if k in through_d else 2e-2
        assert abs(got[k] - losses[rk]) <= tol * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    # the discriminator terms on identical inputs: the oracle's discriminator at the prediction this step produced
    y = out["y_final_lin"].cpu()
    dd = {k: torch.from_numpy(v) for k, v in dis.items()}
    adv_o = float(((N.discriminator(dd, ldr, y, training=False) - 1.0) ** 2).mean())
    dl = ostep.discriminator_losses(dd, ldr, hdr, y, training=True, new_stats={})
    print("   at OUR prediction: adv %.5f (oracle %.5f), disc_generated %.5f (oracle %.5f)" % (got["adv"], adv_o, got["disc_generated"],
                                                                                           float(dl["generated"])))
    assert abs(got["adv"] - adv_o) <= 2e-2 * abs(adv_o), (got["adv"], adv_o)
    assert abs(got["disc_generated"] - float(dl["generated"])) <= 2e-2 * abs(float(dl["generated"])), (got["disc_generated"], float(dl["generated"]))
    a, b = out["y_final_gamma"].cpu().double(), outs["y_final_gamma"].double()
    psnr = float(10 * torch.log10(b.abs().max() ** 2 / ((a - b) ** 2).mean()))
    cos, ratio = _cosine(tr, gg, gs)
    rows = _grad_errors(tr, gg, gs, zero_bias_tol=2e-4)
    big = sorted(rows, key=lambda r: -r[2])[:10]
    print("B = 32 bf16: psnr %.1f dB, gradient cosine %.5f, norm ratio %.4f" % (psnr, cos, ratio))
    print("ten largest tensors (rel max, rel rms, elements, name):", big)
    # norm ratio: the bf16 gradient's norm sits 4-5 % BELOW the oracle's on every build (0.9586 in round 4's and in this round's
    # first build, 0.9499 with the round-5 kernels: it moves with which Grad-CAM arg-max ties and ReLU masks the rounding flips) -
    # asserted at 7 %; the cosine (0.9957-0.9965 measured) is the assertion that says the direction is right
    assert psnr > 40.0 and cos > 0.99 and abs(ratio - 1.0) < 7e-2, (psnr, cos, ratio)
    # per-tensor relative rms error of the ten largest tensors, asserted at 2x what this build measures (VERDICT r4 item 3;
    # round-5 build, gpurun_out/r05a/pytest_s.txt: psnr 48.3 dB, cosine 0.99645, norm ratio 0.9586).  The error is bf16 operand
    # rounding (2^-9 per product term) through ~30 layers of backward pass and the discrete masks: largest where the gradient is
    # the small difference of many terms (the res blocks' kernels, the first Dense layer), smallest for the last Dense layer.
    # (second build of the round - paired decoders, one-launch norm backward, re-measured tiles: 0.114 / 0.043 / 0.057 / 0.069 / 0.127 /
    # 0.109 / 0.127 / 0.098 / 0.111 / 0.093: the table below holds the larger of the two measurements)
    measured = {"sun.fc1.kernel": 0.114, "sun.fc2.kernel": 0.0431, "gen.sun.d4.conv.kernel": 0.0568, "gen.sun.d3.conv.kernel": 0.0688,
                "gen.res.0.conv1.w": 0.1267, "gen.res.0.conv2.w": 0.1090, "gen.res.1.conv1.w": 0.1271, "gen.res.1.conv2.w": 0.0978,
                "gen.res.2.conv1.w": 0.1105, "gen.res.2.conv2.w": 0.0932}
    assert {r[3] for r in big} <= set(measured) | {"gen.res.%d.conv%d.w" % (i, j) for i in range(6) for j in (1, 2)}, big
    for _, rms, _, name in big:
        assert rms < 2.0 * measured.get(name, 0.127), "%s: relative rms error %.4f against the fp32 oracle, measured %.4f in round 5" % (
            name, rms, measured.get(name, 0.127))


def test_hires_train_step_b2_against_the_oracle(dev):
    """128x512 (configs[4]'s size; sun-pose outputs external), B = 2, BF16X3: the per-tensor checks of
    tests/test_hires_gpu.py's B = 1 case with a batch (BatchNorm statistics over two samples, reduce_max over the batch)."""
    from oracle import step as ostep
    import test_hires_gpu as TH
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    tr, (gen, _, dis, vgg) = TH._hires_trainer(dev, "BF16X3", False)
    ldr, hdr = TH._inputs(2, seed=45)
    cmf, gt, cams = TH._sun_inputs(2, seed=23)
    ext = [torch.from_numpy(cmf)] + [torch.from_numpy(c) for c in cams]
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(TH._tt(gen), None, TH._tt(dis), TH._tt(vgg), torch.from_numpy(ldr),
                                                              torch.from_numpy(hdr), torch.from_numpy(gt), sunpose_external=ext)
    d = lambda x: torch.from_numpy(x).to(dev)
    out = tr.step(d(ldr), d(hdr), d(gt), update=False, cmf=d(cmf), cams=[d(c) for c in cams])
    got = tr.loss_dict()
    for k, rk in LOSS_NAMES[:8]:
        assert abs(got[k] - losses[rk]) <= 2e-3 * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    assert_close(out["y_final_gamma"], outs["y_final_gamma"], 1e-3, "y_final_gamma 128x512, B = 2")
    rows = sorted(_grad_errors(tr, gg, {}), reverse=True)
    print("128x512 B = 2 worst gradient tensors:", rows[:6])
    assert rows[0][0] < 5e-2 and np.median([e for e, _, _, _ in rows]) < 3e-3, rows[:5]
    # the discriminator step at OUR prediction with TWO-sample BatchNorm statistics: the worst element keeps the 2e-2 that the
    # single-sample case (tests/test_hires_gpu.py, 5e-2 there: see its comment) cannot
    dr = {k: torch.from_numpy(v).clone().requires_grad_("moving" not in k) for k, v in dis.items()}
    dl = ostep.discriminator_losses(dr, torch.from_numpy(ldr), torch.from_numpy(hdr), out["y_final_lin"].cpu(), training=True, new_stats={})
    names = [k for k in dr if "moving" not in k]
    derr = sorted(((rel_max(tr.ds.g["dis." + k], v), rel_rms(tr.ds.g["dis." + k], v), k)
                   for k, v in zip(names, torch.autograd.grad(dl["total_disc_loss"], [dr[k] for k in names]))), reverse=True)
    print("128x512 B = 2 discriminator gradients (rel max, rel rms, name), worst:", derr[:4])
    assert derr[0][0] < 2e-2 and max(r for _, r, _ in derr) < 5e-3, \
        "discriminator gradients, 128x512 B = 2 (worst element, rms, tensor): %s" % (derr[:4],)
