"""BASELINE configs[4] (128x512 panoramas) on one GPU: the generator encoder with plain and with distortion-aware res
blocks (distortion_aware_ops.py:5-270 in generator.py:14,18's commented-out variant), both decoders, the discriminator,
the VGG16 pools and the DoG / L1 losses against the oracle at B = 1-2 (BF16X3 tight, BF16 at its own level), plus
size-independent properties at the configuration's per-GPU batch of 8 (train.py:535-536 --imheight/--imwidth).
The 12.9 G-parameter sun-pose net of this image size is outside the measured configuration."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import networks as onet, tfsem as T
from util import assert_close, assert_close_bf16, rel_max, rel_rms

pytestmark = pytest.mark.gpu
H, W = 128, 512


def _tt(d):
    return {k: torch.from_numpy(v) for k, v in d.items()}


def _inputs(B, seed=31):
    rng = np.random.default_rng(seed)
    ys = np.linspace(0.2, 0.8, H, dtype=np.float32).reshape(1, H, 1, 1)
    ldr = np.clip(ys + 0.15 * rng.standard_normal((B, H, W, 3)).astype(np.float32), 0.0, 1.0)
    ldr = np.round(ldr * 255.0) / 255.0
    hdr = (ldr ** 2.2 * (1.0 + 3.0 * rng.random((B, H, W, 3)))).astype(np.float32)
    return ldr.astype(np.float32), hdr


@pytest.mark.parametrize("B", [1, 2])
def test_generator_encoder_and_decoders_128x512(dev, B):
    params, engine, K = pkg("params"), pkg("engine"), pkg("kernels")
    gen = params.init_params(params.generator_spec(H, W), 0)
    ldr, _ = _inputs(B)
    torch.set_num_threads(max(torch.get_num_threads(), 4))
    ref_res = onet.gen_encode(_tt(gen), torch.from_numpy(ldr))
    ref_sky = onet.gen_sky_decode(_tt(gen), ref_res, torch.from_numpy(ldr))
    rad = np.random.default_rng(3).random((B, H, W, 3)).astype(np.float32)
    ref_sun = onet.gen_sun_decode(_tt(gen), ref_res, torch.from_numpy(rad))
    nets = engine.Nets(gen, None, device=dev, precise=True, im_height=H, im_width=W)
    x = torch.from_numpy(ldr).to(dev)
    res = engine.encode(nets, x, K.BF16X3)
    assert tuple(res.shape) == (B, H // 4, W // 4, 128)
    assert_close(res, ref_res, 1e-3, "res_out 128x512")
    assert_close(engine.decode(nets, res, "f", x, K.BF16X3), ref_sky, 1e-3, "sky decoder 128x512")
    assert_close(engine.decode(nets, res, "u", torch.from_numpy(rad).to(dev), K.BF16X3), ref_sun, 1e-3, "sun decoder 128x512")
    # bench mode: single bf16 product (the 32x128 res maps do not fit the sample-resident kernel: generic launches)
    res16 = engine.encode(nets, x, K.BF16)
    print("bf16 res_out: rel max %.3e rms %.3e" % (rel_max(res16, ref_res), rel_rms(res16, ref_res)))
    assert rel_rms(res16, ref_res) < 2e-2 and rel_max(res16, ref_res) < 1e-1
    sky16 = engine.decode(nets, res16, "f", x, K.BF16)
    a, b = sky16.cpu().double(), ref_sky.double()
    psnr = float(10 * torch.log10(b.abs().max() ** 2 / ((a - b) ** 2).mean()))
    print("bf16 sky decoder PSNR vs oracle %.1f dB" % psnr)
    assert psnr > 40.0


def test_distortion_aware_res_stack_128x512(dev):
    """The res stack built from distortion_aware_ops.conv2d on the 32x128 quarter-resolution maps of a 128x512 image, B=1
    (oracle: numpy restatement of the gather, oracle/da_ops.py)."""
    params, engine, K = pkg("params"), pkg("engine"), pkg("kernels")
    gen = params.init_params(params.generator_spec(H, W), 0)
    ldr, _ = _inputs(1, seed=5)
    ref = onet.gen_encode(_tt(gen), torch.from_numpy(ldr), distortion_aware=True)
    nets = engine.Nets(gen, None, device=dev, precise=True, im_height=H, im_width=W)
    got = engine.encode(nets, torch.from_numpy(ldr).to(dev), K.BF16X3, distortion_aware=True)
    assert_close(got, ref, 2e-3, "distortion-aware res stack 128x512")
    got16 = engine.encode(nets, torch.from_numpy(ldr).to(dev), K.BF16, distortion_aware=True)
    assert rel_rms(got16, ref) < 2e-2


@pytest.mark.parametrize("B", [1, 2])
def test_discriminator_vgg_and_losses_128x512(dev, B):
    params, K = pkg("params"), pkg("kernels")
    disc_mod, vgg_mod = pkg("discriminator"), pkg("vgg16")
    ldr, hdr = _inputs(B, seed=8)
    dw = params.init_params(params.discriminator_spec(), 2)
    vw = params.init_params(params.vgg_spec(), 3)
    tl, th = torch.from_numpy(ldr), torch.from_numpy(hdr)
    ref_d = onet.discriminator(_tt(dw), tl, th, training=False)
    d = lambda t: t.to(dev)
    dis = disc_mod.model(im_height=H, im_width=W, device=dev, compute=K.BF16X3, weights=dw)
    got_d = dis([d(tl), d(th)], training=False)
    assert tuple(got_d.shape) == tuple(ref_d.shape) == (B, H // 8 - 3, W // 8 - 3, 1)
    assert_close(got_d, ref_d, 1e-3, "discriminator logits 128x512")
    gam = T.hdr_log_compression(th)
    ref_p = onet.vgg16_pools(_tt(vw), gam)
    vgg = vgg_mod.Vgg16(weights=vw, device=dev, compute=K.BF16X3)
    for i, (a, b) in enumerate(zip(vgg(K.tonemap(d(th), False)), ref_p)):
        assert_close(a, b, 1e-3, "VGG pool%d 128x512" % (i + 1))
    vgg16 = vgg_mod.Vgg16(weights=vw, device=dev, compute=K.BF16)
    for i, (a, b) in enumerate(zip(vgg16(K.tonemap(d(th), False)), ref_p)):
        assert rel_rms(a, b) < 3e-2, i
    # losses: L1 and the DoG pyramid (tf_utils.py:61-73) between two 128x512 images
    y = (th * (1.0 + 0.1 * torch.from_numpy(np.random.default_rng(2).standard_normal(hdr.shape).astype(np.float32)))).clamp_min(0)
    slot = torch.zeros(2, device=dev)
    K.l1(d(y), d(th), 1.0, 0.0, slot[0:1])
    K.dog_loss(d(y), d(th), 1.0, slot[1:2], torch.zeros_like(d(y)))
    ref_l1 = float((y - th).abs().mean())
    ref_dog = float(sum((p - q).abs().mean() for p, q in zip(T.dog(y), T.dog(th))))
    got = slot.tolist()
    assert abs(got[0] - ref_l1) <= 1e-4 * ref_l1 and abs(got[1] - ref_dog) <= 1e-3 * ref_dog, (got, ref_l1, ref_dog)


def test_hires_batch8_properties(dev):
    """The configuration's per-GPU batch (8 x 128x512), where the CPU oracle takes minutes: batch independence of the
    encoder / decoder (InstanceNorm couples nothing across samples), plain vs distortion-aware stack on zero offsets, and
    hipGraph replay == eager launches bit for bit."""
    params, engine, K = pkg("params"), pkg("engine"), pkg("kernels")
    B = 8
    gen = params.init_params(params.generator_spec(H, W), 0)
    nets = engine.Nets(gen, None, device=dev, precise=True, im_height=H, im_width=W)
    ldr = torch.from_numpy(_inputs(B, seed=12)[0]).to(dev)
    full = engine.encode(nets, ldr, K.BF16X3)
    sky = engine.decode(nets, full, "f", ldr, K.BF16X3)
    for lo in (0, 5):
        part = engine.encode(nets, ldr[lo:lo + 2].contiguous(), K.BF16X3)
        assert_close(part, full[lo:lo + 2], 3e-4, "encoder batch independence [%d:%d]" % (lo, lo + 2))
        assert_close(engine.decode(nets, part, "f", ldr[lo:lo + 2].contiguous(), K.BF16X3), sky[lo:lo + 2], 3e-3, "decoder")
    # distortion-aware conv with an all-zero offset table == the plain SAME conv (same weights), at this map size
    x = torch.randn(B, H // 4, W // 4, 128, device=dev)
    pw = nets.pk["gen.res.0.conv1"]
    zero = torch.zeros(H // 4, 9, 2, device=dev)
    a = K.da_conv2d(x, pw, nets.gen["res.0.conv1.b"], zero, K.BF16X3)
    b, _ = K.conv2d(x, pw, nets.gen["res.0.conv1.b"], compute=K.BF16X3)
    assert_close(a, b, 2e-4, "DA conv with zero offsets == conv")
    # replay == eager (bench mode)
    nets16 = engine.Nets(gen, None, device=dev, precise=False, im_height=H, im_width=W)
    fn = lambda: engine.decode(nets16, engine.encode(nets16, ldr, K.BF16, distortion_aware=True), "f", ldr, K.BF16)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager = fn().clone()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with pkg("kernels").no_gc(), torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(out, eager) and torch.isfinite(out).all()


# ---------------------------------------------------------------------------------------------------------------------
# configs[4] as a TRAINING step: Trainer(im_height=128, im_width=512, sunpose="external") - train.py:382-415 with the
# 12.9 G-parameter sun-pose net substituted by its outputs (cmf + the three Grad-CAM maps are inputs of the step,
# SURVEY.md section 8d); oracle/step.train_step_grads(sunpose_external=...) is the same substitution on the CPU.
# ---------------------------------------------------------------------------------------------------------------------
def _sun_inputs(B, seed=17):
    rng = np.random.default_rng(seed)
    soft = lambda z: (lambda e: (e / e.sum(1, keepdims=True)).astype(np.float32))(np.exp(z - z.max(1, keepdims=True)))
    cmf = soft(3.0 * rng.standard_normal((B, H * W)))
    gt = soft(3.0 * rng.standard_normal((B, H * W)))
    cams = [np.maximum(rng.standard_normal((B, H >> i, W >> i, 1)), 0.0).astype(np.float32) for i in range(3)]
    return cmf, gt, cams


def _hires_trainer(dev, mode, da=False):
    params, trainer, K = pkg("params"), pkg("trainer"), pkg("kernels")
    nets = (params.init_params(params.generator_spec(H, W), 0), None, params.init_params(params.discriminator_spec(), 2),
            params.init_params(params.vgg_spec(), 3))
    tr = trainer.Trainer(*nets, device=dev, precise=(mode == "BF16X3"), compute=getattr(K, mode), im_height=H, im_width=W,
                         distortion_aware=da, sunpose="external")
    return tr, nets


@pytest.mark.parametrize("da", [False, "res,decoders"])
def test_hires_training_step_gradients_match_oracle(dev, da):
    """One 128x512 training step, B = 1, fp32-class contractions, plain and with distortion-aware res blocks + decoders
    (distortion_aware_ops.py:5-542 where generator.py:14,18 / the decoders would use them): every loss term, the
    generator's gradients and - at OUR prediction, see test_train_gpu - the discriminator's, against the oracle's autograd."""
    from oracle import step as ostep
    torch.set_num_threads(max(torch.get_num_threads(), 4))
    tr, (gen, _, dis, vgg) = _hires_trainer(dev, "BF16X3", da)
    assert tr.ext_sun and tr.fc1 is None and not any(k.startswith("sun.") for k in tr.gs.w)
    ldr, hdr = _inputs(1, seed=41)
    cmf, gt, cams = _sun_inputs(1)
    ext = [torch.from_numpy(cmf)] + [torch.from_numpy(c) for c in cams]
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(_tt(gen), None, _tt(dis), _tt(vgg), torch.from_numpy(ldr),
                                                              torch.from_numpy(hdr), torch.from_numpy(gt), distortion_aware=da,
                                                              sunpose_external=ext)
    assert not gs
    d = lambda a: torch.from_numpy(a).to(dev)
    out = tr.step(d(ldr), d(hdr), d(gt), update=False, cmf=d(cmf), cams=[d(c) for c in cams])
    got = tr.loss_dict()
    for k, rk in (("kl", "kl"), ("perceptual", "perceptual"), ("dog", "dog"), ("l1", "l1"), ("adv", "adv"),
                  ("disc_generated", "generated"), ("disc_real", "real"), ("total_gen_loss", "total_gen_loss")):
        assert abs(got[k] - losses[rk]) <= 2e-3 * abs(losses[rk]) + 1e-6, (k, got[k], losses[rk])
    assert_close(out["y_final_gamma"], outs["y_final_gamma"], 1e-3, "y_final_gamma 128x512 (training mode)")
    assert_close(out["sun_rad_lin"], outs["sun_rad_lin"], 2e-3, "sun_rad_lin 128x512")
    worst = []
    for k, v in gg.items():
        g = tr.gs.g["gen." + k]
        is_bias = k.endswith(".b") or k.endswith("bias_deconv2d")
        if is_bias and not k.startswith("conv1_f") and not k.startswith("conv1_u"):
            wk = k[:-2] + ".w" if k.endswith(".b") else k.replace("bias_deconv2d", "kernel_deconv2d")
            scale = float(gg[wk].abs().max()) * float(np.prod(gg[wk].shape[:3]))
            assert float(g.abs().max()) <= 1e-4 * scale, (k, float(g.abs().max()), scale)     # exactly zero in exact arithmetic
            continue
        worst.append((rel_max(g, v), k))
    worst.sort(reverse=True)
    print("128x512 da=%s worst generator gradient errors:" % (da,), worst[:6])
    assert worst[0][0] < 5e-2 and np.median([e for e, _ in worst]) < 3e-3, worst[:5]
    for k, v in sg.items():
        assert_close(tr.gs.w["gen." + k], v, 1e-4, "gen " + k)
    dr = {k: torch.from_numpy(v).clone().requires_grad_("moving" not in k) for k, v in dis.items()}
    stats = {}
    dl = ostep.discriminator_losses(dr, torch.from_numpy(ldr), torch.from_numpy(hdr), out["y_final_lin"].cpu(), training=True,
                                    new_stats=stats)
    names = [k for k in dr if "moving" not in k]
    # (3e-4 at 32x128, B = 2; here 16x the pixels go through the batch-statistics BatchNorms of a SINGLE sample and the
    # HDR peaks through three LeakyReLU stages: the worst ELEMENT of d1 / d2 sits at 3e-3 .. 3e-2 of the tensor's maximum
    # depending on the build (any change of a summation order upstream moves it: round 4's compiler flag did), every
    # tensor's rms error stays below 5e-3)
    derr = sorted(((rel_max(tr.ds.g["dis." + k], v), rel_rms(tr.ds.g["dis." + k], v), k)
                   for k, v in zip(names, torch.autograd.grad(dl["total_disc_loss"], [dr[k] for k in names]))), reverse=True)
    print("128x512 da=%s discriminator gradients (rel max, rel rms, name), worst:" % (da,), derr[:4])
    # What the two bounds protect.  rms < 5e-3 per tensor is the assertion that says "the gradient is right" (measured 2e-4 ..
    # 2e-3 on every build).  The worst-ELEMENT bound is 5e-2 here and 2e-2 in the B = 2 twin
    # (test_fullsize_gpu.py::test_hires_train_step_b2_against_the_oracle) because of what a SINGLE-sample BatchNorm does: with
    # B = 1 the batch statistics of d2 / d3 / d4 run over one image, the backward's two means (mean dy, mean dy * xhat) are sums
    # over that one image's pixels, and a channel whose single-image variance is small multiplies the fp32-class rounding of
    # its inputs by rstd - the element that carries the maximum error is always in such a channel of d1 / d2's kernel gradient
    # (round 3 build: 1.2e-2; round 4 build with -fno-slp-vectorize, which changed the summation order inside bn_bwd_reduce: d2
    # kernel 2.999e-2, gpurun_out/r04d/pytest.txt; other tensors 3e-3 .. 9e-3).  With two samples the statistics average over
    # both and the same element class stays below 2e-2.  The measured values of THIS build are in the assertion message.
    assert derr[0][0] < 5e-2 and max(r for _, r, _ in derr) < 5e-3, \
        "discriminator gradients, 128x512 B = 1 (worst element, rms, tensor): %s" % (derr[:4],)


@pytest.mark.parametrize("da", [False, "res,decoders"])
def test_hires_training_step_b8_replay_equals_eager(dev, da):
    """The configuration's per-GPU batch of 8 in the bench mode (single bf16 product): three captured replays equal three
    eager steps from the same state (weights of both optimizers, BatchNorm moving statistics, every loss term), the
    weights move, and gradient-only replays from a restored state are bit-reproducible."""
    B = 8
    ldr, hdr = _inputs(B, seed=43)
    cmf, gt, cams = _sun_inputs(B, seed=19)
    d = lambda a: torch.from_numpy(a).to(dev)
    args = (d(ldr), d(hdr), d(gt))
    kw = dict(cmf=d(cmf), cams=[d(c) for c in cams])
    te, _ = _hires_trainer(dev, "BF16", da)
    tc, _ = _hires_trainer(dev, "BF16", da)
    w0 = te.gs.flat.clone()
    tc.capture(*args, **kw)
    for it in range(3):
        te.step(*args, update=True, **kw)
        tc.replay(update=True)
        torch.cuda.synchronize()
        for name, a, b in (("generator", te.gs.flat, tc.gs.flat), ("discriminator", te.ds.flat, tc.ds.flat)):
            assert torch.isfinite(b).all(), (it, name)
            assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()), (it, name)
        le, lc = te.losses.double(), tc.losses.double()
        assert float((le - lc).abs().max()) <= 1e-4 * float(le.abs().max()), (it, te.losses.tolist(), tc.losses.tolist())
    assert float((tc.gs.flat[:tc.gs.ntrain] - w0[:tc.gs.ntrain]).abs().max()) > 1e-5
    wg, wd = tc.gs.flat.clone(), tc.ds.flat.clone()
    ref = None
    for it in range(2):
        tc.gs.flat.copy_(wg); tc.ds.flat.copy_(wd)
        tc.replay(update=False)
        torch.cuda.synchronize()
        snap = (tc.gs.grad.clone(), tc.ds.grad.clone())
        assert torch.isfinite(snap[0]).all() and torch.isfinite(snap[1]).all()
        if ref is not None:
            assert torch.equal(ref[0], snap[0]) and torch.equal(ref[1], snap[1])
        ref = snap


def test_external_sunpose_inputs_are_checked(dev):
    tr, _ = _hires_trainer(dev, "BF16")
    ldr, hdr = _inputs(1)
    cmf, gt, cams = _sun_inputs(1)
    d = lambda a: torch.from_numpy(a).to(dev)
    with pytest.raises(ValueError):
        tr.step(d(ldr), d(hdr), d(gt))                                   # the substituted outputs are mandatory
    with pytest.raises(ValueError):
        tr.step(d(ldr), d(hdr), d(gt), cmf=d(cmf), cams=[d(cams[0])] * 3)   # wrong map sizes
