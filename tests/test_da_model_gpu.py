"""The distortion-aware variants of the model the reference keeps commented out / unwired, end to end: sunposeLayer's
convolutions as distortion_aware_ops.conv2d (sunpose_net.py:11,16 - 7x7 on the first layer), distortion_aware_ops.deconv2d
(:272-542) in both decoders, the res blocks (generator.py:14,18) - forward (+ Grad-CAM through the distortion-aware
sun-pose net) and one training step against the oracle's autograd through the numpy restatement of the layer."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import step as ostep
from util import assert_close, rel_max, rel_rms

pytestmark = pytest.mark.gpu


def _nets():
    P = pkg("params")
    return [P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1), P.init_params(P.discriminator_spec(), 2),
            P.init_params(P.vgg_spec(), 3)]


_tt = lambda dd: {k: torch.from_numpy(v) for k, v in dd.items()}


@pytest.mark.parametrize("parts", ["sunpose", "decoders", "all"])
def test_forward_with_distortion_aware_parts(dev, parts):
    engine, K, synth = pkg("engine"), pkg("kernels"), pkg("synth")
    gen, sun, _, _ = _nets()
    ldr = torch.from_numpy(synth.make_batch(2, seed=77)["ldr"])
    ref = ostep.inference(_tt(gen), _tt(sun), ldr, distortion_aware=parts)
    plain = ostep.inference(_tt(gen), _tt(sun), ldr)
    assert float((plain["y_final_gamma"] - ref["y_final_gamma"]).abs().max()) > 1e-2          # the variant really differs
    nets = engine.Nets(gen, sun, device=dev, precise=True)
    out = engine.generator_forward(nets, ldr.to(dev), compute=K.BF16X3, distortion_aware=parts)
    assert_close(out["sunpose_cmf"], ref["sunpose_cmf"], 2e-3, "cmf")
    for k in (1, 2, 3):
        assert_close(out["actv_maps"][k - 1], ref["actv_maps"][k - 1], 1e-3, "A%d" % k)
        # Grad-CAM: a max-pool tie flips the routing of one gradient element (test_forward_gpu explains) - rms bound
        assert rel_rms(out["sun_cam%d" % k], ref["sun_cam%d" % k]) < 3e-2, k
    assert_close(out["y_final_gamma"], ref["y_final_gamma"], 2e-3, "y_final_gamma")
    out16 = engine.generator_forward(engine.Nets(gen, sun, device=dev, precise=False), ldr.to(dev), compute=K.BF16,
                                     distortion_aware=parts)
    y, r = out16["y_final_gamma"].double().cpu(), ref["y_final_gamma"].double()
    psnr = float(10 * torch.log10(r.abs().max() ** 2 / ((y - r) ** 2).mean()))
    assert psnr > 38.0, psnr


def test_layer_api_distortion_aware_models(dev):
    """sunpose_net.model(distortion_aware=True) / generator.model(distortion_aware="res,decoders") through the mirrors."""
    sunpose_net, generator, K, synth = pkg("sunpose_net"), pkg("generator"), pkg("kernels"), pkg("synth")
    gen, sun, _, _ = _nets()
    ldr = torch.from_numpy(synth.make_batch(2, seed=78)["ldr"])
    ref = ostep.inference(_tt(gen), _tt(sun), ldr, distortion_aware="all")
    sm = sunpose_net.model(weights=sun, device=dev, compute=K.BF16X3, distortion_aware=True)
    cmf, maps = sm.sunposeEstimation(ldr.to(dev))
    assert_close(cmf, ref["sunpose_cmf"], 2e-3, "cmf (layer API)")
    gm = generator.model(weights=gen, device=dev, compute=K.BF16X3, distortion_aware="res,decoders")
    res = gm.encode(ldr.to(dev))
    assert_close(res, ref["res_out"], 1e-3, "res_out (layer API)")
    sky = gm.sky_decode(res, ldr.to(dev))
    from oracle import networks as onet
    assert_close(sky, onet.gen_sky_decode(_tt(gen), ref["res_out"], ldr, "decoders"), 1e-3, "sky_decode (layer API)")


def test_train_step_all_distortion_aware(dev):
    trainer, K, synth = pkg("trainer"), pkg("kernels"), pkg("synth")
    nets = _nets()
    batch = synth.make_batch(2, seed=1234)
    ldr, hdr, gt = (torch.from_numpy(batch[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    losses, gg, gs, gd, sg, sd, outs = ostep.train_step_grads(*[_tt(n) for n in nets], ldr, hdr, gt, distortion_aware="all")
    tr = trainer.Trainer(*nets, device=dev, precise=True, compute=K.BF16X3, distortion_aware="all")
    out = tr.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    got = tr.loss_dict()
    for k in ("kl", "perceptual", "dog", "l1", "adv"):
        assert abs(got[k] - losses[k]) <= 2e-3 * abs(losses[k]) + 1e-6, (k, got[k], losses[k])
    assert_close(out["y_final_gamma"], outs["y_final_gamma"], 2e-3, "y_final_gamma (all distortion-aware)")
    worst = []
    for prefix, ref in (("gen.", gg), ("sun.", gs)):
        for k, v in ref.items():
            is_da = (k.startswith("sunlayer") or k.startswith("res.") or k[:7] in ("conv3_f", "conv2_f", "conv3_u", "conv2_u") or
                     k[:7] in ("norm3_f", "norm2_f", "norm3_u", "norm2_u"))
            if not is_da or k.endswith(".b") or k.endswith("bias_deconv2d"):
                continue      # (conv biases in front of InstanceNorm: exactly-zero gradients, see test_train_gpu)
            worst.append((rel_max(tr.gs.g[prefix + k], v), prefix + k))
    worst.sort(reverse=True)
    print("worst gradient errors of the distortion-aware layers:", worst[:6])
    assert len(worst) >= 6 * 2 + 4 * 3 + 12 * 3
    assert worst[0][0] < 5e-2 and np.median([e for e, _ in worst]) < 3e-3, worst[:6]
    for k in ("conv3_d.w", "conv1_d.w"):          # and what lies behind them in the backward chain
        assert rel_max(tr.gs.g["gen." + k], gg[k]) < 5e-2, k
    assert_close(tr.gs.g["sun.fc1.kernel"], gs["fc1.kernel"], 5e-3, "fc1 kernel gradient")
    tr.apply_gradients()
    assert torch.isfinite(tr.gs.flat).all()
    # bench mode: runs, stays finite, repeats bit for bit
    tr16 = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16, distortion_aware="all")
    w0g, w0d = tr16.gs.flat.clone(), tr16.ds.flat.clone()      # (a step moves the BatchNorm moving statistics)
    tr16.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    g1, d1 = tr16.gs.grad.clone(), tr16.ds.grad.clone()
    tr16.gs.flat.copy_(w0g); tr16.ds.flat.copy_(w0d)
    tr16.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=False)
    assert torch.equal(g1, tr16.gs.grad) and torch.equal(d1, tr16.ds.grad)
    tr16.step(ldr.to(dev), hdr.to(dev), gt.to(dev), update=True)
    assert all(np.isfinite(v) for v in tr16.loss_dict().values()) and torch.isfinite(tr16.gs.flat).all()


def test_sunpose_pretraining_step_distortion_aware(dev):
    """train_sun.py's step with the distortion-aware sun-pose net (sunpose_net.py:11,16): gradients of every layer against
    the oracle's autograd."""
    trainer, K, synth = pkg("trainer"), pkg("kernels"), pkg("synth")
    from oracle import networks as onet, tfsem as T
    sun = _nets()[1]
    batch = synth.make_batch(2, seed=4321)
    ldr, gt = torch.from_numpy(batch["ldr"]), torch.from_numpy(batch["sunpose_gt"])
    p = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sun.items()}
    cmf, _ = onet.sunpose_estimation(p, ldr, "sunpose")
    loss = T.kl_divergence(gt, cmf)
    names = list(p)
    ref = dict(zip(names, torch.autograd.grad(loss, [p[k] for k in names])))
    tr = trainer.SunPoseTrainer(sun, device=dev, precise=True, compute=K.BF16X3, distortion_aware=True)
    tr.step(ldr.to(dev), gt.to(dev), update=False, want_cams=False, dog_weight=0.0)
    assert abs(tr.loss_dict()["kl"] - float(loss.detach())) <= 2e-3 * abs(float(loss.detach()))
    keys = [k for k in ref if not (k.endswith(".b") and k.startswith("sunlayer"))]     # (biases before InstanceNorm: zero)
    worst = sorted(((rel_max(tr.gs.g["sun." + k], ref[k]), k) for k in keys), reverse=True)
    dot = sum(float((tr.gs.g["sun." + k].double().cpu() * ref[k].double()).sum()) for k in keys)
    n1 = sum(float((tr.gs.g["sun." + k].double() ** 2).sum()) for k in keys) ** 0.5
    n2 = sum(float((ref[k].double() ** 2).sum()) for k in keys) ** 0.5
    print("worst:", worst[:4], "cosine", dot / (n1 * n2))
    # the contraction error is fp32-class; single ReLU / max-pool mask flips move individual elements at the 1e-2 level
    # (test_train_gpu.test_sunpose_pretraining_step_matches_oracle picks a batch with a margin; this one does not)
    # (the layers' own backward kernels are held to 5e-2 / 3e-3 by test_train_step_all_distortion_aware; here: the plumbing)
    assert worst[0][0] < 1e-1 and np.median([e for e, _ in worst]) < 1.5e-2, worst[:4]
    assert dot / (n1 * n2) > 0.9998
