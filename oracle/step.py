"""Inference graph and training step restated on torch-CPU autograd.
TEST INFRASTRUCTURE; PARITY UNPINNED (see oracle/__init__.py).

Follows inference.py:81-115 (`generator_in_step`, inference) and train.py:239-415
(`generator_in_step`, `discriminator_in_step`, `train_step`).
"""
import torch

from . import networks as N
from . import tfsem as T

THRESHOLD = 0.12  # inference.py:36, train.py:247


def _alpha_mask(sky_pred_lin, thr=THRESHOLD):
    """inference.py:91-94 / train.py:258-261."""
    alpha = sky_pred_lin.max(dim=3).values
    alpha = torch.minimum(torch.ones_like(alpha), torch.clamp(alpha - 1.0 + thr, min=0.0) / thr)
    return alpha.unsqueeze(-1).repeat(1, 1, 1, 3)


def generator_graph(gen, sun, ldr, y_index=None, training=False, new_stats=None, distortion_aware=False,
                    sunpose_external=None):
    """The generator graph shared by inference.py:81-115 and train.py:239-299.

    y_index: None -> y_c = max_j cmf[b,j] (inference.py:98); LongTensor [B] ->
             y_c = cmf[b, y_index[b]] (train.py:265-267, argmax of sunpose_gt).
    Grad-CAM maps and the alpha mask are constants for gradients
    (train.py:257 `gen_tape.stop_recording()`).
    sunpose_external: (cmf [B,H*W], cam1 [B,H,W,1], cam2 [B,H/2,W/2,1], cam3 [B,H/4,W/4,1]) - the outputs of the
             sun-pose net + Grad-CAM as INPUTS of the graph (constants), `sun` unused: SURVEY.md section 8d's
             substitution for the 128x512 configuration, whose faithful sun-pose net has 12.9 G parameters.
    Returns a dict of every tensor the reference returns from generator_in_step.
    """
    b, h, w, _ = ldr.shape
    res_out = N.gen_encode(gen, ldr, distortion_aware=distortion_aware)   # generator.py:14,18 variant when set
    sky_pred_gamma = N.gen_sky_decode(gen, res_out, ldr, distortion_aware)
    sky_pred_lin = T.hdr_log_decompression(sky_pred_gamma)

    alpha_c3 = _alpha_mask(sky_pred_lin).detach()
    if sunpose_external is not None:
        cmf, cam1, cam2, cam3 = (t.detach() for t in sunpose_external)
        a1 = a2 = a3 = None
    else:
        cmf, (a1, a2, a3) = N.sunpose_estimation(sun, ldr, distortion_aware)
        if y_index is None:
            y_c = cmf.max(dim=1).values
        else:
            y_c = cmf.gather(1, y_index.view(-1, 1)).squeeze(1)
        cam1 = N.grad_cam_layer(y_c, a1).detach()
        cam2 = N.grad_cam_layer(y_c, a2).detach()
        cam3 = N.grad_cam_layer(y_c, a3).detach()
    sunpose_pred = cmf.reshape(-1, h, w, 1)

    sun_rad_lin, gamma, beta = N.gen_sun_rad_estimation(gen, ldr, cam1, cam2, cam3, sunpose_pred,
                                                        training, new_stats)
    sun_rad_gamma = T.hdr_log_compression(sun_rad_lin)
    sun_pred_gamma = N.gen_sun_decode(gen, res_out, sun_rad_gamma, distortion_aware)

    sky_pred_gamma = (1.0 - alpha_c3) * sky_pred_gamma
    sky_pred_lin = T.hdr_log_decompression(sky_pred_gamma)
    sun_pred_gamma = alpha_c3 * sun_pred_gamma
    sun_pred_lin = T.hdr_log_decompression(sun_pred_gamma)
    y_final_gamma = sky_pred_gamma + sun_pred_gamma  # generator.model.blending (generator.py:171-175)
    y_final_lin = T.hdr_log_decompression(y_final_gamma)
    return dict(y_final_lin=y_final_lin, y_final_gamma=y_final_gamma, sky_pred_lin=sky_pred_lin,
                sun_pred_lin=sun_pred_lin, gamma=gamma, beta=beta, alpha_c3=alpha_c3,
                sunpose_cmf=cmf, sunpose_pred=sunpose_pred, sun_cam1=cam1, sun_cam2=cam2,
                sun_cam3=cam3, sun_rad_lin=sun_rad_lin, res_out=res_out, actv_maps=(a1, a2, a3))


def inference(gen, sun, ldr, distortion_aware=False):
    """inference.py:81-119: returns y_final_lin [B,H,W,3] (BGR, linear radiance)."""
    req = {k: v.detach().clone().requires_grad_(True) for k, v in sun.items()}
    out = generator_graph(gen, req, ldr, y_index=None, training=False, distortion_aware=distortion_aware)
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}


def generator_losses(out, dis, vgg, ldr, hdr_t, sunpose_gt):
    """train.py:301-331.  Discriminator runs with training=False here (train.py:302)."""
    hdr_t_gamma = T.hdr_log_compression(hdr_t)
    d_fake = N.discriminator(dis, ldr, out["y_final_lin"], training=False)
    sun_loss = T.kl_divergence(sunpose_gt, out["sunpose_cmf"])
    pools_p = N.vgg16_pools(vgg, out["y_final_gamma"])
    pools_t = N.vgg16_pools(vgg, hdr_t_gamma)
    perceptual = sum((a - b).abs().mean() for a, b in zip(pools_p, pools_t))
    dog_p = T.dog(out["y_final_lin"])
    dog_t = T.dog(hdr_t)
    dog_loss = sum((a - b).abs().mean() for a, b in zip(dog_p, dog_t))
    l1 = (out["y_final_lin"] - hdr_t).abs().mean()
    adv = ((d_fake - 1.0) ** 2).mean()
    total = sun_loss + 1000.0 * dog_loss + adv + 10.0 * l1 + 0.01 * perceptual
    return dict(total_gen_loss=total, kl=sun_loss, perceptual=perceptual, dog=dog_loss, l1=l1, adv=adv)


def discriminator_losses(dis, ldr, hdr_t, y_final_lin, training, new_stats=None):
    """train.py:351-380.  Two calls on the same weights; with training=True both update the
    moving statistics in sequence (real first, then generated)."""
    d_real = N.discriminator(dis, ldr, hdr_t, training, new_stats)
    dis2 = dis
    if training and new_stats:
        dis2 = dict(dis)
        dis2.update(new_stats)
    d_fake = N.discriminator(dis2, ldr, y_final_lin, training, new_stats)
    real_loss = ((d_real - 1.0) ** 2).mean()
    gen_loss = (d_fake ** 2).mean()
    total = (gen_loss + real_loss) * 0.5
    return dict(total_disc_loss=total, real=real_loss, generated=gen_loss)


def train_step_grads(gen, sun, dis, vgg, ldr, hdr_t, sunpose_gt, distortion_aware=False, sunpose_external=None):
    """train.py:382-406 up to (not including) apply_gradients.

    Both tapes see the same pre-update weights.  Returns (losses, grads_gen, grads_sun,
    grads_dis, new_bn_stats_gen, new_bn_stats_dis, outputs).  `ldr` / `hdr_t` are already BGR.
    Trainable variable sets: everything in `gen` and `sun` except BN moving stats; same for `dis`.
    sunpose_external: see generator_graph - `sun` may be None then and grads_sun comes back empty; the KL term is a
    constant of the step (its value is still reported).
    """
    def trainable(d):
        return {k: v for k, v in d.items() if "moving_" not in k}

    gen_r = {k: v.detach().clone().requires_grad_("moving_" not in k) for k, v in gen.items()}
    sun_r = {} if sunpose_external is not None else {k: v.detach().clone().requires_grad_(True) for k, v in sun.items()}
    dis_r = {k: v.detach().clone().requires_grad_("moving_" not in k) for k, v in dis.items()}

    y_index = sunpose_gt.argmax(dim=1)
    stats_gen, stats_dis = {}, {}
    out = generator_graph(gen_r, sun_r, ldr, y_index=y_index, training=True, new_stats=stats_gen,
                          distortion_aware=distortion_aware, sunpose_external=sunpose_external)
    gl = generator_losses(out, dis_r, vgg, ldr, hdr_t, sunpose_gt)
    # train.py:391-396: y_final_lin recomputed from y_final_gamma, still on both tapes
    y_final_lin = T.hdr_log_decompression(out["y_final_gamma"])
    dl = discriminator_losses(dis_r, ldr, hdr_t, y_final_lin, training=True, new_stats=stats_dis)

    gvars = list(trainable(gen_r).items()) + [("sunpose/" + k, v) for k, v in sun_r.items()]
    ggrads = torch.autograd.grad(gl["total_gen_loss"], [v for _, v in gvars], retain_graph=True,
                                 allow_unused=True)
    dvars = list(trainable(dis_r).items())
    dgrads = torch.autograd.grad(dl["total_disc_loss"], [v for _, v in dvars], allow_unused=True)
    grads_gen, grads_sun = {}, {}
    for (k, v), g in zip(gvars, ggrads):
        g = torch.zeros_like(v) if g is None else g
        if k.startswith("sunpose/"):
            grads_sun[k[len("sunpose/"):]] = g
        else:
            grads_gen[k] = g
    grads_dis = {k: (torch.zeros_like(v) if g is None else g) for (k, v), g in zip(dvars, dgrads)}
    losses = {k: float(v.detach()) for k, v in {**gl, **dl}.items()}
    outs = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
    return losses, grads_gen, grads_sun, grads_dis, stats_gen, stats_dis, outs


def test_step(gen, sun, dis, vgg, ldr, hdr_t, sunpose_gt):
    """train.py:417-442 (+ the test branches of generator_in_step / discriminator_in_step): the validation step - the
    training graph with the ground-truth-bin Grad-CAM pick, every BatchNorm in inference mode, no tape, no update.
    Returns (losses, outputs)."""
    req = {k: v.detach().clone().requires_grad_(True) for k, v in sun.items()}     # Grad-CAM differentiates the cmf
    out = generator_graph(gen, req, ldr, y_index=sunpose_gt.argmax(dim=1), training=False)
    gl = generator_losses(out, dis, vgg, ldr, hdr_t, sunpose_gt)
    y_final_lin = T.hdr_log_decompression(out["y_final_gamma"])
    dl = discriminator_losses(dis, ldr, hdr_t, y_final_lin, training=False)
    losses = {k: float(v.detach()) for k, v in {**gl, **dl}.items()}
    return losses, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}


def sun_train_step_grads(sun, ldr, sunpose_gt, dog_weight=1.0):
    """train_sun.py:220-264 up to (not including) apply_gradients: sun-pose pre-training.
    loss = KLDivergence(gt, cmf) + sum_4 mean|DoG_i(cmf image) - DoG_i(gt image)|; the Grad-CAM maps are returned
    but do not enter the loss (computed under stop_recording).  `ldr` is already BGR.
    Returns (losses, grads, outputs)."""
    sun_r = {k: v.detach().clone().requires_grad_(True) for k, v in sun.items()}
    b, h, w, _ = ldr.shape
    cmf, maps = N.sunpose_estimation(sun_r, ldr)
    y_index = sunpose_gt.argmax(dim=1)
    y_c = cmf.gather(1, y_index.view(-1, 1)).squeeze(1)
    cams = [N.grad_cam_layer(y_c, a, create_graph=False).detach() for a in maps]
    kl = T.kl_divergence(sunpose_gt, cmf)
    pred, gt_img = cmf.view(b, h, w, 1), sunpose_gt.view(b, h, w, 1)
    dog = sum((a - c).abs().mean() for a, c in zip(T.dog(pred), T.dog(gt_img)))
    loss = kl + dog_weight * dog          # the reference uses weight 1 (train_sun.py:255); 0 isolates the KL path in tests
    names = list(sun_r.keys())
    grads = torch.autograd.grad(loss, [sun_r[k] for k in names], allow_unused=True)
    grads = {k: (torch.zeros_like(sun_r[k]) if g is None else g) for k, g in zip(names, grads)}
    return dict(kl=float(kl.detach()), dog=float(dog.detach()), sun_loss=float(loss.detach())), grads, \
        dict(sunpose_cmf=cmf.detach(), sun_cam1=cams[0], sun_cam2=cams[1], sun_cam3=cams[2])
