"""CPU tests (no GPU): pin the oracle.  (1) against the committed golden fixtures, (2) against a second,
independent numpy restatement of every risky op (TF SAME padding, half-pixel bilinear resize, IN, reflect-pad
Gaussian), (3) closed-form identities (SURVEY.md section 8c "what pins the build instead")."""
import math
import os

import numpy as np
import torch

from conftest import pkg
from oracle import da_ops, networks as N, step, tfsem as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}


def close(a, b, tol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-30), np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


# ---- independent numpy restatements --------------------------------------------------------------------
def conv2d_loops(x, w, b, stride, same):
    """tf.nn.conv2d semantics with explicit loops (out = ceil(in/s); pad_before = total//2)."""
    B, H, W, C = x.shape
    kh, kw, _, F = w.shape
    if same:
        Ho, Wo = -(-H // stride), -(-W // stride)
        ph = max((Ho - 1) * stride + kh - H, 0); pw = max((Wo - 1) * stride + kw - W, 0)
        pt, pl = ph // 2, pw // 2
    else:
        Ho, Wo = (H - kh) // stride + 1, (W - kw) // stride + 1
        pt = pl = 0
    y = np.zeros((B, Ho, Wo, F), np.float64)
    for oy in range(Ho):
        for ox in range(Wo):
            for ky in range(kh):
                for kx in range(kw):
                    iy, ix = oy * stride - pt + ky, ox * stride - pl + kx
                    if 0 <= iy < H and 0 <= ix < W:
                        y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) @ w[ky, kx].astype(np.float64)
    return y + b


def test_ops_against_golden_and_loops():
    g = np.load(os.path.join(GOLD, "ops_small.npz"))
    x, w3, w4, b = g["x"], g["w3"], g["w4"], g["b"]
    xt = torch.from_numpy(x)
    for key, w, s, same in (("conv3_s1_same", w3, 1, True), ("conv3_s2_same", w3, 2, True), ("conv4_s2_same", w4, 2, True),
                            ("conv4_s1_same", w4, 1, True), ("conv4_s1_valid", w4, 1, False)):
        y = T.conv2d(xt, torch.from_numpy(w), torch.from_numpy(b), s, "SAME" if same else "VALID").numpy()
        close(y, g[key], 1e-5)
        close(y, conv2d_loops(x, w, b, s, same), 1e-5)
    # pads the reference's layers rely on: k3 s2 even -> (0,1); k4 s2 -> (1,1); k4 s1 -> (1,2); k7 s1 -> (3,3)
    assert T.same_pad(32, 3, 2) == (0, 1) and T.same_pad(32, 4, 2) == (1, 1)
    assert T.same_pad(16, 4, 1) == (1, 2) and T.same_pad(32, 7, 1) == (3, 3)
    # resize: torch interpolate vs explicit half-pixel numpy
    r = T.resize_bilinear(xt, 18, 28).numpy()
    close(r, g["resize_2x"], 1e-6)
    close(r, da_ops.resize_bilinear(x, 18, 28), 1e-6)
    # instance norm: vs golden, vs explicit numpy, and its moment identities
    gam, bet = g["in_gamma"], g["in_beta"]
    y = T.instance_norm(xt, torch.from_numpy(gam), torch.from_numpy(bet)).numpy()
    close(y, g["instance_norm"], 1e-5)
    mu = x.mean(axis=(1, 2), keepdims=True); var = x.var(axis=(1, 2), keepdims=True)
    close(y, (x - mu) / np.sqrt(var + 1e-3) * gam + bet, 1e-5)
    close(y.mean(axis=(1, 2)), np.broadcast_to(bet, (2, 5)), 1e-4)
    close(y.var(axis=(1, 2)), (gam ** 2) * var[:, 0, 0] / (var[:, 0, 0] + 1e-3), 1e-4)
    # DoG vs golden and vs an explicit reflect-pad numpy blur
    img = g["dog_in"]
    dogs = [t.numpy() for t in T.dog(torch.from_numpy(img))]
    for i in range(4):
        close(dogs[i], g["dog_%d" % i], 1e-5)

    def blur(a, sigma):
        k = np.exp(-np.array([1.0, 0.0, 1.0]) / (2 * sigma * sigma)); k /= k.sum()
        p = np.pad(a, ((0, 0), (1, 1), (1, 1), (0, 0)), mode="reflect")
        out = np.zeros_like(a, dtype=np.float64)
        for dy in range(3):
            for dx in range(3):
                out += k[dy] * k[dx] * p[:, dy:dy + a.shape[1], dx:dx + a.shape[2], :]
        return out
    base = blur(da_ops.resize_bilinear(img, 12, 20).astype(np.float64), T.DOG_SIGMA_BASE)
    close(dogs[0], blur(base, T.DOG_SIGMAS_2[0]) - blur(base, T.DOG_SIGMAS_1[0]), 1e-4)
    # KL
    p, q = torch.from_numpy(g["kl_p"]), torch.from_numpy(g["kl_q"])
    close(float(T.kl_divergence(p, q)), float(g["kl"]), 1e-6)
    close(float(T.kl_divergence(p, q)), float((g["kl_p"] * np.log(g["kl_p"] / g["kl_q"])).sum(1).mean()), 1e-5)


def test_closed_form_identities():
    x = torch.rand(1000) * 50
    close(T.hdr_log_decompression(T.hdr_log_compression(x)).numpy(), x.numpy(), 1e-5)
    close(float(T.hdr_log_compression(torch.tensor(1.0))), 1.0, 1e-6)   # log(11)/log(11)
    # RMSprop one step closed form from zero state: w - lr*g/(sqrt(0.1 g^2)+eps)
    w, gr = torch.tensor([1.0, -2.0]), torch.tensor([0.5, -4.0])
    w2, ms = T.rmsprop_update(w, gr, torch.zeros(2), 1e-4)
    close(ms.numpy(), 0.1 * gr.numpy() ** 2, 1e-6)
    close(w2.numpy(), w.numpy() - 1e-4 * gr.numpy() / (np.sqrt(0.1) * np.abs(gr.numpy()) + 1e-7), 1e-6)
    # LSGAN on a hand case
    d = torch.tensor([[0.5, 1.5]])
    assert abs(float(((d - 1) ** 2).mean()) - 0.25) < 1e-7
    # alpha mask: 0 below 0.88, 1 above 1.0 (inference.py:91-94)
    sky = torch.tensor([0.5, 0.88, 0.94, 1.0, 3.0]).view(1, 1, 5, 1).repeat(1, 1, 1, 3)
    a = step._alpha_mask(sky)[0, 0, :, 0].numpy()
    close(a, np.array([0, 0, 0.5, 1, 1], np.float32), 1e-5)
    # Dirac-delta head at x == 1: gamma / (beta*sqrt(pi) + 1e-5)
    params = pkg("params")
    gen = tt(params.init_params(params.generator_spec(), 0))
    plz = torch.rand(2, 32, 128, 6)
    rad, gam, bet = N.sun_rad_net(gen, torch.ones(2, 32, 128, 1), plz, training=False)
    close(rad[:, 0, 0, 0].numpy(), (gam / (bet * math.sqrt(math.pi) + 1e-5)).view(-1).numpy(), 1e-5)
    # softmax / vMF targets sum to one
    synth = pkg("synth")
    b = synth.make_batch(3, seed=5)
    close(b["sunpose_gt"].sum(1), np.ones(3), 1e-5)
    assert b["ldr"].min() >= 0 and b["ldr"].max() <= 1 and np.allclose(b["ldr"] * 255, np.round(b["ldr"] * 255), atol=1e-4)


def test_da_conv_identities_and_golden():
    g = np.load(os.path.join(GOLD, "ops_small.npz"))
    off = da_ops.distortion(8, 32)
    close(off, g["da_offsets_h8_w32"], 1e-7)
    # SURVEY.md section 8c appendix sanity values (float32 evaluation, row 0 takes the `x < 0` branch)
    close(off[0, :, 1], [-1.2395, 0, -62.7605, -1, 0, -63, -0.8373, 0, -63.1627], 2e-5)
    close(off[1, :, 0], [-0.2220] * 3 + [0] * 3 + [0.1814] * 3, 3e-4)
    close(off[7, :, 1], [-1.0393, 0, 1.0393, -1, 0, 1, -0.9635, 0, 0.9635], 1e-4)
    y = da_ops.da_conv2d(g["da_x"], g["da_k"], g["da_b"], off)
    close(y, g["da_conv"], 1e-5)
    # zero offsets == SAME stride-1 conv with the kernel reshaped [k,k,C,F]
    y0 = da_ops.da_conv2d(g["da_x"], g["da_k"], g["da_b"], np.zeros_like(off))
    ref = T.conv2d(torch.from_numpy(g["da_x"]), torch.from_numpy(g["da_k"].reshape(3, 3, 4, 6)), torch.from_numpy(g["da_b"]))
    close(y0, ref.numpy(), 1e-5)


def test_forward_against_golden():
    params, synth = pkg("params"), pkg("synth")
    gen = params.init_params(params.generator_spec(), 0)
    sun = params.init_params(params.sunpose_spec(), 1)
    assert params.count_params(params.generator_spec()) == 4891912
    assert params.count_params(params.sunpose_spec()) == 50672544
    assert params.count_params(params.discriminator_spec()) == 2768641
    batch = synth.make_batch(2, seed=1234)
    out = step.inference(tt(gen), tt(sun), torch.from_numpy(batch["ldr"]))
    g = np.load(os.path.join(GOLD, "forward_b2_seed1234.npz"))
    for k in ("y_final_gamma", "y_final_lin", "sunpose_cmf", "gamma", "beta", "sun_rad_lin", "sun_cam3"):
        close(out[k].numpy(), g[k], 2e-4)
    # Grad-CAM maps 1/2 are ill-conditioned w.r.t. last-bit noise (max-pool ties): looser
    for k in ("sun_cam1", "sun_cam2"):
        close(out[k].numpy(), g[k], 5e-2)


def test_gradcam_matches_finite_difference():
    """grad_cam.layer restatement: d y_c / d A_k from autograd vs central finite differences in float64."""
    torch.manual_seed(0)
    params = pkg("params")
    sun = {k: torch.from_numpy(v).double() for k, v in params.init_params(params.sunpose_spec(8, 32), 1).items()}
    x = torch.rand(1, 8, 32, 3, dtype=torch.float64)

    def tail(a3):  # from the third activation map to y_c
        flat = T.flatten_nhwc(T.maxpool2x2(a3))
        f1 = torch.relu(T.dense(flat, sun["fc1.kernel"], sun["fc1.bias"]))
        f2 = torch.relu(T.dense(f1, sun["fc2.kernel"], sun["fc2.bias"]))
        return torch.softmax(f2, -1).max(dim=1).values.sum()
    a1 = N._sunpose_layer(sun, "sunlayer1", x)
    a2 = N._sunpose_layer(sun, "sunlayer2", T.maxpool2x2(a1))
    a3 = N._sunpose_layer(sun, "sunlayer3", T.maxpool2x2(a2)).detach().requires_grad_(True)
    (gr,) = torch.autograd.grad(tail(a3), a3)
    idx = [(0, 0, 1, 5), (0, 1, 3, 77), (0, 0, 0, 127)]
    for i in idx:
        e = torch.zeros_like(a3); e[i] = 1e-6
        fd = (tail(a3.detach() + e) - tail(a3.detach() - e)) / 2e-6
        assert abs(float(fd) - float(gr[i])) <= 1e-5 * max(1e-12, float(gr.abs().max()))


def test_adam_update_closed_form():
    """Keras OptimizerV2 Adam, first step from zero slots: w1 = w0 - lr*sqrt(1-b2)/(1-b1) * (1-b1) g / (sqrt((1-b2) g^2) + eps)."""
    import torch
    from oracle import tfsem as T
    w0, g = torch.tensor([1.0, -2.0, 0.5]), torch.tensor([0.3, -4.0, 1e-3])
    w1, m1, v1 = T.adam_update(w0, g, torch.zeros(3), torch.zeros(3), 1e-3, 1)
    b1, b2, eps = 0.9, 0.999, 1e-7
    expect = w0 - 1e-3 * (1 - b2) ** 0.5 / (1 - b1) * ((1 - b1) * g) / (((1 - b2) * g * g).sqrt() + eps)
    # OptimizerV2 forms 1-beta in float32: float32(1) - float32(0.999) differs from 0.001 by 1.3e-5 relative
    assert torch.allclose(w1, expect, rtol=3e-5, atol=0)
    assert torch.allclose(m1, (1 - b1) * g, rtol=1e-6) and torch.allclose(v1, (1 - b2) * g * g, rtol=3e-5)
    # for |g| >> eps the first step has magnitude ~lr regardless of the gradient scale
    assert abs(float((w0 - w1)[1]) - (-1e-3)) < 1e-6


def test_conv2d_transpose_oracle_against_explicit_construction():
    """tf.nn.conv2d_transpose, SAME, stride 2, k=3: TF pads the FORWARD conv (0,1) => the transpose is a full
    conv_transpose2d cropped at [0 : 2*in]; stride 1 is a SAME conv with the flipped, channel-transposed filter."""
    import torch
    import torch.nn.functional as F
    from oracle import tfsem as T
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.standard_normal((2, 5, 7, 6)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((3, 3, 4, 6)).astype(np.float32))       # [kh,kw,Cout,Cin]
    b = torch.from_numpy(rng.standard_normal(4).astype(np.float32))
    got = T.conv2d_transpose(x, w, b, 10, 14, 2)
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=2)   # weight [Cin, Cout, kh, kw]
    ref = full[:, :, 0:10, 0:14].permute(0, 2, 3, 1) + b
    assert torch.allclose(got, ref, atol=1e-5)
    got1 = T.conv2d_transpose(x, w, None, 5, 7, 1)
    ref1 = T.conv2d(x, w.flip(0, 1).permute(0, 1, 3, 2).contiguous(), None, 1, "SAME")
    assert torch.allclose(got1, ref1, atol=1e-5)


def test_preprocessing_oracle_known_answers():
    """train.py:54-94 restatement: no noise + identity response curve => ldr = round(255*clip(hdr*t))/255; the response
    LUT is interpolated linearly and its top index is clipped; the vMF target sums to 1 and peaks at the sun's bin."""
    from oracle import preproc
    rng = np.random.default_rng(2)
    hdr = (rng.random((2, 4, 8, 3)) * 3).astype(np.float32)
    t = np.array([0.5, 2.0], np.float32)
    z3 = np.zeros((2, 3), np.float32); zn = np.zeros_like(hdr)
    ident = np.tile(np.linspace(0, 1, 1024, dtype=np.float32)[None], (2, 1))
    hdr_t, ldr = preproc.preprocessing(hdr, t, z3, z3, zn, zn, ident)
    assert np.allclose(hdr_t, hdr * t.reshape(2, 1, 1, 1))
    assert np.abs(ldr - np.round(255 * np.clip(hdr_t, 0, 1)) / 255).max() <= 1.0 / 255 + 1e-6
    lut = np.array([[0.0, 1.0, 4.0]], np.float32)                      # k = 3: positions 0, 1, 2
    assert np.allclose(preproc.apply_rf(np.array([[0.0, 0.25, 0.5, 0.75, 1.0]], np.float32), lut), [[0, 0.5, 1, 2.5, 4]])
    pm = preproc.vmf(63.0, 10.0, 32, 128)
    assert abs(float(pm.sum()) - 1.0) < 1e-5
    row, col = divmod(int(pm.argmax()), 128)
    assert abs(col - 63) <= 1 and abs(row - (32 - 10)) <= 1    # azimuth 63 -> column 63; elevation counts rows from the bottom
