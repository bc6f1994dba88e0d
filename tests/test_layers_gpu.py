"""GPU parity of the reference-interface mirrors (ops / generator / sunpose_net / sunrad_net / discriminator / vgg16 /
grad_cam / tf_utils modules of the package) against the oracle.  The inference test is written the way
inference.py:81-115 composes the model methods."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import networks as onet, step as ostep, tfsem as T
from util import TOL_X3, TOL_F32, assert_close, assert_close_bf16, rel_rms

pytestmark = pytest.mark.gpu


def _tt(d):
    return {k: torch.from_numpy(v) for k, v in d.items()}


def test_ops_layers(dev):
    ops, K = pkg("ops"), pkg("kernels")
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 16, 32, 32)).astype(np.float32))
    for stride, k, pad in ((1, 3, "SAME"), (2, 3, "SAME"), (1, 7, "SAME"), (1, 4, "VALID")):
        layer = ops.conv2d(output_channels=64, strides=stride, k_h=k, k_w=k, padding=pad, seed=3, compute=K.BF16X3)
        y = layer(x.to(dev))
        assert tuple(layer.w.shape) == (k, k, 32, 64) and tuple(layer.b.shape) == (64,)
        assert float(layer.b.abs().max()) == 0.0                        # bias_initializer='zeros'
        lim = np.sqrt(6.0 / (k * k * 32 + k * k * 64))                  # glorot_uniform
        assert float(layer.w.abs().max()) <= lim
        layer.assign(layer.w, torch.from_numpy(rng.standard_normal(64).astype(np.float32)))
        y = layer(x.to(dev))
        assert_close(y, T.conv2d(x, layer.w.cpu(), layer.b.cpu(), stride, pad), TOL_X3, "ops.conv2d %s" % ((stride, k, pad),))
    de = ops.deconv2d(output_channels=16, output_imshape=[32, 64], k_h=3, k_w=3, seed=4, compute=K.BF16X3)
    y = de(x.to(dev))
    assert sorted(de.variables) == ["bias_deconv2d", "kernel_deconv2d"]
    assert_close(y, T.deconv2d_resize(x, de.kernel_deconv2d.cpu(), de.bias_deconv2d.cpu(), 32, 64), TOL_X3, "ops.deconv2d")
    for k, pad, oshape in ((3, "SAME", [32, 64]), (4, "SAME", [32, 64]), (3, "SAME", [16, 32])):   # stride 2, 2, 1
        tp = ops.deconv2d(output_channels=48, output_imshape=oshape, k_h=k, k_w=k, padding=pad, method="upsample", seed=6,
                          compute=K.BF16X3)
        y = tp(x.to(dev))
        assert tuple(tp.kernel_deconv2d.shape) == (k, k, 48, 32)                      # [kh, kw, Cout, Cin] (ops.py:77-82)
        tp.assign(tp.kernel_deconv2d, torch.from_numpy(rng.standard_normal(48).astype(np.float32)))
        y = tp(x.to(dev))
        ref = T.conv2d_transpose(x, tp.kernel_deconv2d.cpu(), tp.bias_deconv2d.cpu(), oshape[0], oshape[1], oshape[0] // 16, pad)
        assert_close(y, ref, TOL_X3, "ops.deconv2d upsample k=%d -> %s" % (k, oshape))
    with pytest.raises(ValueError):
        ops.deconv2d(16, [32, 64], 3, 3, method="transpose")
    with pytest.raises(ValueError):
        ops.deconv2d(16, [48, 96], 3, 3)(x.to(dev))
    assert_close(ops.maxpool2d(kernel_size=2)(x.to(dev)), T.maxpool2x2(x), 0.0, "ops.maxpool2d")
    assert_close(ops.relu()(x.to(dev)), torch.relu(x), 0.0, "ops.relu")
    bf = ops.conv2d(64, 1, 3, 3, seed=3)                                # default compute: one bf16 product
    assert_close_bf16(bf(x.to(dev)), T.conv2d(x, bf.w.cpu(), bf.b.cpu(), 1, "SAME"), "ops.conv2d bf16")


def test_tf_utils(dev):
    tfu = pkg("tf_utils")
    rng = np.random.default_rng(6)
    x = torch.from_numpy((rng.random((2, 8, 16, 3)) ** 4 * 50).astype(np.float32))
    assert_close(tfu.hdr_logCompression(x.to(dev)), T.hdr_log_compression(x), TOL_F32, "logCompression")
    g = T.hdr_log_compression(x)
    assert_close(tfu.hdr_logDecompression(g.to(dev)), T.hdr_log_decompression(g), TOL_F32, "logDecompression")
    assert_close(tfu.rgb2bgr(x.to(dev)), T.rgb2bgr(x), 0.0, "rgb2bgr")
    assert_close(tfu.bgr2rgb(tfu.rgb2bgr(x.to(dev))), x, 0.0, "bgr2rgb")
    got, ref = tfu.DoG(x.to(dev)), T.dog(x)
    assert len(got) == 4
    for i in range(4):
        assert tuple(got[i].shape) == (2, 16, 32, 3)
        # differences of nearly equal blurs: compare against the scale of the blurred image, not of the difference
        err = float((got[i].cpu() - ref[i]).abs().max() / x.abs().max())
        assert err < 2e-6, (i, err)
    with pytest.raises(ValueError):
        tfu.hdr_logCompression(x.to(dev), validDR=100.)


@pytest.mark.parametrize("B", [2])
def test_inference_graph_through_the_layer_api(dev, B):
    """inference.py:81-115 composed from the mirrored model methods; BF16X3 so the comparison is fp32-class."""
    K, params, synth = pkg("kernels"), pkg("params"), pkg("synth")
    generator, sunpose_net, grad_cam, tfu = pkg("generator"), pkg("sunpose_net"), pkg("grad_cam"), pkg("tf_utils")
    gen_w = params.init_params(params.generator_spec(), 0)
    sun_w = params.init_params(params.sunpose_spec(), 1)
    batch = synth.make_batch(B, seed=1234)
    ref = ostep.inference(_tt(gen_w), _tt(sun_w), torch.from_numpy(batch["ldr"]))

    _gen = generator.model(batch_size=B, im_height=32, im_width=128, weights=gen_w, device=dev, compute=K.BF16X3)
    _sun = sunpose_net.model(weights=sun_w, device=dev, compute=K.BF16X3)
    ldr = torch.from_numpy(batch["ldr"]).to(dev)
    res_out = _gen.encode(ldr, training=False)
    sky_pred_gamma = _gen.sky_decode(res_out, ldr, training=False)
    sky_pred_lin = tfu.hdr_logDecompression(sky_pred_gamma)
    sunpose_cmf, actv = _sun.sunposeEstimation(ldr, training=False)
    sunpose_pred = sunpose_cmf.reshape(B, 32, 128, 1)
    alpha = torch.clamp((sky_pred_lin.max(dim=3, keepdim=True).values - 1.0 + 0.12) / 0.12, 0.0, 1.0)   # host-side glue
    alpha_c3 = alpha.repeat(1, 1, 1, 3)
    y_c = grad_cam.pick(sunpose_cmf)
    cams = [grad_cam.layer(y_c, a) for a in actv]
    sun_rad_lin, gamma, beta = _gen.sun_rad_estimation(ldr, cams[0], cams[1], cams[2], sunpose_pred, training=False)
    sun_rad_gamma = tfu.hdr_logCompression(sun_rad_lin)
    sun_pred_gamma = _gen.sun_decode(res_out, cams[0], cams[1], cams[2], sun_rad_gamma, training=False)
    y_final_gamma = _gen.blending((1.0 - alpha_c3) * sky_pred_gamma, alpha_c3 * sun_pred_gamma, training=False)
    y_final_lin = tfu.hdr_logDecompression(y_final_gamma)

    assert_close(res_out, ref["res_out"], 1e-3, "res_out")
    assert_close(sunpose_cmf, ref["sunpose_cmf"], 2e-3, "cmf")
    assert_close(y_c.values, ref["sunpose_cmf"].max(dim=1).values, 2e-3, "y_c")
    for i, k in enumerate(("sun_cam1", "sun_cam2", "sun_cam3")):   # tolerance: see test_forward_gpu.py
        assert_close(cams[i], ref[k], 5e-2, k)
        assert rel_rms(cams[i], ref[k]) < 3e-2, k
    assert_close(gamma, ref["gamma"], 2e-3, "gamma")
    assert_close(beta, ref["beta"], 2e-3, "beta")
    assert_close(alpha_c3, ref["alpha_c3"], 5e-3, "alpha")
    assert_close(y_final_gamma, ref["y_final_gamma"], 2e-3, "y_final_gamma")
    assert_close(y_final_lin, ref["y_final_lin"], 2e-3, "y_final_lin")
    # a bare tensor has no graph behind it
    with pytest.raises(TypeError):
        grad_cam.layer(sunpose_cmf.max(dim=1).values, actv[0])
    # assign(): new weights take effect (packed images are rebuilt)
    gen2 = params.init_params(params.generator_spec(), 7)
    _gen.assign(gen2)
    assert_close(_gen.encode(ldr), onet.gen_encode(_tt(gen2), torch.from_numpy(batch["ldr"])), 1e-3, "encode after assign")


@pytest.mark.parametrize("training", [False, True])
def test_discriminator_and_sunradnet_models(dev, training):
    K, params, synth = pkg("kernels"), pkg("params"), pkg("synth")
    discriminator, sunrad_net = pkg("discriminator"), pkg("sunrad_net")
    B = 4
    batch = synth.make_batch(B, seed=77)
    ldr, hdr = torch.from_numpy(batch["ldr"]), torch.from_numpy(batch["hdr_t"])
    dis_w = params.init_params(params.discriminator_spec(), 2)
    rng = np.random.default_rng(9)
    for k in dis_w:   # non-trivial BN state so eval mode is a real test
        if "moving_mean" in k: dis_w[k] = rng.normal(0, 0.05, dis_w[k].shape).astype(np.float32)
        if "moving_variance" in k: dis_w[k] = rng.uniform(0.5, 1.5, dis_w[k].shape).astype(np.float32)
    new_stats = {}
    ref = onet.discriminator(_tt(dis_w), ldr, hdr, training, new_stats)
    _dis = discriminator.model(weights=dis_w, device=dev, compute=K.BF16X3)
    got = _dis([ldr.to(dev), hdr.to(dev)], training=training)
    assert tuple(got.shape) == (B, 1, 13, 1)
    assert_close(got, ref, 2e-3, "discriminator logits")
    for k, v in new_stats.items():     # moving averages updated exactly when training
        assert_close(_dis.variables[k], v, 1e-4, k)
    if not training:
        for k in dis_w:
            if "moving" in k:
                assert_close(_dis.variables[k], dis_w[k], 0.0, k)

    # sunRadNet as a stand-alone model
    gen_w = params.init_params(params.generator_spec(), 0)
    sun_vars = {k[4:]: torch.from_numpy(v).to(dev) for k, v in gen_w.items() if k.startswith("sun.")}
    net = sunrad_net.sunRadNet(variables=sun_vars, compute=K.BF16X3)
    x = torch.from_numpy(rng.random((B, 32, 128, 1)).astype(np.float32)); x /= x.max()
    actv = torch.from_numpy(rng.random((B, 32, 128, 6)).astype(np.float32))
    ref_rad, ref_g, ref_b = onet.sun_rad_net(_tt(gen_w), x, actv, training, {})
    rad, g, b = net(x.to(dev), actv.to(dev), training=training)
    assert_close(g, ref_g, 2e-3, "gamma"); assert_close(b, ref_b, 2e-3, "beta")
    assert_close(rad, ref_rad, 5e-3, "sun radiance")


def test_vgg16_model(dev, tmp_path):
    K, params, vgg16 = pkg("kernels"), pkg("params"), pkg("vgg16")
    w = params.init_params(params.vgg_spec(), 3)
    rng = np.random.default_rng(11)
    for k in w:
        if k.endswith(".b"):
            w[k] = rng.normal(0, 0.1, w[k].shape).astype(np.float32)
    # the reference's file format: pickled dict name -> [W, b]  (vgg16.py:99)
    path = str(tmp_path / "vgg16.npy")
    np.save(path, {n: [w[n + ".w"], w[n + ".b"]] for n, _, _ in params.VGG_CHANNELS}, allow_pickle=True)
    x = torch.from_numpy(rng.random((2, 32, 128, 3)).astype(np.float32))
    ref = onet.vgg16_pools(_tt(w), x)
    got = vgg16.Vgg16(path, device=dev, compute=K.BF16X3)(x.to(dev))
    for i, shape in enumerate(((2, 16, 64, 64), (2, 8, 32, 128), (2, 4, 16, 256))):
        assert tuple(got[i].shape) == shape
        assert_close(got[i], ref[i], 1e-3, "pool%d" % (i + 1))
    with pytest.raises(ValueError):
        vgg16.Vgg16(None)


def test_distortion_aware_res_stack(dev):
    """generator.model(distortion_aware=True): the res blocks built from distortion_aware_ops.conv2d (the variant
    generator.py:14,18 keeps commented out), InstanceNorm statistics from the DA conv epilogue - against the oracle
    composition, and different from the plain encoder."""
    from oracle import networks as N
    P, gen_mod, K = pkg("params"), pkg("generator"), pkg("kernels")
    w = P.init_params(P.generator_spec(), 0)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (2, 32, 128, 3)).astype(np.float32)
    ref = N.gen_encode({k: torch.from_numpy(v) for k, v in w.items()}, torch.from_numpy(x), distortion_aware=True)
    g = gen_mod.model(weights=w, device=dev, compute=K.BF16X3, distortion_aware=True)
    got = g.encode(torch.from_numpy(x).to(dev))
    assert_close(got, ref, 1e-3, "distortion-aware res stack")
    plain = gen_mod.model(weights=w, device=dev, compute=K.BF16X3).encode(torch.from_numpy(x).to(dev))
    assert float((plain - got).abs().max()) > 1e-2 * float(got.abs().max())
    # statistics partials of the DA conv epilogue == statistics of its output
    xin = torch.from_numpy(rng.standard_normal((3, 8, 32, 128)).astype(np.float32)).to(dev)
    y, st = K.da_conv2d(xin, g.nets.pk["gen.res.0.conv1"], g.nets.gen["res.0.conv1.b"], g.nets.da_offsets(8, 32), K.BF16X3,
                        want_stats=True)
    s = st.part.sum(dim=1)
    assert_close(s[:, 0], y.sum(dim=(1, 2)), 1e-4, "sum partials")
    assert_close(s[:, 1], (y * y).sum(dim=(1, 2)), 1e-4, "sumsq partials")
