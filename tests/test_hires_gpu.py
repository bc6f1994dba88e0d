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
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(out, eager) and torch.isfinite(out).all()
