"""hdrsky_jpeg_roundtrip (the JPEG step of train.py:86-92) on the GPU: bit-exact against the libjpeg fixture and against
oracle/jpeg.py (itself pinned to libjpeg by tests/test_jpeg_cpu.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import jpeg as J

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _as_float(u8):
    return torch.from_numpy(u8.astype(np.float32) / 255.0)


def test_matches_libjpeg_fixture(dev):
    K = pkg("kernels")
    g = np.load(os.path.join(HERE, "golden", "jpeg_libjpeg.npz"))
    names = [k[3:] for k in g.files if k.startswith("in_")]
    for name in names:
        im = g["in_" + name]
        qs = (90, 93, 97, 100, 50, 20)
        x = _as_float(np.stack([im] * len(qs))).to(dev)
        out = K.jpeg_roundtrip(x, quality=qs, order="rgb")
        got = torch.round(out * 255.0).to(torch.uint8).cpu().numpy()
        for i, q in enumerate(qs):
            assert np.array_equal(got[i], g["q%d_%s" % (q, name)]), (name, q)
        # the float the reference computes: tf.cast(jpeg_img, tf.float32) / 255.0 (train.py:92)
        assert torch.equal(out.cpu(), _as_float(got))


def test_batch_of_32_reference_quality_ramp_both_channel_orders(dev):
    K, synth = pkg("kernels"), pkg("synth")
    ldr = synth.make_batch(32, seed=3)["ldr"]                                  # BGR, values k/255
    ref = J.jpeg_batch(ldr, "bgr")
    out = K.jpeg_roundtrip(torch.from_numpy(ldr).to(dev))                      # default: bgr, 90..100 ramp
    assert torch.equal(out.cpu(), torch.from_numpy(ref))
    rgb = np.ascontiguousarray(ldr[..., ::-1])
    out2 = K.jpeg_roundtrip(torch.from_numpy(rgb).to(dev), order="rgb")
    assert torch.equal(out2.cpu(), torch.from_numpy(np.ascontiguousarray(ref[..., ::-1])))
    assert float((out.cpu() - torch.from_numpy(ldr)).abs().max()) > 0         # it does change the image
    # in place
    x = torch.from_numpy(ldr).to(dev)
    K.jpeg_roundtrip(x, out=x)
    assert torch.equal(x.cpu(), torch.from_numpy(ref))


def test_other_sizes_random_content_and_errors(dev):
    K = pkg("kernels")
    rng = np.random.default_rng(8)
    # whole MCUs, partial MCUs (H or W not a multiple of 16, odd sizes, W % 4 != 0), chroma planes of <= 2 columns
    for (b, h, w) in ((1, 16, 16), (3, 64, 256), (5, 48, 80), (2, 128, 512), (2, 17, 33), (3, 40, 40), (2, 31, 127), (4, 5, 7),
                      (2, 2, 3), (2, 3, 2), (1, 1, 1), (2, 100, 37), (2, 24, 56), (1, 9, 1), (2, 20, 4)):
        ldr = rng.integers(0, 256, (b, h, w, 3)).astype(np.float32) / 255.0
        qs = [int(q) for q in rng.integers(1, 101, b)]
        ref = np.stack([J.adjust_jpeg_quality(np.rint(ldr[i] * 255).astype(np.uint8), qs[i]) for i in range(b)])
        out = K.jpeg_roundtrip(torch.from_numpy(ldr).to(dev), quality=qs, order="rgb")
        assert np.array_equal(torch.round(out * 255).to(torch.uint8).cpu().numpy(), ref), (b, h, w)
    with pytest.raises(Exception):
        K.jpeg_roundtrip(torch.zeros(1, 16, 32, 4, device=dev))                # three channels only


def test_device_batch_synthesis_applies_jpeg(dev):
    synth, K = pkg("synth"), pkg("kernels")
    a = synth.make_batch_device(8, seed=4, device=dev, jpeg=False)
    b = synth.make_batch_device(8, seed=4, device=dev)
    assert torch.equal(a["hdr_t"], b["hdr_t"]) and torch.equal(a["sunpose_gt"], b["sunpose_gt"])
    assert torch.equal(b["ldr"], K.jpeg_roundtrip(a["ldr"]))
    assert torch.equal(b["ldr"].cpu(), torch.from_numpy(J.jpeg_batch(a["ldr"].cpu().numpy(), "bgr")))
