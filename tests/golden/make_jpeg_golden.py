#!/usr/bin/env python3
"""Generates tests/golden/jpeg_libjpeg.npz: 8-bit images and their JPEG quality round trips computed by libjpeg ITSELF
(through Pillow's JPEG codec: baseline, 4:2:0, slow-integer DCT, fancy upsampling - the defaults TensorFlow's
encode_jpeg / decode_jpeg also use), i.e. by the third-party code behind `tf.image.adjust_jpeg_quality`
(train.py:89).  Unlike the other fixtures this one does NOT come from this repo's restatement: it pins oracle/jpeg.py
and, through it, the HIP kernel.

    python tests/golden/make_jpeg_golden.py
"""
import importlib
import io
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"


def libjpeg_roundtrip(u8_rgb, quality):
    buf = io.BytesIO()
    Image.fromarray(u8_rgb, "RGB").save(buf, "JPEG", quality=int(quality))
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB"))


def images():
    synth = importlib.import_module(PKG + ".synth")
    rng = np.random.default_rng(2024)
    h, w = 32, 128
    sky = np.rint(synth.make_batch(4, h, w, seed=99)["ldr"][..., ::-1] * 255.0).astype(np.uint8)   # BGR -> RGB
    ramp = np.add.outer(np.linspace(0, 200, h), np.linspace(0, 55, w))[..., None] + rng.normal(0, 3, (h, w, 3))
    return {
        "sky0": sky[0], "sky1": sky[1], "sky2": sky[2], "sky3": sky[3],
        "random": rng.integers(0, 256, (h, w, 3), dtype=np.uint8),
        "ramp": np.clip(ramp, 0, 255).astype(np.uint8),
        "saturated": np.where(rng.random((h, w, 3)) > 0.5, 255, 0).astype(np.uint8),
        "tall": rng.integers(0, 256, (64, 48, 3), dtype=np.uint8),
        # partial 16x16 MCUs / tiny chroma planes: libjpeg's edge replication and its plain-upsampling special case
        "ragged17x33": rng.integers(0, 256, (17, 33, 3), dtype=np.uint8),
        "ragged40x40": rng.integers(0, 256, (40, 40, 3), dtype=np.uint8),
        "ragged31x127": sky[0][:31, :127].copy(),
        "tiny3x2": rng.integers(0, 256, (3, 2, 3), dtype=np.uint8),
    }


def main():
    out = {}
    for name, im in images().items():
        out["in_" + name] = im
        for q in (90, 93, 97, 100, 50, 20):
            out["q%d_%s" % (q, name)] = libjpeg_roundtrip(im, q)
    np.savez_compressed(os.path.join(HERE, "jpeg_libjpeg.npz"), **out)
    print("libjpeg-turbo:", features.check_feature("libjpeg_turbo"), "jpeg lib", features.version("jpg"), "-", len(out), "arrays")


if __name__ == "__main__":
    main()
