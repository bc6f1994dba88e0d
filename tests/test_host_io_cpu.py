"""CPU tests of the host-side surface: Radiance .hdr codec and the checkpoint manager (retention, latest, restore)."""
import os

import numpy as np
import pytest

from conftest import pkg


def test_hdr_roundtrip(tmp_path):
    io = pkg("hdr_io")
    rng = np.random.default_rng(0)
    img = (10.0 ** rng.uniform(-3, 4, (32, 128, 3))).astype(np.float32)
    img[0, 0] = 0.0
    p = os.path.join(str(tmp_path), "a.hdr")
    io.write_hdr(p, img)
    back = io.read_hdr(p)
    assert back.shape == img.shape
    # RGBE: 8-bit mantissa shared exponent -> relative error <= 1/128 of the pixel's max channel
    tol = img.max(axis=-1, keepdims=True) / 128.0 + 1e-30
    assert (np.abs(back - img) <= tol).all()
    assert (back[0, 0] == 0).all()
    data = open(p, "rb").read()
    assert data.startswith(b"#?RADIANCE") and b"-Y 32 +X 128" in data


def test_checkpoint_manager_retention_and_restore(tmp_path):
    ck = pkg("checkpoint")
    m = ck.CheckpointManager(os.path.join(str(tmp_path), "SKY"), max_to_keep=5)
    assert m.latest_checkpoint is None and m.restore() == (None, 0)
    for e in range(1, 8):
        m.save({"gen_model/conv1_d/w": np.full((2, 2), e, np.float32), "gen_optimizer/rms": np.zeros(4, np.float32)}, epoch=e * 10)
    files = sorted(os.listdir(m.directory))
    assert len([f for f in files if f.endswith(".npz")]) == 5 and "checkpoint" in files
    assert m.latest_checkpoint.endswith("ckpt-7.npz")
    t, epoch = m.restore()
    assert epoch == 70 and float(t["gen_model/conv1_d/w"][0, 0]) == 7.0
    params = {"conv1_d.w": np.zeros((2, 2), np.float32), "conv1_d.b": np.zeros(2, np.float32)}
    assert ck.load_into(params, t, "gen_model", strict=False) == 1 and params["conv1_d.w"][1, 1] == 7.0
    with pytest.raises(KeyError):                      # a partial restore must not pass silently
        ck.load_into(params, t, "gen_model")
    with pytest.raises(ValueError):                    # nor one written for another geometry
        ck.load_into({"conv1_d.w": np.zeros((3, 2), np.float32)}, t, "gen_model")


def test_load_into_accepts_the_object_graph_attribute_spelling():
    """tf.train.Checkpoint keys a variable by the attribute the layer stores it under: ops.conv2d keeps `self.w` /
    `self.biases` (add_weight names 'w' / 'b', ops.py:30-37), ops.deconv2d `self.kernel` / `self.biases` (names
    'kernel_deconv2d' / 'bias_deconv2d', ops.py:96-108).  Both spellings restore every variable."""
    ck, P = pkg("checkpoint"), pkg("params")
    spec = P.generator_spec()
    ref = P.init_params(spec, 5)
    alias = {"b": "biases", "kernel_deconv2d": "kernel", "bias_deconv2d": "biases"}
    native, graph = {}, {}
    for k, v in ref.items():
        parts = k.split(".")
        native["gen_model/" + "/".join(parts)] = v
        graph["gen_model/" + "/".join(parts[:-1] + [alias.get(parts[-1], parts[-1])])] = v
    for tensors in (native, graph):
        got = P.init_params(spec, 6)
        assert ck.load_into(got, tensors, "gen_model") == len(spec)
        assert all(np.array_equal(got[k], ref[k]) for k in ref)


def test_dorf_curve_file_is_parsed_like_getDoRF(tmp_path):
    """utils.getDoRF (utils.py:105-116): six lines per curve, the sixth holds the response samples; first 175 train."""
    synth = pkg("synth")
    rng = np.random.default_rng(3)
    k, n = 16, 9
    curves = np.sort(rng.uniform(0, 1, (n, k)).astype(np.float32), axis=1)
    curves[:, 0], curves[:, -1] = 0.0, 1.0
    p = os.path.join(str(tmp_path), "dorfCurves.txt")
    with open(p, "w") as f:
        for i, c in enumerate(curves):
            f.write("curve%d\nsRGB\nI =\n%s\nB =\n  %s  \n" % (i, " ".join("%.6f" % v for v in np.linspace(0, 1, k)),
                                                          "   ".join("%.8f" % v for v in c)))
    tr, te = synth.load_dorf(p, n_train=6)
    assert tr.shape == (6, k) and te.shape == (3, k) and tr.dtype == np.float32
    np.testing.assert_allclose(np.concatenate([tr, te]), curves, atol=1e-7)
    pick = synth.pick_crf(tr, 32, seed=11)
    assert pick.shape == (32, k) and all(any(np.array_equal(r, c) for c in tr) for r in pick)
    assert np.array_equal(pick, synth.pick_crf(tr, 32, seed=11))           # seeded
    with open(p, "a") as f:
        f.write("x\ny\nI =\n0 1\nB =\n0 0.5 1\n")                           # a curve of another length: rejected
    with pytest.raises(ValueError):
        synth.load_dorf(p)
