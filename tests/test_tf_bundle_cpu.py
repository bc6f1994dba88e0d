"""tf_bundle.py: the TF tensor-bundle checkpoint format (LevelDB-format index table + raw data shard) written and read
without TensorFlow, and its hook-up to the checkpoint manager.  Unpinned against TF itself;
these tests pin the framing byte for byte against the format definition and the reader against hand-built blocks."""
import os
import struct

import numpy as np
import pytest

from conftest import pkg


def test_crc32c_host_function():
    tfb, tb = pkg("tf_bundle"), pkg("tb_logging")
    for d in (b"", b"123456789", bytes(32), bytes([255] * 32), bytes(range(256)) * 3 + b"xyz"):
        assert tfb.crc32c(d) == tb.crc32c(d)
    assert tfb.crc32c(b"123456789") == 0xE3069283
    a = np.arange(1000, dtype=np.float32)
    assert tfb.crc32c(a) == tb.crc32c(a.tobytes())
    assert tfb.crc32c(a.tobytes()[100:], tfb.crc32c(a.tobytes()[:100])) == tfb.crc32c(a)      # incremental


def test_block_prefix_compression_and_restarts():
    tfb = pkg("tf_bundle")
    ent = [(b"apple", b"1"), (b"applesauce", b"22"), (b"apply", b""), (b"banana", b"4444")]
    blk = tfb.build_block(ent, restart_interval=2)
    # entry 2 shares "apple" (5) with entry 1; entry 3 starts a restart (shared 0); entry 4 shares 0 with "apply"
    assert blk[:9] == bytes([0, 5, 1]) + b"apple" + b"1" and blk[9:19] == bytes([5, 5, 2]) + b"sauce" + b"22"
    assert blk[19:22] == bytes([0, 5, 0]) and struct.unpack("<III", blk[-12:]) == (0, 19, 2)
    assert tfb.parse_block(blk) == ent
    assert tfb.parse_block(tfb.build_block([])) == [] and len(tfb.build_block([])) == 8
    # a hand-written block (not produced by build_block): k1="ab"->"x", k2="abc"->"yz" with shared=2
    hand = bytes([0, 2, 1]) + b"ab" + b"x" + bytes([2, 1, 2]) + b"c" + b"yz" + struct.pack("<II", 0, 1)
    assert tfb.parse_block(hand) == [(b"ab", b"x"), (b"abc", b"yz")]


def test_snappy_decompress_hand_built_stream():
    tfb = pkg("tf_bundle")
    # "abcdabcdabcdX": literal "abcd" (tag (4-1)<<2), copy-1 len 8 offset 4 (overlapping), literal "X"
    s = bytes([13, 3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([0]) + b"X"
    assert tfb.snappy_decompress(s) == b"abcdabcdabcdX"
    # copy-2: "0123456789" then copy len 10 offset 10
    s2 = bytes([20, 9 << 2]) + b"0123456789" + bytes([((10 - 1) << 2) | 2, 10, 0])
    assert tfb.snappy_decompress(s2) == b"0123456789" * 2
    with pytest.raises(ValueError):
        tfb.snappy_decompress(bytes([5, 0 << 2]) + b"a")


def test_table_round_trip_multi_block_and_footer(tmp_path):
    tfb = pkg("tf_bundle")
    rng = np.random.default_rng(0)
    ent = sorted({("key/%04d/%s" % (i, "x" * int(rng.integers(0, 30)))).encode(): rng.bytes(int(rng.integers(0, 200)))
                  for i in range(300)}.items())
    path = str(tmp_path / "t.index")
    tfb.write_table(path, ent, block_size=1024)
    assert tfb.read_table(path) == ent
    raw = open(path, "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57 and len(raw) > 20000
    bad = bytearray(raw); bad[100] ^= 0x40
    open(path, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        tfb.read_table(path)
    with pytest.raises(ValueError):
        tfb.write_table(path, [(b"b", b""), (b"a", b"")])


def test_bundle_round_trip_bytes_and_corruption(tmp_path):
    tfb = pkg("tf_bundle")
    rng = np.random.default_rng(1)
    t = {"gen_model/conv1_d/w/.ATTRIBUTES/VARIABLE_VALUE": rng.standard_normal((7, 7, 3, 32)).astype(np.float32),
         "gen_model/conv1_d/b/.ATTRIBUTES/VARIABLE_VALUE": rng.standard_normal(32).astype(np.float32),
         "epoch/.ATTRIBUTES/VARIABLE_VALUE": np.asarray(40, np.int64),
         "misc/f64": rng.standard_normal((2, 3)), "misc/u8": rng.integers(0, 255, 5, dtype=np.uint8),
         "misc/flag": np.asarray([True, False]), "misc/half": rng.standard_normal(4).astype(np.float16),
         "misc/empty": np.zeros((0, 4), np.float32)}
    prefix = str(tmp_path / "ckpt-4")
    tfb.write_bundle(prefix, t)
    back = tfb.read_bundle(prefix)
    assert set(back) == set(t)
    for k in t:
        assert back[k].dtype == t[k].dtype and back[k].shape == t[k].shape and np.array_equal(back[k], t[k]), k
    # data shard = tensors back to back in key order; the header entry is the 6 bytes num_shards=1, version.producer=1
    order = sorted(t, key=lambda s: s.encode())
    assert open(prefix + ".data-00000-of-00001", "rb").read() == b"".join(np.ascontiguousarray(t[k]).tobytes() for k in order)
    table = tfb.read_table(prefix + ".index")
    assert table[0] == (b"", bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01]))
    e = tfb._parse(dict(table)[b"gen_model/conv1_d/w/.ATTRIBUTES/VARIABLE_VALUE"])
    assert e[1] == [1] and e[5] == [7 * 7 * 3 * 32 * 4] and len(e[6][0]) == 4                # DT_FLOAT, size, fixed32 crc
    assert [tfb._parse(d)[1][0] for d in tfb._parse(e[2][0])[2]] == [7, 7, 3, 32]
    assert tfb.variable_tensors(back).keys() == {"gen_model/conv1_d/w", "gen_model/conv1_d/b", "epoch"}
    # flipped data byte -> tensor checksum mismatch
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); raw[-3] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        tfb.read_bundle(prefix)
    # bfloat16 entries (dtype enum 14) are widened to float32; string entries are skipped
    ent = [(b"", tfb._HEADER),
           (b"_CHECKPOINTABLE_OBJECT_GRAPH", tfb._entry_proto(7, [], 0, 0, 0)),
           (b"w", tfb._entry_proto(14, [2], 0, 4, tfb.mask(tfb.crc32c(struct.pack("<HH", 0x3F80, 0xC000)))))]
    p2 = str(tmp_path / "bf")
    tfb.write_table(p2 + ".index", ent)
    open(p2 + ".data-00000-of-00001", "wb").write(struct.pack("<HH", 0x3F80, 0xC000))
    assert tfb.read_bundle(p2)["w"].tolist() == [1.0, -2.0] and list(tfb.read_bundle(p2)) == ["w"]


def test_checkpoint_manager_reads_and_writes_tf_bundles(tmp_path):
    ckpt, P = pkg("checkpoint"), pkg("params")
    gen = P.init_params(P.generator_spec(), 0)
    native = {"gen_model/" + k.replace(".", "/"): v for k, v in gen.items()}
    native["gen_optimizer/rms"] = np.ones(10, np.float32)
    d = str(tmp_path / "SKY")
    os.makedirs(d)
    ckpt.export_tf_bundle(os.path.join(d, "ckpt-3"), native, epoch=30)
    tfb = pkg("tf_bundle")
    keys = tfb.read_bundle(os.path.join(d, "ckpt-3")).keys()
    assert "gen_model/res/sequence/0/conv1/w/.ATTRIBUTES/VARIABLE_VALUE" in keys          # the reference's object-graph path
    assert "gen_model/sun/d2/conv/kernel/.ATTRIBUTES/VARIABLE_VALUE" in keys and not any("rms" in k for k in keys)
    mgr = ckpt.CheckpointManager(d)
    assert mgr.latest_checkpoint.endswith("ckpt-3.index")
    tensors, epoch = mgr.restore()
    assert epoch == 30 and "save_counter" not in tensors
    gen2 = P.init_params(P.generator_spec(), 5)
    assert ckpt.load_into(gen2, tensors, "gen_model") == len(gen2)
    assert all(np.array_equal(gen[k], gen2[k]) for k in gen)
    # a newer native checkpoint takes precedence
    mgr.save({"gen_model/conv1_d/b": np.zeros(32, np.float32)}, epoch=40)   # becomes ckpt-1.npz: older number than 3
    assert mgr.latest_checkpoint.endswith("ckpt-3.index")
    for _ in range(3):
        mgr.save({"gen_model/conv1_d/b": np.zeros(32, np.float32)}, epoch=50)
    assert mgr.latest_checkpoint.endswith("ckpt-4.npz") and mgr.restore()[1] == 50
