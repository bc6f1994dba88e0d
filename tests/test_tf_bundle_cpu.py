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


# ---- an independent, minimal protobuf reader for the object-graph test (deliberately not tf_bundle._parse) --------------
def _pb_fields(buf):
    i, out = 0, []
    while i < len(buf):
        tag = shift = 0
        while True:
            b = buf[i]; i += 1
            tag |= (b & 0x7F) << shift; shift += 7
            if b < 0x80:
                break
        num, wire = tag >> 3, tag & 7
        if wire == 0:
            val = shift = 0
            while True:
                b = buf[i]; i += 1
                val |= (b & 0x7F) << shift; shift += 7
                if b < 0x80:
                    break
            out.append((num, val))
        elif wire == 2:
            ln = shift = 0
            while True:
                b = buf[i]; i += 1
                ln |= (b & 0x7F) << shift; shift += 7
                if b < 0x80:
                    break
            out.append((num, bytes(buf[i:i + ln]))); i += ln
        else:
            raise AssertionError("unexpected wire type %d" % wire)
    return out


def test_exported_bundle_carries_the_object_graph_checkpoint_restore_walks(tmp_path):
    """export_tf_bundle writes the structure of `tf.train.Checkpoint(epoch, gen_model, dis_model, gen_optimizer,
    disc_optimizer).save` (train.py:208-220): the `_CHECKPOINTABLE_OBJECT_GRAPH` string entry is a TrackableObjectGraph in
    which every saved variable is reached from the root by child edges spelling its attribute path, carries a
    VARIABLE_VALUE attribute whose checkpoint_key is its bundle key, and every optimizer slot hangs off its optimizer's
    node with the right original variable.  Parsed here with a separate minimal protobuf reader.  (Unpinned against TF.)"""
    ckpt, tfb, P = pkg("checkpoint"), pkg("tf_bundle"), pkg("params")
    gen, dis = P.init_params(P.generator_spec(), 0), P.init_params(P.discriminator_spec(), 2)
    native = {"gen_model/" + k.replace(".", "/"): v for k, v in gen.items()}
    native.update({"dis_model/" + k.replace(".", "/"): v for k, v in dis.items()})
    native["gen_optimizer/rms"] = np.ones(10, np.float32)          # flat buffers are not exported
    slots = {"gen_optimizer": {"gen_model/conv1_d/w": np.full((7, 7, 3, 32), 0.5, np.float32),
                               "gen_model/res/3/conv2/b": np.full(128, 0.25, np.float32)},
             "disc_optimizer": {"dis_model/out/kernel": np.full((4, 4, 512, 1), 2.0, np.float32)}}
    prefix = str(tmp_path / "SKY" / "ckpt-7")
    ckpt.export_tf_bundle(prefix, native, epoch=70, slots=slots)
    bundle = tfb.read_bundle(prefix)
    blob = tfb.read_string_entry(prefix, "_CHECKPOINTABLE_OBJECT_GRAPH")
    assert blob is not None and tfb.read_string_entry(prefix, "no/such/key") is None
    # --- decode with the independent reader ---
    nodes = []
    for num, raw in _pb_fields(blob):
        assert num == 1
        ch, at, sl = {}, [], []
        for fnum, val in _pb_fields(raw):
            sub = dict(_pb_fields(val))
            if fnum == 1:
                ch[sub[2].decode()] = sub.get(1, 0)
            elif fnum == 2:
                at.append((sub[1].decode(), sub[2].decode(), sub[3].decode()))
            elif fnum == 3:
                sl.append((sub.get(1, 0), sub[2].decode(), sub.get(3, 0)))
        nodes.append((ch, at, sl))
    # the root is the Checkpoint object: its edges are the keyword arguments of train.py:208-213 (+ save_counter)
    assert set(nodes[0][0]) == {"epoch", "gen_model", "dis_model", "gen_optimizer", "disc_optimizer", "save_counter"}
    assert all(0 < cid < len(nodes) for ch, _, _ in nodes for cid in ch.values())
    # every variable key of the bundle is reachable by walking its attribute path from the root
    var_keys = [k for k in bundle if k.endswith(tfb.SUFFIX) and tfb.SLOT not in k]
    assert len(var_keys) == len(gen) + len(dis) + 2
    for k in var_keys:
        nid = 0
        for comp in k[:-len(tfb.SUFFIX)].split("/"):
            nid = nodes[nid][0][comp]
        (name, full, key), = nodes[nid][1]
        assert name == "VARIABLE_VALUE" and key == k and full
    # attribute spelling of the reference's layers: conv2d keeps `w` / `biases`, deconv2d `kernel` / `biases` (ops.py)
    assert "gen_model/conv1_d/biases" + tfb.SUFFIX in bundle and "gen_model/conv3_f/kernel" + tfb.SUFFIX in bundle
    assert "gen_model/res/sequence/3/conv2/biases" + tfb.SUFFIX in bundle and "dis_model/d2/norm/moving_mean" + tfb.SUFFIX in bundle
    # slot variables: referenced from the optimizer node, pointing at the original variable's node, keyed the TF way
    def node_of(path):
        nid = 0
        for comp in path.split("/"):
            nid = nodes[nid][0][comp]
        return nid
    gslots = nodes[node_of("gen_optimizer")][2]
    assert sorted(s[1] for s in gslots) == ["rms", "rms"] and {s[0] for s in gslots} == {node_of("gen_model/conv1_d/w"), node_of("gen_model/res/sequence/3/conv2/biases")}
    for orig, sname, snode in gslots + nodes[node_of("disc_optimizer")][2]:
        (name, full, key), = nodes[snode][1]
        assert tfb.SLOT in key and key.endswith("/rms" + tfb.SUFFIX) and key in bundle
    assert float(bundle["dis_model/out/kernel" + tfb.SLOT + "disc_optimizer/rms" + tfb.SUFFIX][0, 0, 0, 0]) == 2.0
    # tf_bundle's own parser agrees with the independent one
    own = tfb.parse_object_graph(blob)
    assert [(n["children"], n["attributes"], n["slots"]) for n in own] == [(c, a, s) for c, a, s in nodes]
    # and the file still restores through the checkpoint manager (attribute spelling accepted by load_into)
    tensors, epoch = ckpt.CheckpointManager(str(tmp_path / "SKY")).restore()
    gen2 = P.init_params(P.generator_spec(), 9)
    assert epoch == 70 and ckpt.load_into(gen2, tensors, "gen_model") == len(gen2)
    assert all(np.array_equal(gen[k], gen2[k]) for k in gen)
    # a corrupted string entry is detected
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    pos = raw.find(b"gen_optimizer")
    raw[pos] ^= 0x20
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        tfb.read_string_entry(prefix, "_CHECKPOINTABLE_OBJECT_GRAPH")
