"""The TensorBoard event-file writer (tb_logging.py, replacing tf.summary of tf_utils.py:282-292 / train.py:478-489)."""
import glob
import os
import struct

import pytest

from conftest import pkg


def test_crc32c_known_answers():
    tb = pkg("tb_logging")
    assert tb.crc32c(b"123456789") == 0xE3069283                      # the standard CRC-32C check value
    assert tb.crc32c(b"") == 0
    assert tb.crc32c(bytes(32)) == 0x8A9136AA                         # RFC 3720 B.4: 32 zero bytes
    assert tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                # RFC 3720 B.4: 32 0xFF bytes
    assert tb.masked_crc32c(b"123456789") == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_event_file_round_trip_and_framing(tmp_path):
    tb = pkg("tb_logging")
    train, val, logdir = tb.create_directories(str(tmp_path), "SKY")
    assert os.path.isdir(os.path.join(logdir, "train")) and os.path.isdir(os.path.join(logdir, "val"))
    names = ["gen_total_loss", "gen_l1_loss", "gen_perceptual_loss", "gen_DoG_loss", "gen_adv_loss", "gen_kl_div",
             "disc_total_loss", "disc_generated_loss", "disc_real_loss"]   # train.py:480-489
    for epoch in (1, 2, 300):
        train.scalars({n: 0.5 * epoch + i for i, n in enumerate(names)}, step=epoch)
    train.scalar("g_out", 1.25, step=300)
    train.close(); val.close()
    (path,) = glob.glob(os.path.join(logdir, "train", "events.out.tfevents.*"))
    ev = tb.read_events(path)
    assert ev[0]["file_version"] == "brain.Event:2" and not ev[0]["scalars"]
    assert [e["step"] for e in ev[1:]] == [1, 2, 300, 300]
    assert ev[3]["scalars"] == {n: 150.0 + i for i, n in enumerate(names)}
    assert ev[4]["scalars"] == {"g_out": 1.25}
    # byte-level: first record = u64 length, masked crc, payload starting with field 1 (wall_time, wire type 1)
    raw = open(path, "rb").read()
    (n,) = struct.unpack("<Q", raw[:8])
    assert raw[12] == 0x09 and raw[12 + 9] == 0x1A and raw[12 + 11:12 + n] == b"brain.Event:2"
    # a flipped payload byte is detected
    bad = bytearray(raw); bad[20] ^= 1
    p2 = os.path.join(str(tmp_path), "bad"); open(p2, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        tb.read_events(p2)
