"""TensorBoard-compatible scalar logging without TensorFlow.

The reference logs nine loss scalars per epoch through `tf.summary.create_file_writer` under
`./tensorboard/SKY/<timestamp>/{train,val}` (tf_utils.py:282-292, train.py:478-489,497-506).  This module writes the
same on-disk format TensorBoard reads - an `events.out.tfevents.*` file of TFRecord-framed `Event` protocol buffers
(length, masked CRC-32C of the length, payload, masked CRC-32C of the payload), the first record carrying
`file_version = "brain.Event:2"`, every scalar a `Summary.Value{tag, simple_value}` - with the protobuf wire encoding
done by hand (five fields).  `read_events` parses such a file back (used by the tests; it also checks both CRCs).
"""
import os
import socket
import struct
import time

_CRC_TABLE = []


def _crc_table():
    if not _CRC_TABLE:
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1      # CRC-32C (Castagnoli), reflected
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data):
    t, c = _crc_table(), 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _event(wall_time, step=None, file_version=None, scalars=()):
    # Event: 1 wall_time double, 2 step int64, 3 file_version string, 5 summary Summary
    # Summary: 1 repeated Value;  Value: 1 tag string, 2 simple_value float
    msg = _varint((1 << 3) | 1) + struct.pack("<d", wall_time)
    if step is not None:
        msg += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        msg += _field_bytes(3, file_version.encode())
    if scalars:
        summary = b"".join(_field_bytes(1, _field_bytes(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(v)))
                           for tag, v in scalars)
        msg += _field_bytes(5, summary)
    return msg


class SummaryWriter:
    """`tf.summary.create_file_writer(logdir)` + `tf.summary.scalar(tag, value, step)` for scalars."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s.%d.v2" % (int(time.time()), socket.gethostname(), os.getpid()))
        self._f = open(self.path, "wb")
        self._record(_event(time.time(), file_version="brain.Event:2"))
        self.flush()

    def _record(self, payload):
        head = struct.pack("<Q", len(payload))
        self._f.write(head + struct.pack("<I", masked_crc32c(head)) + payload + struct.pack("<I", masked_crc32c(payload)))

    def scalar(self, tag, value, step):
        self._record(_event(time.time(), step=step, scalars=[(tag, value)]))

    def scalars(self, values, step):
        """Several tags of one step in one record: values = {tag: value}."""
        self._record(_event(time.time(), step=step, scalars=list(values.items())))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()


def create_directories(path, name="SKY"):
    """tf_utils.createDirectories(path, name, dir="tensorboard") (tf_utils.py:282-292): <path>/tensorboard/<name>/<timestamp>/
    {train,val} writers.  Returns (train_writer, val_writer, logdir)."""
    logdir = os.path.join(path, "tensorboard", name, time.strftime("%Y%m%d-%H%M%S"))
    return SummaryWriter(os.path.join(logdir, "train")), SummaryWriter(os.path.join(logdir, "val")), logdir


def _read_varint(buf, pos):
    n = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, pos


def _parse(buf):
    """Minimal protobuf walk: {field number: [values]} with length-delimited fields left as bytes."""
    out, pos = {}, 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        else:
            raise ValueError("unsupported wire type %d" % wt)
        out.setdefault(num, []).append(v)
    return out


def read_events(path):
    """Parses an event file written by SummaryWriter (or by TensorFlow, for simple_value scalars): list of dicts
    {wall_time, step, file_version, scalars: {tag: value}}.  Raises ValueError on a CRC mismatch."""
    events = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        (c1,) = struct.unpack("<I", data[pos + 8:pos + 12])
        payload = data[pos + 12:pos + 12 + n]
        (c2,) = struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])
        if c1 != masked_crc32c(head) or c2 != masked_crc32c(payload) or len(payload) != n:
            raise ValueError("corrupt record at byte %d" % pos)
        pos += 16 + n
        ev = _parse(payload)
        rec = {"wall_time": struct.unpack("<d", ev[1][0])[0], "step": ev.get(2, [0])[0],
               "file_version": ev[3][0].decode() if 3 in ev else None, "scalars": {}}
        for summary in ev.get(5, []):
            for value in _parse(summary).get(1, []):
                v = _parse(value)
                if 2 in v:
                    rec["scalars"][v[1][0].decode()] = struct.unpack("<f", v[2][0])[0]
        events.append(rec)
    return events
