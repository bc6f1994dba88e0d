"""TensorFlow tensor-bundle checkpoint files without TensorFlow.

`tf.train.Checkpoint.save` (train.py:208-220,516-522; tf_utils.py:298-320) writes `<prefix>.index` + `<prefix>.data-00000-
of-00001`.  The index is a LevelDB-format sorted table (prefix-compressed key/value blocks with restart arrays, a 5-byte
trailer per block = compression type + masked CRC-32C, an index block of block handles, a 48-byte footer ending in the
magic 0xdb4775248b80fb57); its values are protocol buffers: key "" -> BundleHeaderProto, every other key ->
BundleEntryProto{dtype, shape, shard_id, offset, size, crc32c} pointing into the data shard, which holds the raw
little-endian tensor bytes back to back.  Object-based checkpoints name variables by attribute path,
`gen_model/conv1_d/w/.ATTRIBUTES/VARIABLE_VALUE`, and add one DT_STRING entry `_CHECKPOINTABLE_OBJECT_GRAPH`.

read_bundle(prefix)          -> {key: ndarray} of every numeric tensor (CRCs of blocks and tensors verified; string
                                entries such as the object graph are skipped); reads uncompressed and snappy blocks.
write_bundle(prefix, dict)   -> the two files, single shard, uncompressed blocks (what TF's BundleWriter emits), readable
                                through TF's low-level reader (`tf.train.load_checkpoint(prefix).get_tensor(key)`); `bytes`
                                values are written as scalar DT_STRING tensors.
object_graph(keys)           -> the serialized TrackableObjectGraph proto (trackable_object_graph.proto) that
                                `tf.train.Checkpoint.restore` walks: one node per object on the variables' attribute
                                paths, `children` edges named after the path components, a VARIABLE_VALUE attribute per
                                variable, `slot_variables` for `<var>/.OPTIMIZER_SLOT/<optimizer>/<slot>` keys; stored
                                under the key `_CHECKPOINTABLE_OBJECT_GRAPH`.
parse_object_graph(bytes)    -> the node list back ({children, attributes, slots}) for tools and tests.
variable_tensors(bundle)     -> strips the `/.ATTRIBUTES/VARIABLE_VALUE` suffix (drops optimizer-slot entries), giving
                                the `gen_model/...`, `dis_model/...`, `lin/...` keys checkpoint.load_into expects.

UNPINNED against TensorFlow itself (not installable here; the reference publishes no checkpoint): written from the format's
public definition (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*, tensor_bundle.proto) and covered
by write -> read round trips, byte-level checks of the framing and hand-built blocks (prefix compression, restarts,
snappy).  CRC-32C comes from libhdrsky (`hdrsky_crc32c`).
"""
import os
import struct

import numpy as np

from . import _lib as L

MAGIC = 0xDB4775248B80FB57
SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
# tensorflow/core/framework/types.proto
_DT = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"), 6: np.dtype("i1"),
       9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"), 22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DT_OF = {v: k for k, v in _DT.items()}
DT_STRING, DT_BFLOAT16 = 7, 14


def crc32c(data, crc=0):
    buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    if buf.size == 0:
        return crc
    buf = np.ascontiguousarray(buf)
    return int(L.load().hdrsky_crc32c(buf.ctypes.data, buf.size, crc))


def mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire helpers ------------------------------------------------------------------------------------
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


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
    out, pos = {}, 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 5:
            v, pos = bytes(buf[pos:pos + 4]), pos + 4
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = bytes(buf[pos:pos + n]), pos + n
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(num, []).append(v)
    return out


def _ld(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _entry_proto(dtype, shape, offset, size, crc):
    # BundleEntryProto: 1 dtype, 2 shape{2 dim{1 size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c (fixed32)
    msg = _varint(1 << 3) + _varint(dtype)
    dims = b"".join(_ld(2, _varint(1 << 3) + _varint(int(d))) for d in shape)
    msg += _ld(2, dims)
    if offset:
        msg += _varint(4 << 3) + _varint(offset)
    msg += _varint(5 << 3) + _varint(size)
    msg += _varint((6 << 3) | 5) + struct.pack("<I", crc)
    return msg


_HEADER = _varint(1 << 3) + _varint(1) + _ld(3, _varint(1 << 3) + _varint(1))   # num_shards=1, LITTLE, version.producer=1


# ---- table (LevelDB format) -------------------------------------------------------------------------------------
def build_block(entries, restart_interval=16):
    """entries: sorted [(key bytes, value bytes)] -> block contents (without the 5-byte trailer)."""
    buf, restarts, counter, last = bytearray(), [0], 0, b""
    for key, value in entries:
        shared = 0
        if counter < restart_interval:
            m = min(len(last), len(key))
            while shared < m and last[shared] == key[shared]:
                shared += 1
        else:
            restarts.append(len(buf))
            counter = 0
        buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        last, counter = key, counter + 1
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def parse_block(contents):
    """Inverse of build_block: [(key, value)] in order."""
    (nrest,) = struct.unpack("<I", contents[-4:])
    end = len(contents) - 4 * (nrest + 1)
    if end < 0:
        raise ValueError("corrupt block (restart array)")
    out, pos, key = [], 0, b""
    while pos < end:
        shared, pos = _read_varint(contents, pos)
        non_shared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        if shared > len(key):
            raise ValueError("corrupt block (shared prefix)")
        key = key[:shared] + bytes(contents[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(contents[pos:pos + vlen])))
        pos += vlen
    return out


def snappy_decompress(src):
    """Raw snappy (the optional block compression of LevelDB tables)."""
    n, pos = _read_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(src[pos:pos + 2], "little")
            pos += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy stream")
        for _ in range(ln):               # byte-wise: copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


def _read_block(data, offset, size):
    contents, trailer = data[offset:offset + size], data[offset + size:offset + size + 5]
    if len(trailer) != 5:
        raise ValueError("truncated table")
    (crc,) = struct.unpack("<I", trailer[1:])
    if mask(crc32c(bytes(contents) + trailer[:1])) != crc:
        raise ValueError("block checksum mismatch at offset %d" % offset)
    if trailer[0] == 0:
        return bytes(contents)
    if trailer[0] == 1:
        return snappy_decompress(bytes(contents))
    raise ValueError("unknown block compression %d" % trailer[0])


def read_table(path):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != MAGIC:
        raise ValueError("%s is not a table file (bad magic)" % path)
    footer = data[-48:]
    _, p = _read_varint(footer, 0)            # metaindex handle (unused)
    _, p = _read_varint(footer, p)
    ioff, p = _read_varint(footer, p)
    isize, p = _read_varint(footer, p)
    out = []
    for _, handle in parse_block(_read_block(data, ioff, isize)):
        off, q = _read_varint(handle, 0)
        size, q = _read_varint(handle, q)
        out.extend(parse_block(_read_block(data, off, size)))
    return out


def write_table(path, entries, block_size=4096):
    """entries: [(key bytes, value bytes)] sorted by key, unique."""
    keys = [k for k, _ in entries]
    if keys != sorted(set(keys)):
        raise ValueError("table keys must be sorted and unique")
    out = bytearray()

    def emit(contents):
        handle = _varint(len(out)) + _varint(len(contents))
        out.extend(contents + b"\x00" + struct.pack("<I", mask(crc32c(contents + b"\x00"))))
        return handle
    index, pending, size = [], [], 0
    for k, v in entries:
        pending.append((k, v))
        size += len(k) + len(v) + 3
        if size >= block_size:
            index.append((pending[-1][0], emit(build_block(pending))))
            pending, size = [], 0
    if pending:
        index.append((pending[-1][0], emit(build_block(pending))))
    meta = emit(build_block([]))
    idx = emit(build_block(index, restart_interval=1))
    footer = meta + idx
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


# ---- bundle -------------------------------------------------------------------------------------------------------
def write_bundle(prefix, tensors):
    """tensors: {key: ndarray} -> <prefix>.index + <prefix>.data-00000-of-00001."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    entries, offset = [(b"", _HEADER)], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for key in sorted(tensors, key=lambda s: s.encode()):
            if not key:
                raise ValueError("the empty key is reserved for the bundle header")
            if isinstance(tensors[key], (bytes, bytearray)):
                # scalar DT_STRING (tensor_bundle.cc WriteStringTensor): [varint64 length][masked crc32c of the lengths, each
                # taken as a little-endian uint32][bytes]; the entry's checksum runs over the uint32 lengths, that 4-byte
                # length checksum and the bytes
                blob = bytes(tensors[key])
                c = crc32c(struct.pack("<I", len(blob)))
                lc = struct.pack("<I", mask(c))
                raw = _varint(len(blob)) + lc + blob
                c = crc32c(blob, crc32c(lc, c))
                f.write(raw)
                entries.append((key.encode(), _entry_proto(DT_STRING, (), offset, len(raw), mask(c))))
                offset += len(raw)
                continue
            arr = np.asarray(tensors[key])            # (ascontiguousarray would turn a scalar into shape (1,))
            dt = arr.dtype.newbyteorder("<") if arr.dtype.byteorder == ">" else arr.dtype
            if np.dtype(dt) not in _DT_OF:
                raise ValueError("unsupported dtype %s for %s" % (arr.dtype, key))
            raw = np.ascontiguousarray(arr.astype(dt, copy=False)).reshape(-1)
            f.write(raw.tobytes())
            entries.append((key.encode(), _entry_proto(_DT_OF[np.dtype(dt)], arr.shape, offset, raw.nbytes, mask(crc32c(raw)))))
            offset += raw.nbytes
    write_table(prefix + ".index", entries)


def read_string_entry(prefix, key):
    """The bytes of a scalar DT_STRING entry (e.g. the object graph), checksum verified; None when absent."""
    table = dict(read_table(prefix + ".index"))
    if key.encode() not in table:
        return None
    e = _parse(table[key.encode()])
    if e.get(1, [0])[0] != DT_STRING:
        raise ValueError("%s is not a string tensor" % key)
    nshards = _parse(table[b""]).get(1, [1])[0]
    shard, offset, size = e.get(3, [0])[0], e.get(4, [0])[0], e.get(5, [0])[0]
    with open("%s.data-%05d-of-%05d" % (prefix, shard, nshards), "rb") as f:
        f.seek(offset)
        raw = f.read(size)
    n, pos = _read_varint(raw, 0)
    lc, blob = raw[pos:pos + 4], raw[pos + 4:pos + 4 + n]
    c = crc32c(struct.pack("<I", n))
    if struct.unpack("<I", lc)[0] != mask(c) or len(blob) != n:
        raise ValueError("string tensor %s: bad length checksum" % key)
    if 6 in e and mask(crc32c(blob, crc32c(lc, c))) != struct.unpack("<I", e[6][0])[0]:
        raise ValueError("string tensor %s: checksum mismatch" % key)
    return blob


def read_bundle(prefix, verify=True):
    """{key: ndarray} of the numeric tensors of a (single- or multi-shard) bundle."""
    table = read_table(prefix + ".index")
    if not table or table[0][0] != b"":
        raise ValueError("bundle header missing")
    header = _parse(table[0][1])
    nshards = header.get(1, [1])[0]
    if header.get(2, [0])[0] != 0:
        raise ValueError("big-endian bundles are not supported")
    shards = {}
    out = {}
    for key, value in table[1:]:
        e = _parse(value)
        dtype = e.get(1, [0])[0]
        if dtype == DT_STRING or 7 in e:          # string tensors (object graph) / sliced tensors: not weights
            continue
        shape = [(_parse(d).get(1, [0])[0]) for d in _parse(e[2][0]).get(2, [])] if 2 in e else []
        shard, offset, size = e.get(3, [0])[0], e.get(4, [0])[0], e.get(5, [0])[0]
        if shard not in shards:
            shards[shard] = np.memmap("%s.data-%05d-of-%05d" % (prefix, shard, nshards), dtype=np.uint8, mode="r")
        raw = np.asarray(shards[shard][offset:offset + size])
        if raw.size != size:
            raise ValueError("tensor %s runs past the end of its shard" % key.decode())
        if verify and 6 in e and mask(crc32c(raw)) != struct.unpack("<I", e[6][0])[0]:
            raise ValueError("tensor checksum mismatch: %s" % key.decode())
        if dtype == DT_BFLOAT16:
            arr = (raw.view("<u2").astype(np.uint32) << 16).view(np.float32)
        elif dtype in _DT:
            arr = raw.view(_DT[dtype])
        else:
            raise ValueError("unsupported dtype enum %d for %s" % (dtype, key.decode()))
        out[key.decode()] = arr.reshape(shape).copy()
    return out


# ---- object graph (tensorflow/core/protobuf/trackable_object_graph.proto) ------------------------------------------
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
SLOT = "/.OPTIMIZER_SLOT/"


def _str(num, text):
    return _ld(num, text.encode())


def object_graph(keys, full_names=None):
    """keys: checkpoint keys of an object-based checkpoint (`a/b/w/.ATTRIBUTES/VARIABLE_VALUE`, optimizer slots as
    `a/b/w/.OPTIMIZER_SLOT/<optimizer path>/<slot>/.ATTRIBUTES/VARIABLE_VALUE`).  Returns the serialized
    TrackableObjectGraph: node 0 is the root (the tf.train.Checkpoint object); every path component is a child edge
    (ObjectReference{node_id, local_name}); a variable node carries SerializedTensor{name: "VARIABLE_VALUE", full_name,
    checkpoint_key}; a slot variable is a node referenced from its optimizer's node through
    SlotVariableReference{original_variable_node_id, slot_name, slot_variable_node_id} (and not through a child edge).
    Nodes are numbered breadth-first from the root with children in sorted order - any consistent numbering is valid."""
    full_names = full_names or {}
    children = [dict()]                 # node id -> {local_name: child id}
    attrs, slots = {}, {}               # node id -> SerializedTensor fields / optimizer node id -> [(orig, slot name, slot node)]

    def walk(path):
        node = 0
        for comp in path:
            nxt = children[node].get(comp)
            if nxt is None:
                nxt = len(children)
                children.append(dict())
                children[node][comp] = nxt
            node = nxt
        return node
    plain = sorted(k for k in keys if k.endswith(SUFFIX) and SLOT not in k)
    for k in plain:
        path = k[:-len(SUFFIX)]
        attrs[walk(path.split("/"))] = ("VARIABLE_VALUE", full_names.get(path, path.split("/")[-1]), k)
    for k in sorted(k for k in keys if k.endswith(SUFFIX) and SLOT in k):
        var_path, rest = k[:-len(SUFFIX)].split(SLOT)
        opt_path, slot_name = rest.rsplit("/", 1)
        var_node, opt_node = walk(var_path.split("/")), walk(opt_path.split("/"))
        slot_node = len(children)
        children.append(dict())
        attrs[slot_node] = ("VARIABLE_VALUE", "%s/%s" % (full_names.get(var_path, var_path.split("/")[-1]), slot_name), k)
        slots.setdefault(opt_node, []).append((var_node, slot_name, slot_node))
    out = b""
    for nid, ch in enumerate(children):
        node = b""
        for name in sorted(ch):
            node += _ld(1, _varint(1 << 3) + _varint(ch[name]) + _str(2, name))                       # children
        if nid in attrs:
            name, full, key = attrs[nid]
            node += _ld(2, _str(1, name) + _str(2, full) + _str(3, key))                              # attributes
        for orig, sname, snode in slots.get(nid, ()):
            node += _ld(3, _varint(1 << 3) + _varint(orig) + _str(2, sname) + _varint(3 << 3) + _varint(snode))   # slot_variables
        out += _ld(1, node)
    return out


def parse_object_graph(blob):
    """[{children: {name: id}, attributes: [(name, full_name, checkpoint_key)], slots: [(orig id, slot name, slot id)]}]"""
    nodes = []
    for raw in _parse(blob).get(1, []):
        f = _parse(raw)
        ch = {}
        for c in f.get(1, []):
            cf = _parse(c)
            ch[cf[2][0].decode()] = cf.get(1, [0])[0]
        at = [tuple(_parse(a).get(i, [b""])[0].decode() for i in (1, 2, 3)) for a in f.get(2, [])]
        sl = []
        for r in f.get(3, []):
            rf = _parse(r)
            sl.append((rf.get(1, [0])[0], rf[2][0].decode(), rf.get(3, [0])[0]))
        nodes.append(dict(children=ch, attributes=at, slots=sl))
    return nodes


def variable_tensors(bundle):
    """Object-based checkpoint keys -> attribute paths: `a/b/w/.ATTRIBUTES/VARIABLE_VALUE` -> `a/b/w`; optimizer slot
    entries (`.../.OPTIMIZER_SLOT/...`) are dropped."""
    return {k[:-len(SUFFIX)]: v for k, v in bundle.items() if k.endswith(SUFFIX) and "/.OPTIMIZER_SLOT/" not in k}


def to_variable_keys(tensors):
    """Attribute paths -> object-based checkpoint keys (the inverse of variable_tensors)."""
    return {k + SUFFIX: v for k, v in tensors.items()}
