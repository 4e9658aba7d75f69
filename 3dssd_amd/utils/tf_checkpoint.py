"""Reading (and, for tests / offline conversion, writing) TensorFlow "tensor bundle" checkpoints without TensorFlow.

SURVEY.md 8f rank 2: the reference restores its weights with `tf.train.Saver.restore(sess, path)`
(`lib/core/trainer.py:157-174`, `lib/core/trainer_utils.py:48-54`); the model zoo of its README ships such
checkpoints (`model.ckpt-N.index` + `model.ckpt-N.data-00000-of-00001`).  TensorFlow is not available here, so this
module restates the on-disk format (TensorFlow `tensor_bundle.cc` / `lib/io/table*.cc`, a LevelDB-style sorted
string table) from its published description:

  <prefix>.index   sorted table, key = variable name, value = serialized BundleEntryProto; key "" = BundleHeaderProto
      block   := entry* restart[num_restarts]:fixed32 num_restarts:fixed32 | type:u8 (0 = raw) | masked crc32c:fixed32
      entry   := shared:varint non_shared:varint value_len:varint key_delta value       (prefix-compressed keys)
      footer  := metaindex handle, index handle (offset:varint size:varint), zero padding to 40 B, magic (8 B LE)
      the index block maps separator keys to data-block handles.
  <prefix>.data-SSSSS-of-NNNNN   raw little-endian tensor bytes at [offset, offset + size) of shard `shard_id`.
  BundleEntryProto: 1 dtype, 2 shape (TensorShapeProto: repeated field 2 {1: size}), 3 shard_id, 4 offset, 5 size,
      6 crc32c (fixed32, masked), 7 slices (partitioned variables -- not supported here).

Format parity is UNPINNED against TensorFlow itself: no TensorFlow-written checkpoint exists in this environment to read.  The reader is checked
against a complete two-shard bundle assembled byte by byte from the format description inside the test (independent of the writer),
against the writer below, against the CRC-32C known-answer vectors, and against hand-assembled blocks
(tests/test_tf_checkpoint.py).  Everything is host-side numpy; the arrays go on to `VariableStore` (weights.py).
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_MASK_DELTA = 0xA282EAD8

# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in DTYPES.items()}


# ------------------------------------------------------------------------------------------- CRC-32C (Castagnoli)
def _make_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t[i] = c
    return t


_CRC_TABLE = [int(v) for v in _make_table()]


def crc32c(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for byte in bytes(data):
        c = tab[(c ^ byte) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    """crc32c::Mask: rotate right by 15 and add a constant (stored CRCs of data that itself contains CRCs)."""
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def _crc32c_array(a):
    """CRC of a whole tensor through the library's slicing-by-8 host routine (sa_host_crc32c, csrc/hostutil.hip)."""
    from . import _native
    a = np.ascontiguousarray(a)
    return int(_native.lib().sa_host_crc32c(a.ctypes.data, a.nbytes, 0))


# ------------------------------------------------------------------------------------------------ wire helpers
def _get_varint(buf, pos):
    out = shift = 0
    while True:
        byte = buf[pos]
        pos += 1
        out |= (byte & 0x7F) << shift
        if byte < 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _parse_proto(buf):
    """-> list of (field, wire_type, value); value = int for varint / fixed, bytes for length-delimited."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((field, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def parse_bundle_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for field, _, v in _parse_proto(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:                                  # Dim
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e["shape"].append(size)
                elif f2 == 3 and v2:
                    raise ValueError("tensor of unknown rank in checkpoint")
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = v
        elif field == 7:
            e["slices"] += 1
    return e


def _encode_bundle_entry(dtype_code, shape, shard_id, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(s) for s in shape))
    out = b"\x08" + _put_varint(dtype_code) + b"\x12" + _put_varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + _put_varint(shard_id)
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


# --------------------------------------------------------------------------------------------------- table read
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    if len(raw) != size or offset + size + 5 > len(data):
        raise ValueError("truncated table block")
    ctype = data[offset + size]
    stored = struct.unpack_from("<I", data, offset + size + 1)[0]
    if verify and mask_crc(crc32c(data[offset:offset + size + 1])) != stored:
        raise ValueError("checkpoint index: block checksum mismatch at offset %d" % offset)
    if ctype != 0:
        raise ValueError("compressed table blocks (type %d) are not supported" % ctype)
    return raw


def _block_entries(block):
    if len(block) < 4:
        raise ValueError("bad table block")
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    if end < 0:
        raise ValueError("bad restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key):
            raise ValueError("bad prefix compression")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a sorted string table, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad magic)" % path)
    foot = data[len(data) - 48:]
    pos = 0
    _, pos = _get_varint(foot, pos)            # metaindex handle (unused)
    _, pos = _get_varint(foot, pos)
    ioff, pos = _get_varint(foot, pos)
    isize, pos = _get_varint(foot, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


# ------------------------------------------------------------------------------------------------- bundle read
def list_variables(prefix):
    """[(name, shape, numpy dtype)] like tf.train.list_variables."""
    out = []
    for key, val in read_table(prefix + ".index"):
        if key == b"":
            continue
        e = parse_bundle_entry(val)
        out.append((key.decode(), tuple(e["shape"]), DTYPES.get(e["dtype"])))
    return out


def load_checkpoint(prefix, names=None, verify=True, skip=r"(/Adam(_\d+)?$|^beta\d_power$|/ExponentialMovingAverage$)"):
    """name -> numpy array for every numeric variable of the bundle `prefix` (or only `names`).  Optimizer slots
    (Adam moments, beta powers: the reference trains with Adam, `lib/core/trainer.py`) are skipped by default."""
    entries = read_table(prefix + ".index", verify)
    if not entries or entries[0][0] != b"":
        raise ValueError("checkpoint index without a header entry")
    hdr = {f: v for f, _, v in _parse_proto(entries[0][1])}
    num_shards = hdr.get(1, 1)
    if hdr.get(2, 0) != 0:
        raise ValueError("big-endian checkpoints are not supported")
    skip_re = re.compile(skip) if skip else None
    shards, out = {}, {}
    for key, val in entries[1:]:
        name = key.decode()
        if names is not None and name not in names:
            continue
        if names is None and skip_re is not None and skip_re.search(name):
            continue
        e = parse_bundle_entry(val)
        if e["slices"]:
            raise ValueError("partitioned variable %r is not supported" % name)
        if e["dtype"] not in DTYPES:
            if names is not None:
                raise ValueError("variable %r has unsupported dtype code %d" % (name, e["dtype"]))
            continue                                           # strings etc.: nothing the kernels consume
        if e["shard_id"] not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, e["shard_id"], num_shards)
            shards[e["shard_id"]] = np.memmap(path, dtype=np.uint8, mode="r")
        shard = shards[e["shard_id"]]
        dt = np.dtype(DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if count * dt.itemsize != e["size"] or e["offset"] + e["size"] > shard.size:
            raise ValueError("variable %r: size/shape mismatch or truncated data file" % name)
        raw = np.array(shard[e["offset"]:e["offset"] + e["size"]])
        if verify and e["crc32c"] is not None and mask_crc(_crc32c_array(raw)) != e["crc32c"]:
            raise ValueError("variable %r: data checksum mismatch" % name)
        out[name] = raw.view(dt).reshape(tuple(e["shape"])).copy()
    if names is not None:
        missing = sorted(set(names) - set(out))
        if missing:
            raise KeyError("variables not in checkpoint: %s" % ", ".join(missing[:8]))
    return out


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by the `checkpoint` state file (trainer.py:160-166)."""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state) as f:
        for line in f:
            m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
            if m:
                p = m.group(1)
                return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


# ------------------------------------------------------------------------------------------------ bundle write
def _build_block(pairs, restart_interval):
    out, restarts, prev, cnt = bytearray(), [], b"", 0
    for key, val in pairs:
        shared = 0
        if cnt % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        prev, cnt = key, cnt + 1
    if not restarts:
        restarts = [0]
    out += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, pairs, block_size=4096, restart_interval=16):
    """Sorted string table with the layout read_table expects (uncompressed blocks)."""
    pairs = sorted(pairs)
    body, index, cur, cur_size = bytearray(), [], [], 0

    def emit(block):
        off = len(body)
        body.extend(block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))

    def flush():
        nonlocal cur, cur_size
        if cur:
            index.append((cur[-1][0], emit(_build_block(cur, restart_interval))))
            cur, cur_size = [], 0

    for key, val in pairs:
        cur.append((key, val))
        cur_size += len(key) + len(val) + 3
        if cur_size >= block_size:
            flush()
    flush()
    meta = emit(_build_block([], 1))
    idx = emit(_build_block(index, 1))
    foot = meta + idx
    foot += b"\x00" * (40 - len(foot)) + struct.pack("<Q", TABLE_MAGIC)
    with open(path, "wb") as f:
        f.write(bytes(body) + foot)


def write_checkpoint(prefix, variables, state_file=True):
    """Single-shard bundle `prefix`.{index,data-00000-of-00001} from name -> array (tests, offline conversion)."""
    pairs, offset = [(b"", b"\x08\x01\x1a\x02\x08\x01")], 0          # num_shards = 1, version.producer = 1
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(variables):
            shape = np.shape(variables[name])                  # (ascontiguousarray turns 0-d into 1-d)
            a = np.ascontiguousarray(variables[name])
            if a.dtype not in _DTYPE_CODES:
                raise ValueError("dtype %s of %r cannot be stored" % (a.dtype, name))
            raw = a.tobytes()
            f.write(raw)
            pairs.append((name.encode(), _encode_bundle_entry(_DTYPE_CODES[a.dtype], shape, 0, offset, len(raw),
                                                              mask_crc(_crc32c_array(a)))))
            offset += len(raw)
    write_table(prefix + ".index", pairs)
    if state_file:
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
            base = os.path.basename(prefix)
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
