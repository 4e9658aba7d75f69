"""TF-1 checkpoint import without TensorFlow: CRC-32C known answers (RFC 3720 B.4), a
hand-assembled table block, writer -> reader round trips (multi-block index, optimizer slots, scalars, integer
tensors), corruption detection, and the reference's variable names through VariableStore.  Runs on CPU; the library
is only loaded for its host-side CRC routine."""
import os
import struct

import numpy as np
import pytest

from conftest import pkg


def test_crc32c_known_answers_python_and_native():
    C, N = pkg("utils.tf_checkpoint"), pkg("utils._native")
    kats = [(b"123456789", 0xE3069283), (bytes(32), 0x8A9136AA), (b"\xff" * 32, 0x62A8AB43),
            (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C), (b"", 0)]
    for data, want in kats:
        assert C.crc32c(data) == want
        buf = np.frombuffer(data, np.uint8) if data else np.zeros(0, np.uint8)
        assert N.lib().sa_host_crc32c(buf.ctypes.data, buf.size, 0) == want
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, 100003, dtype=np.uint8)
    for off, ln in ((0, 100003), (1, 77), (3, 4096), (5, 8), (7, 99991)):
        view = big[off:off + ln]
        assert N.lib().sa_host_crc32c(view.ctypes.data, view.size, 0) == C.crc32c(view.tobytes())
    # continuation: crc(a + b) == crc(b, seed = crc(a))
    a, b = big[:1000], big[1000:5000]
    part = N.lib().sa_host_crc32c(a.ctypes.data, a.size, 0)
    assert N.lib().sa_host_crc32c(b.ctypes.data, b.size, part) == C.crc32c(big[:5000].tobytes())
    assert C.mask_crc(0) == 0xA282EAD8 and C.mask_crc(0x00008000) == (0xA282EAD8 + 1) & 0xFFFFFFFF


def test_hand_assembled_block_with_prefix_compression():
    C = pkg("utils.tf_checkpoint")
    # entries: ("layer1/a","X"), ("layer1/b","YZ") sharing 7 key bytes, ("m","") ; one restart point at 0
    blk = (bytes([0, 8, 1]) + b"layer1/a" + b"X" + bytes([7, 1, 2]) + b"b" + b"YZ" + bytes([0, 1, 0]) + b"m" +
           struct.pack("<II", 0, 1))
    assert list(C._block_entries(blk)) == [(b"layer1/a", b"X"), (b"layer1/b", b"YZ"), (b"m", b"")]
    # BundleEntryProto by hand: dtype 1 (float), shape [3,5], offset 12, size 60, crc 0xDEADBEEF
    ent = (b"\x08\x01" + b"\x12\x08" + b"\x12\x02\x08\x03" + b"\x12\x02\x08\x05" + b"\x20\x0c" + b"\x28\x3c" +
           b"\x35" + struct.pack("<I", 0xDEADBEEF))
    e = C.parse_bundle_entry(ent)
    assert (e["dtype"], e["shape"], e["shard_id"], e["offset"], e["size"], e["crc32c"]) == (1, [3, 5], 0, 12, 60, 0xDEADBEEF)
    assert C._encode_bundle_entry(1, (3, 5), 0, 12, 60, 0xDEADBEEF) == ent
    assert C._get_varint(C._put_varint(300), 0) == (300, 2) and C._put_varint(300) == b"\xac\x02"


def _variables(rng, n_layers=40):
    v = {"global_step": np.array(80000, np.int64), "beta1_power": np.array(0.5, np.float32)}
    for i in range(n_layers):
        sc = "layer%d/conv0_%d" % (i // 3 + 1, i % 3)
        v[sc + "/weights"] = rng.normal(0, 1, (1, 1, 7 + i, 16)).astype(np.float32)
        v[sc + "/biases"] = rng.normal(0, 1, 16).astype(np.float32)
        v[sc + "/weights/Adam"] = np.zeros((1, 1, 7 + i, 16), np.float32)
        v[sc + "/weights/Adam_1"] = np.ones((1, 1, 7 + i, 16), np.float32)
        for k in ("gamma", "beta", "moving_mean", "moving_variance"):
            v[sc + "/bn/" + k] = rng.uniform(0.5, 1.5, 16).astype(np.float32)
    v["counts"] = rng.integers(-5, 5, (3, 0, 2)).astype(np.int32)          # an empty tensor
    v["half"] = rng.normal(0, 1, (4, 4)).astype(np.float16)
    return v


def test_round_trip_with_own_writer_multi_block_and_filters(tmp_path):
    C = pkg("utils.tf_checkpoint")
    rng = np.random.default_rng(1)
    v = _variables(rng)
    prefix = str(tmp_path / "ckpt" / "model-80000")
    C.write_checkpoint(prefix, v)
    assert os.path.getsize(prefix + ".index") > 2 * 4096                     # several data blocks behind the index block
    listed = C.list_variables(prefix)
    assert [n for n, _, _ in listed] == sorted(v)
    assert dict((n, s) for n, s, _ in listed)["layer1/conv0_0/weights"] == (1, 1, 7, 16)
    got = C.load_checkpoint(prefix)
    kept = {k for k in v if "/Adam" not in k and k != "beta1_power"}
    assert set(got) == kept
    for k in kept:
        assert got[k].dtype == v[k].dtype and got[k].shape == v[k].shape and np.array_equal(got[k], v[k])
    assert got["global_step"].shape == () and int(got["global_step"]) == 80000
    full = C.load_checkpoint(prefix, skip=None)
    assert set(full) == set(v)
    one = C.load_checkpoint(prefix, names=["layer2/conv0_1/weights/Adam_1"])
    assert list(one) == ["layer2/conv0_1/weights/Adam_1"] and (one["layer2/conv0_1/weights/Adam_1"] == 1).all()
    with pytest.raises(KeyError):
        C.load_checkpoint(prefix, names=["no/such/variable"])
    assert C.latest_checkpoint(str(tmp_path / "ckpt")) == prefix
    assert C.latest_checkpoint(str(tmp_path)) is None


def test_corruption_of_files_from_own_writer_is_detected(tmp_path):
    C = pkg("utils.tf_checkpoint")
    v = _variables(np.random.default_rng(2), n_layers=6)
    prefix = str(tmp_path / "model")
    C.write_checkpoint(prefix, v, state_file=False)
    data_path, index_path = prefix + ".data-00000-of-00001", prefix + ".index"
    raw = bytearray(open(data_path, "rb").read())
    raw[len(raw) // 2] ^= 0x10
    open(data_path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        C.load_checkpoint(prefix, skip=None)                                  # (the byte may sit in an optimizer slot)
    assert len(C.load_checkpoint(prefix, verify=False)) > 0                   # explicit opt-out still parses
    raw[len(raw) // 2] ^= 0x10
    open(data_path, "wb").write(bytes(raw[:-8]))                             # truncated data file
    with pytest.raises(ValueError):
        C.load_checkpoint(prefix, skip=None)
    open(data_path, "wb").write(bytes(raw))
    idx = bytearray(open(index_path, "rb").read())
    idx[40] ^= 0x01
    open(index_path, "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        C.load_checkpoint(prefix)
    idx[40] ^= 0x01
    idx[-1] ^= 0xFF                                                           # magic
    open(index_path, "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="magic"):
        C.load_checkpoint(prefix)


def test_backbone_variables_through_a_checkpoint_from_own_writer(tmp_path):
    """The reference's variable names (layers_util.py:175, tf_util.py:96,111,439-442) written as a checkpoint with
    Adam slots, read back, folded: bitwise the same (W', b') as from the in-memory dict."""
    C, W = pkg("utils.tf_checkpoint"), pkg("utils.weights")
    cfgs, syn = pkg("configs"), pkg("synthetic")
    params = syn.random_backbone_params(cfgs.KITTI_3DSSD_ARCH)
    saved = dict(params)
    for k in list(params):
        if k.endswith("/weights") or k.endswith("/biases"):
            saved[k + "/Adam"] = np.zeros_like(params[k])
            saved[k + "/Adam_1"] = np.zeros_like(params[k])
    saved["global_step"] = np.array(123, np.int64)
    C.write_checkpoint(str(tmp_path / "model-123"), saved)
    store = W.VariableStore.from_checkpoint(str(tmp_path), "cpu")
    assert set(store.params) == set(params) | {"global_step"}
    for scope in ("layer1/conv0_0", "layer4/conv1_2", "vote/vote_layer_0"):
        a, b = W.fold_conv_bn(store.params, scope), W.fold_conv_bn(params, scope)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    w, b = W.fold_conv_bn(store.params, "vote/vote_offsets", bn=False)
    assert w.shape[1] == 3


def test_hand_assembled_two_shard_bundle_with_restart_points(tmp_path):
    """A complete bundle assembled BYTE BY BYTE here from the published format (TensorFlow tensor_bundle.cc /
    table_builder.cc), without any of the module's writer functions: a data block whose second restart point starts an
    uncompressed key, keys prefix-compressed against their predecessor in between, a header entry announcing TWO
    shards, entries that point into `.data-00000-of-00002` and `.data-00001-of-00002`, an index block with one
    separator key, the 48-byte footer.  Pins the READER against the spec as far as this sandbox allows (no
    TensorFlow-written file exists here; checksums use the module's CRC-32C, itself pinned by the RFC 3720 vectors)."""
    C = pkg("utils.tf_checkpoint")

    def varint(v):
        out = bytearray()
        while True:
            b = v & 0x7F
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)

    def entry_proto(dtype, shape, shard, offset, size, crc):
        dims = b"".join(b"\x12" + varint(len(d)) + d for d in (b"\x08" + varint(x) for x in shape))
        out = b"\x08" + varint(dtype) + b"\x12" + varint(len(dims)) + dims
        if shard:
            out += b"\x18" + varint(shard)
        if offset:
            out += b"\x20" + varint(offset)
        return out + b"\x28" + varint(size) + b"\x35" + struct.pack("<I", crc)

    def masked(raw):
        c = C.crc32c(raw)
        return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF

    a = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.5              # shard 0, offset 0
    b = np.array([7, -3, 2 ** 40], np.int64)                             # shard 1, offset 16 (after 16 junk bytes)
    c = np.array(2.5, np.float32)                                        # shard 0, offset 48, scalar (no dims)
    shard0 = a.tobytes() + c.tobytes()
    shard1 = b"\xEE" * 16 + b.tobytes()
    vals = {b"": b"\x08\x02\x1a\x02\x08\x01",                          # BundleHeaderProto: num_shards 2, version{producer 1}
            b"layer1/conv0_0/biases": entry_proto(1, [3, 4], 0, 0, 48, masked(a.tobytes())),
            b"layer1/conv0_0/steps": entry_proto(9, [3], 1, 16, 24, masked(b.tobytes())),
            b"layer2/scale": entry_proto(1, [], 0, 48, 4, masked(c.tobytes()))}
    keys = sorted(vals)
    # data block, restart interval 2: entries 0 and 2 are restart points (full keys), 1 and 3 prefix-compressed
    blk, restarts, prev = bytearray(), [], b""
    for i, k in enumerate(keys):
        shared = 0
        if i % 2 == 0:
            restarts.append(len(blk))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        blk += varint(shared) + varint(len(k) - shared) + varint(len(vals[k])) + k[shared:] + vals[k]
        prev = k
    assert restarts[1] > 0 and blk[restarts[1]] == 0                      # the second restart entry shares 0 bytes
    blk += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    data_block = bytes(blk)
    f = bytearray(data_block + b"\x00" + struct.pack("<I", masked(data_block + b"\x00")))
    meta_off = len(f)                                                     # empty metaindex block
    meta = struct.pack("<II", 0, 1)
    f += meta + b"\x00" + struct.pack("<I", masked(meta + b"\x00"))
    idx_off = len(f)
    handle = varint(0) + varint(len(data_block))
    sep = b"m"                                                            # any key >= the last key of the data block
    iblk = varint(0) + varint(len(sep)) + varint(len(handle)) + sep + handle + struct.pack("<II", 0, 1)
    f += iblk + b"\x00" + struct.pack("<I", masked(iblk + b"\x00"))
    foot = varint(meta_off) + varint(len(meta)) + varint(idx_off) + varint(len(iblk))
    f += foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", 0xDB4775248B80FB57)
    prefix = str(tmp_path / "model.ckpt-7")
    open(prefix + ".index", "wb").write(bytes(f))
    open(prefix + ".data-00000-of-00002", "wb").write(shard0)
    open(prefix + ".data-00001-of-00002", "wb").write(shard1)
    assert [k for k, _ in C.read_table(prefix + ".index")] == keys
    assert C.list_variables(prefix) == [("layer1/conv0_0/biases", (3, 4), np.float32), ("layer1/conv0_0/steps", (3,), np.int64),
                                        ("layer2/scale", (), np.float32)]
    got = C.load_checkpoint(prefix)
    assert np.array_equal(got["layer1/conv0_0/biases"], a) and np.array_equal(got["layer1/conv0_0/steps"], b)
    assert got["layer2/scale"].shape == () and got["layer2/scale"] == np.float32(2.5)
    # a flipped byte in the second shard is caught by the per-tensor checksum
    bad = bytearray(shard1)
    bad[20] ^= 1
    open(prefix + ".data-00001-of-00002", "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="checksum"):
        C.load_checkpoint(prefix)
