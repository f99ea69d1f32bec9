"""CPU: wide_deep_amd/tf_checkpoint.py -- TensorFlow's checkpoint container (leveldb-format index table + raw data shard).

No TF-written file exists offline, so the CONTAINER is pinned against itself only (write -> read, hand-built blocks); its
primitives are pinned against published known answers: CRC-32C vectors of RFC 3720 / leveldb's crc32c_test.cc, the snappy
format description, protobuf varints."""
import os
import struct

import numpy as np
import pytest

from wide_deep_amd import tf_checkpoint as T


def test_crc32c_known_answers_and_mask():
    assert T.crc32c(b"123456789") == 0xE3069283                       # the standard check value of CRC-32C
    assert T.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4 / leveldb crc32c_test.cc
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert T.crc32c(b"world", T.crc32c(b"hello ")) == T.crc32c(b"hello world")      # leveldb: Extend
    c = T.crc32c(b"foo")
    assert T.mask_crc(c) != c and T.mask_crc(T.mask_crc(c)) != c
    assert T.unmask_crc(T.mask_crc(c)) == c and T.unmask_crc(T.unmask_crc(T.mask_crc(T.mask_crc(c)))) == c


def test_varints():
    for v, enc in ((0, b"\x00"), (1, b"\x01"), (127, b"\x7f"), (128, b"\x80\x01"), (300, b"\xac\x02"),
                   ((1 << 32) - 1, b"\xff\xff\xff\xff\x0f"), ((1 << 64) - 1, b"\xff" * 9 + b"\x01")):
        assert T.put_varint(v) == enc
        assert T.get_varint(enc + b"zz", 0) == (v, len(enc))
    with pytest.raises(ValueError):
        T.get_varint(b"\xff" * 11, 0)


def test_snappy_streams_built_from_the_format_description():
    # literal 'abc' + copy (2-byte offset) of 9 bytes from 3 back: an overlapping copy = run-length expansion
    s = bytes([12, (3 - 1) << 2]) + b"abc" + bytes([((9 - 1) << 2) | 2, 3, 0])
    assert T.snappy_uncompress(s) == b"abcabcabcabc"
    # copy with 1-byte offset: length 5 (4 + 1), offset 3
    s = bytes([8, (3 - 1) << 2]) + b"xyz" + bytes([((5 - 4) << 2) | 1, 3])
    assert T.snappy_uncompress(s) == b"xyzxyzxy"
    # long literal: 100 bytes -> tag 60 << 2, one extra length byte (len - 1)
    body = bytes(range(100))
    assert T.snappy_uncompress(bytes([100, 60 << 2, 99]) + body) == body
    # 4-byte offset copy
    s = bytes([6, (3 - 1) << 2]) + b"abc" + bytes([((3 - 1) << 2) | 3, 3, 0, 0, 0])
    assert T.snappy_uncompress(s) == b"abcabc"
    with pytest.raises(ValueError):
        T.snappy_uncompress(bytes([5, (3 - 1) << 2]) + b"abc")          # declared length 5, produced 3
    with pytest.raises(ValueError):
        T.snappy_uncompress(bytes([4, ((4 - 4) << 2) | 1, 1]))          # copy before any output


def test_table_round_trip_prefix_compression_many_blocks_and_a_compressed_block(tmp_path):
    keys = sorted({("dnn/dnn_1/hiddenlayer_%d/%s%s" % (i, k, s)).encode() for i in range(40) for k in ("kernel", "bias")
                   for s in ("", "/Adagrad")} | {b""})
    entries = [(k, (b"v" + k) * (1 + i % 5)) for i, k in enumerate(keys)]
    p = str(tmp_path / "t.index")
    T.write_table(p, entries, block_size=512)
    got = T.read_table(p)
    assert got == entries and len(open(p, "rb").read()) < sum(len(k) + len(v) for k, v in entries) + 4096
    # a flipped byte in a data block is caught by the block checksum
    raw = bytearray(open(p, "rb").read())
    raw[10] ^= 0x40
    open(p, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        T.read_table(p)
    assert T.read_table(p, verify_checksums=False) != entries
    with pytest.raises(ValueError, match="magic"):
        open(p, "wb").write(b"x" * 100)
        T.read_table(p)
    # a snappy-compressed data block (type byte 1), as TF's table builder may write: all-literal stream of the raw block
    blk = T._build_block(entries[:7])
    comp = T.put_varint(len(blk)) + bytes([60 << 2 | 0 if False else (61 << 2)]) + struct.pack("<H", len(blk) - 1) + blk
    f = bytearray(comp) + bytes([1]) + struct.pack("<I", T.mask_crc(T.crc32c(comp + b"\x01")))
    idx = T._build_block([(entries[6][0], T.put_varint(0) + T.put_varint(len(comp)))], restart_interval=1)
    ioff = len(f)
    f += idx + b"\0" + struct.pack("<I", T.mask_crc(T.crc32c(idx + b"\0")))
    meta = T._build_block([])
    moff = len(f)
    f += meta + b"\0" + struct.pack("<I", T.mask_crc(T.crc32c(meta + b"\0")))
    footer = T.put_varint(moff) + T.put_varint(len(meta)) + T.put_varint(ioff) + T.put_varint(len(idx))
    f += footer + b"\0" * (40 - len(footer)) + struct.pack("<Q", T.MAGIC)
    open(p, "wb").write(bytes(f))
    assert T.read_table(p) == entries[:7]


def test_bundle_round_trip_with_the_engines_variable_names(tmp_path):
    rng = np.random.default_rng(3)
    state = {
        "dnn/input_from_feature_columns/input_layer/C00_embedding/embedding_weights": rng.standard_normal((50, 16)).astype(np.float32),
        "dnn/input_from_feature_columns/input_layer/C00_embedding/embedding_weights/Adagrad": np.full((50, 16), 0.1, np.float32),
        "dnn/dnn_1/hiddenlayer_0/kernel": rng.standard_normal((29, 8)).astype(np.float32),
        "dnn/dnn_1/hiddenlayer_0/bias": np.zeros(8, np.float32),
        "dnn/dnn_1/logits/kernel": rng.standard_normal((8, 1)).astype(np.float32),
        "linear/linear_model/C00/weights": rng.standard_normal((50, 1)).astype(np.float32),
        "linear/linear_model/bias_weights": np.asarray([0.25], np.float32),
        "global_step": np.asarray(1234, dtype=np.int64),
        "beta1_power": np.asarray(0.9 ** 7, dtype=np.float32),
        "flags": np.asarray([True, False, True]),
        "d": rng.standard_normal(5),
    }
    prefix = str(tmp_path / "m" / "model.ckpt-1234")
    assert T.write_tf_checkpoint(prefix, state) == prefix
    assert os.path.isfile(prefix + ".index") and os.path.isfile(prefix + ".data-00000-of-00001")
    got = T.read_tf_checkpoint(prefix)
    assert sorted(got) == sorted(state)
    for k, v in state.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert got["global_step"].shape == () and int(got["global_step"]) == 1234
    # keys are stored in byte order; the header entry comes first; the variables of the dnn / linear scopes are stored the way
    # the reference's partitioner scopes make TensorFlow store them: an entry with the full shape + a slice, the data under the slice key
    table = T.read_table(prefix + ".index")
    keys = [k for k, _ in table]
    assert keys[0] == b"" and keys[1:] == sorted(keys[1:])
    assert sorted(k for k in keys[1:] if k[:1] != b"\0") == sorted(k.encode() for k in state)
    ent = {k: T.parse_entry(v) for k, v in table[1:]}
    nm = b"dnn/dnn_1/hiddenlayer_0/kernel"
    assert ent[nm]["slices"] == [[(0, 29), (0, 8)]] and ent[nm]["shape"] == (29, 8) and ent[nm]["size"] == 0
    sk = T.encode_tensor_name_slice(nm, [(0, 29), (0, 8)])
    assert ent[sk]["shape"] == (29, 8) and ent[sk]["size"] == 29 * 8 * 4 and not ent[sk]["slices"]
    assert not ent[b"global_step"]["slices"] and not ent[b"d"]["slices"]
    # the state file tf.train.latest_checkpoint reads
    assert open(str(tmp_path / "m" / "checkpoint")).read().startswith('model_checkpoint_path: "model.ckpt-1234"')
    # corrupted tensor bytes are caught by the entry checksum
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[3] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="data checksum"):
        T.read_tf_checkpoint(prefix)
    assert sorted(T.read_tf_checkpoint(prefix, verify_checksums=False)) == sorted(state)
    # string tensors are refused, not misread
    with pytest.raises(ValueError, match="dtype"):
        T.write_tf_checkpoint(str(tmp_path / "s"), {"s": np.asarray(["a"])})


def test_latest_tf_checkpoint_follows_the_state_file(tmp_path):
    d = str(tmp_path)
    assert T.latest_tf_checkpoint(d) is None
    for step in (10, 200, 30):
        T.write_tf_checkpoint(os.path.join(d, "model.ckpt-%d" % step), {"global_step": np.asarray(step, np.int64)}, state_file=False)
    assert T.latest_tf_checkpoint(d) == os.path.join(d, "model.ckpt-200")         # no state file: the highest step
    T.write_tf_checkpoint(os.path.join(d, "model.ckpt-30"), {"global_step": np.asarray(30, np.int64)})
    assert T.latest_tf_checkpoint(d) == os.path.join(d, "model.ckpt-30")          # the writer's own state file points at its prefix
    open(os.path.join(d, "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt-30"\nall_model_checkpoint_paths: "model.ckpt-10"\n')
    assert T.latest_tf_checkpoint(d) == os.path.join(d, "model.ckpt-30")
    open(os.path.join(d, "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt-99"\n')      # stale pointer
    assert T.latest_tf_checkpoint(d) == os.path.join(d, "model.ckpt-200")


def test_ordered_code_and_slice_keys_hand_assembled():
    """tensorflow/core/lib/strings/ordered_code.cc, restated: NumIncreasing = length byte + big-endian bytes; String = escaped
    + 00 01; SignedNumIncreasing = one byte 0x80 ^ v for -64 <= v < 64, else ceil-ish(bits / 7) + 1 bytes of the two's complement
    with a unary length header (0xc0 for 2 bytes, 0xe0 for 3 ...; inverted for negatives)."""
    assert T._oc_num_increasing(0) == b"\x00" and T._oc_num_increasing(2) == b"\x01\x02" and T._oc_num_increasing(0x1234) == b"\x02\x12\x34"
    assert T._oc_string(b"a\x00b\xffc") == b"a\x00\xffb\xff\x00c\x00\x01"
    for v, enc in ((0, "80"), (1, "81"), (63, "bf"), (-1, "7f"), (-64, "40"), (64, "c040"), (-65, "3fbf"), (8191, "dfff"), (8192, "e02000"),
                   (1000000, "ef4240"), (-1000000, "10bdc0")):
        assert T._oc_signed_num_increasing(v).hex() == enc, v
    # increasing: the byte strings order like the numbers
    vals = [-(1 << 40), -70000, -8193, -8192, -65, -64, -1, 0, 1, 63, 64, 8191, 8192, 1 << 20, 1 << 40]
    encs = [T._oc_signed_num_increasing(v) for v in vals]
    assert encs == sorted(encs)
    # the key of slice [0, 1000000) x [0, 16) of "a/b": 0 | "a/b" | rank 2 | (0, 1000000) | (0, 16)
    assert T.encode_tensor_name_slice("a/b", [(0, 1000000), (0, 16)]).hex() == "00" + "612f62" + "0001" + "0102" + "80" + "ef4240" + "80" + "90"
    # a slice that takes a whole dimension is written with start 0 ... here as TF's TensorSlice (-1 = full) would: (0, -1) -> 80 7f
    assert T.encode_tensor_name_slice("v", [(0, -1)]).hex() == "00" + "76" + "0001" + "0101" + "80" + "7f"


def test_partitioned_variables_multi_slice_round_trip_and_missing_slice(tmp_path):
    rng = np.random.default_rng(5)
    state = {"linear/linear_model/C01/weights": rng.standard_normal((1001, 1)).astype(np.float32),
             "dnn/input_from_feature_columns/input_layer/C01_embedding/embedding_weights": rng.standard_normal((1001, 8)).astype(np.float32),
             "dnn/dnn_1/hiddenlayer_0/bias": rng.standard_normal(7).astype(np.float32),
             "global_step": np.asarray(9, np.int64)}
    prefix = str(tmp_path / "model.ckpt-9")
    T.write_tf_checkpoint(prefix, state, partitions=3)
    table = dict(T.read_table(prefix + ".index"))
    e = T.parse_entry(table[b"linear/linear_model/C01/weights"])
    assert e["slices"] == [[(0, 333), (0, 1)], [(333, 334), (0, 1)], [(667, 334), (0, 1)]] and e["shape"] == (1001, 1)
    assert T.parse_entry(table[b"dnn/dnn_1/hiddenlayer_0/bias"])["slices"] == [[(0, 2)], [(2, 2)], [(4, 3)]]
    got = T.read_tf_checkpoint(prefix)
    for k, v in state.items():
        assert np.array_equal(got[k], v), k
    # an index without one of the slices' data entries is refused
    pairs = [(k, v) for k, v in T.read_table(prefix + ".index")
             if k != T.encode_tensor_name_slice("linear/linear_model/C01/weights", [(333, 334), (0, 1)])]
    T.write_table(prefix + ".index", pairs)
    with pytest.raises(ValueError, match="slice"):
        T.read_tf_checkpoint(prefix)


def test_large_tensors_carry_a_checksum(tmp_path):
    """tensors above 4 MB used to be written with crc32c = 0, which TensorFlow's BundleReader rejects."""
    a = np.arange(3 << 20, dtype=np.float32).reshape(-1, 16)         # 12 MB
    prefix = str(tmp_path / "model.ckpt-1")
    T.write_tf_checkpoint(prefix, {"dnn/t/embedding_weights": a})
    table = dict(T.read_table(prefix + ".index"))
    sk = T.encode_tensor_name_slice("dnn/t/embedding_weights", [(0, a.shape[0]), (0, 16)])
    e = T.parse_entry(table[sk])
    assert e["crc32c"] != 0 and T.unmask_crc(e["crc32c"]) == T.crc32c(a.tobytes())
    assert np.array_equal(T.read_tf_checkpoint(prefix)["dnn/t/embedding_weights"], a)
