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
    # keys are stored in byte order; the header entry comes first
    keys = [k for k, _ in T.read_table(prefix + ".index")]
    assert keys[0] == b"" and keys[1:] == sorted(k.encode() for k in state)
    # corrupted tensor bytes are caught by the entry checksum
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[3] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="data checksum"):
        T.read_tf_checkpoint(prefix)
    assert sorted(T.read_tf_checkpoint(prefix, verify_checksums=False)) == sorted(state)
    # partitioned variables (entries with `slices`) and string tensors are refused, not misread
    e = T.encode_entry(1, (4,), 0, 16, 0) + T._pb_bytes(7, b"\x0a\x02\x08\x01")
    assert T.parse_entry(e)["slices"] == 1
    with pytest.raises(ValueError, match="dtype"):
        T.write_tf_checkpoint(str(tmp_path / "s"), {"s": np.asarray(["a"])})


def test_latest_tf_checkpoint_follows_the_state_file(tmp_path):
    d = str(tmp_path)
    assert T.latest_tf_checkpoint(d) is None
    for step in (10, 200, 30):
        T.write_tf_checkpoint(os.path.join(d, "model.ckpt-%d" % step), {"global_step": np.asarray(step, np.int64)})
    assert T.latest_tf_checkpoint(d) == os.path.join(d, "model.ckpt-200")
    open(os.path.join(d, "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt-30"\nall_model_checkpoint_paths: "model.ckpt-10"\n')
    assert T.latest_tf_checkpoint(d) == os.path.join(d, "model.ckpt-30")
    open(os.path.join(d, "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt-99"\n')      # stale pointer
    assert T.latest_tf_checkpoint(d) == os.path.join(d, "model.ckpt-200")
