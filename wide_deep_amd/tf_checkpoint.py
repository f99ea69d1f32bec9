"""Reader / writer of TensorFlow's checkpoint container (tensor bundle V2: `<prefix>.index` + `<prefix>.data-00000-of-00001`).

SURVEY section 8 row f3 ("optional TF-checkpoint reader for importing reference-trained weights", python/train.py:188-191:
`keep_train` resumes from whatever tf.estimator left in model_dir).  TensorFlow cannot be installed here and no TF-written
checkpoint exists offline, so **the format is restated from its published description and is NOT pinned against a file TF wrote**
(parity unpinned; tests/test_tf_checkpoint.py pins the primitives -- CRC-32C, varints, snappy -- against their published
known answers and the container against itself):

  * `.index` is a LevelDB-format sorted string table (tensorflow/core/lib/io/table_*.cc = leveldb table format): data blocks
    of prefix-compressed (shared, non_shared, value_len varint32s + key suffix + value) entries with a restart array, each
    block followed by a 1-byte compression type (0 raw, 1 snappy) and a masked CRC-32C; a metaindex block, an index block
    (last key of each data block -> BlockHandle{offset, size} varint64s) and a 48-byte footer ending in the magic
    0xdb4775248b80fb57.
  * key "" -> BundleHeaderProto {num_shards = 1, endianness = 2, version = 3}; every other key is a variable name ->
    BundleEntryProto {dtype = 1, shape = 2 (TensorShapeProto: repeated dim = 2 {size = 1}), shard_id = 3, offset = 4,
    size = 5, crc32c = 6 (fixed32, masked), slices = 7}  (tensorflow/core/protobuf/tensor_bundle.proto).
  * `.data-SSSSS-of-NNNNN` holds the raw little-endian tensor bytes at [offset, offset + size).

  * PARTITIONED variables -- every variable of the reference's `dnn` and `linear` scopes, which are created under a partitioner
    (python/lib/joint.py:140-143, 165-168, 186-189), even with a single partition -- are stored as SLICES: the entry under the
    variable's full name carries dtype, the FULL shape and `slices` (repeated TensorSliceProto {repeated Extent {start = 1,
    length = 2}}) and no data; each slice's data sits under the key  EncodeTensorNameSlice(full name, slice)
    (tensorflow/core/util/saved_tensor_slice_util.cc) = OrderedCode[ NumIncreasing(0), String(name), NumIncreasing(rank),
    then per dimension SignedNumIncreasing(start), SignedNumIncreasing(length) ]  with its own BundleEntryProto (shape = the
    slice's).  The reader assembles the full tensor; the writer stores the `dnn/` and `linear/` variables that way (one slice
    covering the variable, or `partitions` row ranges).  Optimizer slots of a partitioned variable are sliced under
    "<full name>/<slot>" (tf.train slot_creator), i.e. the names export_state() already uses.

Only what the reference's checkpoints contain is handled: float / double / int32 / int64 / bool tensors.  Variable names are the
reference's, i.e. exactly the keys of WideDeepEngine.export_state() / import_state().
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 10: np.dtype("bool")}
_DTYPE_IDS = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9,
              np.dtype("bool"): 10}


# ---- primitives ---------------------------------------------------------------------------------------------------------------
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC = _crc_table()


_native_crc = None


def _native():
    """wd_crc32c of libwd_ingest.so (slice-by-8 in C: gigabytes of embedding tables per second), or False."""
    global _native_crc
    if _native_crc is None:
        _native_crc = False
        try:
            import ctypes
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libwd_ingest.so")
            if os.path.exists(path):
                fn = ctypes.CDLL(path).wd_crc32c
                fn.restype = ctypes.c_uint32
                fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32]
                _native_crc = fn
        except (OSError, AttributeError):
            _native_crc = False
    return _native_crc


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), the checksum of leveldb tables and bundle entries."""
    fn = _native()
    if fn and len(data) >= 64:
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        return int(fn(a.ctypes.data, a.size, crc))
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = _CRC[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - 0xa282ead8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


def put_varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def get_varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def snappy_uncompress(src):
    """Raw snappy block format (https://github.com/google/snappy/blob/main/format_description.txt)."""
    src = bytes(src)
    n, pos = get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                   # copy, 1-byte offset
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:                                 # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 2], "little")
            pos += 2
        else:                                           # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("snappy: bad copy offset")
        for _ in range(ln):                             # byte-wise: source and destination may overlap
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch (%d != %d)" % (len(out), n))
    return bytes(out)


# ---- minimal protobuf wire format -----------------------------------------------------------------------------------------------
def _pb_fields(buf):
    """yield (field number, wire type, value) of one message; value = int (varint / fixed) or bytes (length-delimited)."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = get_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _pb_varint(num, v):
    return put_varint(num << 3) + put_varint(v)


def _pb_bytes(num, b):
    return put_varint((num << 3) | 2) + put_varint(len(b)) + b


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
    dims = []
    for num, wt, v in _pb_fields(buf):
        if num == 2 and wt == 2:                        # repeated Dim dim = 2
            size = 0
            for n2, w2, v2 in _pb_fields(v):
                if n2 == 1 and w2 == 0:
                    size = _signed64(v2)
            dims.append(size)
        elif num == 3 and wt == 0 and v:
            raise ValueError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def _encode_shape(shape):
    return b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)


# ---- OrderedCode (tensorflow/core/lib/strings/ordered_code.cc) and slice keys -----------------------------------------------------
def _oc_num_increasing(v):
    """WriteNumIncreasing: one length byte n (0..8) + the n significant big-endian bytes of the unsigned value."""
    b = int(v).to_bytes(8, "big").lstrip(b"\0")
    return bytes([len(b)]) + b


def _oc_string(sv):
    """WriteString: 0x00 -> 00 ff, 0xff -> ff 00, terminated by 00 01."""
    return b"".join((b"\x00\xff" if c == 0 else b"\xff\x00" if c == 0xff else bytes([c])) for c in bytes(sv)) + b"\x00\x01"


_OC_HEADER = [(0, 0), (0x80, 0), (0xc0, 0), (0xe0, 0), (0xf0, 0), (0xf8, 0), (0xfc, 0), (0xfe, 0), (0xff, 0), (0xff, 0x80),
              (0xff, 0xc0)]


def _oc_signed_num_increasing(val):
    """WriteSignedNumIncreasing: the value's two's-complement big-endian bytes, shortened to len = bits / 7 + 1 bytes, the
    leading bits replaced by a unary length header (x < 64: the single byte 0x80 ^ val)."""
    val = int(val)
    x = ~val if val < 0 else val
    if x < 64:
        return bytes([(0x80 ^ val) & 0xff])
    bits = x.bit_length()
    ln = bits // 7 + 1
    buf = bytearray((b"\xff" if val < 0 else b"\x00") * 2 + (val & ((1 << 64) - 1)).to_bytes(8, "big"))
    begin = len(buf) - ln
    buf[begin] ^= _OC_HEADER[ln][0]
    buf[begin + 1] ^= _OC_HEADER[ln][1]
    return bytes(buf[begin:])


def encode_tensor_name_slice(name, extents):
    """Key of a slice's data entry; extents = [(start, length)] per dimension ((0, -1) = the whole dimension)."""
    out = _oc_num_increasing(0) + _oc_string(name.encode() if isinstance(name, str) else name) + _oc_num_increasing(len(extents))
    for start, length in extents:
        out += _oc_signed_num_increasing(start) + _oc_signed_num_increasing(length)
    return out


def _parse_slice(buf):
    """TensorSliceProto -> [(start, length)], length -1 = the full dimension."""
    ext = []
    for num, wt, v in _pb_fields(buf):
        if num == 1 and wt == 2:
            start, length = 0, -1
            for n2, w2, v2 in _pb_fields(v):
                if n2 == 1 and w2 == 0:
                    start = _signed64(v2)
                elif n2 == 2 and w2 == 0:
                    length = _signed64(v2)
            ext.append((start, length))
    return ext


def _encode_slice(extents):
    out = b""
    for start, length in extents:
        e = (_pb_varint(1, start) if start else b"") + (_pb_varint(2, length) if length >= 0 else b"")
        out += _pb_bytes(1, e)
    return out


def parse_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": []}
    for num, wt, v in _pb_fields(buf):
        if num == 1 and wt == 0:
            e["dtype"] = v
        elif num == 2 and wt == 2:
            e["shape"] = _parse_shape(v)
        elif num == 3 and wt == 0:
            e["shard_id"] = v
        elif num == 4 and wt == 0:
            e["offset"] = v
        elif num == 5 and wt == 0:
            e["size"] = v
        elif num == 6 and wt == 5:
            e["crc32c"] = v
        elif num == 7 and wt == 2:
            e["slices"].append(_parse_slice(v))
    return e


def encode_entry(dtype_id, shape, offset, size, crc_masked, slices=None):
    out = _pb_varint(1, dtype_id) + _pb_bytes(2, _encode_shape(shape))
    if slices:          # the entry of a partitioned variable: full shape + its slices, no data of its own
        return out + b"".join(_pb_bytes(7, _encode_slice(sl)) for sl in slices)
    if offset:
        out += _pb_varint(4, offset)
    out += _pb_varint(5, size) + put_varint((6 << 3) | 5) + struct.pack("<I", crc_masked)
    return out


# ---- leveldb-format table -----------------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify):
    raw = data[offset: offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if unmask_crc(stored) != crc32c(data[offset: offset + size + 1]):
            raise ValueError("table block checksum mismatch at offset %d" % offset)
    if ctype == 0:
        return bytes(raw)
    if ctype == 1:
        return snappy_uncompress(raw)
    raise ValueError("unknown block compression type %d" % ctype)


def _block_entries(block):
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = get_varint(block, pos)
        non_shared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        key = key[:shared] + block[pos: pos + non_shared]
        pos += non_shared
        yield key, block[pos: pos + vlen]
        pos += vlen


def read_table(path, verify_checksums=True):
    """all (key, value) pairs of a leveldb-format table file, in key order"""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != MAGIC:
        raise ValueError("%s is not a table file (bad magic)" % path)
    footer = data[-48:]
    _, p = get_varint(footer, 0)            # metaindex handle (unused)
    _, p = get_varint(footer, p)
    ioff, p = get_varint(footer, p)
    isize, p = get_varint(footer, p)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify_checksums)):
        off, q = get_varint(handle, 0)
        size, q = get_varint(handle, q)
        out.extend(_block_entries(_read_block(data, off, size, verify_checksums)))
    return out


def _build_block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        out += put_varint(shared) + put_varint(len(k) - shared) + put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, entries, block_size=4096):
    """entries: (key bytes, value bytes) sorted by key.  Raw (uncompressed) blocks; readable by any leveldb-format reader."""
    f = bytearray()

    def emit(block):
        off = len(f)
        f.extend(block)
        f.append(0)                                             # kNoCompression
        f.extend(struct.pack("<I", mask_crc(crc32c(block + b"\0"))))
        return off, len(block)
    index, cur, cur_bytes = [], [], 0
    for k, v in entries:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 8
        if cur_bytes >= block_size:
            off, size = emit(_build_block(cur))
            index.append((cur[-1][0], put_varint(off) + put_varint(size)))
            cur, cur_bytes = [], 0
    if cur or not index:
        off, size = emit(_build_block(cur))
        index.append((cur[-1][0] if cur else b"", put_varint(off) + put_varint(size)))
    moff, msize = emit(_build_block([]))                        # empty metaindex block
    ioff, isize = emit(_build_block(index, restart_interval=1))
    footer = put_varint(moff) + put_varint(msize) + put_varint(ioff) + put_varint(isize)
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    f.extend(footer)
    with open(path, "wb") as fh:
        fh.write(bytes(f))


# ---- the bundle ---------------------------------------------------------------------------------------------------------------
def _data_path(prefix, shard, nshards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, nshards)


def read_tf_checkpoint(prefix, verify_checksums=True):
    """`prefix` = path without extension (e.g. model_dir/model.ckpt-1200).  Returns {variable name: numpy array}."""
    entries = read_table(prefix + ".index", verify_checksums)
    if not entries or entries[0][0] != b"":
        raise ValueError("%s.index has no bundle header" % prefix)
    nshards, endian = 1, 0
    for num, wt, v in _pb_fields(entries[0][1]):
        if num == 1 and wt == 0:
            nshards = v
        elif num == 2 and wt == 0:
            endian = v
    if endian != 0:
        raise ValueError("big-endian checkpoints are not supported")
    shards = {}
    parsed = {key: parse_entry(val) for key, val in entries[1:]}

    def load(key, e, what):
        if e["dtype"] not in _DTYPES:
            raise ValueError("variable `%s`: unsupported dtype enum %d" % (what, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap(_data_path(prefix, sid, nshards), dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]: e["offset"] + e["size"]]
        dt = _DTYPES[e["dtype"]]
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        if n * dt.itemsize != e["size"]:
            raise ValueError("variable `%s`: %d bytes for shape %s" % (what, e["size"], e["shape"]))
        if verify_checksums and e["crc32c"] and (e["size"] <= (1 << 22) or _native()):   # (pure-Python CRC: small tensors only)
            if unmask_crc(e["crc32c"]) != crc32c(raw):
                raise ValueError("variable `%s`: data checksum mismatch" % what)
        return np.frombuffer(raw.tobytes(), dtype=dt).reshape(e["shape"]).copy()

    out = {}
    for key, e in parsed.items():
        if key[:1] == b"\0":            # the data entry of a slice: read through its variable's entry
            continue
        name = key.decode()
        if not e["slices"]:
            out[name] = load(key, e, name)
            continue
        # a partitioned variable: assemble the full tensor from its slices (disjoint hyper-rectangles covering it)
        if e["dtype"] not in _DTYPES:
            raise ValueError("variable `%s`: unsupported dtype enum %d" % (name, e["dtype"]))
        full = np.zeros(e["shape"], dtype=_DTYPES[e["dtype"]])
        covered = 0
        for ext in e["slices"]:
            skey = encode_tensor_name_slice(key, ext)
            if skey not in parsed:
                raise ValueError("variable `%s`: the data of slice %s is missing from the index" % (name, ext))
            part = load(skey, parsed[skey], "%s %s" % (name, ext))
            idx = tuple(slice(st, None if ln < 0 else st + ln) for st, ln in ext)
            if full[idx].shape != part.shape:
                raise ValueError("variable `%s`: slice %s has shape %s" % (name, ext, part.shape))
            full[idx] = part
            covered += part.size
        if covered != full.size:
            raise ValueError("variable `%s`: its slices cover %d of %d elements" % (name, covered, full.size))
        out[name] = full
    return out


def reference_partitioned(name):
    """The variables the reference creates under a partitioner scope (python/lib/joint.py:140-143, 165-168, 186-189): everything
    of the `dnn` and `linear` scopes, optimizer slots included -- stored as slices even when there is one partition."""
    return name.startswith("dnn/") or name.startswith("linear/")


def write_tf_checkpoint(prefix, tensors, sliced=reference_partitioned, partitions=1, state_file=True):
    """Write {name: array} as a one-shard bundle.  `sliced(name)`: store the variable the way TensorFlow stores a partitioned
    one -- `partitions` row ranges (axis 0; scalars and 1-row variables: one), each under its slice key -- default: what the
    reference partitions.  state_file: also write `<dir>/checkpoint` (tf.train.CheckpointState) pointing at this prefix, which
    is how tf.estimator / tf.train.latest_checkpoint find it."""
    names = sorted(tensors, key=lambda s: s.encode())
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    index = [(b"", _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1)))]      # num_shards = 1, little endian, version.producer = 1
    off = 0
    if not _native():
        import warnings
        warnings.warn("libwd_ingest.so is not built: CRC-32C of the checkpoint tensors in pure Python (slow for large tables)")
    with open(_data_path(prefix, 0, 1), "wb") as fd:
        for nm in names:
            a = np.asarray(tensors[nm])                 # (not ascontiguousarray: it would turn a scalar into shape (1,))
            if a.dtype not in _DTYPE_IDS:
                raise ValueError("variable `%s`: dtype %s cannot be stored" % (nm, a.dtype))
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            did = _DTYPE_IDS[a.dtype]
            if not (sliced and sliced(nm)) or a.ndim == 0:
                raw = a.tobytes(order="C")
                fd.write(raw)
                index.append((nm.encode(), encode_entry(did, a.shape, off, len(raw), mask_crc(crc32c(raw)))))
                off += len(raw)
                continue
            nparts = max(1, min(int(partitions), a.shape[0]))
            bounds = [a.shape[0] * k // nparts for k in range(nparts + 1)]
            slices = []
            for k in range(nparts):
                ext = [(bounds[k], bounds[k + 1] - bounds[k])] + [(0, d) for d in a.shape[1:]]
                part = np.ascontiguousarray(a[bounds[k]: bounds[k + 1]])
                raw = part.tobytes(order="C")
                fd.write(raw)
                index.append((encode_tensor_name_slice(nm, ext), encode_entry(did, part.shape, off, len(raw), mask_crc(crc32c(raw)))))
                off += len(raw)
                slices.append(ext)
            index.append((nm.encode(), encode_entry(did, a.shape, 0, 0, 0, slices=slices)))
    index.sort(key=lambda kv: kv[0])
    write_table(prefix + ".index", index)
    if state_file:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as fh:
            fh.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix


def latest_tf_checkpoint(model_dir):
    """model_dir/checkpoint (tf.train.CheckpointState text proto) -> prefix of the newest checkpoint, else the highest
    model.ckpt-<step>.index found; None if there is none."""
    state = os.path.join(model_dir, "checkpoint")
    if os.path.isfile(state):
        for line in open(state, "r", errors="replace"):
            line = line.strip()
            if line.startswith("model_checkpoint_path:"):
                p = line.split(":", 1)[1].strip().strip('"')
                p = p if os.path.isabs(p) else os.path.join(model_dir, p)
                if os.path.isfile(p + ".index"):
                    return p
    best, step = None, -1
    if os.path.isdir(model_dir):
        for fn in os.listdir(model_dir):
            if fn.startswith("model.ckpt-") and fn.endswith(".index"):
                try:
                    s = int(fn[len("model.ckpt-"):-len(".index")])
                except ValueError:
                    continue
                if s > step:
                    best, step = os.path.join(model_dir, fn[:-len(".index")]), s
    return best
