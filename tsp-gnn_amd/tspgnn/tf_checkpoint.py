"""Reader / writer for TensorFlow "tensor bundle" checkpoints (the ``model.ckpt.index`` +
``model.ckpt.data-00000-of-00001`` pair that ``tf.train.Saver`` writes, which is what the reference's
``util.save_weights`` / ``load_weights`` produce and consume, /root/reference/util.py:5-37), in pure
Python + numpy -- TensorFlow is not needed, so weights trained with the reference can be loaded into
this implementation and vice versa.

Format (restated from the published TensorFlow / LevelDB sources; there is no TensorFlow in this
image to cross-check against, the tests pin the known-answer pieces -- CRC-32C, masking, varints,
footer magic -- and the write -> read round trip):
  * ``.index`` is a LevelDB-style sorted string table: data blocks of prefix-compressed entries
    (varint shared, varint non_shared, varint value_len, key suffix, value) followed by the restart
    array and its length, each block trailed by 1 type byte (0 = uncompressed) and a masked CRC-32C;
    an (empty) metaindex block, an index block mapping separator keys to block handles (varint offset,
    varint size), and a 48-byte footer (two handles padded to 40 bytes, magic 0xdb4775248b80fb57).
  * key ""  -> BundleHeaderProto {num_shards = 1, endianness = little, version {producer = 1}};
    key <variable name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, masked crc32c}.
  * ``.data-NNNNN-of-MMMMM`` holds the raw little-endian tensor bytes at [offset, offset + size).
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _crc_table():
    poly = 0x82F63B78  # CRC-32C (Castagnoli), reflected
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table.append(c)
    return table


_TABLE = _crc_table()


def crc32c(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    t = _TABLE
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc):
    """leveldb / TF crc32c::Mask: rotate right by 15 bits and add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def _put_varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


# ------------------------------------------------------------------------------------- protobuf
def _parse_proto(buf):
    """Minimal wire-format decoder -> {field: [values]} (varint -> int, fixed32 -> int, bytes -> bytes)."""
    fields, pos = {}, 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _get_varint(buf, pos)
        elif wire == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = _get_varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        fields.setdefault(field, []).append(val)
    return fields


def _field(num, wire, payload):
    return _put_varint((num << 3) | wire) + payload


def _shape_proto(shape):
    out = b""
    for s in shape:
        dim = _field(1, 0, _put_varint(int(s)))
        out += _field(2, 2, _put_varint(len(dim)) + dim)
    return out


def _entry_proto(dtype_code, shape, offset, size, masked_crc):
    out = _field(1, 0, _put_varint(dtype_code))
    sp = _shape_proto(shape)
    out += _field(2, 2, _put_varint(len(sp)) + sp)
    # shard_id = 0 and offset = 0 are proto3 defaults (omitted on the wire)
    if offset:
        out += _field(4, 0, _put_varint(offset))
    out += _field(5, 0, _put_varint(size))
    out += _field(6, 5, struct.pack("<I", masked_crc))
    return out


def _header_proto():
    version = _field(1, 0, _put_varint(1))                     # VersionDef.producer = 1
    return _field(1, 0, _put_varint(1)) + _field(3, 2, _put_varint(len(version)) + version)   # num_shards = 1


# ------------------------------------------------------------------------------------- table
def _read_block(buf, offset, size):
    contents = buf[offset:offset + size]
    btype = buf[offset + size]
    stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
    if unmask_crc(stored) != crc32c(buf[offset:offset + size + 1]):
        raise ValueError("checkpoint index: block checksum mismatch at offset %d" % offset)
    if btype != 0:
        raise NotImplementedError("checkpoint index: compressed block (type %d); TensorFlow writes bundles uncompressed" % btype)
    n_restarts = struct.unpack_from("<I", contents, len(contents) - 4)[0]
    end = len(contents) - 4 - 4 * n_restarts
    entries, pos, key = [], 0, b""
    while pos < end:
        shared, pos = _get_varint(contents, pos)
        non_shared, pos = _get_varint(contents, pos)
        vlen, pos = _get_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + non_shared])
        pos += non_shared
        entries.append((key, bytes(contents[pos:pos + vlen])))
        pos += vlen
    return entries


def _build_block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (key, value) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _with_trailer(block):
    return block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))


# ------------------------------------------------------------------------------------- public
def read_bundle(prefix):
    """{variable name: numpy array} of the checkpoint written as ``prefix.index`` / ``prefix.data-*``."""
    with open(prefix + ".index", "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError("%s.index is not a TensorFlow bundle index (bad footer magic)" % prefix)
    footer = buf[len(buf) - 48:]
    _, pos = _get_varint(footer, 0)          # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    entries = []
    for _, handle in _read_block(buf, ioff, isize):
        off, p = _get_varint(handle, 0)
        size, p = _get_varint(handle, p)
        entries += _read_block(buf, off, size)
    if not entries or entries[0][0] != b"":
        raise ValueError("%s.index has no bundle header" % prefix)
    header = _parse_proto(entries[0][1])
    n_shards = header.get(1, [0])[0]
    if header.get(2, [0])[0] != 0:
        raise NotImplementedError("big-endian bundle")
    shards = {}
    out = {}
    for key, value in entries[1:]:
        e = _parse_proto(value)
        dtype = _DTYPES.get(e.get(1, [0])[0])
        if dtype is None:
            raise NotImplementedError("variable %r has an unsupported dtype code %r" % (key, e.get(1)))
        if 7 in e:
            raise NotImplementedError("variable %r is stored in slices (partitioned variable)" % key)
        shape = []
        if 2 in e:
            for dim in _parse_proto(e[2][0]).get(2, []):
                shape.append(_parse_proto(dim).get(1, [0])[0])
        shard, offset, size = e.get(3, [0])[0], e.get(4, [0])[0], e.get(5, [0])[0]
        if shard not in shards:
            with open("%s.data-%05d-of-%05d" % (prefix, shard, n_shards), "rb") as f:
                shards[shard] = f.read()
        raw = shards[shard][offset:offset + size]
        if len(raw) != size:
            raise ValueError("variable %r: data shard truncated" % key)
        if 6 in e and unmask_crc(e[6][0]) != crc32c(raw):
            raise ValueError("variable %r: data checksum mismatch" % key)
        out[key.decode("utf-8")] = np.frombuffer(raw, dtype=np.dtype(dtype).newbyteorder("<")).reshape(shape).copy()
    return out


def write_bundle(prefix, tensors):
    """Write {name: array} as a single-shard TensorFlow bundle ``prefix.index`` + ``prefix.data-00000-of-00001``
    (what tf.train.Saver().save(sess, prefix) leaves on disk, minus the optional .meta graph)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = sorted(((name.encode("utf-8"), np.asarray(arr, order="C")) for name, arr in tensors.items()), key=lambda kv: kv[0])
    entries, offset = [(b"", _header_proto())], 0
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for key, arr in items:
            if arr.dtype not in _DTYPE_CODES:
                raise NotImplementedError("dtype %s of %r" % (arr.dtype, key))
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            data.write(raw)
            entries.append((key, _entry_proto(_DTYPE_CODES[arr.dtype], arr.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    out = bytearray()
    block = _build_block(entries)
    data_handle = _put_varint(0) + _put_varint(len(block))
    out += _with_trailer(block)
    meta = _build_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out += _with_trailer(meta)
    # one data block -> one index entry; its key must be >= the block's last key
    index = _build_block([(entries[-1][0] + b"\x00", data_handle)], restart_interval=1)
    index_handle = _put_varint(len(out)) + _put_varint(len(index))
    out += _with_trailer(index)
    handles = meta_handle + index_handle
    out += handles + b"\x00" * (40 - len(handles)) + struct.pack("<Q", _MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
