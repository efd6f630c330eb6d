"""TensorFlow bundle checkpoint reader / writer (tspgnn/tf_checkpoint.py): known-answer pieces of the format and
the write -> read round trip.  (No TensorFlow here: what is pinned is the published format, piece by piece.)"""
import os
import struct

import numpy as np
import pytest

from tspgnn import tf_checkpoint as T


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC-32C
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    # incremental == one shot
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == T.crc32c(b"123456789")


def test_crc_masking_is_leveldbs():
    crc = T.crc32c(b"foo")
    m = T.mask_crc(crc)
    assert m != crc and T.unmask_crc(m) == crc
    assert m == ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF
    assert T.mask_crc(m) != crc and T.unmask_crc(T.unmask_crc(T.mask_crc(m))) == crc


def test_varint_and_proto_roundtrip():
    for n in (0, 1, 127, 128, 300, 2 ** 31 - 1, 2 ** 40 + 5):
        enc = T._put_varint(n)
        assert T._get_varint(enc, 0) == (n, len(enc))
    assert T._put_varint(300) == b"\xac\x02"
    e = T._parse_proto(T._entry_proto(1, (128, 256), 4096, 131072, 0xDEADBEEF))
    assert e[1] == [1] and e[4] == [4096] and e[5] == [131072] and e[6] == [0xDEADBEEF]
    dims = [T._parse_proto(d)[1][0] for d in T._parse_proto(e[2][0])[2]]
    assert dims == [128, 256]
    h = T._parse_proto(T._header_proto())
    assert h[1] == [1] and T._parse_proto(h[3][0])[1] == [1]


def test_block_prefix_compression_roundtrip():
    keys = [b"", b"TSP/E_cell/kernel", b"TSP/E_cell/kernel/Adam", b"TSP/E_cell/kernel/Adam_1", b"TSP/V_cell/kernel", b"V_init"]
    entries = [(k, bytes([i]) * (i + 1)) for i, k in enumerate(keys)]
    for interval in (1, 2, 16):
        block = T._build_block(entries, restart_interval=interval)
        buf = T._with_trailer(block)
        assert T._read_block(buf, 0, len(block)) == entries
    bad = bytearray(T._with_trailer(T._build_block(entries)))
    bad[3] ^= 1
    with pytest.raises(ValueError):
        T._read_block(bytes(bad), 0, len(bad) - 5)


def test_bundle_roundtrip_and_layout(tmp_path):
    rng = np.random.RandomState(0)
    tensors = {
        "TSP/E_cell/layer_norm_basic_lstm_cell/kernel": rng.randn(128, 256).astype(np.float32),
        "TSP/E_cell/layer_norm_basic_lstm_cell/kernel/Adam": rng.randn(128, 256).astype(np.float32),
        "V_init": rng.randn(1, 64).astype(np.float32),
        "beta1_power": np.float32(0.81),
        "E_vote_MLP_layer_4/bias": rng.randn(1).astype(np.float32),
        "global_step": np.int64(7),
    }
    prefix = str(tmp_path / "epoch=3" / "model.ckpt")
    T.write_bundle(prefix, tensors)
    assert sorted(os.listdir(os.path.dirname(prefix))) == ["checkpoint", "model.ckpt.data-00000-of-00001", "model.ckpt.index"]
    back = T.read_bundle(prefix)
    assert sorted(back) == sorted(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == np.asarray(v).dtype and back[k].shape == np.asarray(v).shape
        assert np.array_equal(back[k], v)
    # footer: 48 bytes, LevelDB table magic; data file = tensors in key order, back to back
    idx = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", idx[-8:])[0] == 0xdb4775248b80fb57
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    assert len(data) == sum(np.asarray(v).nbytes for v in tensors.values())
    first = sorted(tensors)[0]
    assert data[:np.asarray(tensors[first]).nbytes] == np.asarray(tensors[first]).tobytes()
    # a flipped data byte is detected through the per-tensor checksum
    corrupt = bytearray(data)
    corrupt[10] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(corrupt))
    with pytest.raises(ValueError):
        T.read_bundle(prefix)


def test_many_variables_span_restart_points(tmp_path):
    tensors = {"scope/var_%03d/kernel" % i: np.full((3, 2), i, dtype=np.float32) for i in range(100)}
    prefix = str(tmp_path / "model.ckpt")
    T.write_bundle(prefix, tensors)
    back = T.read_bundle(prefix)
    assert all(np.array_equal(back[k], v) for k, v in tensors.items()) and len(back) == 100
