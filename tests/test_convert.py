"""TF-free model conversion: the checkpoint reader and the weight re-emitter must reproduce the
reference's shipped trt_weights.bin files byte for byte.  Needs the reference checkout (checkpoints are too large to
commit), so it runs where /root/reference exists and is skipped on the GPU box; the table reader itself is also
checked on a synthetic bundle built in the test."""
import os
import struct

import numpy as np
import pytest

from redtail_amd.convert import tf_bundle, weights

MODELS = "/root/reference/stereoDNN/models"
CASES = [("ResNet-18_2D", "model-inference-513x257-0"), ("NVTiny", "model-inference-513x161-0")]


@pytest.mark.parametrize("model,ckpt", CASES)
@pytest.mark.parametrize("fp16", [False, True])
def test_checkpoint_reproduces_shipped_weights(tmp_path, model, ckpt, fp16):
    prefix = os.path.join(MODELS, model, "TensorFlow", ckpt)
    shipped = os.path.join(MODELS, model, "TensorRT", "trt_weights_fp16.bin" if fp16 else "trt_weights.bin")
    if not (os.path.exists(prefix + ".index") and os.path.exists(shipped)):
        pytest.skip("reference models not present")
    out = tmp_path / "w.bin"
    weights.write_trt_weights(str(out), weights.trt_weights_from_checkpoint(prefix), fp16=fp16)
    assert out.read_bytes() == open(shipped, "rb").read()


def _varint(n):
    b = bytearray()
    while True:
        if n < 0x80:
            b.append(n)
            return bytes(b)
        b.append((n & 0x7F) | 0x80)
        n >>= 7


def _entry_proto(dtype, shape, offset, size):
    dims = b"".join(b"\x12" + _varint(len(d)) + d for d in (b"\x08" + _varint(s) for s in shape))
    return b"\x08" + _varint(dtype) + b"\x12" + _varint(len(dims)) + dims + b"\x20" + _varint(offset) + b"\x28" + _varint(size)


def _block(entries):
    """one table block with prefix compression against the previous key and a single restart point"""
    body, prev = b"", b""
    for k, v in entries:
        shared = 0
        while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
            shared += 1
        body += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        prev = k
    return body + struct.pack("<I", 0) + struct.pack("<I", 1)


def test_bundle_reader_on_synthetic_table(tmp_path):
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    b = np.array([7, -1], dtype=np.int64)
    data = a.tobytes() + b.tobytes()
    entries = [(b"", b"\x08\x01"),
               (b"model/enc/conv1/biases", _entry_proto(9, b.shape, a.nbytes, b.nbytes)),
               (b"model/enc/conv1/weights", _entry_proto(1, a.shape, 0, a.nbytes))]
    blk = _block(entries)
    index_blk = _block([(b"model/enc/conv1/weights\xff", _varint(0) + _varint(len(blk)))])
    meta_blk = _block([])
    table = blk + b"\0" + b"\0\0\0\0"
    moff = len(table)
    table += meta_blk + b"\0" + b"\0\0\0\0"
    ioff = len(table)
    table += index_blk + b"\0" + b"\0\0\0\0"
    footer = _varint(moff) + _varint(len(meta_blk)) + _varint(ioff) + _varint(len(index_blk))
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    prefix = str(tmp_path / "ckpt")
    open(prefix + ".index", "wb").write(table + footer)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    got = tf_bundle.read_checkpoint(prefix)
    assert set(got) == {"model/enc/conv1/biases", "model/enc/conv1/weights"}
    assert np.array_equal(got["model/enc/conv1/weights"], a) and np.array_equal(got["model/enc/conv1/biases"], b)
    with pytest.raises(ValueError):
        open(prefix + ".index", "wb").write(table + footer[:-8] + b"\0" * 8)
        tf_bundle.read_checkpoint(prefix)
