"""Reader for TensorFlow checkpoint bundles (`<prefix>.index` + `<prefix>.data-00000-of-0000N`) without TensorFlow.

The reference's converter needs TF 1.5 only to pull the variables out of a checkpoint
(scripts/tensorrt_model_builder.py: `conv_op.inputs[1].eval()`); the container format itself is simple:
  * `.index` is a LevelDB-style sorted string table: prefix-compressed key/value blocks, an index block that maps
    keys to block handles, and a 48-byte footer with the two top-level handles and the magic 0xdb4775248b80fb57;
  * the value of key "" is a BundleHeaderProto, every other value a BundleEntryProto
    {1: dtype, 2: shape{2: dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c};
  * `.data-*` files hold the raw little-endian tensor bytes at [offset, offset + size).
Only what the Stereo DNN checkpoints use is implemented: uncompressed blocks, DT_FLOAT / DT_INT32 / DT_INT64 tensors.
"""
import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 19: np.float16}


def _varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if b < 0x80:
            return val, pos
        shift += 7


def _block(data, offset, size):
    """entries of one table block: list of (key bytes, value bytes)"""
    if data[offset + size] != 0:
        raise ValueError("compressed table blocks are not supported (type %d)" % data[offset + size])
    blk = data[offset:offset + size]
    (num_restarts,) = struct.unpack_from("<I", blk, size - 4)
    limit = size - 4 - 4 * num_restarts
    out, pos, key = [], 0, b""
    while pos < limit:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(blk[pos:pos + vlen])))
        pos += vlen
    return out


def _proto(buf):
    """flat protobuf decode: {field: [values]}; length-delimited fields stay bytes"""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(val)
    return out


def read_index(prefix):
    """{variable name: dict(dtype, shape, shard, offset, size)} and the number of data shards"""
    data = open(prefix + ".index", "rb").read()
    footer = data[-48:]
    if struct.unpack("<Q", footer[-8:])[0] != _MAGIC:
        raise ValueError("%s.index is not a TensorFlow bundle index (bad magic)" % prefix)
    _, pos = _varint(footer, 0)            # metaindex handle (offset, size): unused
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    entries, shards = {}, 1
    for _, handle in _block(data, ioff, isize):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, val in _block(data, boff, bsize):
            msg = _proto(val)
            if key == b"":
                shards = msg.get(1, [1])[0]
                continue
            shape = []
            for sh in msg.get(2, []):
                for dim in _proto(sh).get(2, []):
                    shape.append(_proto(dim).get(1, [0])[0])
            entries[key.decode()] = dict(dtype=msg.get(1, [0])[0], shape=tuple(shape), shard=msg.get(3, [0])[0],
                                         offset=msg.get(4, [0])[0], size=msg.get(5, [0])[0])
    return entries, shards


def read_checkpoint(prefix):
    """{variable name: numpy array} of every tensor in the bundle"""
    entries, shards = read_index(prefix)
    files = {}
    out = {}
    for name, e in entries.items():
        if e["dtype"] not in _DTYPES:
            raise ValueError("%s: unsupported dtype %d" % (name, e["dtype"]))
        if e["shard"] not in files:
            files[e["shard"]] = open("%s.data-%05d-of-%05d" % (prefix, e["shard"], shards), "rb").read()
        raw = files[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        arr = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]])
        out[name] = arr.reshape(e["shape"]) if e["shape"] else arr.reshape(())
    return out


if __name__ == "__main__":
    import sys
    for k, v in sorted(read_checkpoint(sys.argv[1]).items()):
        print("%-70s %-22s %s" % (k, v.shape, v.dtype))
