#!/usr/bin/env python3
"""Pack the reference's golden test tensors into one compressed archive.

The reference keeps 58 TensorFlow-generated tensors under
``stereoDNN/tests/data/*.bin`` (format: int32 ndims, int32 dims[], float32
data[] -- written by ``stereoDNN/scripts/test_data_generator.py:34-39`` and
read back by ``stereoDNN/tests/tests_main.cpp:259-275``).  They are the only
known-answer vectors that pin the plugin ops (ELU, Conv3D, Conv3DTranspose,
cost volumes, softargmin/max) to TensorFlow 1.5 semantics.

/root/reference does not exist on the GPU box, so this script (run once, here)
re-packs them as ``tests/golden/redtail_fixtures.npz``; key = file stem.
Nothing but the numeric payload is carried over.

    python tests/golden/make_golden.py [/root/reference]
"""
import glob
import os
import struct
import sys

import numpy as np


def read_bin(path):
    with open(path, "rb") as f:
        raw = f.read()
    (nd,) = struct.unpack_from("<i", raw, 0)
    dims = struct.unpack_from("<%di" % nd, raw, 4)
    data = np.frombuffer(raw, dtype="<f4", offset=4 + 4 * nd)
    assert data.size == int(np.prod(dims)), (path, dims, data.size)
    return data.reshape(dims).copy()


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "stereoDNN", "tests", "data")
    files = sorted(glob.glob(os.path.join(src, "*.bin")))
    assert len(files) == 58, "expected the reference's 58 fixture files, found %d" % len(files)
    out = {os.path.splitext(os.path.basename(p))[0]: read_bin(p) for p in files}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "redtail_fixtures.npz")
    np.savez_compressed(dst, **out)
    print("wrote %s: %d tensors, %d bytes" % (dst, len(out), os.path.getsize(dst)))


if __name__ == "__main__":
    main()
