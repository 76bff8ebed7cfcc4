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
    sample_image(ref)


BAND = (160, 288)       # destination rows kept from img_left.bin (their source rows straddle y = 256, where float32 coordinates coarsen)


def sample_image(ref):
    """The sample application's network input as the reference ships it: sample_app/data/img_left.png (1242 x 375, 8-bit) and
    img_left.bin (3 x 321 x 1025 float32 CHW RGB in [0, 1]: what readImgFile, sample_app/main.cpp:83-98, is to produce from the PNG) --
    the only known-answer vector for the pre-processing.  Kept: the decoded PNG and rows BAND of the .bin (the whole files are staged next
    to the weights by redtail_amd/model_files.py and compared in full where they are present)."""
    from PIL import Image
    data = os.path.join(ref, "stereoDNN", "sample_app", "data")
    png = np.asarray(Image.open(os.path.join(data, "img_left.png")))
    assert png.shape == (375, 1242, 3) and png.dtype == np.uint8
    chw = np.fromfile(os.path.join(data, "img_left.bin"), dtype="<f4").reshape(3, 321, 1025)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "redtail_sample_image.npz")
    np.savez_compressed(dst, img_left_png_rgb=png, img_left_bin_rows=chw[:, BAND[0]:BAND[1]].copy(), band=np.int32(BAND))
    print("wrote %s: %d bytes" % (dst, os.path.getsize(dst)))


if __name__ == "__main__":
    main()
