"""TensorFlow checkpoint -> `trt_weights.bin`, without TensorFlow.

Restates the weight path of the reference's converter (scripts/tensorrt_model_builder.py): file layout :52-60,
identity scale triples :121-139, 2-D kernels TF [h,w,in,out] -> KCRS and transposed 2-D kernels [h,w,out,in] ->
(Cin,Cout,R,S) through the same `rsck_to_kcrs` (:220, :278), 3-D kernels TF [d,h,w,in,out] -> KVCRS (:342, :430).
The reference walks the TF graph to find the variables; here the checkpoint's own variable names carry the same
information:  model/<scope>/<layer>/{weights,biases}  ->  <layer>_k / <layer>_b, the shared 2-D encoder
(scope `encoder2D`) is emitted once per image side as left_* / right_*, residual blocks
resblockN/res_convM -> resblockN_convM.  Output is byte-identical to the shipped models/*/TensorRT/*.bin
(tests/test_convert.py).

    python -m redtail_amd.convert.weights <checkpoint prefix> <out.bin> [--fp16]
"""
import collections
import re
import struct
import sys

import numpy as np

from . import tf_bundle


def _order_key(layer):
    """model definition order inside a scope: conv1 < conv2 < ... < resblock1/res_conv1 < ... < encoder2D_out;
    conv3D_1a < conv3D_1b < conv3D_1ds < conv3D_2a ..."""
    m = re.match(r"resblock(\d+)/res_conv(\d+)$", layer)
    if m:
        return (1, int(m.group(1)), int(m.group(2)), "")
    m = re.match(r"[A-Za-z0-9]*?_?(\d+)([a-z]*)$", layer)
    if m:
        return (0, int(m.group(1)), 0, m.group(2))
    return (2, 0, 0, layer)           # encoder2D_out and anything unnumbered go last


def _kernel(v):
    if v.ndim == 4:
        return np.ascontiguousarray(v.transpose(3, 2, 0, 1))        # RSCK -> KCRS (also [h,w,out,in] -> (in,out,h,w))
    if v.ndim == 5:
        return np.ascontiguousarray(v.transpose(4, 0, 3, 1, 2))     # VRSCK -> KVCRS
    raise ValueError("unexpected kernel rank %d" % v.ndim)


def trt_weights_from_checkpoint(prefix):
    """Ordered {name: float32 array} exactly as the reference writes trt_weights.bin for this checkpoint."""
    ckpt = tf_bundle.read_checkpoint(prefix)
    scopes = collections.OrderedDict()
    for name in ckpt:
        parts = name.split("/")
        if len(parts) < 4 or parts[0] != "model" or parts[-1] not in ("weights", "biases"):
            continue                                                  # global_step, optimizer slots, ...
        scopes.setdefault(parts[1], set()).add("/".join(parts[2:-1]))
    if "encoder2D" not in scopes:
        raise ValueError("checkpoint has no model/encoder2D scope: not a Stereo DNN model")
    out = collections.OrderedDict()
    for side in ("left", "right"):                                    # identity input scaling (:121-139)
        out[side + "_scale_shift"] = np.float32([0.0])
        out[side + "_scale_scale"] = np.float32([1.0])
        out[side + "_scale_power"] = np.float32([1.0])

    def emit(scope, layer, prefix_):
        trt = prefix_ + layer.replace("/res_conv", "_conv")
        out[trt + "_k"] = _kernel(ckpt["model/%s/%s/weights" % (scope, layer)])
        out[trt + "_b"] = np.ascontiguousarray(ckpt["model/%s/%s/biases" % (scope, layer)])

    # shared weights, one copy per image side; the model scripts build the two towers side by side, one unit (a plain
    # convolution or a whole residual block) at a time: left unit, right unit, next unit ...
    units = collections.OrderedDict()
    for layer in sorted(scopes["encoder2D"], key=_order_key):
        units.setdefault(layer.split("/")[0], []).append(layer)
    for layers in units.values():
        for side in ("left_", "right_"):
            for layer in layers:
                emit("encoder2D", layer, side)
    rank = {"bneck_encoder2D": 0, "encoder3D": 0, "bneck_decoder2D": 1, "decoder3D": 1}
    for scope in sorted((s for s in scopes if s != "encoder2D"), key=lambda s: (rank.get(s, 2), s)):
        for layer in sorted(scopes[scope], key=_order_key):
            emit(scope, layer, "")
    return out


def write_trt_weights(path, weights, fp16=False):
    """repeated {name '\\0', uint32 count, count x (float32 | float16)}  (tensorrt_model_builder.py:52-60)"""
    with open(path, "wb") as f:
        for name, v in weights.items():
            flat = np.asarray(v).reshape(-1)
            f.write(name.encode() + b"\0")
            f.write(struct.pack("<I", flat.size))
            f.write(flat.astype("<f2" if fp16 else "<f4").tobytes())


def main(argv):
    if len(argv) < 3:
        print(__doc__)
        return 2
    w = trt_weights_from_checkpoint(argv[1])
    write_trt_weights(argv[2], w, fp16="--fp16" in argv[3:])
    print("%d tensors, %d parameters -> %s" % (len(w), sum(v.size for v in w.values()), argv[2]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
