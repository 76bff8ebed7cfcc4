"""CPU oracle for the Stereo DNN inference hot path  --  TEST INFRASTRUCTURE ONLY.

This module restates, on the CPU (torch fp32/fp64 + numpy), the arithmetic of the
reference's ``stereoDNN/lib`` plugin layer and of the TensorRT-native 2-D layers the
generated networks call.  Nothing in the product path (``redtail_amd/``) may import
it; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and only as the checker / the timed CPU baseline.

Parity status: PINNED for every plugin op -- ``tests/test_oracle_golden.py`` checks
each function below against the reference's own TF-generated tensors
(``stereoDNN/tests/data/*.bin`` re-packed by ``tests/golden/make_golden.py``).
The TensorRT-native 2-D layers (conv/deconv/add/concat/sigmoid) have no reference
fixture and no end-to-end golden disparity exists: for those,
parity is pinned only by TF semantics ("parity unpinned" by reference vectors).

Every function cites the reference file:line whose semantics it follows
(paths relative to /root/reference/stereoDNN).
"""
import struct

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# Plugin-level ops (layouts exactly as the plugins see them; leading dim = batch)
# --------------------------------------------------------------------------------------
def elu(x):
    """ELU, alpha = 1 (lib/elu_plugin.cpp:93,132; scripts/test_data_generator.py:41-60)."""
    return torch.where(x > 0, x, torch.expm1(x))


def corr_cost_volume(left, right, max_disp):
    """Correlation cost volume (lib/kernels.cu:168-200; test_data_generator.py:242-259).

    left/right: (N, C, H, W)  ->  (N, D, H, W);  cv[d,y,x] = sum_c L[c,y,x] * R[c,y,x-d],
    zero where x < d; disparities ordered min -> max.
    """
    n, c, h, w = left.shape
    out = left.new_zeros((n, max_disp, h, w))
    for d in range(min(max_disp, w)):
        out[:, d, :, d:] = (left[:, :, :, d:] * right[:, :, :, : w - d]).sum(1)
    return out


def to_nc2hw2(x):
    """(N,C,H,W) float -> TensorRT kNC2HW2 fp16 packing: (N, ceil(C/2), H, W, 2) float16, channel pair (2i, 2i+1)
    of a pixel in one 4-byte slot, odd C zero padded (lib/kernels.cu:203-250 reads it as __half2 per channel pair)."""
    n, c, h, w = x.shape
    xp = torch.zeros((n, (c + 1) // 2 * 2, h, w), dtype=torch.float16)
    xp[:, :c] = x.to(torch.float16)
    return xp.reshape(n, (c + 1) // 2, 2, h, w).permute(0, 1, 3, 4, 2).contiguous()


def from_nc2hw2(x, c):
    """inverse of to_nc2hw2: (N, ceil(C/2), H, W, 2) float16 -> (N, C, H, W) float32"""
    n, c2, h, w, _ = x.shape
    return x.permute(0, 1, 4, 2, 3).reshape(n, 2 * c2, h, w)[:, :c].to(torch.float32)


def corr_cost_volume_fp16(left_h2, right_h2, c, max_disp):
    """corrCostVolumeFP16NC2HW2Kernel (lib/kernels.cu:203-250): fp16 NC2HW2 in and out, fp32 arithmetic
    (:219-221 'using FP32 arithmetic for better precision'), one rounding to half at the end (:247)."""
    cv = corr_cost_volume(from_nc2hw2(left_h2, c), from_nc2hw2(right_h2, c), max_disp)
    return to_nc2hw2(cv)


def cost_volume(left, right, max_disp):
    """Default (concatenation) cost volume (lib/kernels.cu:50-97,136-161;
    test_data_generator.py:223-240).  (N,C,H,W) x2 -> (N, D, 2C, H, W):
    cv[d, 0:C] = L ; cv[d, C:2C, y, x] = R[:, y, x-d] (0 for x < d)."""
    n, c, h, w = left.shape
    out = left.new_zeros((n, max_disp, 2 * c, h, w))
    for d in range(max_disp):
        out[:, d, :c] = left
        if d < w:
            out[:, d, c:, :, d:] = right[:, :, :, : w - d]
    return out


def softargmax(vol, is_min):
    """Soft-argmax / soft-argmin over the disparity axis (lib/softargmax_plugin.cpp:167-205;
    test_data_generator.py:300-315).  vol: (N, D, H, W) [a (N,D,1,H,W) input is squeezed,
    softargmax_plugin.cpp:66-72] -> (N, 1, H, W) = sum_d d * softmax_d(+-vol)."""
    if vol.dim() == 5:
        assert vol.shape[2] == 1
        vol = vol[:, :, 0]
    p = torch.softmax(-vol if is_min else vol, dim=1)
    idx = torch.arange(vol.shape[1], dtype=vol.dtype).view(1, -1, 1, 1)
    return (p * idx).sum(1, keepdim=True)


def conv3d_tf(x, w, bias, stride, pad_start, pad_end):
    """Conv3DPlugin, Conv3DType::kTensorFlow (lib/conv3d_plugin.cpp:187-216 with the
    descriptor reshape of lib/conv_utils.cpp:27-32,58-72).

    x: (N, D, C, H, W)   w: (K, V, C, R, S)   bias: (K,) or None
    stride/pad_*: (d, h, w).  cuDNN is given pad_start only (conv3d_plugin.cpp:86-87), so
    the (TF-asymmetric) extra end padding in D must already be in x (PaddingPlugin) --
    exactly like the reference.  Returns (N, K, Do, Ho, Wo).
    """
    xt = x.permute(0, 2, 1, 3, 4)            # N C D H W
    wt = w.permute(0, 2, 1, 3, 4)            # K C V R S
    return F.conv3d(xt, wt, bias, stride=tuple(stride), padding=tuple(pad_start))


def conv3d_transpose_tf(y, w, bias, out_dims, stride, pad_start, pad_end):
    """Conv3DTransposePlugin (lib/conv3d_transpose_plugin.cpp:205-243): cuDNN backward-data
    of the conv above.  y: (N, K, Dy, Hy, Wy)  w: (K, V, C, R, S)  bias: (C,) or None
    out_dims: (Dx, C, Hx, Wx) as passed to createConv3DTransposePlugin.  Returns
    (N, Dx, C, Hx, Wx).  Bias is per C of the output (kernels.cu:292-308)."""
    dx, c, hx, wx = out_dims
    wt = w.permute(0, 2, 1, 3, 4)            # (K=in, C=out, V, R, S)
    k = wt.shape[2:]
    got = [(y.shape[2 + i] - 1) * stride[i] - 2 * pad_start[i] + k[i] for i in range(3)]
    opad = [t - g for t, g in zip((dx, hx, wx), got)]
    assert all(0 <= o < s for o, s in zip(opad, stride)), (got, out_dims)
    x = F.conv_transpose3d(y, wt, bias, stride=tuple(stride), padding=tuple(pad_start),
                           output_padding=tuple(opad))
    return x.permute(0, 2, 1, 3, 4).contiguous()


def transform(x, order=(1, 0, 2, 3)):
    """TransformPlugin (lib/transform_plugin.cpp:94-108,135-165); only {1,0,2,3} is used."""
    return x.permute(0, *[o + 1 for o in order]).contiguous()


def pad_d(x, n_end=1):
    """PaddingPlugin: append zero slices at the end of the outermost non-batch dim
    (lib/padding_plugin.cpp:79-94)."""
    z = x.new_zeros((x.shape[0], n_end) + tuple(x.shape[2:]))
    return torch.cat([x, z], 1)


def slice_d(x, start, end):
    """SlicePlugin: keep [start, end) of the outermost non-batch dim (lib/slice_plugin.cpp:80-92)."""
    return x[:, start:end].contiguous()


# --------------------------------------------------------------------------------------
# TensorRT-native 2-D layers used by the generated nets (no source in the reference;
# semantics = TensorFlow's, see scripts/tensorrt_model_builder.py:140-288)
# --------------------------------------------------------------------------------------
def tf_same_pad(n, k, s):
    """scripts/tensorrt_model_builder.py:140-147."""
    pad = max(k - s, 0) if n % s == 0 else max(k - (n % s), 0)
    return pad // 2, pad - pad // 2


def conv2d(x, w, b, stride, pad):
    """addConvolution: cross-correlation, weights KCRS (tensorrt_model_builder.py:149-228)."""
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def deconv2d(x, w, b, stride, pad):
    """addDeconvolution: weights (Cin, Cout, R, S) = TF conv2d_transpose/Conv2DBackpropInput
    (tensorrt_model_builder.py:230-288)."""
    return F.conv_transpose2d(x, w, b, stride=stride, padding=pad)


# --------------------------------------------------------------------------------------
# Weight files (scripts/tensorrt_model_builder.py:52-60; reader sample_app/main.cpp:111-134)
# --------------------------------------------------------------------------------------
def _area_table(ssize, dsize):
    """per destination index: list of (source index, weight) -- OpenCV's computeResizeAreaTab for scale >= 1
    (the INTER_AREA path cv::resize takes when shrinking; modules/imgproc/src/resize.cpp)"""
    scale = ssize / dsize
    tab = []
    for dx in range(dsize):
        f1 = dx * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = int(np.ceil(f1)), min(int(np.floor(f2)), ssize)
        s1 = min(s1, s2)
        taps = []
        if s1 - f1 > 1e-3:
            taps.append((s1 - 1, (s1 - f1) / cell))
        for sx in range(s1, s2):
            taps.append((sx, 1.0 / cell))
        if f2 - s2 > 1e-3 and s2 < ssize:
            taps.append((s2, min(min(f2 - s2, 1.0), cell) / cell))
        tab.append(taps)
    return tab


def resize_area_tf(img, dst_h, dst_w):
    """TensorFlow 1.5 `tf.image.resize_area` (tensorflow/core/kernels/resize_area_op.cc, not in /root/reference: restated from its published
    algorithm): the overlap integral of every source cell with the destination cell, source coordinates and weights in FLOAT32, row sums
    first, no threshold on small taps.  PINNED: the reference's shipped network inputs sample_app/data/img_{left,right}.bin are this filter
    applied to its img_{left,right}.png, / 255, to 3e-7 (tests/test_oracle_golden.py; fixture tests/golden/redtail_sample_image.npz) --
    which is what anchors the area filter of readImgFile below.  img: (H, W, C) any real type; returns float32 (dst_h, dst_w, C)."""
    f32 = np.float32
    ih, iw, cn = img.shape
    hs, ws = f32(ih) / f32(dst_h), f32(iw) / f32(dst_w)

    def cells(n_out, scale):
        out = []
        for x in range(n_out):
            a, b = f32(x) * scale, f32(x + 1) * scale
            start, end = int(np.floor(a)), int(np.ceil(b))
            wts = []
            for v in range(start, end):
                v = f32(v)
                if v < a:
                    wts.append(scale if v + f32(1) > b else f32(v + f32(1) - a))
                else:
                    wts.append(f32(b - v) if v + f32(1) > b else f32(1.0))
            out.append((start, wts))
        return out

    x = np.asarray(img).astype(f32)
    rows = np.zeros((ih, dst_w, cn), f32)
    for dx, (start, wts) in enumerate(cells(dst_w, ws)):
        acc = x[:, min(max(start, 0), iw - 1), :] * wts[0]
        for k in range(1, len(wts)):
            acc = acc + x[:, min(max(start + k, 0), iw - 1), :] * wts[k]
        rows[:, dx, :] = acc
    out = np.zeros((dst_h, dst_w, cn), f32)
    scale = f32(1.0) / f32(hs * ws)
    for dy, (start, wts) in enumerate(cells(dst_h, hs)):
        acc = np.zeros((dst_w, cn), f32)
        for k in range(len(wts)):
            acc = acc + rows[min(max(start + k, 0), ih - 1)] * wts[k]
        out[dy] = acc * scale
    return out


def area_taps_below_threshold(ssize, dsize):
    """destination indices at which OpenCV's INTER_AREA table (computeResizeAreaTab) drops a tap because its share of the cell is below
    1e-3 source pixels -- the only places where it differs from the overlap integrals by more than rounding"""
    scale = ssize / dsize
    out = []
    for dx in range(dsize):
        f1 = dx * scale
        f2 = f1 + scale
        s1, s2 = int(np.ceil(f1)), min(int(np.floor(f2)), ssize)
        s1 = min(s1, s2)
        if 0 < s1 - f1 <= 1e-3 or (0 < f2 - s2 <= 1e-3 and s2 < ssize):
            out.append(dx)
    return out


def preprocess_bgr8(img_u8, dst_h, dst_w):
    """readImgFile (sample_app/main.cpp:83-98): u8 BGR HWC -> float32, cv::resize(INTER_AREA), BGR -> RGB, HWC -> CHW,
    / 255.  OpenCV is not available in this environment; the area filter restates its table construction (computeResizeAreaTab) and is
    PINNED against the reference's own data: on sample_app/data/img_left.png it reproduces the shipped img_left.bin (3 x 321 x 1025) to
    8.3e-4 max / 1.0e-5 mean -- everywhere to < 1e-4 except at the two destination columns where OpenCV's table drops a tap of less than
    1e-3 source pixels (area_taps_below_threshold: 222 and 802 for 1242 -> 1025), the convention the .bin's producer (resize_area_tf, which
    matches the file to 3e-7) does not have (tests/test_oracle_golden.py).  Only shrinking / same size (what the apps do with KITTI frames)."""
    sh, sw, _ = img_u8.shape
    x = img_u8.astype(np.float64)
    if (sh, sw) != (dst_h, dst_w):
        ty, tx = _area_table(sh, dst_h), _area_table(sw, dst_w)
        rows = np.stack([sum(wgt * x[sy] for sy, wgt in taps) for taps in ty])                 # (dst_h, sw, 3)
        x = np.stack([sum(wgt * rows[:, sx] for sx, wgt in taps) for taps in tx], axis=1)      # (dst_h, dst_w, 3)
    return (x[:, :, ::-1].transpose(2, 0, 1) / 255.0).astype(np.float32)


def disparity_to_u16(disp, scale):
    """main.cpp:324-330: img_f *= 256 (* w); convertTo(CV_16U) = saturate_cast<ushort>(cvRound(x)), round half to even"""
    return np.clip(np.rint(disp.astype(np.float32) * np.float32(scale)), 0, 65535).astype(np.uint16)


def read_weights(path, fp16=False):
    raw = open(path, "rb").read()
    off, out = 0, {}
    dt = np.dtype("<f2") if fp16 else np.dtype("<f4")
    while off < len(raw):
        end = raw.index(b"\0", off)
        name = raw[off:end].decode()
        (cnt,) = struct.unpack_from("<I", raw, end + 1)
        off = end + 5
        out[name] = np.frombuffer(raw, dtype=dt, count=cnt, offset=off).astype(np.float32)
        off += cnt * dt.itemsize
    return out


def write_weights(path, weights, fp16=False):
    with open(path, "wb") as f:
        for name, v in weights.items():
            f.write(name.encode() + b"\0")
            flat = np.asarray(v).reshape(-1)
            f.write(struct.pack("<I", flat.size))
            f.write(flat.astype("<f2" if fp16 else "<f4").tobytes())


from redtail_amd.synth import (NVSMALL_3D, NVTINY_3D, RESNET18_3D, synth_pair, synth_weights_3d,  # noqa: E402,F401
                               synth_weights_resnet18_2d)


# --------------------------------------------------------------------------------------
# Whole networks, op for op as the generated builders call them
# --------------------------------------------------------------------------------------
def _t(w, name, shape):
    return torch.from_numpy(np.ascontiguousarray(w[name])).reshape(shape)


def _conv2d_named(x, w, name, k, stride):
    cin = x.shape[1]
    b = _t(w, name + "_b", (-1,))
    kk = _t(w, name + "_k", (b.numel(), cin, k, k)).to(x.dtype)
    ph = tf_same_pad(x.shape[2], k, stride)
    pw = tf_same_pad(x.shape[3], k, stride)
    assert ph[0] == ph[1] and pw[0] == pw[1], "asymmetric 2-D pad (tensorrt_model_builder.py:203-207)"
    return conv2d(x, kk, b.to(x.dtype), stride, (ph[0], pw[0]))


def _resnet_encoder(x, w, side):
    """sample_app/resnet18_2D_513x257_net.cpp:48-598 (scripts/model_resnet18.py:20-45)."""
    cur = elu(_conv2d_named(x, w, side + "_conv1", 5, 2))
    conv1_act = cur
    for i in range(1, 9):
        p = "%s_resblock%d" % (side, i)
        t = elu(_conv2d_named(cur, w, p + "_conv1", 3, 1))
        t = _conv2d_named(t, w, p + "_conv2", 3, 1)
        cur = elu(t + cur)
    return _conv2d_named(cur, w, side + "_encoder2D_out", 3, 1), conv1_act


def resnet18_2d(left, right, w, max_disp=48, return_intermediates=False):
    """ResNet-18 2D network, sample_app/resnet18_2D_513x257_net.cpp:21-777
    (scripts/model_resnet18_2D.py:16-45).  left/right: (N,3,H,W) in [0,1], H,W = 1 (mod 8).
    Returns (N,1,H,W) sigmoid output = disparity / width."""
    lf, l1 = _resnet_encoder(left, w, "left")
    rf, _ = _resnet_encoder(right, w, "right")
    cv = corr_cost_volume(lf, rf, max_disp)
    sa = softargmax(cv, is_min=False)
    cur = torch.cat([l1, sa], 1)
    acts = {}
    for name, k, s in (("conv2D_1", 3, 1), ("conv2D_2", 3, 1), ("conv2D_3ds", 3, 2), ("conv2D_4", 3, 1),
                       ("conv2D_5", 3, 1), ("conv2D_6ds", 3, 2), ("conv2D_7", 3, 1), ("conv2D_8", 3, 1)):
        cur = elu(_conv2d_named(cur, w, name, k, s))
        acts[name] = cur
    for name, skip in (("deconv2D_1", "conv2D_5"), ("deconv2D_2", "conv2D_2"), ("deconv2D_3", None)):
        b = _t(w, name + "_b", (-1,)).to(cur.dtype)
        kk = _t(w, name + "_k", (cur.shape[1], b.numel(), 3, 3)).to(cur.dtype)
        cur = deconv2d(cur, kk, b, 2, 1)
        if skip is not None:
            cur = elu(cur + acts[skip])
    out = torch.sigmoid(cur)
    if return_intermediates:
        return out, dict(left_feat=lf, right_feat=rf, cost_vol=cv, softargmax=sa)
    return out


def stereo3d(left, right, w, cfg, max_disp, plugin_fp16=False):
    """NVSmall / NVTiny / ResNet-18 (3-D) networks: sample_app/nvsmall_1025x321_net.cpp,
    nvtiny_513x161_net.cpp, resnet18_1025x321_net.cpp (scripts/model_nvsmall.py:18-73,
    scripts/model_resnet18.py:46-85).  max_disp is the half-resolution D of the cost volume.
    Returns (N,1,H,W) disparity in pixels (softargmin).  float64 inputs evaluate the whole graph in fp64.

    plugin_fp16=True restates what the reference's Conv3D / Conv3DTranspose plugins do when they are created with fp16 weights
    (lib/conv3d_plugin.cpp:187-216, 247-274; lib/conv3d_transpose_plugin.cpp): the fp32 input is converted to fp16, cuDNN
    convolves fp16 tensors (the output is an fp16 tensor), the bias is added to that fp16 tensor (cudnnAddTensor), and the
    result is converted back to fp32 -- everything between the plugins (ELU, transform, slice, skip add, soft-argmin) stays fp32."""
    r16 = lambda t: t.half().to(t.dtype)

    def conv3d_p(x, kk, b, *a):
        if not plugin_fp16:
            return conv3d_tf(x, kk, b, *a)
        y = r16(conv3d_tf(r16(x), kk, None, *a))                  # (N, K, D, H, W) fp16 tensor
        return r16(y + b.view(1, -1, 1, 1, 1))

    def deconv3d_p(y, kk, b, *a):
        if not plugin_fp16:
            return conv3d_transpose_tf(y, kk, b, *a)
        x = r16(conv3d_transpose_tf(r16(y), kk, None, *a))        # (N, D, C, H, W)
        return r16(x + b.view(1, 1, -1, 1, 1))

    def enc(x, side):
        if cfg["enc2d"] == "resnet":
            return _resnet_encoder(x, w, side)[0]
        cur = elu(_conv2d_named(x, w, side + "_conv1", 5, 2))
        for l in ("conv2", "conv3", "conv4"):
            cur = elu(_conv2d_named(cur, w, "%s_%s" % (side, l), 3, 1))
        return _conv2d_named(cur, w, side + "_conv5", 3, 1)

    cur = cost_volume(enc(left, "left"), enc(right, "right"), max_disp)    # N D 2C H W
    acts = {}
    last = cfg["conv3d"][-1][0]
    for name, k, c, s in cfg["conv3d"]:
        kk = _t(w, name + "_k", (k, 3, c, 3, 3)).to(cur.dtype)
        b = _t(w, name + "_b", (k,)).to(cur.dtype)
        if s == 2:
            cur = pad_d(cur, 1)                      # Pad plugin is emitted before every *ds conv
            pd = tf_same_pad(cur.shape[1] - 1, 3, 2)
            ph = tf_same_pad(cur.shape[3], 3, 2)
            pw = tf_same_pad(cur.shape[4], 3, 2)
            cur = conv3d_p(cur, kk, b, (2, 2, 2), (pd[0], ph[0], pw[0]), (pd[1], ph[1], pw[1]))
        else:
            cur = conv3d_p(cur, kk, b, (1, 1, 1), (1, 1, 1), (1, 1, 1))
        if name != last:
            cur = transform(cur)                     # KDHW -> DKHW
        cur = elu(cur)
        acts[name] = cur
    for i, (name, k, c, skip) in enumerate(cfg["deconv3d"]):
        kk = _t(w, name + "_k", (k, 3, c, 3, 3)).to(cur.dtype)
        b = _t(w, name + "_b", (c,)).to(cur.dtype)
        dy, hy, wy = cur.shape[2], cur.shape[3], cur.shape[4]
        if skip is not None:
            dx, hx, wx = acts[skip].shape[1], acts[skip].shape[3], acts[skip].shape[4]
        else:
            dx, hx, wx = 2 * dy, 2 * hy - 1, 2 * wy - 1
        pd = tf_same_pad(dx, 3, 2)
        if pd[0] != pd[1]:                           # tensorrt_model_builder.py:422-440
            cur = deconv3d_p(cur, kk, b, (dx + 1, c, hx, wx), (2, 2, 2), (0, 1, 1), (0, 1, 1))
            cur = slice_d(cur, 0, dx)
        else:
            cur = deconv3d_p(cur, kk, b, (dx, c, hx, wx), (2, 2, 2), (pd[0], 1, 1), (pd[1], 1, 1))
        if skip is not None:
            cur = elu(cur + acts[skip])
            cur = transform(cur)                     # DKHW -> KDHW
    return softargmax(cur, is_min=True)
