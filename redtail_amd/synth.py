"""Seeded synthetic inputs for benchmarks, smoke tests and parity tests: KITTI-shaped stereo pairs
 and He-normal weights with the tensor names / shapes of the reference's weight files
(scripts/tensorrt_model_builder.py:52-60).  No network: there are no datasets or checkpoints on the GPU box."""
import numpy as np


def _he(rng, shape, fan_in, gain=1.0):
    return (rng.standard_normal(shape) * gain * np.sqrt(2.0 / fan_in)).astype(np.float32)


def _add_conv(w, rng, name, cout, cin, k, gain=1.0):
    w[name + "_k"] = _he(rng, (cout, cin) + k, cin * int(np.prod(k)), gain).reshape(-1)
    w[name + "_b"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)


def synth_weights_resnet18_2d(seed=7):
    """Seeded He-normal weights with the tensor names/shapes of
    sample_app/resnet18_2D_513x257_net.cpp (both sides share weights, as the real file does)."""
    rng = np.random.default_rng(seed)
    w = {}
    for s in ("left", "right"):
        w[s + "_scale_shift"] = np.zeros(1, np.float32)
        w[s + "_scale_scale"] = np.ones(1, np.float32)
        w[s + "_scale_power"] = np.ones(1, np.float32)
    enc = {}
    _add_conv(enc, rng, "conv1", 32, 3, (5, 5))
    for i in range(1, 9):
        _add_conv(enc, rng, "resblock%d_conv1" % i, 32, 32, (3, 3))
        _add_conv(enc, rng, "resblock%d_conv2" % i, 32, 32, (3, 3), gain=0.5)
    _add_conv(enc, rng, "encoder2D_out", 32, 32, (3, 3), gain=0.7)
    for s in ("left", "right"):
        for k, v in enc.items():
            w[s + "_" + k] = v.copy()
    for name, co, ci in (("conv2D_1", 32, 33), ("conv2D_2", 32, 32), ("conv2D_3ds", 64, 32),
                         ("conv2D_4", 64, 64), ("conv2D_5", 64, 64), ("conv2D_6ds", 128, 64),
                         ("conv2D_7", 128, 128), ("conv2D_8", 128, 128)):
        _add_conv(w, rng, name, co, ci, (3, 3))
    # deconv kernels are stored (Cin, Cout, R, S)
    for name, ci, co in (("deconv2D_1", 128, 64), ("deconv2D_2", 64, 32), ("deconv2D_3", 32, 1)):
        w[name + "_k"] = _he(rng, (ci, co, 3, 3), ci * 9 / 4.0).reshape(-1)
        w[name + "_b"] = rng.uniform(-0.1, 0.1, co).astype(np.float32)
    w["deconv2D_3_k"] *= np.float32(0.1)      # keep the pre-sigmoid output O(1) so the sigmoid does not hide errors
    return w


# 3-D models: (name, K, C, stride) for conv3D_*, (name, K_in, C_out) for deconv3D_*
NVSMALL_3D = dict(
    feat=32, enc2d=("conv1", "conv2", "conv3", "conv4", "conv5"),
    conv3d=[("conv3D_1", 32, 64, 1), ("conv3D_2", 32, 32, 1), ("conv3D_3ds", 64, 32, 2),
            ("conv3D_4", 64, 64, 1), ("conv3D_5", 64, 64, 1), ("conv3D_6ds", 128, 64, 2),
            ("conv3D_7", 128, 128, 1), ("conv3D_8", 128, 128, 1)],
    deconv3d=[("deconv3D_1", 128, 64, "conv3D_5"), ("deconv3D_2", 64, 32, "conv3D_2"),
              ("deconv3D_3", 32, 1, None)])
NVTINY_3D = dict(
    feat=8, enc2d=("conv1", "conv2", "conv3", "conv4", "conv5"),
    conv3d=[("conv3D_1", 16, 16, 1), ("conv3D_2", 16, 16, 1), ("conv3D_3ds", 32, 16, 2),
            ("conv3D_4", 32, 32, 1), ("conv3D_5", 32, 32, 1), ("conv3D_6ds", 64, 32, 2),
            ("conv3D_7", 64, 64, 1), ("conv3D_8", 64, 64, 1)],
    deconv3d=[("deconv3D_1", 64, 32, "conv3D_5"), ("deconv3D_2", 32, 16, "conv3D_2"),
              ("deconv3D_3", 16, 1, None)])
RESNET18_3D = dict(
    feat=32, enc2d="resnet",
    conv3d=[("conv3D_1a", 32, 64, 1), ("conv3D_1b", 32, 32, 1), ("conv3D_1ds", 64, 32, 2),
            ("conv3D_2a", 64, 64, 1), ("conv3D_2b", 64, 64, 1), ("conv3D_2ds", 64, 64, 2),
            ("conv3D_3a", 64, 64, 1), ("conv3D_3b", 64, 64, 1), ("conv3D_3ds", 64, 64, 2),
            ("conv3D_4a", 64, 64, 1), ("conv3D_4b", 64, 64, 1), ("conv3D_4ds", 128, 64, 2),
            ("conv3D_5a", 128, 128, 1), ("conv3D_5b", 128, 128, 1)],
    deconv3d=[("deconv3D_1", 128, 64, "conv3D_4b"), ("deconv3D_2", 64, 64, "conv3D_3b"),
              ("deconv3D_3", 64, 64, "conv3D_2b"), ("deconv3D_4", 64, 32, "conv3D_1b"),
              ("deconv3D_5", 32, 1, None)])


def synth_weights_3d(cfg, seed=7):
    """Seeded weights with the names/shapes of sample_app/{nvsmall,nvtiny,resnet18}_*_net.cpp."""
    rng = np.random.default_rng(seed)
    w = {}
    for s in ("left", "right"):
        w[s + "_scale_shift"] = np.zeros(1, np.float32)
        w[s + "_scale_scale"] = np.ones(1, np.float32)
        w[s + "_scale_power"] = np.ones(1, np.float32)
    enc = {}
    if cfg["enc2d"] == "resnet":
        _add_conv(enc, rng, "conv1", 32, 3, (5, 5))
        for i in range(1, 9):
            _add_conv(enc, rng, "resblock%d_conv1" % i, 32, 32, (3, 3))
            _add_conv(enc, rng, "resblock%d_conv2" % i, 32, 32, (3, 3), gain=0.5)
        _add_conv(enc, rng, "encoder2D_out", 32, 32, (3, 3), gain=0.7)
    else:
        _add_conv(enc, rng, "conv1", 32, 3, (5, 5))
        for l in ("conv2", "conv3", "conv4"):
            _add_conv(enc, rng, l, 32, 32, (3, 3))
        _add_conv(enc, rng, "conv5", cfg["feat"], 32, (3, 3), gain=0.7)
    for s in ("left", "right"):
        for k, v in enc.items():
            w[s + "_" + k] = v.copy()
    for name, k, c, _s in cfg["conv3d"]:
        w[name + "_k"] = _he(rng, (k, 3, c, 3, 3), 27 * c).reshape(-1)
        w[name + "_b"] = rng.uniform(-0.1, 0.1, k).astype(np.float32)
    for name, k, c, _skip in cfg["deconv3d"]:
        w[name + "_k"] = _he(rng, (k, 3, c, 3, 3), 27 * k / 8.0).reshape(-1)
        w[name + "_b"] = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    return w



# --------------------------------------------------------------------------------------
# KITTI-shaped synthetic stereo pairs
# --------------------------------------------------------------------------------------
def synth_pair(h, w, seed=1234):
    """Textured left image + right image = left warped by a smooth disparity field
    d(y,x) = 4 + 60*(y/H) + 8*sin(2*pi*x/W) px, scaled by w/1257.  Returns (3,H,W) x2 in [0,1]."""
    rng = np.random.default_rng(seed)
    img = np.zeros((3, h, w), np.float64)
    for scale, amp in ((1, 0.25), (2, 0.25), (4, 0.2), (8, 0.15), (16, 0.15)):
        hh, ww = -(-h // scale) + 1, -(-w // scale) + 1
        n = rng.uniform(0, 1, (3, hh, ww))
        up = np.kron(n, np.ones((1, scale, scale)))[:, :h, :w]
        img += amp * up
    img += 0.1 * np.linspace(0, 1, w)[None, None, :]
    img = (img - img.min()) / (img.max() - img.min())
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    disp = (4 + 60 * (yy / h) + 8 * np.sin(2 * np.pi * xx / w)) * (w / 1257.0)
    # right[y, x] = left[y, x + d]  (a point at x in the right view sits at x + d in the left)
    src = xx + disp
    x0 = np.floor(src).astype(int)
    fr = src - x0
    valid = (x0 >= 0) & (x0 + 1 < w)
    x0c = np.clip(x0, 0, w - 2)
    right = (1 - fr) * img[:, yy, x0c] + fr * img[:, yy, x0c + 1]
    right = np.where(valid[None], right, 0.0)
    # fancy indexing above leaves `right` in an HWC-strided layout: hand out dense CHW arrays
    return np.ascontiguousarray(img, dtype=np.float32), np.ascontiguousarray(right, dtype=np.float32)


def synth_disparity(h, w):
    """Ground truth of synth_pair: the disparity (in pixels) of every LEFT-image pixel.  synth_pair defines the field on the right view,
    right[y, x] = left[y, x + d(y, x)]; a left pixel x_l is seen at the x_r with x_r + d(y, x_r) = x_l and its disparity is d(y, x_r) --
    solved by fixed-point iteration (d is smooth, |dd/dx| < 0.05).  0 = no ground truth (the match falls outside the right image), the
    KITTI convention (redtail_amd/kitti.py)."""
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    field = lambda x: (4 + 60 * (yy / h) + 8 * np.sin(2 * np.pi * x / w)) * (w / 1257.0)
    d = field(xx.astype(np.float64))
    for _ in range(30):
        d = field(xx - d)
    xr = xx - d
    return np.where((xr >= 0) & (xr <= w - 1), d, 0.0).astype(np.float32)
