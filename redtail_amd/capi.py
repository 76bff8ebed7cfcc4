"""ctypes binding of the C ABI in include/rt_stereo.h (+ the net-level ABI in include/rt_stereo_net.h).

This is plumbing for tests and bench.py: PyTorch (or numpy under the test-only emulator) owns the
buffers, the C ABI does the work.  There is no Python/torch compute fallback: if the HIP library is
missing, or no GPU is visible, loading fails loudly.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# RT_LIB_DIR: another build of the two libraries (redtail_amd/build.py: build_variant) for A/B measurements with bench.py / tools
_LIB_DIR = os.environ.get("RT_LIB_DIR") or os.path.join(ROOT, "redtail_amd", "lib")
HIP_LIB = os.path.join(_LIB_DIR, "librt_stereo_hip.so")
HOST_LIB = os.path.join(_LIB_DIR, "libnvstereo_inference.so")

RT_F32, RT_F16 = 0, 1
RT_NCHW, RT_NC2HW2 = 0, 1
RT_ACT_NONE, RT_ACT_ELU, RT_ACT_SIGMOID = 0, 1, 2


RT_HINT_THROUGHPUT = 1     # include/rt_stereo.h
RT_CONV_EXACT_FP32 = 1     # rtConv2dDesc.flags / rtConv3dDesc.flags / rtNetOptions.flags


class RtError(RuntimeError):
    pass


class Conv2dDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("Cin", "Cout", "Hin", "Win", "KH", "KW", "stride", "pad_h", "pad_w", "act",
                                     "has_residual", "dtype", "flags")]


class Conv3dDesc(ctypes.Structure):
    _fields_ = [("C", c_int), ("K", c_int), ("D", c_int), ("H", c_int), ("W", c_int), ("kernel", c_int * 3),
                ("stride", c_int * 3), ("pad_start", c_int * 3), ("pad_end", c_int * 3), ("act", c_int),
                ("out_dchw", c_int), ("has_residual", c_int), ("dtype", c_int), ("out_depth", c_int),
                ("in_pad_end", c_int), ("cv_fold", c_int), ("flags", c_int)]


# name -> (restype, argtypes); every symbol declared in include/rt_stereo.h
KERNEL_SYMBOLS = {
    "rt_last_error_string": (c_char_p, []),
    "rt_backend_name": (c_char_p, []),
    "rt_device_count": (c_int, []),
    "rt_set_device": (c_int, [c_int]),
    "rt_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "rt_free": (c_int, [c_void_p]),
    "rt_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "rt_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "rt_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "rt_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "rt_stream_create": (c_int, [POINTER(c_void_p)]),
    "rt_stream_destroy": (c_int, [c_void_p]),
    "rt_stream_sync": (c_int, [c_void_p]),
    "rt_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "rt_event_create": (c_int, [POINTER(c_void_p)]),
    "rt_event_create_ordering": (c_int, [POINTER(c_void_p)]),
    "rt_event_destroy": (c_int, [c_void_p]),
    "rt_event_record": (c_int, [c_void_p, c_void_p]),
    "rt_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "rt_elu": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rt_add_act": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "rt_activation": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "rt_corr_cost_volume": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "rt_corr_cost_volume_flags": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [ctypes.c_uint, c_void_p]),
    "rt_cost_volume": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "rt_softargmax": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "rt_corr_softargmax": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_int64, c_int, c_void_p]),
    "rt_corr_softargmax_pitched": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_int64, c_int, c_void_p]),
    "rt_corr_softargmax_il": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_int64, c_void_p]),
    "rt_corr_softargmax_il_slot": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_int64, c_int, c_void_p]),
    "rt_corr_softargmax_il8_f16": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_int64, c_int, c_void_p]),
    "rt_permute4d": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [POINTER(c_int), c_int, c_void_p]),
    "rt_convert_format": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "rt_pad_d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "rt_slice_d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "rt_concat_channels": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "rt_conv2d_plan_create": (c_int, [POINTER(c_void_p), POINTER(Conv2dDesc), c_void_p, c_void_p]),
    "rt_deconv2d_plan_create": (c_int, [POINTER(c_void_p), POINTER(Conv2dDesc), c_void_p, c_void_p]),
    "rt_resblock_plan_create": (c_int, [POINTER(c_void_p), POINTER(Conv2dDesc), c_void_p, c_void_p, POINTER(Conv2dDesc), c_void_p,
                                        c_void_p]),
    "rt_conv3d_plan_create": (c_int, [POINTER(c_void_p), POINTER(Conv3dDesc), c_void_p, c_void_p]),
    "rt_conv3d_transpose_plan_create": (c_int, [POINTER(c_void_p), POINTER(Conv3dDesc), POINTER(c_int), c_void_p,
                                                c_void_p]),
    "rt_preprocess_bgr8": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rt_disparity_to_u16": (c_int, [c_void_p, c_void_p, c_int64, ctypes.c_float, c_void_p]),
    "rt_conv_plan_input_limit": (c_int, [c_void_p, POINTER(c_float)]),
    "rt_has_experimental": (c_int, []),
    "rt_graph_begin_capture": (c_int, [c_void_p]),
    "rt_graph_end_capture": (c_int, [c_void_p, POINTER(c_void_p)]),
    "rt_graph_launch": (c_int, [c_void_p, c_void_p]),
    "rt_graph_destroy": (c_int, [c_void_p]),
    "rt_check_range": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_float, POINTER(c_float), POINTER(c_int64), c_void_p]),
    "rt_hash_buffer": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "rt_conv_plan_out_dims": (c_int, [c_void_p, POINTER(c_int)]),
    "rt_conv_plan_set_pitch": (c_int, [c_void_p, c_int, c_int]),
    "rt_conv_plan_set_batch_strides": (c_int, [c_void_p, c_int64, c_int64, c_int64]),
    "rt_conv_plan_set_io_types": (c_int, [c_void_p, c_int, c_int]),
    "rt_conv_plan_supports_il8": (c_int, [c_void_p]),
    "rt_conv_plan_set_layouts": (c_int, [c_void_p, c_int, c_int, c_int]),
    "rt_conv_plan_set_softarg": (c_int, [c_void_p, c_int]),
    "rt_conv_plan_supports_twin_input": (c_int, [c_void_p]),
    "rt_conv_enqueue_twin_input": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "rt_resblock_plan_supports_split": (c_int, [c_void_p]),
    "rt_resblock_plan_set_split": (c_int, [c_void_p, c_int, c_int]),
    "rt_conv_enqueue": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rt_conv_enqueue_hint": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "rt_conv_plan_workspace_bytes": (ctypes.c_size_t, [c_void_p, c_int]),
    "rt_conv_enqueue_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, ctypes.c_size_t, c_void_p, c_int]),
    "rt_conv_plan_destroy": (c_int, [c_void_p]),
    # multi-GPU: RCCL communicator + byte broadcast (librccl is loaded on first use)
    "rt_comm_unique_id": (c_int, [c_void_p]),
    "rt_comm_init_rank": (c_int, [POINTER(c_void_p), c_int, c_int, c_void_p]),
    "rt_comm_init_all": (c_int, [POINTER(c_void_p), c_int, POINTER(c_int)]),
    "rt_comm_adopt": (c_int, [POINTER(c_void_p), c_void_p]),
    "rt_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "rt_comm_broadcast": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "rt_comm_group_start": (c_int, []),
    "rt_comm_group_end": (c_int, []),
    "rt_comm_destroy": (c_int, [c_void_p]),
}
RT_COMM_ID_BYTES = 128


def _ptr(x):
    """Raw device address of a torch tensor / numpy array (emulator) / int."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        return x.ctypes.data
    raise TypeError("cannot take the address of %r" % type(x))


def _host_weights(a, dtype, what):
    """Plan constructors read host weight / bias arrays through a raw pointer: refuse an array of another element type (a float32
    array divided by np.sqrt(...) is float64 under NumPy 2 and would be read as garbage)."""
    if a is not None and hasattr(a, "dtype") and hasattr(a, "ctypes"):
        want = "float16" if dtype == RT_F16 else "float32"
        if a.dtype.name != want:
            raise TypeError("%s: expected a %s array, got %s" % (what, want, a.dtype))
        if not a.flags["C_CONTIGUOUS"]:
            raise TypeError("%s: array is not C-contiguous" % what)
    return a


class KernelLib:
    """The op-level C ABI.  `path=None` loads the gfx950 build and requires a visible GPU."""

    def __init__(self, path=None):
        emulated = path is not None
        path = path or HIP_LIB
        if not os.path.exists(path):
            raise RtError("native library %s is missing -- run `python -m redtail_amd.build` "
                          "(there is no Python fallback for the HIP path)" % path)
        self.path = path
        self.lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in KERNEL_SYMBOLS.items():
            fn = getattr(self.lib, name)      # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        if not emulated and self.lib.rt_device_count() < 1:
            raise RtError("no HIP device visible: the Stereo DNN path runs on MI355X only")

    # -- helpers ---------------------------------------------------------------------------------
    def check(self, rc, what=""):
        if rc != 0:
            raise RtError("%s failed (%d): %s" % (what, rc, self.lib.rt_last_error_string().decode()))

    def backend(self):
        return self.lib.rt_backend_name().decode()

    # -- ops (x/y are torch CUDA tensors, or numpy arrays under the emulator) -----------------------
    def elu(self, x, y, n, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_elu(_ptr(x), _ptr(y), n, dtype, stream), "rt_elu")

    def add_act(self, a, b, y, n, act, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_add_act(_ptr(a), _ptr(b), _ptr(y), n, act, dtype, stream), "rt_add_act")

    def activation(self, x, y, n, act, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_activation(_ptr(x), _ptr(y), n, act, dtype, stream), "rt_activation")

    def corr_cost_volume(self, l, r, cv, batch, C, H, W, D, dtype=RT_F32, fmt=RT_NCHW, stream=None, flags=0):
        self.check(self.lib.rt_corr_cost_volume_flags(_ptr(l), _ptr(r), _ptr(cv), batch, C, H, W, D, dtype, fmt, flags, stream)
                   if flags else self.lib.rt_corr_cost_volume(_ptr(l), _ptr(r), _ptr(cv), batch, C, H, W, D, dtype, fmt, stream),
                   "rt_corr_cost_volume")

    def preprocess_bgr8(self, src, src_h, src_w, dst, dst_h, dst_w, batch=1, stream=None):
        self.check(self.lib.rt_preprocess_bgr8(_ptr(src), src_h, src_w, _ptr(dst), dst_h, dst_w, batch, stream),
                   "rt_preprocess_bgr8")

    def disparity_to_u16(self, disp, out, n, scale, stream=None):
        self.check(self.lib.rt_disparity_to_u16(_ptr(disp), _ptr(out), n, scale, stream), "rt_disparity_to_u16")

    def corr_softargmax_pitched(self, l, r, out, batch, C, H, W, D, is_min, in_pitch, out_pitch, out_bstride=0,
                                dtype=RT_F32, stream=None):
        self.check(self.lib.rt_corr_softargmax_pitched(_ptr(l), _ptr(r), _ptr(out), batch, C, H, W, D, int(is_min),
                                                       in_pitch, out_pitch, out_bstride, dtype, stream),
                   "rt_corr_softargmax_pitched")

    def corr_softargmax_il(self, l, r, out, batch, C, H, W, D, is_min, in_pitch=0, out_pitch=0, out_bstride=0, stream=None, out_slot=1):
        """out_slot = 4: the map as lane 0 of the 16-byte slots of an interleaved group, (H, out_pitch, 4), zeros in lanes 1..3"""
        self.check(self.lib.rt_corr_softargmax_il_slot(_ptr(l), _ptr(r), _ptr(out), batch, C, H, W, D, int(is_min), in_pitch, out_pitch,
                                                       out_bstride, out_slot, stream), "rt_corr_softargmax_il_slot")

    def corr_softargmax_il8_f16(self, l, r, out, batch, C, H, W, D, is_min, in_pitch=0, out_pitch=0, out_bstride=0, stream=None, out_slot=1):
        """half2 mode: fp16 (C/8, H, pitch, 8) feature maps, fp16 map out (out_slot = 8: lane 0 of the 16-byte slots of a group of 8)"""
        self.check(self.lib.rt_corr_softargmax_il8_f16(_ptr(l), _ptr(r), _ptr(out), batch, C, H, W, D, int(is_min), in_pitch, out_pitch,
                                                       out_bstride, out_slot, stream), "rt_corr_softargmax_il8_f16")

    def cost_volume(self, l, r, cv, batch, C, H, W, D, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_cost_volume(_ptr(l), _ptr(r), _ptr(cv), batch, C, H, W, D, dtype, stream),
                   "rt_cost_volume")

    def softargmax(self, vol, out, batch, D, H, W, is_min, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_softargmax(_ptr(vol), _ptr(out), batch, D, H, W, int(is_min), dtype, stream),
                   "rt_softargmax")

    def corr_softargmax(self, l, r, out, batch, C, H, W, D, is_min, out_bstride=0, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_corr_softargmax(_ptr(l), _ptr(r), _ptr(out), batch, C, H, W, D, int(is_min),
                                               out_bstride, dtype, stream), "rt_corr_softargmax")

    def permute4d(self, x, y, batch, dims, order, dtype=RT_F32, stream=None):
        o = (c_int * 4)(*order)
        self.check(self.lib.rt_permute4d(_ptr(x), _ptr(y), batch, dims[0], dims[1], dims[2], dims[3], o, dtype,
                                         stream), "rt_permute4d")

    def pad_d(self, x, y, batch, D, inner, pad_end, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_pad_d(_ptr(x), _ptr(y), batch, D, inner, pad_end, dtype, stream), "rt_pad_d")

    def slice_d(self, x, y, batch, D, inner, start, end, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_slice_d(_ptr(x), _ptr(y), batch, D, inner, start, end, dtype, stream), "rt_slice_d")

    def concat_channels(self, x, y, batch, C, Ctot, c_off, inner, dtype=RT_F32, stream=None):
        self.check(self.lib.rt_concat_channels(_ptr(x), _ptr(y), batch, C, Ctot, c_off, inner, dtype, stream),
                   "rt_concat_channels")

    # -- convolution plans -------------------------------------------------------------------------
    def has_experimental(self):
        """the library carries the rejected kernel families (RT_EXPERIMENTAL build: the emulator library of the CPU test tier)"""
        return bool(self.lib.rt_has_experimental())

    def check_range(self, x, rows, valid, pitch, dtype=RT_F32, limit=65504.0, stream=None):
        """(max finite |x|, number of elements with |x| >= limit or non-finite) of a device tensor (rt_check_range)"""
        mx, bad = c_float(), c_int64()
        self.check(self.lib.rt_check_range(_ptr(x), rows, valid, pitch, dtype, limit, ctypes.byref(mx), ctypes.byref(bad), stream), "rt_check_range")
        return mx.value, bad.value

    def conv2d_plan(self, w_host, b_host, Cin, Cout, Hin, Win, k, stride, pad, act=0, has_residual=False,
                    dtype=RT_F32, transposed=False, flags=0):
        d = Conv2dDesc(Cin, Cout, Hin, Win, k, k, stride, pad, pad, act, int(has_residual), dtype, flags)
        _host_weights(w_host, dtype, "conv2d_plan weights"); _host_weights(b_host, dtype, "conv2d_plan bias")
        plan = c_void_p()
        fn = self.lib.rt_deconv2d_plan_create if transposed else self.lib.rt_conv2d_plan_create
        self.check(fn(ctypes.byref(plan), ctypes.byref(d), _ptr(w_host), _ptr(b_host)), "conv2d plan")
        return ConvPlan(self, plan)

    def resblock_plan(self, w1, b1, w2, b2, C, Cmid, H, W, act1=RT_ACT_ELU, act2=RT_ACT_ELU, dtype=RT_F32):
        """fused residual block y = act2(conv3x3(act1(conv3x3(x) + b1)) + b2 + x)"""
        d1 = Conv2dDesc(C, Cmid, H, W, 3, 3, 1, 1, 1, act1, 0, dtype)
        d2 = Conv2dDesc(Cmid, C, H, W, 3, 3, 1, 1, 1, act2, 1, dtype)
        plan = c_void_p()
        self.check(self.lib.rt_resblock_plan_create(ctypes.byref(plan), ctypes.byref(d1), _ptr(w1), _ptr(b1), ctypes.byref(d2),
                                                    _ptr(w2), _ptr(b2)), "resblock plan")
        return ConvPlan(self, plan)

    def conv3d_plan(self, w_host, b_host, C, K, dims, kernel, stride, pad_start, pad_end, act=0, out_dchw=False,
                    has_residual=False, dtype=RT_F32, transposed_in_dims=None, out_depth=0, in_pad_end=0, cv_fold=0):
        d = Conv3dDesc(C, K, dims[0], dims[1], dims[2], (c_int * 3)(*kernel), (c_int * 3)(*stride),
                       (c_int * 3)(*pad_start), (c_int * 3)(*pad_end), act, int(out_dchw), int(has_residual), dtype,
                       out_depth, in_pad_end, cv_fold)
        plan = c_void_p()
        if transposed_in_dims is None:
            rc = self.lib.rt_conv3d_plan_create(ctypes.byref(plan), ctypes.byref(d), _ptr(w_host), _ptr(b_host))
        else:
            ind = (c_int * 3)(*transposed_in_dims)
            rc = self.lib.rt_conv3d_transpose_plan_create(ctypes.byref(plan), ctypes.byref(d), ind, _ptr(w_host),
                                                          _ptr(b_host))
        self.check(rc, "conv3d plan")
        return ConvPlan(self, plan)


class ConvPlan:
    def __init__(self, klib, handle):
        self.klib, self.handle = klib, handle
        dims = (c_int * 4)()
        klib.check(klib.lib.rt_conv_plan_out_dims(handle, dims), "rt_conv_plan_out_dims")
        self.out_dims = tuple(dims)

    def set_pitch(self, in_pitch, out_pitch):
        self.klib.check(self.klib.lib.rt_conv_plan_set_pitch(self.handle, in_pitch, out_pitch), "rt_conv_plan_set_pitch")

    def set_batch_strides(self, x_bstride, y_bstride, r_bstride=0):
        """per-sample strides in elements of input / output / residual (0 = leave as it is)"""
        self.klib.check(self.klib.lib.rt_conv_plan_set_batch_strides(self.handle, x_bstride, y_bstride, r_bstride), "rt_conv_plan_set_batch_strides")

    def set_io_types(self, x_dtype, y_dtype):
        self.klib.check(self.klib.lib.rt_conv_plan_set_io_types(self.handle, x_dtype, y_dtype), "rt_conv_plan_set_io_types")

    def input_limit(self):
        """bound on |x| the plan's arithmetic needs (65504 for fp16-pipe plans, inf for exact fp32)"""
        v = c_float()
        self.klib.check(self.klib.lib.rt_conv_plan_input_limit(self.handle, ctypes.byref(v)), "rt_conv_plan_input_limit")
        return v.value

    def supports_il8(self):
        return bool(self.klib.lib.rt_conv_plan_supports_il8(self.handle))

    def il_caps(self):
        """which tensors may be channel-interleaved: bit 0 input, bit 1 output, bit 2 residual, bit 3 output only with an interleaved input,
        bit 4 input with its channel count padded to a whole group"""
        return int(self.klib.lib.rt_conv_plan_supports_il8(self.handle))

    def set_layouts(self, x_il8, y_il8, r_il8=False):
        """channel-interleaved (C/8, H, pitch, 8) fp16 tensors: input / output / residual"""
        self.klib.check(self.klib.lib.rt_conv_plan_set_layouts(self.handle, int(x_il8), int(y_il8), int(r_il8)), "rt_conv_plan_set_layouts")

    def enqueue_twin_input(self, x, x2, y, batch=1, stream=None, hints=0):
        """first layer of both towers: samples [0, batch) from x, [batch, 2 batch) from x2"""
        self.klib.check(self.klib.lib.rt_conv_enqueue_twin_input(self.handle, _ptr(x), _ptr(x2), _ptr(y), batch, stream, hints), "rt_conv_enqueue_twin_input")

    def supports_split(self):
        return bool(self.klib.lib.rt_resblock_plan_supports_split(self.handle))

    def set_split(self, x_split, y_split):
        """pre-split tensors (C/8, H, pitch, [8 hi | 8 lo]) as input / output of a tower block"""
        self.klib.check(self.klib.lib.rt_resblock_plan_set_split(self.handle, int(x_split), int(y_split)), "rt_resblock_plan_set_split")

    def set_softarg(self, mode):
        """end the launch in the soft-argmax (1) / soft-argmin (2) over the output depth (last Conv3DTranspose of a 3-D model): y becomes the
        (batch, 1, H, W) map; raises RtError (RT_E_UNSUPPORTED) for a plan that has no such form"""
        self.klib.check(self.klib.lib.rt_conv_plan_set_softarg(self.handle, int(mode)), "rt_conv_plan_set_softarg")

    def enqueue(self, x, y, residual=None, batch=1, stream=None, hints=0):
        self.klib.check(self.klib.lib.rt_conv_enqueue_hint(self.handle, _ptr(x), _ptr(y), _ptr(residual), batch, stream, hints),
                        "rt_conv_enqueue")

    def destroy(self):
        if self.handle:
            self.klib.lib.rt_conv_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------
# whole-network ABI (include/rt_stereo_net.h, libnvstereo_inference.so)
# ---------------------------------------------------------------------------------------------------
RT_MODEL_RESNET18_2D, RT_MODEL_NVSMALL, RT_MODEL_NVTINY, RT_MODEL_RESNET18 = 0, 1, 2, 3
MODEL_IDS = {"resnet18_2D": RT_MODEL_RESNET18_2D, "nvsmall": RT_MODEL_NVSMALL, "nvtiny": RT_MODEL_NVTINY,
             "resnet18": RT_MODEL_RESNET18}

NET_SYMBOLS = {
    "rt_net_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, c_char_p]),
    "rt_net_create_from_memory": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                          c_size_t]),
    "rt_net_serialize": (c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_size_t)]),
    "rt_net_create_from_plan": (c_int, [POINTER(c_void_p), c_void_p, c_size_t]),
    "rt_net_execute": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rt_net_profile": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_char_p, c_size_t]),
    "rt_net_num_layers": (c_int, [c_void_p]),
    "rt_net_num_launches": (c_int, [c_void_p]),
    "rt_net_set_streams": (c_int, [c_void_p, c_int]),
    "rt_net_set_graph": (c_int, [c_void_p, c_int]),
    "rt_net_create_broadcast": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int]),
    "rt_net_weights_crc32": (c_int, [c_void_p, POINTER(ctypes.c_uint32)]),
    "rt_net_weights_image": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]),
    "rt_net_create_opt": (c_int, [POINTER(c_void_p), c_void_p]),
    "rt_net_set_debug": (c_int, [c_void_p, c_int]),
    "rt_net_set_launch_trace": (c_int, [c_void_p, c_int]),
    "rt_net_read_launch_trace": (c_int, [c_void_p, POINTER(ctypes.c_ulonglong), c_int]),
    "rt_net_launch_name": (c_char_p, [c_void_p, c_int]),
    "rt_net_read_launch_output": (ctypes.c_longlong, [c_void_p, c_int, c_void_p, ctypes.c_longlong]),
    "rt_net_destroy": (c_int, [c_void_p]),
    "rt_net_last_error": (c_char_p, []),
}


class NetOptions(ctypes.Structure):
    """rtNetOptions (include/rt_stereo_net.h)"""
    _fields_ = [("model", c_int), ("width", c_int), ("height", c_int), ("max_batch", c_int), ("weights_dtype", c_int), ("max_disp", c_int),
                ("weights_path", c_char_p), ("blob", c_void_p), ("bytes", c_size_t), ("flags", ctypes.c_uint), ("comm", c_void_p), ("root", c_int)]


def pack_weights(weights, fp16=False):
    """dict name -> array  ==>  bytes in trt_weights.bin layout (scripts/tensorrt_model_builder.py:52-60)."""
    import struct

    import numpy as np
    out = bytearray()
    for name, v in weights.items():
        flat = np.asarray(v).reshape(-1)
        out += name.encode() + b"\0" + struct.pack("<I", flat.size)
        out += flat.astype("<f2" if fp16 else "<f4").tobytes()
    return bytes(out)


def read_weights(path, fp16=False):
    """trt_weights.bin / trt_weights_fp16.bin -> dict name -> float32 array (inverse of pack_weights; the reader of
    sample_app/main.cpp:111-134: name, NUL, uint32 count, count elements)."""
    import struct

    import numpy as np
    raw = open(path, "rb").read()
    off, out = 0, {}
    dt = np.dtype("<f2") if fp16 else np.dtype("<f4")
    while off < len(raw):
        end = raw.index(b"\0", off)
        (cnt,) = struct.unpack_from("<I", raw, end + 1)
        out[raw[off:end].decode()] = np.frombuffer(raw, dtype=dt, count=cnt, offset=end + 5).astype(np.float32)
        off = end + 5 + cnt * dt.itemsize
    return out


class NetLib:
    """libnvstereo_inference.so: the NvInfer.h shim + plugins + executor, through the C ABI."""

    def __init__(self, host_path=None, kernels_path=None):
        self.kernels = KernelLib(kernels_path)          # loads (and checks) the kernel library first
        path = host_path or HOST_LIB
        if not os.path.exists(path):
            raise RtError("native library %s is missing -- run `python -m redtail_amd.build`" % path)
        self.lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in NET_SYMBOLS.items():
            fn = getattr(self.lib, name)
            fn.restype, fn.argtypes = res, args

    def check(self, rc, what):
        if rc != 0:
            raise RtError("%s failed (%d): %s" % (what, rc, self.lib.rt_net_last_error().decode()))

    def create(self, model, width, height, max_batch=1, weights=None, weights_path=None, fp16_weights=False,
               max_disp=0, flags=0):
        h = c_void_p()
        dt = RT_F16 if fp16_weights else RT_F32
        mid = MODEL_IDS[model] if isinstance(model, str) else model
        if flags:                                          # options beyond the plain entries: rt_net_create_opt
            blob = None if weights_path is not None else (weights if isinstance(weights, (bytes, bytearray)) else pack_weights(weights, fp16_weights))
            keep = ctypes.create_string_buffer(bytes(blob), len(blob)) if blob is not None else None
            o = NetOptions(mid, width, height, max_batch, dt, max_disp, weights_path.encode() if weights_path else None,
                           ctypes.cast(keep, c_void_p) if keep is not None else None, len(blob) if blob is not None else 0, flags, None, 0)
            self.check(self.lib.rt_net_create_opt(ctypes.byref(h), ctypes.byref(o)), "rt_net_create_opt")
            return StereoNet(self, h, width, height)
        if weights_path is not None:
            rc = self.lib.rt_net_create(ctypes.byref(h), mid, width, height, max_batch, dt, max_disp,
                                        weights_path.encode())
        else:
            blob = weights if isinstance(weights, (bytes, bytearray)) else pack_weights(weights, fp16_weights)
            rc = self.lib.rt_net_create_from_memory(ctypes.byref(h), mid, width, height, max_batch, dt, max_disp,
                                                    blob, len(blob))
        self.check(rc, "rt_net_create")
        return StereoNet(self, h, width, height)

    # ---- multi-GPU start-up through the native RCCL entry (include/rt_stereo.h: rt_comm_*, rt_stereo_net.h: rt_net_create_broadcast)
    def comm_unique_id(self):
        """rank 0: the 128 bytes every other rank needs for comm_init_rank (ncclGetUniqueId)"""
        buf = ctypes.create_string_buffer(RT_COMM_ID_BYTES)
        self.kernels.check(self.kernels.lib.rt_comm_unique_id(buf), "rt_comm_unique_id")
        return buf.raw

    def comm_init_rank(self, world, rank, unique_id):
        """ncclCommInitRank on the current device (rt_set_device first); returns an opaque handle"""
        h = c_void_p()
        self.kernels.check(self.kernels.lib.rt_comm_init_rank(ctypes.byref(h), world, rank, unique_id), "rt_comm_init_rank")
        return h

    def comm_init_all(self, ndev, devices=None):
        """ncclCommInitAll: one communicator per device of THIS process (apps/stereo_throughput.cpp: one thread per device)"""
        comms = (c_void_p * ndev)()
        devs = (c_int * ndev)(*devices) if devices is not None else None
        self.kernels.check(self.kernels.lib.rt_comm_init_all(ctypes.cast(comms, POINTER(c_void_p)), ndev, devs), "rt_comm_init_all")
        return [c_void_p(comms[i]) for i in range(ndev)]

    def comm_destroy(self, comm):
        self.kernels.check(self.kernels.lib.rt_comm_destroy(comm), "rt_comm_destroy")

    def create_broadcast(self, model, width, height, comm, root=0, max_batch=1, blob=None, fp16_weights=False, max_disp=0):
        """every rank calls this; rank `root` passes the weight-file image, the others receive it over RCCL (ncclBroadcast)"""
        h = c_void_p()
        dt = RT_F16 if fp16_weights else RT_F32
        mid = MODEL_IDS[model] if isinstance(model, str) else model
        rc = self.lib.rt_net_create_broadcast(ctypes.byref(h), mid, width, height, max_batch, dt, max_disp, blob, len(blob) if blob else 0, comm, root)
        self.check(rc, "rt_net_create_broadcast")
        return StereoNet(self, h, width, height)

    def create_from_plan(self, plan, width, height):
        """IRuntime::deserializeCudaEngine on bytes returned by StereoNet.serialize()"""
        h = c_void_p()
        self.check(self.lib.rt_net_create_from_plan(ctypes.byref(h), plan, len(plan)), "rt_net_create_from_plan")
        return StereoNet(self, h, width, height)


class StereoNet:
    def __init__(self, netlib, handle, width, height):
        self.netlib, self.handle, self.width, self.height = netlib, handle, width, height

    @property
    def num_layers(self):
        return self.netlib.lib.rt_net_num_layers(self.handle)

    @property
    def num_launches(self):
        return self.netlib.lib.rt_net_num_launches(self.handle)

    def execute(self, left, right, disp, batch=1, stream=None):
        self.netlib.check(self.netlib.lib.rt_net_execute(self.handle, _ptr(left), _ptr(right), _ptr(disp), batch, stream),
                          "rt_net_execute")

    def set_debug(self, on=True):
        """IExecutionContext::setDebugSync: synchronise every launch and range-check the input of every fp16-pipe convolution"""
        self.netlib.check(self.netlib.lib.rt_net_set_debug(self.handle, int(on)), "rt_net_set_debug")

    def weights_crc32(self):
        """crc32 of the weight-file image this engine was built from (== zlib.crc32 of the file)"""
        c = ctypes.c_uint32()
        self.netlib.check(self.netlib.lib.rt_net_weights_crc32(self.handle, ctypes.byref(c)), "rt_net_weights_crc32")
        return c.value

    def weights_image(self):
        """bytes of the weight-file image the engine holds (what a non-root rank received by broadcast)"""
        p, n = c_void_p(), c_size_t()
        self.netlib.check(self.netlib.lib.rt_net_weights_image(self.handle, ctypes.byref(p), ctypes.byref(n)), "rt_net_weights_image")
        return ctypes.string_at(p.value, n.value)

    def set_launch_trace(self, on=True):
        """hash every launch's output on its own stream (debugging aid): read_launch_trace() after a pass"""
        self.netlib.check(self.netlib.lib.rt_net_set_launch_trace(self.handle, int(on)), "rt_net_set_launch_trace")

    def read_launch_trace(self):
        n = self.num_launches
        buf = (ctypes.c_ulonglong * n)()
        got = self.netlib.lib.rt_net_read_launch_trace(self.handle, buf, n)
        if got < 0:
            raise RtError("rt_net_read_launch_trace failed: " + self.netlib.lib.rt_net_last_error().decode())
        return list(buf[:got])

    def launch_name(self, i):
        s = self.netlib.lib.rt_net_launch_name(self.handle, i)
        return s.decode() if s else None

    def read_launch_output(self, i):
        """raw bytes of launch i's output tensor as stored on the device (numpy uint8 array)"""
        import numpy as np
        n = self.netlib.lib.rt_net_read_launch_output(self.handle, i, None, 0)
        if n < 0:
            raise RtError("rt_net_read_launch_output failed: " + self.netlib.lib.rt_net_last_error().decode())
        out = np.empty(n, np.uint8)
        if self.netlib.lib.rt_net_read_launch_output(self.handle, i, out.ctypes.data, n) != n:
            raise RtError("rt_net_read_launch_output failed: " + self.netlib.lib.rt_net_last_error().decode())
        return out

    def set_streams(self, n):
        """1: all launches on the caller's stream (throughput with several contexts); 2: second stream for the right encoder"""
        self.netlib.check(self.netlib.lib.rt_net_set_streams(self.handle, n), "rt_net_set_streams")

    def set_graph(self, on):
        """graph mode: repeated executes with the same pointers replay one captured hipGraph (rt_net_set_graph)"""
        self.netlib.check(self.netlib.lib.rt_net_set_graph(self.handle, int(bool(on))), "rt_net_set_graph")

    def profile(self, left, right, disp, batch=1):
        buf = ctypes.create_string_buffer(1 << 16)
        self.netlib.check(self.netlib.lib.rt_net_profile(self.handle, _ptr(left), _ptr(right), _ptr(disp), batch, buf,
                                                         len(buf)), "rt_net_profile")
        rows = []
        for line in buf.value.decode().splitlines():
            name, ms = line.rsplit("\t", 1)
            rows.append((name, float(ms)))
        return rows

    def serialize(self):
        """engine plan bytes (ICudaEngine::serialize)"""
        n = ctypes.c_size_t()
        self.netlib.check(self.netlib.lib.rt_net_serialize(self.handle, None, 0, ctypes.byref(n)), "rt_net_serialize")
        buf = ctypes.create_string_buffer(n.value)
        self.netlib.check(self.netlib.lib.rt_net_serialize(self.handle, buf, n.value, ctypes.byref(n)), "rt_net_serialize")
        return buf.raw

    def destroy(self):
        if self.handle:
            self.netlib.lib.rt_net_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
