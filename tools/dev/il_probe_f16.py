# proxy for the 3-D layers of half2 mode as 2-D fp16 plans (conv_f16mma_kernel): merged (depth taps x channels) -> K, one launch over D slices
import os, sys, ctypes, numpy as np, torch
os.environ["RT_DEV_KNOBS"]="1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from redtail_amd import capi
k = capi.KernelLib()
def timeit(fn, iters=10):
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    k.lib.rt_event_create(ctypes.byref(e0)); k.lib.rt_event_create(ctypes.byref(e1))
    for _ in range(2): fn()
    torch.cuda.synchronize()
    k.lib.rt_event_record(e0, None)
    for _ in range(iters): fn()
    k.lib.rt_event_record(e1, None)
    torch.cuda.synchronize()
    ms = ctypes.c_float(); k.lib.rt_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value/iters
for name, cin, cout, h, w, b in (("conv3D_2 (3x32 -> 32, 48 slices of 513x161)", 96, 32, 161, 513, 48), ("conv3D_1-like (3x64 -> 32)", 192, 32, 161, 513, 48),
                                 ("conv3D_4/5 (3x64 -> 64, 24 slices of 257x81)", 192, 64, 81, 257, 24), ("conv3D_7/8 (3x128 -> 128, 12 slices of 129x41)", 384, 128, 41, 129, 12)):
    wt = (np.random.randn(cout*cin*9)/np.sqrt(cin*9)).astype(np.float32); bias = np.zeros(cout, np.float32)
    for il in (0, 1):
        plan = k.conv2d_plan(wt, bias, cin, cout, h, w, 3, 1, 1, act=capi.RT_ACT_ELU)
        pitch = (w + 63)//64*64
        plan.set_pitch(pitch, pitch)
        plan.set_io_types(capi.RT_F16, capi.RT_F16)
        if il: plan.set_layouts(1, 1, 0)
        x = torch.randn(b, cin, h, pitch, device="cuda").half(); y = torch.empty(b, cout, h, pitch, device="cuda").half()
        t = timeit(lambda: plan.enqueue(x, y, None, b))
        gflop = 2.0*cin*cout*9*h*w*b/1e9
        print("%-52s %s: %.3f ms  (%.0f TFLOP/s, %.2f of 2.5 PF)" % (name, "il8   " if il else "planar", t, gflop/t, gflop/t/2.5e3))
        plan.destroy()
