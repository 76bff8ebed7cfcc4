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
# proxy for conv3D_2 of NVSmall: merged (3 depth taps x 32 ch) = 96 -> 32, 161x513, 48 output slices = batch 48
for cin, cout, h, w, b in ((96, 32, 161, 513, 48), (192, 32, 161, 513, 48), (192, 64, 81, 257, 24)):
    wt = (np.random.randn(cout*cin*9)/np.sqrt(cin*9)).astype(np.float32); bias = np.zeros(cout, np.float32)
    for il in (0, 1):
        plan = k.conv2d_plan(wt, bias, cin, cout, h, w, 3, 1, 1, act=capi.RT_ACT_ELU)
        pitch = (w + 31)//32*32
        plan.set_pitch(pitch, pitch)
        if il: plan.set_layouts(1, 1, 0)
        x = torch.randn(b, cin, h, pitch, device="cuda"); y = torch.empty(b, cout, h, pitch, device="cuda")
        t = timeit(lambda: plan.enqueue(x, y, None, b))
        gflop = 2.0*cin*cout*9*h*w*b/1e9
        print("%d->%d @%dx%d x%d  %s: %.3f ms  (%.0f TFLOP/s direct form)" % (cin, cout, w, h, b, "interleaved" if il else "planar    ", t, gflop/t/1e3))
        plan.destroy()
