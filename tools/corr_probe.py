import os
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi, synth
from oracle import stereo_oracle as O
k = capi.KernelLib()
w = synth.synth_weights_resnet18_2d(seed=7)
l, r = synth.synth_pair(369, 1257, 1234)
with torch.no_grad():
    _, im = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], w, return_intermediates=True)
lf, rf = im["left_feat"].cuda().contiguous(), im["right_feat"].cuda().contiguous()
print("feat stats", lf.abs().mean().item(), lf.abs().max().item(), "denormal frac", ((lf.abs() < 1.2e-38) & (lf != 0)).float().mean().item())
def timeit(fn, iters=30):
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    k.lib.rt_event_create(ctypes.byref(e0)); k.lib.rt_event_create(ctypes.byref(e1))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    k.lib.rt_event_record(e0, None)
    for _ in range(iters): fn()
    k.lib.rt_event_record(e1, None)
    ms = ctypes.c_float(); k.check(k.lib.rt_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "el")
    return ms.value * 1e3 / iters
C, H, W, D = 32, 185, 629, 48
sa = torch.empty(1, 1, H, W, device="cuda"); cv = torch.empty(1, D, H, W, device="cuda")
for name, a, b in (("randn", torch.randn_like(lf), torch.randn_like(lf)), ("features", lf, rf), ("zeros", torch.zeros_like(lf), torch.zeros_like(lf)),
                   ("feat*1e-3", lf * 1e-3, rf * 1e-3), ("randn*100", torch.randn_like(lf) * 100, torch.randn_like(lf) * 100)):
    t1 = timeit(lambda: k.corr_softargmax(a, b, sa, 1, C, H, W, D, False))
    t2 = timeit(lambda: k.corr_cost_volume(a, b, cv, 1, C, H, W, D))
    print("%-12s fused %.1f us   plain %.1f us" % (name, t1, t2))
# clock / state hypothesis: corr timed right after a burst of MFMA convolutions on the same stream
wt = (np.random.randn(32 * 32 * 9).astype(np.float32) / 17.0); bias = np.zeros(32, np.float32)
plan = k.conv2d_plan(wt, bias, 32, 32, H, W, 3, 1, 1, act=capi.RT_ACT_ELU, has_residual=True)
x, y, rs = torch.randn(1, 32, H, W, device="cuda"), torch.empty(1, 32, H, W, device="cuda"), torch.randn(1, 32, H, W, device="cuda")
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
k.lib.rt_event_create(ctypes.byref(e0)); k.lib.rt_event_create(ctypes.byref(e1))
for burst in (0, 5, 40):
    tot = 0.0
    for rep in range(10):
        for _ in range(burst): plan.enqueue(x, y, rs, 1)
        k.lib.rt_event_record(e0, None)
        k.corr_softargmax(lf, rf, sa, 1, C, H, W, D, False)
        k.lib.rt_event_record(e1, None)
        torch.cuda.synchronize()
        ms = ctypes.c_float(); k.lib.rt_event_elapsed_ms(e0, e1, ctypes.byref(ms)); tot += ms.value
    print("corr after %d convs: %.1f us" % (burst, tot / 10 * 1e3))
