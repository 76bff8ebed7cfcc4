#!/usr/bin/env python3
"""Host-side cost of one IExecutionContext::enqueue (49 launches + cross-stream events) of ResNet-18 2D."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi, synth
lib = capi.NetLib()
net = lib.create("resnet18_2D", 1257, 369, max_batch=1, weights=synth.synth_weights_resnet18_2d())
l, r = synth.synth_pair(369, 1257)
L, R = torch.from_numpy(l[None]).cuda(), torch.from_numpy(r[None]).cuda()
out = torch.empty(1, 1, 369, 1257, device="cuda")
st = torch.cuda.Stream()
for _ in range(5): net.execute(L, R, out, 1, stream=st.cuda_stream)
torch.cuda.synchronize()
ts, lat = [], []
for _ in range(200):
    t0 = time.perf_counter(); net.execute(L, R, out, 1, stream=st.cuda_stream); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
print("host time per enqueue (GPU idle at call): median %.0f us  min %.0f us" % (np.median(ts) * 1e6, min(ts) * 1e6))
print("synchronous loop (the reference's: execute, wait, next; sample_app/main.cpp:303-309): median %.0f us per pair = %.0f pairs/s" % (
    np.median(lat[50:]) * 1e6, 1.0 / np.median(lat[50:])))
lat = []
for _ in range(200):
    t0 = time.perf_counter(); net.execute(L, R, out, 1); lat.append(time.perf_counter() - t0)
print("IExecutionContext::execute() in a loop: median %.0f us per pair = %.0f pairs/s" % (np.median(lat[50:]) * 1e6, 1.0 / np.median(lat[50:])))
# the same calls back to back (the bench loop): does the host run ahead of the GPU, or does a call block until the previous step
# has drained?  host time per call and the time of the whole loop (GPU-bound if the host is ahead)
ts = []
t_all = time.perf_counter()
for _ in range(60):
    t0 = time.perf_counter(); net.execute(L, R, out, 1, stream=st.cuda_stream); ts.append(time.perf_counter() - t0)
t_issue = time.perf_counter() - t_all
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
print("pipelined: host time per enqueue median %.0f us, first ten %s; all 60 issued after %.1f ms, finished after %.1f ms" % (
    np.median(ts) * 1e6, " ".join("%.0f" % (t * 1e6) for t in ts[:10]), t_issue * 1e3, t_all * 1e3))
# does execute()'s NULL-stream ordering get more expensive with more contexts alive in the process?
extra = [lib.create("resnet18_2D", 1257, 369, max_batch=1, weights=synth.synth_weights_resnet18_2d()) for _ in range(3)]
outs = [torch.empty_like(out) for _ in extra]
for n2, o2 in zip(extra, outs):
    n2.execute(L, R, o2, 1, stream=st.cuda_stream)
torch.cuda.synchronize()
lat = []
for _ in range(200):
    t0 = time.perf_counter(); net.execute(L, R, out, 1); lat.append(time.perf_counter() - t0)
print("execute() loop with three more contexts alive: median %.0f us per pair" % (np.median(lat[50:]) * 1e6))
