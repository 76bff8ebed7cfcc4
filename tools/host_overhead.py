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
ts = []
for _ in range(50):
    t0 = time.perf_counter(); net.execute(L, R, out, 1, stream=st.cuda_stream); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("host time per enqueue (GPU idle at call): median %.0f us  min %.0f us" % (np.median(ts) * 1e6, min(ts) * 1e6))
