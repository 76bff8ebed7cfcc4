#!/usr/bin/env python3
"""Where does the end-to-end error of a deep 3-D model come from?  Compares the GPU result (Winograd on / off) and the
fp32 oracle against an fp64 evaluation of the same graph (ResNet-18 3D, 1025x321, synthetic weights)."""
import os, sys, time
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stereo_oracle as O
from redtail_amd import capi, synth
model, cfg, w_, h_, disp = "resnet18", synth.RESNET18_3D, 1025, 321, 68
weights = synth.synth_weights_3d(cfg)
l, r = synth.synth_pair(h_, w_, 1234)
L, R = torch.from_numpy(l[None]), torch.from_numpy(r[None])
torch.set_num_threads(16)
t0 = time.time()
with torch.no_grad():
    ref32 = O.stereo3d(L, R, weights, cfg, disp)
    w64 = {k: np.asarray(v, np.float64) for k, v in weights.items()}
    ref64 = O.stereo3d(L.double(), R.double(), w64, cfg, disp)
print("oracles %.0f s; fp32 oracle vs fp64: %.3e px" % (time.time() - t0, (ref32.double() - ref64).abs().max().item()))
for nowino in ("0", "1"):
    os.environ["RT_CONV_NO_WINO"] = nowino
    lib = capi.NetLib()
    net = lib.create(model, w_, h_, weights=weights, max_disp=disp)
    out = torch.empty(1, 1, h_, w_, device="cuda")
    net.execute(L.cuda(), R.cuda(), out, 1); torch.cuda.synchronize()
    o = out.cpu().double()
    print("GPU (winograd %s) vs fp64: %.3e px   vs fp32 oracle: %.3e px" % ("off" if nowino == "1" else "on", (o - ref64).abs().max().item(), (o - ref32.double()).abs().max().item()))
    net.destroy()
