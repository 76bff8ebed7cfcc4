import os, sys, numpy as np, torch
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
sys.path.insert(0, '/root/repo')
from redtail_amd import capi, synth
lib = capi.NetLib()
net = lib.create("resnet18_2D", 1257, 369, max_batch=1, weights=synth.synth_weights_resnet18_2d())
l, r = synth.synth_pair(369, 1257)
L = torch.from_numpy(l[None]).cuda(); R = torch.from_numpy(r[None]).cuda(); out = torch.empty(1, 1, 369, 1257, device="cuda")
for _ in range(5): net.execute(L, R, out, 1)
for _ in range(10): net.profile(L, R, out, 1)
os.environ["RT_PROFILE_TIMELINE"] = "1"
sys.stderr.write("BEGIN\n")
net.profile(L, R, out, 1)
