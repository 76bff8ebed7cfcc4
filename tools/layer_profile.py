#!/usr/bin/env python3
"""Per-launch times of ResNet-18 2D at 1257x369 through the executor's IProfiler path (production
two-stream schedule).  RT_CONV_VARIANT selects the conv tile variant."""
import os
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi, model_files, synth  # noqa: E402

W, H, b = 1257, 369, int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = capi.NetLib()
net = lib.create("resnet18_2D", W, H, max_batch=b, weights_path=model_files.weight_file("resnet18_2D"))
l, r = synth.synth_pair(H, W)
L = torch.from_numpy(np.stack([l] * b)).cuda()
R = torch.from_numpy(np.stack([r] * b)).cuda()
out = torch.empty(b, 1, H, W, device="cuda")
for _ in range(5):
    net.execute(L, R, out, b)
acc = {}
order = []
runs = 10
for _ in range(runs):
    for name, ms in net.profile(L, R, out, b):
        if name not in acc:
            acc[name] = 0.0
            order.append(name)
        acc[name] += ms
groups = {}
for name in order:
    key = name.replace("left_", "L/R ").replace("right_", "L/R ")
    if "resblock" in key:
        key = "L/R resblock* (block = one launch)" if "+" in name else "L/R resblock*_conv*"
    groups.setdefault(key, [0, 0.0])
    groups[key][0] += 1
    groups[key][1] += acc[name] / runs * 1e3
tot = sum(v[1] for v in groups.values())
for k, (n, us) in groups.items():
    print("%-28s x%-3d %8.1f us total %7.1f us each" % (k, n, us, us / n))
print("sum of launch times %.1f us (two streams overlap)" % tot)
