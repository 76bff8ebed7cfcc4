#!/usr/bin/env python3
"""A few passes of a model for rocprofv3 counter runs (tools/pmc_3d.sh): python tools/iso_3d.py [model] [passes] [--half2] [--batch=N] [--mark]
(the 3-D models, and ResNet-18 2D at the sizes of bench.py's secondary lines)"""
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi, synth  # noqa: E402

CASES = {"nvtiny": (513, 161, synth.NVTINY_3D), "nvsmall": (1025, 321, synth.NVSMALL_3D), "resnet18": (1025, 321, synth.RESNET18_3D),
         # the 2-D model at the sizes of bench.py's secondary lines (C3: 1257x369 half2 batch 8; the reference's published 513x257)
         "resnet18_2D": (1257, 369, None), "resnet18_2D_513": (513, 257, None)}
args = [a for a in sys.argv[1:] if not a.startswith("--")]
model = args[0] if args else "nvsmall"
passes = int(args[1]) if len(args) > 1 else 3
half2 = "--half2" in sys.argv
batch = max([int(a[8:]) for a in sys.argv if a.startswith("--batch=")] + [1])
w, h, cfg = CASES[model]
lib = capi.NetLib()
if cfg is None:
    from redtail_amd import model_files
    net = lib.create("resnet18_2D", w, h, max_batch=batch, weights_path=model_files.weight_file("resnet18_2D", half2), fp16_weights=half2)
    model = "resnet18_2D"
else:
    net = lib.create(model, w, h, max_batch=batch, weights=synth.synth_weights_3d(cfg), fp16_weights=half2)
net.set_streams(1)
if "--mark" in sys.argv:       # the launch trace hashes every launch's output on the launch's stream: one rt::hash_words_kernel dispatch
    net.set_launch_trace(True)  # after each launch, which is what tools/pmc_3d.sh segments the dispatch list by
l, r = synth.synth_pair(h, w, 1234)
L, R = torch.from_numpy(np.stack([l] * batch)).cuda(), torch.from_numpy(np.stack([r] * batch)).cuda()
out = torch.empty(batch, 1, h, w, device="cuda")
for _ in range(passes):
    net.execute(L, R, out, batch)
torch.cuda.synchronize()
print("launches per pass:", [net.launch_name(i) for i in range(net.num_launches)])
