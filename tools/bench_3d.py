#!/usr/bin/env python3
"""End-to-end timing of the 3-D Stereo DNN models (NVTiny 513x161, NVSmall / ResNet-18 3D 1025x321) on one MI355X:
ms per pair through IExecutionContext::enqueue with the inputs resident in HBM, plus the per-launch profile."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi, synth  # noqa: E402

CASES = {"nvtiny": (513, 161, synth.NVTINY_3D), "nvsmall": (1025, 321, synth.NVSMALL_3D), "resnet18": (1025, 321, synth.RESNET18_3D)}


def main():
    lib = capi.NetLib()
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    half2 = "--half2" in sys.argv            # fp16 weight file -> half2 mode: the 3-D tensors are stored / multiplied as fp16
    batch = max([int(a[8:]) for a in sys.argv if a.startswith("--batch=")] + [1])
    for model in (args or ["nvtiny", "nvsmall"]):
        w, h, cfg = CASES[model]
        net = lib.create(model, w, h, max_batch=batch, weights=synth.synth_weights_3d(cfg), fp16_weights=half2)
        l, r = synth.synth_pair(h, w, 1234)
        L, R = torch.from_numpy(np.stack([l] * batch)).cuda(), torch.from_numpy(np.stack([r] * batch)).cuda()
        out = torch.empty(batch, 1, h, w, device="cuda")
        for _ in range(3):
            net.execute(L, R, out, batch)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            net.execute(L, R, out, batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3 / batch
        prof = net.profile(L, R, out, batch)
        tot = sum(t for _, t in prof)
        print("%-9s %dx%d%s batch %d: %8.2f ms/pair = %.1f pairs/s  (%d launches, sum of launch times %.2f ms per batch)  finite=%s" % (
            model, w, h, " half2" if half2 else "", batch, ms, 1e3 / ms, len(prof), tot, bool(torch.isfinite(out).all())))
        for name, t in sorted(prof, key=lambda p: -p[1])[:64]:
            print("      %-28s %8.3f ms" % (name, t))
        net.destroy()


if __name__ == "__main__":
    main()
