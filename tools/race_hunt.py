#!/usr/bin/env python3
"""Run-to-run determinism of the whole network on the GPU: the same pair through the same engine N times (and through
several engines on several streams at once, as bench.py does), every output compared bit for bit with the first and
with the oracle.  Kernel / layout / schedule knobs are environment variables of the native library (read at plan
creation), so each configuration runs in its own process:

    python tools/race_hunt.py                      # all configurations
    python tools/race_hunt.py one <runs> <ctx>     # this process, current environment
"""
import json
import os
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = [
    ("default", {}),
    ("single stream", {"RT_SINGLE_STREAM": "1"}),
    ("persistent split kernel for the 32->32 layers", {"RT_S3P": "1"}),
    ("no split kernels (round-1 fp32 kernels)", {"RT_CONV_EXACT_FP32": "1"}),
    ("no split kernels, single stream", {"RT_CONV_EXACT_FP32": "1", "RT_SINGLE_STREAM": "1"}),
    ("planar tensors", {"RT_NO_IL8": "1"}),
    ("fused residual blocks", {"RT_RB": "1"}),
]


def one(runs, nctx):
    import torch
    from oracle import stereo_oracle as O
    from redtail_amd import capi, model_files, synth
    W, H = 1257, 369
    lib = capi.NetLib()
    path = model_files.weight_file("resnet18_2D")
    weights = capi.read_weights(path)
    l, r = synth.synth_pair(H, W, 1234)
    L, R = torch.from_numpy(l)[None].cuda(), torch.from_numpy(r)[None].cuda()
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], weights)
    nets = [lib.create("resnet18_2D", W, H, weights_path=path) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in nets]
    outs = [torch.full((1, 1, H, W), float("nan"), device="cuda") for _ in nets]
    first, bad, worst, where = None, 0, 0.0, None
    for it in range(runs):
        for c, net in enumerate(nets):
            net.execute(L, R, outs[c], 1, stream=streams[c].cuda_stream)
        if it % 4 == 3 or it == runs - 1:          # several executes in flight per context between checks
            torch.cuda.synchronize()
            for c in range(nctx):
                o = outs[c].cpu()
                if first is None:
                    first = o.clone()
                if not torch.equal(o, first):
                    bad += 1
                    d = (o - first).abs()
                    if float(d.max()) >= worst:
                        worst = float(d.max())
                        ys, xs = torch.nonzero(d[0, 0] > 0, as_tuple=True)
                        where = [int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()), int(len(ys))]
    err = float((first - ref).abs().max())
    print(json.dumps({"runs": runs, "contexts": nctx, "mismatching_outputs": bad, "max_diff_between_runs": worst,
                      "bbox_y0_y1_x0_x1_count": where, "max_abs_err_vs_oracle": err}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    for name, env in CONFIGS:
        for nctx in (1, 4):
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", "40", str(nctx)], env=e, capture_output=True, text=True)
            line = [x for x in out.stdout.splitlines() if x.startswith("{")]
            print("%-48s ctx %d  %s" % (name, nctx, line[-1] if line else "FAILED: " + out.stderr[-400:]), flush=True)
