#!/usr/bin/env python3
"""Locates a run-to-run deviation of the network inside the network: N engines run the SAME pair side by side (as tools/race_hunt.py
does) with the executor's launch trace on (IExecutionContext::setLaunchTrace: every launch's output is hashed on its own stream), and
every pass's hash vector is compared with the first.  The first launch whose hash differs names the kernel that computed something
else; its output tensor is then fetched from the deviating context and from a context that agrees with the reference, and the
differing elements are described (how many, where in the tensor, how large, what the wrong values look like).

    python tools/race_locate.py <passes> <contexts> [exact|split] [streams]
      RT_VARIANT_DIR=<dir>   another build of the two libraries (librt_stereo_hip.so + libnvstereo_inference.so side by side)
      RACE_NO_TRACE=1        trace off: only the disparity maps are compared (does the trace's own launches change the rate?)
      RACE_NEIGHBOURS=split  contexts 1.. run the DEFAULT (split-fp16) engine instead of the observed one: which co-runners does it take?
"""
import json
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# (C, H, pitch) of the interleaved fp32 tensors (C/4, H, pitch, 4) the exact engine's Winograd launches write at 1257 x 369
SHAPES = {"resblock": (32, 185, 640), "conv2D_1": (32, 185, 640), "conv2D_2": (32, 185, 640), "conv2D_4": (64, 93, 320), "conv2D_5": (64, 93, 320),
          "conv2D_7": (128, 47, 160), "conv2D_8": (128, 47, 160)}


def decode(name, word):
    """word index in a launch's output -> sample, channel, row, column, and the position inside the Winograd kernel's wave: output (a, b) of
    the lane's 2 x 2 tile, accumulator register i (channel % 4) and lane quarter k4 (conv_wino.hip.h: lane (k4, t) holds channels 16*cb + 4*k4 + i)"""
    for key, (C, H, P) in SHAPES.items():
        if key in name:
            per = (C // 4) * H * P * 4
            smp, w = divmod(word, per)
            slot, e = divmod(w, 4)
            g, r = divmod(slot, H * P)
            y, x = divmod(r, P)
            c = 4 * g + e
            return {"sample": smp, "channel": c, "y": y, "x": x, "a": y % 2, "b": x % 2, "i": e, "k4": (c % 16) // 4, "cb": (c % 32) // 16, "t": (x % 32) // 2, "tg": (y % 4) // 2}
    return None


def describe(net_bad, net_good, k, name):
    bad = net_bad.read_launch_output(k).view(np.float32)
    good = net_good.read_launch_output(k).view(np.float32)
    n = min(len(bad), len(good))
    bad, good = bad[:n], good[:n]
    neq = np.nonzero(bad.view(np.uint32) != good.view(np.uint32))[0]
    if len(neq) == 0:
        return {"differing_words": 0}
    d = np.abs(bad[neq].astype(np.float64) - good[neq].astype(np.float64))
    info = {"differing_words": int(len(neq)), "first_word": int(neq[0]), "last_word": int(neq[-1]),
            "max_abs_diff": float(np.nanmax(d)) if np.isfinite(d).any() else None,
            "nan_in_bad": int(np.isnan(bad[neq]).sum()), "zero_in_bad": int((bad[neq] == 0).sum()),
            "sample_bad": [float(x) for x in bad[neq[:6]]], "sample_good": [float(x) for x in good[neq[:6]]]}
    # contiguous runs of differing words (a tile of an interleaved tensor is runs of 4 * 32 words per row)
    runs = np.split(neq, np.nonzero(np.diff(neq) != 1)[0] + 1)
    info["runs"] = len(runs)
    info["run_lengths"] = sorted({int(len(r)) for r in runs})[:12]
    info["run_starts"] = [int(r[0]) for r in runs[:12]]
    # where do the wrong bits come from?  every position of the GOOD tensor that holds exactly a wrong word's bit pattern
    gb = good.view(np.uint32)
    order = np.argsort(gb, kind="stable")
    sorted_bits = gb[order]
    src = []
    for wd in neq[:16]:
        bits = bad.view(np.uint32)[wd]
        lo, hi = np.searchsorted(sorted_bits, bits, "left"), np.searchsorted(sorted_bits, bits, "right")
        hits = [int(order[i]) for i in range(lo, min(hi, lo + 4))]
        src.append({"word": int(wd), "at": decode(name, int(wd)), "same_bits_in_good_tensor_at": [{"word": h, "delta_words": h - int(wd), "pos": decode(name, h)} for h in hits]})
    info["sources"] = src
    dec = [decode(name, int(w)) for w in neq[:64]]
    if dec and dec[0]:
        info["first"] = dec[0]
        for key in ("a", "b", "i", "k4", "cb", "tg", "y", "channel"):
            info["set_" + key] = sorted({d[key] for d in dec})
        info["t_values"] = sorted({d["t"] for d in dec})
    return info


def main():
    import torch
    from redtail_amd import capi, model_files, synth
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    mode = sys.argv[3] if len(sys.argv) > 3 else "exact"
    streams = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    W, H = 1257, 369
    vdir = os.environ.get("RT_VARIANT_DIR")
    lib = capi.NetLib(os.path.join(vdir, "libnvstereo_inference.so"), os.path.join(vdir, "librt_stereo_hip.so")) if vdir else capi.NetLib()
    path = model_files.weight_file("resnet18_2D")
    l, r = synth.synth_pair(H, W, 1234)
    L, R = torch.from_numpy(l)[None].cuda(), torch.from_numpy(r)[None].cuda()
    flags = capi.RT_CONV_EXACT_FP32 if mode == "exact" else 0
    nb = os.environ.get("RACE_NEIGHBOURS", "same")
    nets = [lib.create("resnet18_2D", W, H, weights_path=path, flags=flags if (c == 0 or nb == "same") else 0) for c in range(nctx)]
    trace = os.environ.get("RACE_NO_TRACE", "0") == "0"
    for n in nets:
        n.set_streams(streams)
        n.set_launch_trace(trace)
    tstreams = [torch.cuda.Stream() for _ in nets]
    outs = [torch.full((1, 1, H, W), float("nan"), device="cuda") for _ in nets]
    names = [nets[0].launch_name(i) for i in range(nets[0].num_launches)]
    ref = None
    stats = {"mode": mode, "contexts": nctx, "streams": streams, "passes": passes, "outputs": 0, "deviating_passes": 0, "first_deviating_launch": {}}
    shown = 0
    for it in range(passes):
        for c, net in enumerate(nets):
            net.execute(L, R, outs[c], 1, stream=tstreams[c].cuda_stream)
        if not trace:
            torch.cuda.synchronize()
            cur = [o.clone() for o in outs]
            if ref is None:
                ref = cur[0]
            for c in range(nctx):
                stats["outputs"] += 1
                stats["deviating_passes"] += int(not torch.equal(cur[c], ref))
            continue
        traces = [net.read_launch_trace() for net in nets]
        if ref is None:
            ref = traces[0]
            ref_nb = traces[1] if nctx > 1 else None
        if nb != "same":                        # only context 0 is the observed engine; the neighbours are compared among themselves
            stats["neighbour_deviations"] = stats.get("neighbour_deviations", 0) + sum(1 for c in range(1, nctx) if traces[c] != ref_nb)
        good = [c for c in range(nctx) if traces[c] == ref]
        for c in range(nctx if nb == "same" else 1):
            stats["outputs"] += 1
            if traces[c] == ref:
                continue
            stats["deviating_passes"] += 1
            k = next(i for i in range(len(ref)) if traces[c][i] != ref[i])
            later = [i for i in range(len(ref)) if traces[c][i] != ref[i]]
            key = "%d %s" % (k, names[k])
            stats["first_deviating_launch"][key] = stats["first_deviating_launch"].get(key, 0) + 1
            if shown < int(os.environ.get('RACE_SHOW', '12')) and good:
                shown += 1
                info = describe(nets[c], nets[good[0]], k, names[k])
                print(json.dumps({"pass": it, "context": c, "launch": k, "name": names[k], "launches_after_it_that_differ": len(later) - 1,
                                  "output": info}), flush=True)
    print(json.dumps(stats), flush=True)
    print("launches:", json.dumps(names))


if __name__ == "__main__":
    main()
