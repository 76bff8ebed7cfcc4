#!/usr/bin/env python3
"""The reference's loop (execute, wait, next) under rocprofv3 --kernel-trace: `python tools/sync_trace.py run` is the traced program,
`python tools/sync_trace.py show <dir>` prints the last step's launches (start, duration, gap to the previous end on any queue)."""
import csv
import glob
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import numpy as np
    import torch
    from redtail_amd import capi, model_files, synth
    lib = capi.NetLib()
    net = lib.create("resnet18_2D", 1257, 369, max_batch=1, weights_path=model_files.weight_file("resnet18_2D"))
    l, r = synth.synth_pair(369, 1257)
    L, R = torch.from_numpy(l[None]).cuda(), torch.from_numpy(r[None]).cuda()
    out = torch.empty(1, 1, 369, 1257, device="cuda")
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
        net.execute(L, R, out, 1)
        torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("rt::", "void rt::"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    last = rows[-n:]
    t0, end = int(last[0]["Start_Timestamp"]), 0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = "%2s,%2s,%2s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
        print("%8.1f %7.1f  gap %6.1f  q%-2s wg %-10s x%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3 if end else 0.0, r["Queue_Id"], g,
                                                                   r["Workgroup_Size_X"], r["Kernel_Name"][:60]))
        end = max(end, e)
    print("span %.1f us, sum of durations %.1f us" % ((end - t0) / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e3))
