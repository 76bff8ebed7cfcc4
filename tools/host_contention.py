#!/usr/bin/env python3
"""Host-side headroom of the multi-GPU set-up (one process per GPU, each issuing ~23 launches per pair through Python + ctypes): N
processes side by side -- here all on the ONE GPU of the box, so the GPU is N-fold oversubscribed and only the HOST figures mean
anything -- each measuring the host time of IExecutionContext::enqueue on its own stream.  At 2 600 pairs/s a rank has 380 us per step.
    python tools/host_contention.py [processes = 8] [steps = 300]"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(idx, steps, q, go):
    import numpy as np
    import torch
    from redtail_amd import capi, synth
    lib = capi.NetLib()
    net = lib.create("resnet18_2D", 1257, 369, max_batch=1, weights=synth.synth_weights_resnet18_2d())
    net.set_streams(1)
    l, r = synth.synth_pair(369, 1257, 1234 + idx)
    L, R = torch.from_numpy(l[None]).cuda(), torch.from_numpy(r[None]).cuda()
    out = torch.empty(1, 1, 369, 1257, device="cuda")
    st = torch.cuda.Stream()
    for _ in range(5):
        net.execute(L, R, out, 1, stream=st.cuda_stream)
    torch.cuda.synchronize()
    go.wait()
    ts = []
    t_all = time.perf_counter()
    for i in range(steps):
        t0 = time.perf_counter()
        net.execute(L, R, out, 1, stream=st.cuda_stream)
        ts.append(time.perf_counter() - t0)
        if i % 20 == 19:
            torch.cuda.synchronize()          # keep the queue short: the figure wanted is the cost of issuing, not of a full queue
    torch.cuda.synchronize()
    q.put((idx, float(np.median(ts)) * 1e6, float(np.percentile(ts, 95)) * 1e6, (time.perf_counter() - t_all) / steps * 1e6))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    mp.set_start_method("spawn")
    for procs in (1, n):
        q, go = mp.Queue(), mp.Barrier(procs)
        ps = [mp.Process(target=worker, args=(i, steps, q, go)) for i in range(procs)]
        for p in ps:
            p.start()
        rows = sorted(q.get() for _ in ps)
        for p in ps:
            p.join()
        print("%d process(es) on %d host cores: host time per enqueue (23 launches) median %s us, 95th percentile %s us; wall per step %s us (one shared GPU)" % (
            procs, len(os.sched_getaffinity(0)), " ".join("%.0f" % r[1] for r in rows), " ".join("%.0f" % r[2] for r in rows), " ".join("%.0f" % r[3] for r in rows)))


if __name__ == "__main__":
    main()
