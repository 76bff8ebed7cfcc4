#!/usr/bin/env python3
"""How long does a HIP stream that waits for an event of another stream take to resume once the event completes?

    rocprofv3 --kernel-trace --output-format csv -d out -o x -- python tools/micro/xstream_probe.py
    python tools/micro/xstream_probe.py --parse out/x_kernel_trace.csv

Stream A runs a spin kernel of a given length and records an event; stream B (idle until then, its wait queued long before)
waits for it and runs a tiny kernel (fork lag = its start - the spin kernel's end); A then waits for B (join lag).  The
executor's two-stream schedule (engine.cpp: ev_fork_ / ev_join_) pays the first once per step on the side stream.
"""
import csv
import sys


def run():
    import torch
    a = torch.zeros(256, device="cuda")
    b = torch.zeros(256, device="cuda", dtype=torch.int32)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for cycles in (20000, 50000, 100000, 200000, 400000, 800000, 1600000):
        for _ in range(12):
            ev, ev2 = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.stream(sa):
                torch.cuda._sleep(cycles)
                ev.record(sa)
            sb.wait_event(ev)
            with torch.cuda.stream(sb):
                b.add_(1)
                ev2.record(sb)
            sa.wait_event(ev2)
            with torch.cuda.stream(sa):
                a.add_(1.0)
        torch.cuda.synchronize()


def parse(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    spins = [r for r in rows if "sleep" in r["Kernel_Name"].lower() or "spin" in r["Kernel_Name"].lower()]
    adds = [r for r in rows if "elementwise" in r["Kernel_Name"]]
    qa = spins[0]["Queue_Id"]
    addb = [r for r in adds if r["Queue_Id"] != qa]
    adda = [r for r in adds if r["Queue_Id"] == qa]
    print("spin_us  fork_lag_us  join_lag_us")
    for s, fb, ja in zip(spins, addb, adda):
        d = (int(s["End_Timestamp"]) - int(s["Start_Timestamp"])) / 1e3
        print("%8.1f %10.1f %10.1f" % (d, (int(fb["Start_Timestamp"]) - int(s["End_Timestamp"])) / 1e3,
                                       (int(ja["Start_Timestamp"]) - int(fb["End_Timestamp"])) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
