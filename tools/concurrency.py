#!/usr/bin/env python3
"""How many kernels run at the same time in a rocprofv3 kernel trace (bench.py in throughput mode): share of wall time with 0, 1, 2, ... kernels
in flight over the middle of the run, and per kernel the share of its duration it ran alone.
    python tools/concurrency.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rt::", "").replace("rt::", "")[:48]))
rows = [r for r in rows if not r[2].startswith("at::") and "rocclr" not in r[2]]
rows.sort()
t0, t1 = rows[len(rows) // 4][0], rows[3 * len(rows) // 4][0]          # the middle half
ev = []
for s, e, n in rows:
    s, e = max(s, t0), min(e, t1)
    if e > s:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = collections.Counter()
cur, last = 0, t0
for t, d in ev:
    hist[cur] += t - last
    last = t
    cur += d
hist[cur] += t1 - last
tot = float(t1 - t0)
print("kernels in flight over %.1f ms: " % (tot / 1e6) + "  ".join("%d: %.1f %%" % (k, 100 * v / tot) for k, v in sorted(hist.items())))
print("mean concurrency %.2f" % (sum(k * v for k, v in hist.items()) / tot))
dur = collections.defaultdict(float); cnt = collections.Counter()
for s, e, n in rows:
    if s >= t0 and e <= t1:
        dur[n] += e - s; cnt[n] += 1
for n, d in sorted(dur.items(), key=lambda x: -x[1]):
    print("%-50s %6d launches  avg %7.1f us  %5.1f %% of the summed durations" % (n, cnt[n], d / cnt[n] / 1e3, 100 * d / sum(dur.values())))
