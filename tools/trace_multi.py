#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv of tools/dev_tracker_multi.py: per kernel name the mean duration, and over the whole trace how many
kernels run at a time, per hardware queue how busy it is, and how long a kernel waited behind the previous kernel of ITS stream's queue.
usage: trace_multi.py <dir-or-csv> [t0_fraction t1_fraction]   (the window of the trace to look at, default 0.5 1.0: the last run)"""
import csv, glob, sys
from collections import defaultdict
path = sys.argv[1]
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
t_lo, t_hi = ev[0][0], max(e[1] for e in ev)
a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
ev = [e for e in ev if e[0] >= a and e[1] <= b]
span = (max(e[1] for e in ev) - min(e[0] for e in ev)) / 1e3
print("%d kernels in %.1f us; sum of durations %.1f us -> %.2f kernels in flight on average" % (len(ev), span, sum(e[1] - e[0] for e in ev) / 1e3, sum(e[1] - e[0] for e in ev) / 1e3 / span))
per = defaultdict(lambda: [0, 0.0])
for s, e, n, q, st in ev:
    per[n][0] += 1; per[n][1] += e - s
for n in sorted(per, key=lambda n: -per[n][1])[:14]:
    print("  %-28s n %5d  mean %7.2f us  total %8.1f us" % (n[:28], per[n][0], per[n][1] / per[n][0] / 1e3, per[n][1] / 1e3))
qs = defaultdict(lambda: [0, 0.0, set()])
for s, e, n, q, st in ev:
    qs[q][0] += 1; qs[q][1] += e - s; qs[q][2].add(st)
for q in sorted(qs):
    print("  queue %-4s kernels %5d  busy %5.1f %%  streams %d" % (q, qs[q][0], 100 * qs[q][1] / 1e3 / span, len(qs[q][2])))
# concurrency histogram
pts = sorted([(s, 1) for s, e, *_ in ev] + [(e, -1) for s, e, *_ in ev])
hist = defaultdict(float); cur = 0; last = pts[0][0]
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("  in flight: " + ", ".join("%d: %.0f %%" % (k, 100 * v / tot) for k, v in sorted(hist.items()) if v / tot > 0.005))
