#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv of the bench: how the solver kernels behave while a front-end kernel is running.
Per solver kernel name: dispatches, mean duration and mean gap to the previous solver kernel, split by whether a front-end
kernel overlapped the dispatch.  usage: trace_overlap.py <dir-or-csv>"""
import bisect, csv, glob, sys
from collections import defaultdict
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0]
SOLVER = ("k_ba_", "k_chol_")
fe, ba = [], []
for r in rows:
    n = short(r["Kernel_Name"]); s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    (ba if n.startswith(SOLVER) else fe).append((s, e, n))
fe.sort(); ba.sort()
fs = [x[0] for x in fe]
# running maximum of front-end end times so that "some front-end kernel covers t" is a bisect
run_end = []; m = 0
for s, e, _ in fe:
    m = max(m, e); run_end.append(m)
def covered(t):
    i = bisect.bisect_right(fs, t) - 1
    return i >= 0 and run_end[i] > t
acc = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
tot = {True: [0, 0.0, 0.0], False: [0, 0.0, 0.0]}
for s, e, n in ba:
    ov = covered(s) or covered(e)
    gap = (s - prev_end) if prev_end is not None and s - prev_end < 400000 else 0.0      # longer: the host between two solves
    prev_end = e
    for a in (acc[(n, ov)], tot[ov]):
        a[0] += 1; a[1] += e - s; a[2] += max(gap, 0.0)
for (n, ov) in sorted(acc):
    c, d, g = acc[(n, ov)]
    print("%-18s %-9s n %6d  dur %7.2f us  gap %6.2f us" % (n, "beside-FE" if ov else "alone", c, d / c / 1e3, g / c / 1e3))
for ov in (False, True):
    c, d, g = tot[ov]
    if c: print("ALL %-9s n %6d  dur %7.2f us  gap %6.2f us" % ("beside-FE" if ov else "alone", c, d / c / 1e3, g / c / 1e3))

# which front-end kernel a slowed k_chol_pair ran beside
by = defaultdict(lambda: [0, 0.0])
for s_, e_, n in ba:
    if n != "k_chol_pair": continue
    names = sorted({fn for fs_, fe_, fn in fe if fs_ < e_ and fe_ > s_})
    key = "+".join(names) if names else "(alone)"
    by[key][0] += 1; by[key][1] += e_ - s_
for k in sorted(by, key=lambda k: -by[k][0]):
    print("k_chol_pair beside %-60s n %5d  dur %7.2f us" % (k, by[k][0], by[k][1] / by[k][0] / 1e3))
