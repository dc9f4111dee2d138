import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=12)
seq = synth.StereoSequence(1280, 720, 0)
for i in range(6):
    l, r = seq.frame(i); ctx.upload(2 * i, l); ctx.upload(2 * i + 1, r)
for rep in range(2):
    ctx.stage("pyramid", 12); ctx.stage("fast", 12); ctx.sync()
    ctx.timer_begin(0); ctx.stage("distribute", 12); ctx.timer_end(0); ctx.sync()
print("kernel", ctx.timer_ms(0) * 1e3, "us")
for lvl in (0, 3, 7):
    buf = (C.c_ulonglong * 200)()
    ctx.lib.lpslam_hip_dev_stamps(ctx.h, 0, lvl, buf, 200)
    st = [buf[i] for i in range(200)]
    n = st.index(0) if 0 in st else 200
    st = st[:n]
    d = [(b - a) / 100.0 for a, b in zip(st, st[1:])]
    print("level", lvl, "n stamps", n, "total %.1f us" % ((st[-1] - st[0]) / 100.0))
    print("  ", " ".join("%.1f" % x for x in d))
