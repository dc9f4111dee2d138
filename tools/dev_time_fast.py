import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=12)
seq = synth.StereoSequence(1280, 720, 0)
for i in range(6):
    l, r = seq.frame(i); ctx.upload(2 * i, l); ctx.upload(2 * i + 1, r)
ctx.stage("pyramid", 12); ctx.sync()
for dbg in (0,):
    os.environ["LPSLAM_FAST_DBG"] = str(dbg)
    ts = []
    for it in range(6):
        ctx.timer_begin(0); ctx.stage("fast", 12); ctx.timer_end(0); ctx.sync(); ts.append(ctx.timer_ms(0))
    print("dbg %2d: %.1f us" % (dbg, 1e3 * min(ts)))
