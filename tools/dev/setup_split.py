"""development: where a window's set-up goes -- host preparation (lpslam_hip_ba_prepare), the device build chain (lpslam_hip_ba_build_batch, until the stream is idle)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
for kind in ("random", "contiguous"):
    p = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0, tracks=kind, top_up=True) if kind == "contiguous" else synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0)
    obs = hip.ba_obs_array(p)
    tp, tb, tt = [], [], []
    for i in range(12):
        t0 = time.perf_counter()
        b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], obs, p["cam"], build=False)
        t1 = time.perf_counter()
        hip.ba_build_batch([b]); b.state()
        t2 = time.perf_counter()
        b.close()
        tp.append(1e3 * (t1 - t0)); tb.append(1e3 * (t2 - t1)); tt.append(1e3 * (t2 - t0))
    print("%-10s prepare %.3f ms, build + state read %.3f ms, total %.3f ms (medians of 11)" % (kind, np.median(tp[1:]), np.median(tb[1:]), np.median(tt[1:])))
