"""development: where a keyframe's 1.22 ms go in bench.py's pipelined mapping loop (host-side clocks around each call)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
probs = [synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=s, tracks="random", top_up=True) for s in range(4)]
obs = [hip.ba_obs_array(p) for p in probs]
def new(v, build=True):
    p = probs[v % 4]
    return hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], obs[v % 4], p["cam"], build=build)
T = {k: 0.0 for k in ("new", "end", "set_state", "begin", "state", "close")}
def tic(): return time.perf_counter()
n_kf = 60
cur = new(0); cur.set_state(probs[0]["poses"], probs[0]["points"]); cur.optimize_begin(True, 10)
t_all = tic()
for i in range(n_kf):
    t = tic(); nxt = new(i + 1) if i + 1 < n_kf else None; T["new"] += tic() - t
    t = tic(); cur.optimize_end(); T["end"] += tic() - t
    if nxt is not None:
        p = probs[(i + 1) % 4]
        t = tic(); nxt.set_state(p["poses"], p["points"]); T["set_state"] += tic() - t
        t = tic(); nxt.optimize_begin(True, 10); T["begin"] += tic() - t
    t = tic(); cur.state(); T["state"] += tic() - t
    t = tic(); cur.close(); T["close"] += tic() - t
    cur = nxt
t_all = tic() - t_all
print("per keyframe %.3f ms: " % (1e3 * t_all / n_kf) + ", ".join("%s %.3f" % (k, 1e3 * v / n_kf) for k, v in T.items()))
