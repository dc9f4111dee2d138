"""development: where a keyframe's time goes in bench.py's pipelined mapping loop -- host time of every call of the loop, window alone on the
card (no front end beside it).  usage: dev_pipeline_times.py [random|contiguous] [keyframes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "random"
n_kf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
probs = [synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=s, tracks=kind, top_up=True) for s in range(4)]
obs = [hip.ba_obs_array(p) for p in probs]
def new(v):
    p = probs[v % 4]
    return hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], obs[v % 4], p["cam"])
T = {}
def tick(name, t0):
    t1 = time.perf_counter(); T.setdefault(name, []).append(t1 - t0); return t1
for rep in range(3):
    T.clear()
    v = 0
    cur = new(v); p = probs[0]; cur.set_state(p["poses"], p["points"]); cur.optimize_begin(True, 10)
    t_all = time.perf_counter()
    for i in range(n_kf):
        v += 1
        t = time.perf_counter()
        nxt = new(v); t = tick("new_problem", t)
        cur.optimize_end(); t = tick("optimize_end (wait)", t)
        p = probs[v % 4]
        nxt.set_state(p["poses"], p["points"]); t = tick("set_state", t)
        nxt.optimize_begin(True, 10); t = tick("optimize_begin", t)
        cur.state(); t = tick("state", t)
        cur.close(); t = tick("close", t)
        cur = nxt
    cur.optimize_end(); cur.close()
    tot = time.perf_counter() - t_all
    print("%s, %d keyframes: %.1f us per keyframe" % (kind, n_kf, 1e6 * tot / n_kf))
    for k, a in T.items():
        print("   %-22s median %7.1f us  mean %7.1f" % (k, 1e6 * np.median(a), 1e6 * np.mean(a)))
