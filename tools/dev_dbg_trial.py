import sys, os
sys.path.insert(0, os.getcwd())
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
p = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0)
b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hip.ba_obs_array(p), p["cam"])
b.optimize(True, 3); b.reset()
k, iters, dim = b.optimize_profiled(True, 10)
ms, marks, per = k["k_ba_trial"]
print("dbg %s: trial %.2f us per launch (%d launches)" % (os.environ.get("LPSLAM_DBG_SPEC", "0"), 1e3 * ms / marks, marks))
