"""development: wall time per LM iteration of banded windows of several sizes (k_chol_band: twisted from BC_TWIST_MIN strips on)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
for kf in (12, 20, 26, 30, 34, 40, 50):
    p = synth.ba_problem(kf, 100 * kf, 800 * kf, 1280, 720, seq_id=3, tracks="contiguous", top_up=True)
    b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hip.ba_obs_array(p), p["cam"])
    for _ in range(3):
        b.reset(); b.optimize(True, 10)
    ts = []
    for _ in range(8):
        b.reset(); b.state()
        t0 = time.perf_counter(); b.optimize(True, 10); ts.append(time.perf_counter() - t0)
    k, iters, dim = b.optimize_profiled(True, 10)
    print("%2d keyframes (dim %3d, %2d strips) %s: %.1f us per iteration (wall), factor %.1f us (event)" % (kf, dim, (dim + 15) // 16, b.solver(), 1e5 * np.median(ts), 1e3 * k["chol"][0] / max(iters, 1)))
    b.close()
