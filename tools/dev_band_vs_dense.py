import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpslam_amd import hip, synth
from oracle import oracle as O
ctx = hip.Context(320, 240, 400, 1.2, 4, max_images=1)
for seed, kf in ((42, 8), (41, 8), (45, 10), (56, 10), (52, 8)):
    prob = synth.ba_problem(kf, 200, 1000, 640, 480, seq_id=seed, pose_noise=(0.5, 3.0), point_noise=3.0, tracks="contiguous")
    op, ox, olog = O.ba_optimize(prob["poses"], prob["fixed"], prob["points"], O.ba_obs(prob), prob["cam"], True, 10)
    for name in ("band", "dense"):
        ba = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
        ba.set_solver(name)
        glog = ba.optimize(True, 10)
        n = min(len(glog), len(olog))
        rel = np.abs(glog["chi2_after"][:n] - olog["chi2_after"][:n]) / olog["chi2_after"][:n]
        print(seed, kf, name, "trials", glog["trials"], "o", olog["trials"], "rel", " ".join("%.1e" % r for r in rel))
        ba.close()
