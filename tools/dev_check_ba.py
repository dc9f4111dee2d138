"""Developer diagnostic: HIP bundle adjustment vs oracle (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from lpslam_amd import hip, synth

def rot_err(q1, q2):
    d = np.abs(np.sum(q1 * q2, axis=1)).clip(0, 1)
    return 2 * np.arccos(d)

def main(n_kf=8, n_pts=200, n_obs=1200, iters=10):
    prob = synth.ba_problem(n_kf, n_pts, n_obs, 640, 480)
    obs = O.ba_obs(prob)
    print("problem", n_kf, n_pts, len(obs))
    t = time.time(); op, ox, olog = O.ba_optimize(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], True, iters); to = time.time() - t
    ctx = hip.Context(640, 480, 500, 1.2, 4, max_images=1)
    ba = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
    t = time.time(); glog = ba.optimize(True, iters); tg = time.time() - t
    gp, gx = ba.state()
    print("oracle time %.3f s, gpu time %.4f s" % (to, tg))
    for a, b in zip(olog, glog):
        print("  chi2 %.6f -> %.6f | %.6f -> %.6f  lam %.3e %.3e trials %d %d" % (a["chi2_before"], a["chi2_after"], b["chi2_before"], b["chi2_after"], a["lambda"], b["lambda"], a["trials"], b["trials"]))
    print("max rot diff", rot_err(op[:, :4], gp[:, :4]).max(), "max t diff", np.abs(op[:, 4:] - gp[:, 4:]).max(), "max pt diff", np.abs(ox - gx).max())
    print("vs gt: rot", rot_err(gp[:, :4], prob["poses_gt"][:, :4]).max(), "t", np.abs(gp[:, 4:] - prob["poses_gt"][:, 4:]).max())
    # second call timing (warm)
    ba2 = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
    t = time.time(); ba2.optimize(True, iters); print("warm gpu time %.4f s" % (time.time() - t))
    # local BA flow
    op2, ox2, oout = O.ba_local(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"])
    ba3 = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
    gout = ba3.local(5, 10); gp3, gx3 = ba3.state()
    print("local: outliers", oout.sum(), gout.sum(), "equal", np.array_equal(oout, gout), "rot", rot_err(op2[:, :4], gp3[:, :4]).max(), "t", np.abs(op2[:, 4:] - gp3[:, 4:]).max())

if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:]])
