#!/usr/bin/env python3
"""One-off randomised BA parity sweep on a GPU box: random window sizes, observation counts, mono share, inactive share, noise
levels (rejected trials), robust on / off; chi2 trajectory, lambda control and final state against the CPU oracle
(tolerances of tests/test_ba_gpu.py).  usage: fuzz_ba.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lpslam_amd import hip, synth                                  # noqa: E402
from oracle import oracle as O                                      # noqa: E402

ROT_TOL, TRANS_TOL, CHI_RTOL = 1e-4, 1e-3, 1e-9
def rot_err(q1, q2): return 2 * np.arccos(np.clip(np.abs(np.sum(q1 * q2, axis=1)), 0, 1))

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip.load()
ctx = hip.Context(320, 240, 400, 1.2, 4, max_images=1)
bad = 0; t0 = time.time()
for case in range(n_cases):
    n_kf = int(rng.integers(2, 52)); n_pts = int(rng.integers(20, 1500))      # from 31 free keyframes on the band factorisation runs from both ends
    n_obs = int(min(n_kf * n_pts, rng.integers(2 * n_pts, 8 * n_pts + 1)))
    robust = bool(rng.integers(0, 2)); iters = int(rng.integers(1, 12))
    noise = float(rng.choice([1.0, 1.0, 3.0, 8.0]))
    tracks = "contiguous" if rng.integers(0, 3) else "random"        # two thirds as a tracker makes them: the band path (ba_band.inl)
    prob = synth.ba_problem(n_kf, n_pts, n_obs, 640, 480, seq_id=int(rng.integers(1000)), pose_noise=(0.01 * noise, 0.05 * noise), point_noise=0.05 * noise,
                            tracks=tracks, top_up=bool(rng.integers(0, 2)))
    m = len(prob["obs_pose"])
    if rng.integers(0, 4) == 0:                                       # landmarks seen twice by a keyframe
        dup = rng.choice(m, max(1, m // 50), replace=False)
        for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma2"): prob[key] = np.concatenate([prob[key], prob[key][dup]])
        prob["obs_uvr"][m:, :2] += rng.normal(0, 0.3, (len(dup), 2)); m = len(prob["obs_pose"])
    if rng.integers(0, 3) == 0:                                       # caller order shuffled
        perm = rng.permutation(m)
        for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma2"): prob[key] = prob[key][perm]
    if rng.integers(0, 2): prob["obs_uvr"][rng.random(m) < 0.3, 2] = -1.0
    active = None
    if rng.integers(0, 2): active = (rng.random(m) > 0.15).astype(np.uint8)
    fx = int(rng.integers(0, 4))
    if fx == 0: prob["fixed"][rng.random(n_kf) < 0.3] = 1
    elif fx == 1: prob["fixed"][:int(rng.integers(1, max(2, n_kf // 2)))] = 1      # a tracker's window: the oldest observers are fixed
    elif fx == 2: prob["fixed"][:] = 0; prob["fixed"][int(rng.integers(0, n_kf))] = 1
    tag = "case %d: %s %d KF (%d free) %d pts %d obs robust %d iters %d noise %.0f" % (case, tracks, n_kf, int((prob["fixed"] == 0).sum()), n_pts, m, robust, iters, noise)
    try:
        obs = O.ba_obs(prob)
        op, ox, olog = O.ba_optimize(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], robust, iters, active)
        ba = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
        if active is not None: ba.set_active(active)
        glog = ba.optimize(robust, iters); gp, gx = ba.state()
        tag += " [%s %d]" % ba.solver()
        checks = {"len": len(glog) == len(olog)}
        if checks["len"]:
            checks.update(chi_before=np.allclose(glog["chi2_before"], olog["chi2_before"], rtol=CHI_RTOL),
                          chi_after=np.allclose(glog["chi2_after"], olog["chi2_after"], rtol=CHI_RTOL),
                          trials=np.array_equal(glog["trials"], olog["trials"]), status=np.array_equal(glog["status"], olog["status"]),
                          lam=np.allclose(glog["lambda"], olog["lambda"], rtol=1e-6),
                          rot=rot_err(gp[:, :4], op[:, :4]).max() < ROT_TOL, trans=np.abs(gp[:, 4:] - op[:, 4:]).max() < TRANS_TOL,
                          pts=np.abs(gx - ox).max() < TRANS_TOL)
        ok = all(checks.values())
        if not ok: tag += "  failed: " + ",".join(k for k, v in checks.items() if not v) + "  chi0 gpu %r oracle %r" % (float(glog["chi2_before"][0]) if len(glog) else None, float(olog["chi2_before"][0]) if len(olog) else None)
        print(("ok   " if ok else "FAIL ") + tag + "  trials %s" % (list(glog["trials"]),), flush=True)
        bad += not ok
        ba.close() if hasattr(ba, "close") else None
    except Exception as e:                                           # noqa: BLE001
        print("ERR  " + tag + ": " + repr(e), flush=True); bad += 1
print("%d cases, %d bad, %.1f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
