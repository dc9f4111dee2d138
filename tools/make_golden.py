#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (G1-G5).

The reference ships no fixture for this path (parity unpinned), so these vectors are produced by the oracle itself
and pin it against regressions; the GPU tests must reproduce G1-G4 bit-exactly and G5 within 1e-4 rad / 1e-3 m.
Usage: python tools/make_golden.py      (deterministic: seeded inputs, single-threaded C oracle)
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from lpslam_amd import synth            # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    # G1: 96x96 image -> 3 pyramid levels
    img = synth.random_image(96, 96, seed=11)
    p = O.params(60, 1.2, 3)
    lw, lh = O.pyramid_sizes(96, 96, p)
    lv = [img]
    for l in range(1, 3):
        lv.append(O.resize(lv[-1], lw[l], lh[l]))
    np.savez_compressed(os.path.join(OUT, "g1_pyramid.npz"), image=img, level1=lv[1], level2=lv[2])
    # G2: FAST corners (x, y, score) of one level, cells + thresholds 20/7
    img2 = synth.random_image(160, 120, seed=12)
    c = O.fast_level(img2, 20, 7)
    np.savez_compressed(os.path.join(OUT, "g2_fast.npz"), image=img2, x=c["x"], y=c["y"], score=c["score"])
    # G3: full extraction 160x120, 3 levels, 150 keypoints
    p3 = O.params(150, 1.2, 3)
    kp, desc, cc, _ = O.extract(img2, p3)
    np.savez_compressed(os.path.join(OUT, "g3_orb.npz"), image=img2, x=kp["x"], y=kp["y"], size=kp["size"], angle=kp["angle"],
                        response=kp["response"], octave=kp["octave"], desc=desc, cand_count=cc)
    # G4: 64 x 64 descriptor sets (with duplicates and near-duplicates to exercise the tie rules)
    rng = np.random.Generator(np.random.PCG64(44))
    t = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    q = t[rng.permutation(64)].copy()
    flip = rng.integers(0, 256, (64, 32), dtype=np.uint8) & rng.integers(0, 256, (64, 32), dtype=np.uint8) & rng.integers(0, 256, (64, 32), dtype=np.uint8)
    q ^= flip
    q[5] = q[6]; t[10] = t[11]; q[20] = t[10]
    bi, bd, sd = O.match_bf_knn2(q, t)
    mq, mt, md = O.match_bf(q, t, 50, 0.9, True)
    np.savez_compressed(os.path.join(OUT, "g4_bf.npz"), q=q, t=t, best_idx=bi, best_dist=bd, second_dist=sd, mq=mq, mt=mt, md=md)
    # G5: BA toy problem 4 KF / 60 points
    prob = synth.ba_problem(4, 60, 200, 640, 480, seq_id=5)
    obs = O.ba_obs(prob)
    poses, points, log = O.ba_optimize(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], True, 10)
    cam = np.array([prob["cam"][k] for k in ("fx", "fy", "cx", "cy", "fxb")])
    np.savez_compressed(os.path.join(OUT, "g5_ba.npz"), poses0=prob["poses"], points0=prob["points"], fixed=prob["fixed"],
                        obs_pose=prob["obs_pose"], obs_point=prob["obs_point"], obs_uvr=prob["obs_uvr"],
                        obs_inv_sigma2=prob["obs_inv_sigma2"], cam=cam, poses=poses, points=points,
                        chi2_before=log["chi2_before"], chi2_after=log["chi2_after"], lam=log["lambda"], trials=log["trials"])
    # G6: stereo match of a small pair
    seq = synth.StereoSequence(320, 240, 3, n_points=1500)
    l, r = seq.frame(0)
    p6 = O.params(400, 1.2, 4)
    kl, dl, _, pl = O.extract(l, p6, True)
    kr, dr, _, pr = O.extract(r, p6, True)
    k = synth.intrinsics(320, 240)
    xr, dep, bi, nv = O.match_stereo(pl, pr, p6, kl, dl, kr, dr, k["fxb"], k["baseline"])
    np.savez_compressed(os.path.join(OUT, "g6_stereo.npz"), left=l, right=r, x_right=xr, depth=dep, best_idx=bi, n_left=len(kl), n_right=len(kr))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
