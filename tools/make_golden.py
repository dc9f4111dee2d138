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
    # G7: Sim3 pose graph, 12 keyframes, loop closed (10 Levenberg iterations, scale fixed) + one free-scale variant
    pg = synth.pose_graph_problem(12, 7, n_loop=2)
    e7 = O.sim3_edges(pg["edge_i"], pg["edge_j"], pg["meas"])
    v7, log7 = O.sim3_graph_optimize(pg["verts"], pg["fixed"], e7, True, 10)
    pg2 = synth.pose_graph_problem(12, 8, drift_scale=0.01, n_loop=2)
    e8 = O.sim3_edges(pg2["edge_i"], pg2["edge_j"], pg2["meas"])
    v8, log8 = O.sim3_graph_optimize(pg2["verts"], pg2["fixed"], e8, False, 10)
    np.savez_compressed(os.path.join(OUT, "g7_sim3.npz"), verts0=pg["verts"], fixed=pg["fixed"], edge_i=pg["edge_i"], edge_j=pg["edge_j"],
                        meas=pg["meas"], verts=v7, chi2_after=log7["chi2_after"], lam=log7["lambda"], trials=log7["trials"],
                        f_verts0=pg2["verts"], f_edge_i=pg2["edge_i"], f_edge_j=pg2["edge_j"], f_meas=pg2["meas"], f_verts=v8,
                        f_chi2_after=log8["chi2_after"],
                        exp_in=np.array([0.3, -0.2, 0.1, 1.0, -2.0, 0.5, 0.2]), exp_out=O.sim3_exp([0.3, -0.2, 0.1, 1.0, -2.0, 0.5, 0.2]))
    # G8: the benchmark's brute-force shape, 2000 x 2000 (the train set spans two 1024-descriptor tiles of the HIP kernel):
    # train = random descriptors, query = a permutation of them with a few flipped bits, plus planted exact ties
    rng = np.random.Generator(np.random.PCG64(88))
    t8 = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    q8 = t8[rng.permutation(2000)].copy()
    q8 ^= (rng.integers(0, 256, (2000, 32), dtype=np.uint8) & rng.integers(0, 256, (2000, 32), dtype=np.uint8) &
           rng.integers(0, 256, (2000, 32), dtype=np.uint8) & rng.integers(0, 256, (2000, 32), dtype=np.uint8))
    t8[1030] = t8[5]; q8[0] = t8[5]; t8[1024] = t8[1023]; q8[1] = t8[1023]; t8[1999] = t8[0]; q8[3] = t8[0]; q8[1999] = q8[17]
    bi, bd, sd = O.match_bf_knn2(q8, t8)
    mq, mt, md = O.match_bf(q8, t8, 100, 0.9, True)
    np.savez_compressed(os.path.join(OUT, "g8_bf2000.npz"), q=q8, t=t8, best_idx=bi, best_dist=bd, second_dist=sd, mq=mq, mt=mt, md=md)
    # G9: a 1280x720 stereo pair at the benchmark's configuration (2000 keypoints, 8 levels).  The images come from the committed
    # generator (lpslam_amd/synth.py, sequence 9, frame 2) and are pinned by their SHA-256; the fixture holds the oracle's
    # keypoints, descriptors, x_right and depth.
    import hashlib
    seq9 = synth.StereoSequence(1280, 720, 9)
    l9, r9 = seq9.frame(2)
    p9 = O.params(2000, 1.2, 8)
    kl9, dl9, _, pl9 = O.extract(l9, p9, True)
    kr9, dr9, _, pr9 = O.extract(r9, p9, True)
    k9 = synth.intrinsics(1280, 720)
    xr9, dep9, bi9, nv9 = O.match_stereo(pl9, pr9, p9, kl9, dl9, kr9, dr9, k9["fxb"], k9["baseline"])
    np.savez_compressed(os.path.join(OUT, "g9_stereo720.npz"), sha_left=hashlib.sha256(l9.tobytes()).hexdigest(),
                        sha_right=hashlib.sha256(r9.tobytes()).hexdigest(), kl=kl9, dl=dl9, kr=kr9, dr=dr9, x_right=xr9, depth=dep9, best_idx=bi9, n_valid=nv9)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
