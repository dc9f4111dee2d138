#!/usr/bin/env python3
"""One-off randomised parity sweep on a GPU box (not part of the test suite): random image sizes, keypoint budgets, level
counts, scale factors and FAST thresholds; ORB extraction, brute-force and stereo matching against the CPU oracle, bit for bit.
usage: fuzz_parity.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpslam_amd import hip, synth                                  # noqa: E402
from oracle import oracle as O                                      # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip.load()
bad = 0
t0 = time.time()
for case in range(n_cases):
    w = int(rng.integers(96, 900)); h = int(rng.integers(96, 600))
    if case % 7 == 3: w, h = int(rng.integers(700, 1600)), int(rng.integers(96, 170))        # very wide: many quad-tree roots
    if case % 11 == 5: w, h = int(rng.integers(96, 170)), int(rng.integers(500, 1000))       # very tall
    levels = int(rng.integers(1, 9))
    scale = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
    while levels > 1 and min(w, h) / scale ** (levels - 1) < 64: levels -= 1
    kpts = int(rng.integers(20, 1500))
    ini = int(rng.integers(5, 40)); mn = int(rng.integers(2, ini + 1))
    kind = int(rng.integers(0, 3))
    if kind == 0: img = synth.random_image(w, h, seed=int(rng.integers(1 << 30)))
    elif kind == 1: img = rng.integers(0, 256, (h, w)).astype(np.uint8)                 # white noise: corners everywhere
    else:
        img = synth.random_image(w, h, seed=int(rng.integers(1 << 30))); img[:, : w // 3] = 77   # a flat third: empty cells, min-threshold retries
    tag = "case %d: %dx%d levels %d scale %.1f kpts %d thr %d/%d kind %d" % (case, w, h, levels, scale, kpts, ini, mn, kind)
    try:
        p = O.params(kpts, scale, levels, ini, mn)
        okp, od, occ, opyr = O.extract(img, p, True)
        ctx = hip.Context(w, h, kpts, scale, levels, ini, mn, max_images=2)
        reserve = int(rng.choice([0, 0, 4, 8, 16]))                  # > 0: the extraction kernels run as persistent work-queue grids
        if reserve: ctx.set_mapping_reserve(reserve)
        tag += " reserve %d" % reserve
        ctx.upload(0, img); ctx.upload(1, np.roll(img, -3, axis=1)); ctx.extract(2)
        ok = all(np.array_equal(ctx.pyramid_level(0, l), opyr[l]) for l in range(levels))
        gkp, gd = ctx.keypoints(0)
        ok = ok and len(gkp) == len(okp) and all(np.array_equal(okp[f], gkp[f]) for f in okp.dtype.names) and np.array_equal(od, gd)
        # brute-force matching image 0 -> image 1 against the oracle on the GPU's own keypoints of image 1
        kp1, d1 = ctx.keypoints(1)
        if len(gkp) and len(kp1):
            ctx.match_bf(0, 1)
            gq, gt, gdist = ctx.bf_matches(0, 1, 64, 0.8, True)
            oq, ot, odist = O.match_bf(gd, d1, 64, 0.8, True)
            ok = ok and np.array_equal(gq, oq) and np.array_equal(gt, ot) and np.array_equal(gdist, odist)
        # stereo: image 1 is image 0 shifted 3 px to the left, so disparities of 3 px exist; oracle on its own extraction of both
        right_img = np.roll(img, -3, axis=1)
        rkp, rd, _, rpyr = O.extract(right_img, p, True)
        fxb, base = 40.0 * w / 640.0, 0.1
        if len(okp) and len(rkp):
            oxr, odep, obi, _ = O.match_stereo(opyr, rpyr, p, okp, od, rkp, rd, fxb, base)
            ctx.match_stereo(0, 1, fxb, base)
            gxr, gdep, gbi = ctx.stereo(0)
            ok = ok and np.array_equal(gxr, oxr) and np.array_equal(gdep, odep) and np.array_equal(gbi, obi)
        print(("ok   " if ok else "FAIL ") + tag + "  -> %d keypoints" % len(gkp), flush=True)
        bad += not ok
        del ctx
    except Exception as e:                                           # noqa: BLE001
        print("ERR  " + tag + ": " + repr(e), flush=True); bad += 1
print("%d cases, %d bad, %.1f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
