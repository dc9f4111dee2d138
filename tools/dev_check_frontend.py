"""Developer diagnostic: HIP front end vs oracle on one synthetic frame (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from lpslam_amd import hip, synth

def main(w=1280, h=720, kpts=2000, levels=8):
    p = O.params(kpts, 1.2, levels)
    seq = synth.StereoSequence(w, h, 0, n_points=max(200, int(20000 * w * h / (1280 * 720))))
    L, R = seq.frame(0)
    ctx = hip.Context(w, h, kpts, 1.2, levels, max_images=2)
    ctx.upload(0, L); ctx.upload(1, R)
    t = time.time(); ctx.extract(2); ctx.sync(); print("extract wall", time.time() - t)
    okp, od, occ, opyr = O.extract(L, p, True)
    okr, odr, _, opyr_r = O.extract(R, p, True)
    for l in range(levels):
        g = ctx.pyramid_level(0, l)
        print("level", l, g.shape, "pyr equal", np.array_equal(g, opyr[l]), end=" ")
        oc = O.fast_level(opyr[l])
        gc = ctx.candidates(0, l)
        same = len(oc) == len(gc) and np.array_equal(oc, gc)
        print("cand", len(oc), len(gc), same)
        if not same and len(oc) and len(gc):
            n = min(len(oc), len(gc))
            bad = np.nonzero((oc[:n] != gc[:n]))[0]
            print("   first diff at", bad[:5], oc[bad[:3]], gc[bad[:3]])
    gkp, gd = ctx.keypoints(0)
    print("kpts", len(okp), len(gkp))
    n = min(len(okp), len(gkp))
    for f in okp.dtype.names:
        eq = np.array_equal(okp[f][:n], gkp[f][:n])
        print("  field", f, eq, "" if eq else np.nonzero(okp[f][:n] != gkp[f][:n])[0][:8])
    print("  desc equal", np.array_equal(od[:n], gd[:n]), (od[:n] != gd[:n]).any(axis=1).sum())
    # stereo
    k = synth.intrinsics(w, h)
    ctx.match_stereo(0, 1, k["fxb"], k["baseline"])
    gxr, gdep, gbi = ctx.stereo(0)
    oxr, odep, obi, nv = O.match_stereo(opyr, opyr_r, p, okp, od, okr, odr, k["fxb"], k["baseline"])
    print("stereo idx equal", np.array_equal(gbi, obi), "xr", np.array_equal(gxr, oxr), "depth", np.array_equal(gdep, odep), nv, (gdep > 0).sum())
    # bf
    ctx.match_bf(0, 1)
    gb = ctx.bf_knn2(0)
    ob = O.match_bf_knn2(od, odr)
    print("bf equal", [np.array_equal(a, b) for a, b in zip(gb, ob)])

if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    main(*a)
