#!/usr/bin/env python3
"""Generates tests/golden/g10_track.npz: per-frame poses of a 24-frame 640x480 stereo sequence from the closed-loop oracle
(oracle/tracker.py).  The images come from the committed generator (lpslam_amd/synth.py, sequence 4) and are pinned by a hash."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O           # noqa: E402
from oracle import tracker as T          # noqa: E402
from lpslam_amd import synth             # noqa: E402

W, H, N = 640, 480, 24
CFG = dict(max_keypoints=1000, num_levels=4, scale_factor=1.2, keyframe_interval=4, local_window=10)


def main():
    O.build()
    k = synth.intrinsics(W, H)
    seq = synth.StereoSequence(W, H, 4, n_points=6000)
    trk = T.StereoTracker(W, H, k, **CFG)
    poses, sha = [], hashlib.sha256()
    t0 = time.time()
    for i in range(N):
        l, r = seq.frame(i)
        sha.update(l.tobytes()); sha.update(r.tobytes())
        poses.append(trk.feed(l, r))
    print("tracked %d frames in %.1f s; statistics %s; landmarks %d" % (N, time.time() - t0, trk.stats, len(trk.landmarks)))
    poses = np.array(poses)
    print("last pose", poses[-1])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g10_track.npz"), poses=poses, sha=sha.hexdigest(), frames=N,
                        keyframes=trk.stats["keyframes"], **{"stat_" + k_: v for k_, v in trk.stats.items()})


if __name__ == "__main__":
    main()
