#!/usr/bin/env python3
"""Generates the closed-loop goldens tests/golden/g10..g16_*.npz: per-frame poses of synthetic stereo sequences from the closed-loop
oracle (oracle/tracker.py).  The images come from the committed generator (lpslam_amd/synth.py) and are pinned by a hash.

  g10_track        640x480, 1000 keypoints, 4 levels, 24 frames, asyncMapping false                     (round 2)
  g11_track_async  the same sequence with asyncMapping true -- the product's default
  g12_track720     1280x720, 2000 keypoints, 8 levels (the configuration the benchmark is quoted on), 20 frames, asyncMapping true
  g13_track_lost   640x480, 20 frames of which 10..12 are blank: Lost -> the map is kept -> relocalisation, asyncMapping true
  g15_track_mono   640x480 monocular, 30 frames of the three-wall scene: two-view initialisation, tracking, triangulated keyframes
  g14_track_loop   640x480, 156 frames of a turn on the spot (a full turn and 108 degrees: the candidate must be detected at four keyframes in a row), loopClosure true: voting, continuity, Sim3 verification, pose graph, fusion, global BA

  g16_mono_loop    640x480 monocular, 210 frames of the three-wall scene on a rectangular path (right, up, left, down, back at the start):
                   a monocular loop -- Sim3 with a free scale from the Sim3 solver, pose graph, fusion, global BA

usage: make_golden_track.py [g10 g11 g12 g13 g14 g15 g16]      (default: all)"""
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

CASES = {
    "g10_track": dict(w=640, h=480, n=24, seq=4, points=6000, blank=(),
                      cfg=dict(max_keypoints=1000, num_levels=4, scale_factor=1.2, keyframe_interval=4, local_window=10, async_mapping=False)),
    "g11_track_async": dict(w=640, h=480, n=24, seq=4, points=6000, blank=(),
                            cfg=dict(max_keypoints=1000, num_levels=4, scale_factor=1.2, keyframe_interval=4, local_window=10, async_mapping=True)),
    "g12_track720": dict(w=1280, h=720, n=20, seq=4, points=None, blank=(),
                         cfg=dict(max_keypoints=2000, num_levels=8, scale_factor=1.2, keyframe_interval=6, local_window=10, async_mapping=True)),
    "g13_track_lost": dict(w=640, h=480, n=20, seq=4, points=6000, blank=(10, 11, 12),
                           cfg=dict(max_keypoints=1000, num_levels=4, scale_factor=1.2, keyframe_interval=4, local_window=10, async_mapping=True)),
    "g15_track_mono": dict(w=640, h=480, n=30, seq="walls", points=None, blank=(), mono=True,
                           cfg=dict(max_keypoints=2000, num_levels=3, scale_factor=1.2, keyframe_interval=4, local_window=10, async_mapping=True)),
    "g16_mono_loop": dict(w=640, h=480, n=210, seq="rectangle", points=None, blank=(), mono=True,
                          cfg=dict(max_keypoints=2000, num_levels=3, scale_factor=1.2, keyframe_interval=4, local_window=6, async_mapping=True, loop_closure=True)),
    "g14_track_loop": dict(w=640, h=480, n=156, seq="turn", points=None, blank=(),
                           cfg=dict(max_keypoints=1000, num_levels=4, scale_factor=1.2, keyframe_interval=3, local_window=4, async_mapping=True, loop_closure=True)),
}


def frames_of(case):
    c = CASES[case]
    if c["seq"] == "turn":
        return [list(f) for f in synth.turning_sequence(c["w"], c["h"], c["n"])[0]]
    if c["seq"] in ("walls", "rectangle"):
        walls = synth.WallSequence(c["w"], c["h"], 11, rectangle=(50, 45) if c["seq"] == "rectangle" else None)
        return [[walls.frame(i), None] for i in range(c["n"])]
    seq = synth.StereoSequence(c["w"], c["h"], c["seq"], n_points=c["points"]) if c["points"] else synth.StereoSequence(c["w"], c["h"], c["seq"])
    frames = [list(seq.frame(i)) for i in range(c["n"])]
    blank = np.full((c["h"], c["w"]), 110, np.uint8)
    for i in c["blank"]:
        frames[i] = [blank.copy(), blank.copy()]
    return frames


def make(case):
    c = CASES[case]
    k = synth.intrinsics(c["w"], c["h"])
    trk = (T.MonoTracker if c.get("mono") else T.StereoTracker)(c["w"], c["h"], k, **c["cfg"])
    poses, valid, sha = [], [], hashlib.sha256()
    t0 = time.time()
    for i, (l, r) in enumerate(frames_of(case)):
        sha.update(l.tobytes())
        if r is not None:
            sha.update(r.tobytes())
        p = trk.feed(l, r, 0.04 * (i + 1)) if r is not None else trk.feed(l, 0.04 * (i + 1))
        valid.append(p is not None)
        poses.append(p if p is not None else np.zeros(7))
    print("%s: %d frames in %.1f s; statistics %s; landmarks %d" % (case, c["n"], time.time() - t0, trk.stats, len(trk.landmarks)))
    poses = np.array(poses)
    print("  last pose", poses[-1], "valid", "".join("1" if v else "0" for v in valid))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", case + ".npz"), poses=poses, valid=np.array(valid), sha=sha.hexdigest(), frames=c["n"],
                        **{"stat_" + k_: v for k_, v in trk.stats.items()})


if __name__ == "__main__":
    O.build()
    for case in (sys.argv[1:] or list(CASES)):
        name = [k for k in CASES if k.startswith(case)][0]
        make(name)
