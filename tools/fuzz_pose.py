#!/usr/bin/env python3
"""Open-ended fuzz of lpslam_hip_pose_optimize against the oracle's pose_optimize (GPU box): random observation counts (0 ... beyond
what the kernel keeps in LDS), mono / stereo mixes, gross outliers, start poses off the truth.  The cases are tests/fuzz_cases.py's
(a bounded slice runs inside `-m gpu`: tests/test_fuzz_gpu.py).  usage: fuzz_pose.py [cases] [seed] [first_case]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lpslam_amd import hip                                          # noqa: E402
from oracle import oracle as O                                      # noqa: E402
import fuzz_cases                                                   # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
O.build()
ctx = hip.Context(640, 480, 500, 1.2, 4, max_images=2)
bad = 0; worst_r = worst_t = 0.0
t0 = time.time()
for case in range(first, first + cases):
    ok, tag, dr, dt = fuzz_cases.pose_case(O, ctx, seed, case)
    worst_r = max(worst_r, dr); worst_t = max(worst_t, dt)
    if not ok:
        bad += 1; print("BAD  " + tag, flush=True)
    elif case % 20 == 0:
        print("ok   " + tag, flush=True)
print("%d cases, %d bad, worst |dq| %.2e |dt| %.2e, %.1f s" % (cases, bad, worst_r, worst_t, time.time() - t0))
sys.exit(1 if bad else 0)
