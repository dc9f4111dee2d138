#!/usr/bin/env python3
"""Open-ended randomised parity sweep on a GPU box: random image sizes, keypoint budgets, level counts, scale factors, FAST
thresholds and mapping reserves; ORB extraction (pyramid, FAST candidates, keypoints, descriptors), brute-force and stereo matching
against the CPU oracle, bit for bit.  The cases are tests/fuzz_cases.py's (a bounded slice of them runs inside `-m gpu`:
tests/test_fuzz_gpu.py).  usage: fuzz_parity.py [n_cases] [seed] [first_case]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lpslam_amd import hip                                          # noqa: E402
from oracle import oracle as O                                      # noqa: E402
import fuzz_cases                                                   # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
O.build(); hip.load()
bad = 0
t0 = time.time()
for case in range(first, first + n_cases):
    try:
        ok, tag = fuzz_cases.frontend_case(O, seed, case)
        print(("ok   " if ok else "FAIL ") + tag, flush=True)
        bad += not ok
    except Exception as e:                                           # noqa: BLE001
        print("ERR  case %d: %r" % (case, e), flush=True); bad += 1
print("%d cases, %d bad, %.1f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
