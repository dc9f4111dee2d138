"""Scratch: phase stamps of k_chol_wg (LPSLAM_CW_STAMP=1)."""
import sys, os
os.environ["LPSLAM_CW_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
p = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0)
b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hip.ba_obs_array(p), p["cam"])
b.optimize(True, 3)
out = np.zeros(256, np.uint64)
hip._check(hip.load().lpslam_hip_debug_ba_scratch(b.h, C.c_int64(16384), C.c_int64(256), out.ctypes.data_as(C.c_void_p)))
w0, w7 = out[:128].astype(np.int64), out[128:].astype(np.int64)
t0 = w0[0]
print("wave0 stamps (cycles of 100 MHz memtime => x10 ns):")
idx = [i for i in range(128) if w0[i] > 0]
print([(i, int(w0[i] - t0)) for i in idx])
idx = [i for i in range(128) if w7[i] > 0]
print("wave7:", [(i, int(w7[i] - t0)) for i in idx])
