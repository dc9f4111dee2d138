"""development: in-kernel wall-clock stamps of k_chol_band (build with LPSLAM_HIP_EXTRA_FLAGS=-DLPSLAM_BC_STAMPS)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
p = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0, tracks="contiguous", top_up=True)
b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hip.ba_obs_array(p), p["cam"])
for _ in range(3):
    b.reset(); b.optimize(True, 3)
ptr, n = b.reduced_buffer()
rt = C.CDLL("libamdhip64.so")
dim, npad = 294, 320
buf = np.zeros(8 * 23)
rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
rc = rt.hipMemcpy(buf.ctypes.data, C.c_void_p(ptr + 8 * (dim + 2) * npad), buf.nbytes, 2)
st = buf.reshape(23, 8); ex = st[20]; gs = st[22]
t0 = st[0, 0]
names = ["start", "loaded", "factored", "published", "after B1", "after E1", "w2 tiles done", "w2 rows stored"]
for s in range(19):
    print("strip %2d: " % s + "  ".join("%s %6.2f" % (nm, (st[s, k] - t0) / 100.0) for k, nm in enumerate(names)))
print("per strip us:", (st[18, 0] - st[0, 0]) / 100.0 / 18)
print("entry -> loop start %.2f us, loop %.2f us, -> back operands + prefetch %.2f us, back loop %.2f us" % ((ex[0] - ex[4]) / 100, (ex[1] - ex[0]) / 100, (ex[2] - ex[1]) / 100, (ex[3] - ex[2]) / 100))
print("k_schur_group WG 0: entry -> zeroed %.2f, -> loads issued+returned %.2f, -> staged %.2f, -> mfma+stored %.2f us" % tuple((gs[i + 1] - gs[i]) / 100 for i in range(4)))
hb = st[21]
print("helper: entry %.2f us after the main chain's entry; entry -> loop start %.2f, loop %.2f, flush + dump + flag %.2f us" % ((hb[4] - ex[4]) / 100, (hb[0] - hb[4]) / 100, (hb[1] - hb[0]) / 100, (hb[5] - hb[1]) / 100))
