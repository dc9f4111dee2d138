"""development: where a k_fast_cells workgroup's cycles go (build with LPSLAM_HIP_EXTRA_FLAGS=-DLPSLAM_FAST_STAMPS; thread 0's clock,
summed over the workgroups of a 32-image launch).  usage: dev_fast_stamps.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
n = 32
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=n)
seq = synth.StereoSequence(1280, 720, 0)
for i in range(n // 2):
    l, r = seq.frame(i % 8)
    ctx.upload(2 * i, l); ctx.upload(2 * i + 1, r)
ctx.stage("pyramid", n); ctx.sync()
NWG = 32768
buf = np.zeros(8 * NWG, dtype=np.uint64)
ptr = buf.ctypes.data_as(C.c_void_p)
names = ["geometry + mask + staging", "pre-test + queue", "strength", "wait for the other wavefronts", "suppression", "offsets + keys"]
for rep in range(3):
    ctx.lib.lpslam_hip_debug_fast_stamps(ptr, NWG, 1)
    ctx.timer_begin(0); ctx.stage("fast", n); ctx.timer_end(0); ctx.sync()
    ctx.lib.lpslam_hip_debug_fast_stamps(ptr, NWG, 0)
    a = buf.reshape(-1, 8).astype(np.float64); a = a[a[:, 7] > 0]
    v = a.sum(axis=0)
    tot = v[:6].sum()
    print("launch %.1f us, %d workgroups, %.0f cycles per workgroup: " % (1e3 * ctx.timer_ms(0), v[7], tot / max(v[7], 1)) + ", ".join("%s %.0f (%.0f %%)" % (nm, v[k] / max(v[7], 1), 100 * v[k] / tot) for k, nm in enumerate(names)))
