"""development: when every workgroup of k_ba_schur starts and ends (build with LPSLAM_HIP_EXTRA_FLAGS=-DLPSLAM_SCHUR_STAMPS).
usage: dev_schur_stamps.py [random|contiguous]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "random"
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
prob = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0, tracks=kind, top_up=True)
ba = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
n_poses, n_free = 50, 49
n_blocks = n_free * (n_free + 1) // 2
lead = n_poses * 8
efirst = int(os.environ.get("EFIRST", 3 * n_free + 32))          # what lpslam_hip_ba_prepare expects (diagonal blocks' parts + 32)
total = 8192
for rep in range(3):
    ba.reset(); ba.optimize(True, 3)
    buf = np.zeros(8 * 8192, dtype=np.uint64)
    ctx.lib.lpslam_hip_debug_schur_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
    st = buf.reshape(-1, 8).astype(np.int64)
    live = st[(st[:, 0] > 0)]
    t0 = live[:, 0].min()
    def show(name, lo, hi):
        a = st[lo:hi]
        a = a[a[:, 1] >= a[:, 0]]
        a = a[a[:, 0] >= t0]
        if not len(a): return
        s, e, w = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01, (a[:, 2] - t0) * 0.01
        print("  %-22s n %5d  start %.2f .. %.2f  end %.2f .. %.2f (mean %.2f)  duration mean %.2f max %.2f   wait starts mean %.2f, waited mean %.2f max %.2f"
              % (name, len(a), s.min(), s.max(), e.min(), e.max(), e.mean(), (e - s).mean(), (e - s).max(), w.mean(), (e - w).mean(), (e - w).max()))
    def phases(name, lo, hi):
        a = st[lo:hi]
        a = a[(a[:, 1] >= a[:, 0]) & (a[:, 0] >= t0) & (a[:, 7] >= a[:, 0])]
        if not len(a): return
        d = lambda x, y: (a[:, x] - a[:, y]).mean() * 0.01
        print("  %-22s phases (mean us): view / flags / list bounds %.2f | records + first fetch issued %.2f | first round in LDS %.2f | rest of the loop %.2f | lane sums %.2f | to the end %.2f"
              % (name, d(3, 0), d(4, 3), d(5, 4), d(6, 5), d(7, 6), d(1, 7)))
    print("launch %d (us from the first workgroup's start)" % rep)
    show("pose side (lead)", 0, lead)
    show("further parts (front)", lead, lead + efirst)
    show("pairs part 0", lead + efirst, lead + efirst + n_blocks)
    phases("pairs part 0", lead + efirst, lead + efirst + n_blocks)
    phases("further parts (front)", lead, lead + efirst)
    show("further parts (back)", lead + efirst + n_blocks, total)
print(ba.optimize_profiled(True, 10)[0])
