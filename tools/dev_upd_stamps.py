"""development: phase stamps inside k_ba_update (build with LPSLAM_HIP_EXTRA_FLAGS=-DLPSLAM_UPD_STAMPS).  usage: dev_upd_stamps.py [contiguous|random]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "contiguous"
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
prob = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0, tracks=kind, top_up=True)
ba = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
for rep in range(3):
    ba.reset(); ba.optimize(True, 3)
    buf = np.zeros(32)
    ctx.lib.lpslam_hip_debug_upd_stamps(buf.ctypes.data_as(C.c_void_p))
    t0 = buf[10]
    rel = lambda k: (buf[k] - t0) * 0.01
    print("land block 0 (us from its first instruction): view/flags %.2f | loads issued %.2f | poses done %.2f | W+xp terms %.2f | barrier %.2f | seg sums %.2f | solve %.2f | published %.2f | pass2 math %.2f | pass2 done %.2f | before ticket %.2f"
          % tuple(rel(k) for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11)))
    print("   pass 2 of tid 0: residual + weight done %.2f | Jacobians + H_ll shares done %.2f | W formed and stored (stamp 8) %.2f" % (rel(14), rel(15), rel(8)))
    if buf[12]: print("   pass 2 repeated (hot): starts %.2f, chunk math done %.2f (stamp 8 is overwritten by the repeat), ends %.2f" % (rel(12), rel(8), rel(13)))
    print("   deciding block: ticket won %.2f | totals %.2f | decided %.2f ;   keyframe block 0: start %.2f | preloaded %.2f | wait over %.2f | math done %.2f | stored %.2f"
          % tuple(rel(k) for k in (16, 17, 18, 24, 25, 26, 27, 28)))
print(ba.optimize_profiled(True, 10)[0])
