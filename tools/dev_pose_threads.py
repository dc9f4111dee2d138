"""development: do latency-bound single-workgroup kernels of independent contexts add up?  N threads, each with its own context, call the
pose optimiser in a loop.  usage: [LPSLAM_DEV_FLAT=1] dev_pose_threads.py 1,2,4,8"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
if os.environ.get("LPSLAM_DEV_FLAT"): hip.set_flat_priorities(True)
counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
prob = synth.ba_problem(2, 60, 120, 640, 480, seq_id=60)
sel = prob["obs_pose"] == 1
obs = hip.ba_obs_array(prob)[sel]; pose = prob["poses"][1].copy()
CALLS = 400
for n in counts:
    ctxs = [hip.Context(320, 240, 400, 1.2, 4, max_images=1) for _ in range(n)]
    for c in ctxs: hip.pose_optimize(c, pose, prob["points"], obs, prob["cam"])
    def work(c):
        for _ in range(CALLS): hip.pose_optimize(c, pose, prob["points"], obs, prob["cam"])
    th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%2d threads: %.1f us per call and thread, %.0f calls/s aggregate" % (n, 1e6 * dt / CALLS, n * CALLS / dt))
    for c in ctxs: c.close()
