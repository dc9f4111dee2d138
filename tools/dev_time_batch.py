"""Scratch: BA set-up time and batched-solve throughput (local BA 50 KF / 5000 landmarks / ~39k observations, 10 LM iterations)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
TRACKS = os.environ.get("TRACKS", "random")
probs = [synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=s, tracks=TRACKS, top_up=True) for s in range(4)]
obs = [hip.ba_obs_array(p) for p in probs]
def make(i):
    p = probs[i % 4]
    return hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], obs[i % 4], p["cam"])
# set-up
for rep in range(3):
    t = time.perf_counter(); b = make(0); b.state(); t1 = time.perf_counter() - t
    b.close()
ts = []
for rep in range(10):
    t = time.perf_counter(); b = make(rep); t_enq = time.perf_counter() - t; b.state(); ts.append((t_enq, time.perf_counter() - t)); b.close()
print("create: enqueue %.3f ms, until ready %.3f ms (median)" % (1e3 * np.median([a for a, _ in ts]), 1e3 * np.median([b for _, b in ts])))
t = time.perf_counter(); b = make(0); b.optimize(True, 10); t1 = time.perf_counter() - t; b.close()
print("create + first optimize(10): %.3f ms" % (1e3 * t1))
for B in (tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1, 2, 4, 8, 16, 32, 64)):
    bas = [make(i) for i in range(B)]
    hip.ba_optimize_batch(bas, True, 10)
    tt = []
    for rep in range(5):
        hip.ba_reset_batch(bas)
        t = time.perf_counter(); logs = hip.ba_optimize_batch(bas, True, 10); tt.append(time.perf_counter() - t)
    m = np.median(tt)
    print("batch %3d: %.3f ms per batch, %.4f ms per problem-iteration, %.1f problems/s; iters %s" % (B, 1e3 * m, 1e3 * m / B / 10, B / m, sorted(set(len(l) for l in logs))))
    for b in bas:
        b.close()
if os.environ.get("ROUNDS"):
    S = 16
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(int(os.environ["THREADS"])) if os.environ.get("THREADS") else None
    def session_round(v0):
        t0 = time.perf_counter()
        bas = list(pool.map(make, range(v0, v0 + S))) if pool else [make(v0 + i) for i in range(S)]
        t1 = time.perf_counter()
        for i, b in enumerate(bas):
            p = probs[(v0 + i) % 4]
            b.set_state(p["poses"], p["points"])
        t1b = time.perf_counter()
        hip.ba_optimize_batch(bas, True, 10)
        t2 = time.perf_counter()
        for b in bas:
            b.state()
        t2b = time.perf_counter()
        for b in bas:
            b.close()
        t3 = time.perf_counter()
        return 1e3 * (t1 - t0), 1e3 * (t1b - t1), 1e3 * (t2 - t1b), 1e3 * (t2b - t2), 1e3 * (t3 - t2b)
    session_round(0)
    for r in range(4):
        print("round %d: create %.3f ms, set_state %.3f, batch %.3f ms, state %.3f, close %.3f ms" % ((r,) + session_round(16 * r)))
