"""development: bench.py's multi_session mapping side alone -- rounds of 16 fresh windows (prepared on 8 threads, built by one chain beside the
previous round's solve, set_state, one batched solve, read back, released) -- with the time of every phase of every round, three invocations
in a row (the first round of an invocation starts an empty pipeline).  usage: dev_session_rounds.py [random|contiguous] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from lpslam_amd import hip, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "contiguous"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
S = 16
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
probs = [synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=s, tracks=kind, top_up=True) for s in range(4)]
obs = [hip.ba_obs_array(p) for p in probs]
creators = ThreadPoolExecutor(8); ahead = ThreadPoolExecutor(1)
def new(v):
    p = probs[v % 4]
    return hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], obs[v % 4], p["cam"], build=False)
def make_all(v0):
    t0 = time.perf_counter()
    bas = list(creators.map(new, range(v0, v0 + S)))
    t1 = time.perf_counter()
    hip.ba_build_batch(bas)
    return bas, 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)
for inv in range(3):
    fut = ahead.submit(make_all, 0)
    t_all = time.perf_counter()
    for r in range(rounds):
        ta = time.perf_counter()
        bas, t_prep, t_build = fut.result()
        tb = time.perf_counter()
        if r + 1 < rounds: fut = ahead.submit(make_all, (r + 1) * S)
        ps = [probs[(r * S + i) % 4] for i in range(S)]
        hip.ba_set_state_batch(bas, [p["poses"] for p in ps], [p["points"] for p in ps])
        tc = time.perf_counter()
        hip.ba_optimize_batch(bas, True, 10)
        td = time.perf_counter()
        hip.ba_state_batch(bas)
        te = time.perf_counter()
        for b in bas: b.close()
        tf = time.perf_counter()
        print("invocation %d round %d: waited %.2f ms (prepare %.2f, build call %.2f), set_state %.2f, batch %.2f, state %.2f, close %.2f" % (inv, r, 1e3 * (tb - ta), t_prep, t_build, 1e3 * (tc - tb), 1e3 * (td - tc), 1e3 * (te - td), 1e3 * (tf - te)))
    print("invocation %d: %.2f ms per round" % (inv, 1e3 * (time.perf_counter() - t_all) / rounds))
