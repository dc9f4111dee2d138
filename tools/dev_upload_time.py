"""where the PCIe-inclusive leg of bench.py loses time: copy rate alone, front end with / without uploads, BA beside uploads"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

PACE = float(os.environ.get('PACE', '0'))

def main():
    wl = bench.Workload(0, 0, 16, with_ba=True)
    wl.ctx.set_mapping_reserve(int(os.environ.get('RESERVE', '0')))
    F = wl.F
    # (a) copies alone
    for rep in range(3):
        wl.ctx.sync(); t0 = time.perf_counter()
        for s in range(6):
            wl.upload_step(s)
        t_host = time.perf_counter() - t0
        wl.ctx.extract_range(0, 2); wl.ctx.sync()      # waits for slot 0's last copy only; drain everything:
        for s in range(3):
            wl.ctx.extract_range(2 * ((s * F) % wl.R), 2)
        wl.ctx.sync()
        t_all = time.perf_counter() - t0
        print("copies: 6 steps, host enqueue %.3f ms/step, done after %.3f ms/step -> %.1f GB/s" % (1e3 * t_host / 6, 1e3 * t_all / 6, 6 * 2 * F * bench.W * bench.H / t_all / 1e9))
    # (b) front end only
    for upload in (False, True, False, True):
        wl.ctx.sync(); t0 = time.perf_counter()
        wl.front_end_steps(0, 12, upload); wl.ctx.sync()
        print("front end only, upload=%s: %.3f ms/step" % (upload, 1e3 * (time.perf_counter() - t0) / 12))
    # (c) BA only, with a thread that uploads beside it
    for upload in (False, True, False, True):
        stop = threading.Event()
        def up():
            s = 0
            while not stop.is_set():
                wl.upload_step(s); s += 1
                wl.ctx.sync()           # the copy stream only: nothing else is enqueued on the context
                if PACE: time.sleep(PACE)
        th = threading.Thread(target=up) if upload else None
        if th: th.start()
        t0 = time.perf_counter(); wl.bundle_adjust_pipelined(16); t = (time.perf_counter() - t0) / 16
        stop.set()
        if th: th.join()
        print("BA pipelined, uploads beside=%s: %.3f ms/keyframe" % (upload, 1e3 * t))
    # (d) full
    for upload in (False, True, False, True):
        t0 = time.perf_counter(); wl.run_steps(3, 12, upload=upload); print("full, upload=%s: %.3f ms/step" % (upload, 1e3 * (time.perf_counter() - t0) / 12))

main()
