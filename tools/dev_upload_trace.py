"""short run of the PCIe-inclusive loop for a rocprofv3 timeline (kernel + memory-copy trace)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = bench.Workload(0, 0, 16, with_ba=True)
up = os.environ.get("UPLOAD", "1") == "1"
wl.run_steps(0, 3, upload=up)
t0 = time.perf_counter(); wl.run_steps(3, 6, upload=up); print("upload=%s %.3f ms/step" % (up, 1e3 * (time.perf_counter() - t0) / 6))
