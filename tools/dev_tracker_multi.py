"""development: N LpSlamManager instances in one process on one GPU, aggregate frames/s (bench.py's tracker_multi alone).
usage: [GPU_MAX_HW_QUEUES=..] dev_tracker_multi.py 1,4,8,16 [frames]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpslam_amd import manager, synth, _build, hip
if os.environ.get("WITH_TORCH") == "1":              # what bench.py's process looks like by the time it reaches tracker_multi
    import torch
    torch.cuda.init(); _x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
if os.environ.get("WITH_CTX") == "1":
    _ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=96)
    _ctx.set_mapping_reserve(16)
if os.environ.get("WITH_BA") == "1":                 # ... after its mapping thread has solved windows: a high-priority stream stays in the context's pool
    _p = synth.ba_problem(10, 300, 1500, 1280, 720, seq_id=0, tracks="contiguous", top_up=True)
    _c2 = _ctx if os.environ.get("WITH_CTX") == "1" else hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
    for _ in range(int(os.environ.get("WITH_BA_N", "1"))):
        _b = hip.BundleAdjuster(_c2, _p["poses"], _p["fixed"], _p["points"], hip.ba_obs_array(_p), _p["cam"]); _b.optimize(True, 3); _b.close()
    if os.environ.get("WITH_BA_CLOSE") == "1": _c2.close(); _c2 = None
if os.environ.get("LPSLAM_DEV_BLOCKING_SYNC"):        # hipStreamSynchronize by interrupt instead of a spin (hipDeviceScheduleBlockingSync), before the runtime makes its context
    import ctypes
    print("hipSetDeviceFlags ->", ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(ctypes.c_uint(4)))
if os.environ.get("LPSLAM_DEV_FLAT"): hip.set_flat_priorities(True)
_build.host_library()
W, H, KPTS, LEVELS, KF = 1280, 720, 2000, 8, 6
counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,4,8,16").split(",")]
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
k = synth.intrinsics(W, H)
seq = synth.StereoSequence(W, H, 4)
frames = [seq.frame(i) for i in range(n_frames)]
extra = os.environ.get("LPSLAM_DEV_TRACKER_CFG", "")


def make():
    mg = manager.Manager()
    for num in (0, 1):
        c = manager.default_camera()
        c.camera_number = num; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
        c.resolution_x = W; c.resolution_y = H; c.focal_x_baseline = k["fxb"]
        mg.set_camera(c)
    mg.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": %d, "numLevels": %d, "keyframeInterval": %d, "device": 0%s}' % (KPTS, LEVELS, KF, extra))
    mg.count_results(); mg.provide_odometry(native=True)
    if os.environ.get("LPSLAM_DEV_STATS"):
        import tempfile
        mg._log = os.path.join(tempfile.mkdtemp(prefix="lpslam_multi_"), "slam.log"); mg.log_to_file(mg._log)
    mg.start()
    return mg


def run(n):
    mgs = [make() for _ in range(n)]

    def feed(mg):
        for i, (l, r) in enumerate(frames):
            mg.add_stereo((i + 1) * 40_000_000, l, r)
    th = [threading.Thread(target=feed, args=(mg,)) for mg in mgs]
    t0 = time.perf_counter()
    for t in th: t.start()
    want = n * len(frames)
    while sum(mg.result_counts()[0] for mg in mgs) < want and time.perf_counter() - t0 < 120:
        time.sleep(0.001)
    dt = time.perf_counter() - t0
    for t in th: t.join()
    got = sum(mg.result_counts()[0] for mg in mgs); valid = sum(mg.result_counts()[1] for mg in mgs)
    for mg in mgs: mg.stop()
    try:
        st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        print("   cpu.stat: usage %.2f s, throttled %d periods / %.3f s" % (int(st["usage_usec"]) * 1e-6, int(st["nr_throttled"]), int(st["throttled_usec"]) * 1e-6))
    except Exception:
        pass
    if os.environ.get("LPSLAM_DEV_STATS"):
        st = [mg.tracker_statistics() for mg in mgs]      # (each manager's own line: the log file is process-wide)
        keys = ("ms_per_frame", "ms_front_end", "ms_track", "ms_local_map", "ms_keyframe", "ms_dev_extract", "ms_dev_match", "ms_dev_pose", "ms_dev_get", "ms_prefetch_wait", "ms_prefetch_busy", "ms_kf_wait", "ms_kf_apply", "ms_kf_insert", "ms_kf_loop", "ms_kf_prepare", "ms_map_solve", "prefetched")
        print("   per-manager statistics (mean over %d managers): " % n + ", ".join("%s %.3f" % (k, sum(x.get(k, 0) for x in st) / len(st)) for k in keys))
    return got, valid, dt


run(1)
for n in counts:
    got, valid, dt = run(n)
    print("GPU_MAX_HW_QUEUES=%s  %2d managers: %5d results (%d valid) in %.3f s = %.0f frames/s aggregate, %.0f per manager" % (os.environ.get("GPU_MAX_HW_QUEUES", "-"), n, got, valid, dt, got / dt, got / dt / n), flush=True)
