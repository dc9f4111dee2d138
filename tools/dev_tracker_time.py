"""Scratch: stage times of the stereo tracker on the bench sequence."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpslam_amd import manager, _build, synth
_build.host_library()
if os.environ.get("WITH_TORCH") == "1":
    import torch
    torch.cuda.init(); _x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
if os.environ.get("WITH_CTX") == "1":
    from lpslam_amd import hip
    _ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=32)
    _ctx.set_mapping_reserve(16)
W, H, KPTS, LEVELS, KF = 1280, 720, 2000, 8, 6
k = synth.intrinsics(W, H)
for async_map in ("true", "false", "true", "false"):
    mg = manager.Manager()
    for num in (0, 1):
        c = manager.default_camera()
        c.camera_number = num; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
        c.resolution_x = W; c.resolution_y = H; c.focal_x_baseline = k["fxb"]
        mg.set_camera(c)
    mg.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": %d, "numLevels": %d, "keyframeInterval": %d, "asyncMapping": %s, "mappingReserve": %s%s}' % (KPTS, LEVELS, KF, async_map, os.environ.get("RESERVE", "0"), os.environ.get("TRACKER_CFG", "")))
    arrivals = []
    mg.collect_results(on_result=lambda: arrivals.append(time.perf_counter())); mg.provide_odometry(native=os.environ.get("NATIVE_ODOM", "1") == "1")
    log = os.path.join(tempfile.mkdtemp(), "slam.log")
    mg.log_to_file(log)
    seq = synth.StereoSequence(W, H, 4)
    frames = [seq.frame(i) for i in range(int(os.environ.get("FRAMES", "90")))]
    late = os.environ.get("ENQUEUE_AFTER_START", "0") == "1"
    if late:
        mg.start()
    ta = time.perf_counter()
    t0 = ta
    for i, (l, r) in enumerate(frames):
        mg.add_stereo((i + 1) * 40_000_000, l, r)
    print("enqueue of %d frames: %.2f ms" % (len(frames), 1e3 * (time.perf_counter() - ta)))
    if not late:
        t0 = time.perf_counter()
        mg.start()
    while len(mg.results) < len(frames) and time.perf_counter() - t0 < 60:
        time.sleep(0.0005)
    dt = time.perf_counter() - t0
    mg.stop()
    print("first result after %.2f ms; gaps of the next 12 results, ms: %s; median gap of the rest %.3f ms" % (1e3 * (arrivals[0] - t0), " ".join("%.2f" % (1e3 * (b - a)) for a, b in zip(arrivals[:12], arrivals[1:13])), 1e3 * float(__import__("numpy").median([b - a for a, b in zip(arrivals[13:-1], arrivals[14:])]))))
    print([l.strip() for l in open(log, errors="replace") if "Worker statistics" in l][:3])
    print("asyncMapping", async_map, "%.1f frames/s" % (len(frames) / dt), manager.Manager.statistics(log))
