#!/usr/bin/env python3
"""bench.py -- throughput of the lpslam hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] + configs[2], the configuration the metric is quoted on): 1280x720 stereo,
2000 keypoints / image, 8 pyramid levels.  One "step" = one keyframe interval of the path:
    6 stereo frames: ORB extraction of 12 images, 6 stereo matches, 6 temporal brute-force matches (2000 x 2000),
    1 local bundle adjustment: 50 keyframes / 5000 landmarks / ~40k stereo observations, 10 LM iterations (Huber).
Inputs (frames, BA problem) are synthetic and resident in HBM before the timed region.
value = frames/s over the whole job (all ranks); N > 1 runs one independent sequence per GPU (replicas, no data-path
collective, "weak" scaling).  The JSON line also carries the roofline of the dominant front-end kernel (HIP-event
timed on the stream it runs on) and the CPU oracle timed on a bounded sample (rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H, KPTS, LEVELS = 1280, 720, 2000, 8
FRAMES_PER_STEP = 6
BA_KF, BA_PTS, BA_OBS, BA_ITERS = 50, 5000, 40000, 10
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)

# timer slots
T_PYR, T_FAST, T_DIST, T_DESC, T_STEREO, T_BF, T_BA = range(7)
STAGE_NAMES = ["pyramid", "fast", "distribute", "describe", "stereo", "bf", "ba"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_STEP, help="stereo frames per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-ba", action="store_true", help="front end only (BASELINE configs[1])")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (front end alone, BA spread, pose graph): profiling runs")
    return ap.parse_args()


class Workload:
    def __init__(self, device, seq_id, frames, with_ba=True):
        from lpslam_amd import hip, synth
        self.hip, self.synth = hip, synth
        self.F = frames
        self.k = synth.intrinsics(W, H)
        self.ctx = hip.Context(W, H, KPTS, 1.2, LEVELS, max_images=2 * frames, device=device)
        seq = synth.StereoSequence(W, H, seq_id)
        self.host_frames = [seq.frame(i) for i in range(frames)]
        for i, (l, r) in enumerate(self.host_frames):
            self.ctx.upload(2 * i, l); self.ctx.upload(2 * i + 1, r)
        self.ba = None
        if with_ba:
            self.prob = synth.ba_problem(BA_KF, BA_PTS, BA_OBS, W, H, seq_id)
            self.ba = hip.BundleAdjuster(self.ctx, self.prob["poses"], self.prob["fixed"], self.prob["points"],
                                         hip.ba_obs_array(self.prob), self.prob["cam"])
        self.ctx.sync()

    def front_end(self):
        c, F = self.ctx, self.F
        c.extract(2 * F)
        c.match_stereo_strided(0, 1, 2, F, self.k["fxb"], self.k["baseline"])
        if F > 1:
            c.match_bf_strided(2, 0, 2, F - 1)
        c.match_bf(0, 2 * F - 2)                  # first frame of this interval against the last of the previous one

    def bundle_adjust(self):
        self.ba.reset()
        self.ba.optimize(True, BA_ITERS)

    def step(self, timers=False):
        c, F = self.ctx, self.F
        if not timers:
            # the local BA of this interval runs on its own stream beside the front end of the interval's frames, as the
            # reference's mapping thread runs beside tracking: enqueue it, enqueue the frames, then wait for both
            if self.ba is not None:
                self.ba.reset()
                self.ba.optimize_begin(True, BA_ITERS)
            self.front_end()
            if self.ba is not None:
                self.ba.optimize_end()
            return
        for slot, stage in ((T_PYR, "pyramid"), (T_FAST, "fast"), (T_DIST, "distribute"), (T_DESC, "describe")):
            c.timer_begin(slot); c.stage(stage, 2 * F); c.timer_end(slot)
        c.timer_begin(T_STEREO); c.match_stereo_strided(0, 1, 2, F, self.k["fxb"], self.k["baseline"]); c.timer_end(T_STEREO)
        c.timer_begin(T_BF)
        if F > 1:
            c.match_bf_strided(2, 0, 2, F - 1)
        c.match_bf(0, 2 * F - 2)
        c.timer_end(T_BF)
        if self.ba is not None:
            c.sync()
            t0 = time.perf_counter(); self.bundle_adjust(); self.ba_wall_ms = 1e3 * (time.perf_counter() - t0)

    def extract_bytes(self):
        """B_img = (P - P_7) + (P - P_0) + P + 2P + 2 K 31^2 + 60 K per image (17 236 083 B at 1280x720 / 2000)."""
        P = [w * h for w, h in zip(self.ctx.level_w, self.ctx.level_h)]
        Ps = sum(P)
        return 2 * self.F * ((Ps - P[-1]) + (Ps - P[0]) + Ps + 2 * Ps + 2 * KPTS * 31 * 31 + 60 * KPTS)

    def algorithmic_bytes(self):
        """per-image pass-structured bytes of each front-end kernel group."""
        P = [w * h for w, h in zip(self.ctx.level_w, self.ctx.level_h)]
        Psum = sum(P)
        n_img = 2 * self.F
        K = KPTS
        return {
            "pyramid": n_img * ((Psum - P[-1]) + (Psum - P[0])),
            "fast": n_img * Psum,
            "describe": n_img * (2 * K * 31 * 31 + K * 60),
        }


_CPU_MT = None


def cpu_baseline(frames_sample=12, ba_solves=2):
    """Oracle (CPU restatement, 1 thread, -O3 without -march=native) on a bounded sample of the same workload."""
    from oracle import oracle as O
    from lpslam_amd import synth
    p = O.params(KPTS, 1.2, LEVELS)
    seq = synth.StereoSequence(W, H, 0)
    k = synth.intrinsics(W, H)
    frames = [seq.frame(i) for i in range(frames_sample)]
    prob = synth.ba_problem(BA_KF, BA_PTS, BA_OBS, W, H, 0)
    obs = O.ba_obs(prob)
    t0 = time.perf_counter()
    prev = None
    for l, r in frames:
        kl, dl, _, pl = O.extract(l, p, True)
        kr, dr, _, pr = O.extract(r, p, True)
        O.match_stereo(pl, pr, p, kl, dl, kr, dr, k["fxb"], k["baseline"])
        if prev is not None:
            O.match_bf_knn2(dl, prev)
        else:
            O.match_bf_knn2(dl, dl)
        prev = dl
    t_front = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(ba_solves):
        O.ba_optimize(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], True, BA_ITERS)
    t_ba = (time.perf_counter() - t0) / ba_solves
    per_frame = t_front / frames_sample + t_ba / FRAMES_PER_STEP
    # the reference's own threading: left / right extraction on two threads (the std::async pair of
    # src/Trackers/OpenVSLAMStereoTracker.cpp:199-213), the local BA on a third (OpenVSLAM's mapping thread); ctypes releases the GIL
    import threading
    n_mt = 6

    def ba_thread():
        O.ba_optimize(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], True, BA_ITERS)
    t0 = time.perf_counter()
    th_ba = threading.Thread(target=ba_thread); th_ba.start()
    prev = None
    for l, r in frames[:n_mt]:
        res = {}
        th_r = threading.Thread(target=lambda: res.__setitem__("r", O.extract(r, p, True))); th_r.start()
        kl, dl, _, pl = O.extract(l, p, True)
        th_r.join()
        kr, dr, _, pr = res["r"]
        O.match_stereo(pl, pr, p, kl, dl, kr, dr, k["fxb"], k["baseline"])
        O.match_bf_knn2(dl, prev if prev is not None else dl)
        prev = dl
    th_ba.join()
    t_mt = time.perf_counter() - t0
    global _CPU_MT
    _CPU_MT = {"value": round(n_mt / t_mt, 3), "unit": "frames/s", "cores": 3, "kind": "port",
               "sample": "%d stereo frames with left / right extraction on two threads beside one local-BA solve of %d LM iterations on a third "
                         "(one keyframe interval of the workload)" % (n_mt, BA_ITERS)}
    return {"value": round(1.0 / per_frame, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d stereo frames (extract L+R, stereo match, 2000x2000 BF) + %d local-BA solves of %d LM iterations "
                      "amortised 1 per %d frames; single thread" % (frames_sample, ba_solves, BA_ITERS, FRAMES_PER_STEP),
            "front_end_ms_per_frame": round(1e3 * t_front / frames_sample, 2), "ba_ms_per_iter": round(1e3 * t_ba / BA_ITERS, 3)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch            # noqa: F401  (before the HIP library: both must share one HIP runtime, torch's loads first)
        import torch.distributed  # noqa: F401
    from lpslam_amd import hip
    ndev = hip.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    device = local_rank % ndev
    backend = os.environ.get("LPSLAM_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for single-GPU rehearsals
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)

    wl = Workload(device, rank, args.frames, with_ba=not args.no_ba)

    def sync_tensor(values, op=None):
        import torch
        t = torch.tensor(values, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
        if backend == "nccl":
            torch.cuda.synchronize()
        return t

    def barrier():
        wl.ctx.sync()
        if dist is not None:
            sync_tensor([0.0])

    import threading

    def run_steps(k):
        # the local BA runs on its own stream and host thread beside the front end of the following frames, as the
        # reference's mapping thread does; both finish their k units before the step count is met
        if wl.ba is None:
            for _ in range(k):
                wl.front_end()
            wl.ctx.sync()
            return
        err = []

        def ba_loop():
            try:
                for _ in range(k):
                    wl.bundle_adjust()
            except Exception as e:      # noqa: BLE001
                err.append(e)
        th = threading.Thread(target=ba_loop)
        th.start()
        for _ in range(k):
            wl.front_end()
        wl.ctx.sync()
        th.join()
        if err:
            raise err[0]

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        elapsed = float(sync_tensor([elapsed], dist.ReduceOp.MAX).item())

    # instrumented pass (same steps, HIP events around every stage on the context stream)
    stage_ms = np.zeros(len(STAGE_NAMES))
    n_inst = max(3, min(args.steps, 10))
    for _ in range(n_inst):
        wl.step(timers=True)
        wl.ctx.sync()
        for s in range(len(STAGE_NAMES)):
            if s == T_BA:
                stage_ms[s] += wl.ba_wall_ms if wl.ba is not None else 0.0      # BA: host wall time of reset + 10 LM iterations
                continue
            stage_ms[s] += wl.ctx.timer_ms(s)
    stage_ms /= n_inst

    # PCIe-inclusive rate (host frames uploaded inside the loop) -- reported beside, never as `value`
    t1 = time.perf_counter()
    n_pcie = max(2, min(args.steps, 5))
    for _ in range(n_pcie):
        for i, (l, r) in enumerate(wl.host_frames):
            wl.ctx.upload(2 * i, l); wl.ctx.upload(2 * i + 1, r)
        wl.step()
    wl.ctx.sync()
    pcie_fps = n_pcie * args.frames / (time.perf_counter() - t1)

    # SURVEY.md 8(d) extras, outside the timed region: front end alone in batched and single-frame-latency mode, the spread of
    # the BA time per LM iteration, and the Sim3 pose graph of BASELINE config 5's keyframe count
    extras = {}
    if rank == 0 and not args.no_extras and world == 1:          # the scaling runs (N > 1) print the timed line only
        n_fe = max(3, min(args.steps, 10))
        wl.ctx.sync(); t2 = time.perf_counter()
        for _ in range(n_fe):
            wl.front_end()
        wl.ctx.sync()
        fe_batched = n_fe * args.frames / (time.perf_counter() - t2)
        lat = []
        for _ in range(20):
            t2 = time.perf_counter()
            wl.ctx.extract(2)
            wl.ctx.match_stereo_strided(0, 1, 2, 1, wl.k["fxb"], wl.k["baseline"])
            wl.ctx.match_bf(0, 2)
            wl.ctx.sync()
            lat.append(1e3 * (time.perf_counter() - t2))
        extras["front_end"] = {"batched_frames_per_s": round(fe_batched, 1), "frames_per_launch": args.frames,
                               "single_frame_latency_ms": round(float(np.median(lat)), 4)}
        # K0 on-device undistort / rectify: 12 remaps of a staged raw frame, 8 B per pixel algorithmic
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        for eye in (0, 1):
            wl.ctx.set_rectify_map(eye, xx * 0.97 + 15 + 3 * np.sin(yy / 50), yy * 0.97 + 9 + 3 * np.cos(xx / 70))
        wl.ctx.upload_raw(0, 0, wl.host_frames[0][0])
        for i in range(2 * args.frames):
            wl.ctx.remap_staged(i, i & 1)
        wl.ctx.sync(); t2 = time.perf_counter()
        for _ in range(10):
            for i in range(2 * args.frames):
                wl.ctx.remap_staged(i, i & 1)
        wl.ctx.sync()
        t_rm = (time.perf_counter() - t2) / (10 * 2 * args.frames)
        extras["remap"] = {"us_per_image": round(1e6 * t_rm, 2), "algorithmic_GBps": round(8.0 * W * H / t_rm / 1e9, 1)}
        for i, (l, r) in enumerate(wl.host_frames):          # restore the resident frames
            wl.ctx.upload(2 * i, l); wl.ctx.upload(2 * i + 1, r)
        if wl.ba is not None:
            per = []
            for _ in range(10):
                t2 = time.perf_counter(); wl.bundle_adjust(); per.append(1e3 * (time.perf_counter() - t2) / BA_ITERS)
            extras["ba_ms_per_iter_spread"] = {"mean": round(float(np.mean(per)), 4), "p50": round(float(np.median(per)), 4)}
        # several independent SLAM sessions on the one GPU (each its own context, streams and BA): the single session above is
        # latency bound and leaves most CUs idle; this is what one MI355X sustains when it serves S sequences at once
        S = 4
        others = [Workload(device, 100 + i, args.frames, with_ba=wl.ba is not None) for i in range(S - 1)]
        sessions = [wl] + others

        def session_steps(w_, k_):
            ths = []
            if w_.ba is not None:
                ths.append(threading.Thread(target=lambda: [w_.bundle_adjust() for _ in range(k_)]))
                ths[-1].start()
            for _ in range(k_):
                w_.front_end()
            w_.ctx.sync()
            for th in ths:
                th.join()
        for w_ in sessions:
            session_steps(w_, 1)
        k_ms = max(5, min(args.steps, 20))
        t2 = time.perf_counter()
        ths = [threading.Thread(target=session_steps, args=(w_, k_ms)) for w_ in sessions]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        extras["multi_session"] = {"sessions_per_gpu": S, "frames_per_s": round(S * k_ms * args.frames / (time.perf_counter() - t2), 1)}
        del others
        # the integrated path: the same sequence through the drop-in boundary (LpSlamManager -> stereo tracker: upload, extract,
        # stereo + projection matching, one-launch pose optimiser, keyframe every 6th frame with a windowed local BA)
        try:
            from lpslam_amd import manager, _build
            _build.host_library()
            mg = manager.Manager()
            for num in (0, 1):
                c = manager.default_camera()
                c.camera_number = num; c.f_x = wl.k["fx"]; c.f_y = wl.k["fy"]; c.c_x = wl.k["cx"]; c.c_y = wl.k["cy"]
                c.resolution_x = W; c.resolution_y = H; c.focal_x_baseline = wl.k["fxb"]
                mg.set_camera(c)
            mg.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": %d, "numLevels": %d, "keyframeInterval": %d, "device": %d}' % (KPTS, LEVELS, FRAMES_PER_STEP, device))
            mg.collect_results(); mg.provide_odometry()
            mg.start()
            seq_t = wl.synth.StereoSequence(W, H, 4)
            tr_frames = [seq_t.frame(i) for i in range(30)]
            t2 = time.perf_counter()
            for i, (l, r) in enumerate(tr_frames):
                mg.add_stereo((i + 1) * 40_000_000, l, r)
            while len(mg.results) < len(tr_frames) and time.perf_counter() - t2 < 60:
                time.sleep(0.0005)
            t_tr = time.perf_counter() - t2
            st_tr = mg.status()
            mg.stop()
            extras["tracker"] = {"frames": len(mg.results), "valid": int(sum(r["valid"] for r in mg.results)), "frames_per_s": round(len(mg.results) / t_tr, 1),
                                 "last_frame_ms": round(1e3 * st_tr.frame_time, 3), "key_frames": int(st_tr.key_frames)}
        except Exception as e:      # noqa: BLE001 -- an extra must not take the benchmark line down
            extras["tracker"] = {"error": str(e)}
        # the monocular tracker on the same boundary: two-view initialisation, then tracking with triangulated keyframes
        try:
            mg = manager.Manager()
            c = manager.default_camera()
            c.camera_number = 0; c.f_x = wl.k["fx"]; c.f_y = wl.k["fy"]; c.c_x = wl.k["cx"]; c.c_y = wl.k["cy"]; c.resolution_x = W; c.resolution_y = H
            mg.set_camera(c)
            mg.add_tracker("VSLAMMono", '{"cameraSetup": "monocular", "slamKeypoints": %d, "numLevels": 3, "keyframeInterval": %d, "device": %d}' % (KPTS, FRAMES_PER_STEP, device))
            mg.collect_results(); mg.provide_odometry()
            mg.start()
            walls = wl.synth.WallSequence(W, H, 4, step=0.05)
            mono_frames = [walls.frame(i) for i in range(30)]
            t2 = time.perf_counter()
            for i, img in enumerate(mono_frames):
                mg.add_image((i + 1) * 40_000_000, img)
            while len(mg.results) < len(mono_frames) and time.perf_counter() - t2 < 60:
                time.sleep(0.0005)
            t_tr = time.perf_counter() - t2
            st_tr = mg.status()
            mg.stop()
            extras["tracker_mono"] = {"frames": len(mg.results), "valid": int(sum(r["valid"] for r in mg.results)),
                                      "frames_per_s": round(len(mg.results) / t_tr, 1), "key_frames": int(st_tr.key_frames), "landmarks": int(st_tr.feature_points)}
        except Exception as e:      # noqa: BLE001
            extras["tracker_mono"] = {"error": str(e)}
        # BASELINE configs[4]'s global BA on ONE GPU (all landmarks on this rank; the partitioned solve adds one all-reduce of
        # the 1200^2 reduced system per trial): 200 keyframes, 30 k landmarks, ~240 k observations, 10 LM iterations
        try:
            gprob = wl.synth.ba_problem(200, 30000, 240000, 1920, 1080, seq_id=2, kf_stride=2)
            gba = wl.hip.BundleAdjuster(wl.ctx, gprob["poses"], gprob["fixed"], gprob["points"], wl.hip.ba_obs_array(gprob), gprob["cam"])
            gba.optimize(True, 2); gba.reset()
            t2 = time.perf_counter(); glog2 = gba.optimize(True, BA_ITERS); t_g = time.perf_counter() - t2
            extras["global_ba"] = {"keyframes": 200, "landmarks": 30000, "observations": int(gba.n_obs), "ms_per_iter": round(1e3 * t_g / max(len(glog2), 1), 4),
                                   "chi2_first": float(glog2["chi2_before"][0]), "chi2_last": float(glog2["chi2_after"][-1])}
            gba.close()
        except Exception as e:      # noqa: BLE001
            extras["global_ba"] = {"error": str(e)}
        pg = wl.synth.pose_graph_problem(200, 0)
        graph = wl.hip.PoseGraph(wl.ctx, pg["verts"], pg["fixed"], wl.hip.sim3_edges(pg["edge_i"], pg["edge_j"], pg["meas"]), True)
        graph.optimize(2)
        t2 = time.perf_counter(); glog = graph.optimize(10); t_pg = time.perf_counter() - t2
        extras["pose_graph"] = {"keyframes": 200, "edges": int(len(pg["edge_i"])), "ms_per_iter": round(1e3 * t_pg / max(len(glog), 1), 4)}
        graph.close()

    if rank == 0:
        frames_total = world * args.frames * args.steps
        value = frames_total / elapsed
        ab = wl.algorithmic_bytes()
        dom = max(ab.keys(), key=lambda k_: stage_ms[STAGE_NAMES.index(k_)])
        dom_ms = stage_ms[STAGE_NAMES.index(dom)]
        launches = {"pyramid": 1, "fast": 1, "describe": 1, "distribute": 1}[dom]
        achieved = ab[dom] / (dom_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel group from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs of this command, condensed by tools/pmc_summary.py into profiles/<round>_pmc.json: KB per launch as
        # rocprofv3 reports them, no x2 correction for these dword / dwordx2 loads -- profiles/r01b_pmc_hbm.json calibrates that);
        # per step = per launch x launches of the group; only quoted for the launch shape it was measured on
        traffic = None
        try:
            pmc_files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json"))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_files[-1])))["kernels"]
            kname = {"pyramid": "k_pyr_bands", "fast": "k_fast_cells", "describe": "k_describe", "distribute": "k_distribute"}[dom]
            if args.frames == FRAMES_PER_STEP:
                traffic = int(round(launches * 1024.0 * (pmc[kname]["FETCH_SIZE"]["mean_per_launch"] + pmc[kname]["WRITE_SIZE"]["mean_per_launch"])))
        except (OSError, KeyError, ValueError, IndexError):
            pass
        out = {
            "metric": "frames/sec (ORB+match+local-BA), 1280x720 stereo",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 front end / f64 BA", "data": "synthetic",
            "config": {"workload": "configs[1]+configs[2]: 1280x720 stereo, 2000 kpts, 8 levels; step = %d stereo frames "
                                   "(extract L+R, stereo match, 2000x2000 BF temporal match) + one 50-KF/5k-landmark/%d-obs "
                                   "local BA of %d LM iterations" % (args.frames, wl.ba.n_obs if wl.ba else 0, BA_ITERS),
                       "frames_per_step": args.frames, "replicas": world, "parallelism": "replicas x%d" % world},
            "ba_ms_per_iter": round(stage_ms[T_BA] / BA_ITERS, 4) if wl.ba is not None else None,
            "roofline": {"bound": "hbm", "kernel": dom, "launches_per_step": launches,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_step": int(ab[dom]), "avg_ms_per_step": round(float(dom_ms), 4)},
            "stage_ms_per_step": {n: round(float(v), 4) for n, v in zip(STAGE_NAMES, stage_ms)},
            # SURVEY.md 8(d) whole-extraction figure: B_img = pyramid + FAST + blur + patches + outputs per image, over the
            # summed time of the four extraction kernels (the blur's 2P bytes are part of B_img although it is fused away here)
            "front_end_roofline": (lambda b_img, t_ms: {"algorithmic_bytes_per_step": int(b_img), "extract_ms_per_step": round(float(t_ms), 4),
                                                         "achieved": round(b_img / (t_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                                                         "frac": round(b_img / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})(
                wl.extract_bytes(), stage_ms[T_PYR] + stage_ms[T_FAST] + stage_ms[T_DIST] + stage_ms[T_DESC]),
            "pcie_inclusive_frames_per_s": round(pcie_fps, 2),
        }
        out.update(extras)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 2)
            if _CPU_MT:
                out["cpu_baseline_threads"] = _CPU_MT
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
