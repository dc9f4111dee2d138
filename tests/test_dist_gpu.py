"""GPU tests of the multi-process paths on a single MI355X: two ranks share the GPU and talk over gloo (127.0.0.1)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _torchrun(args, env_extra, timeout=150):
    env = dict(os.environ); env.update(env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _write_problem(path, hiplib, prob, robust=1):
    """flat binary problem file of the C++ test drivers (tests/cpp/*.cpp)"""
    obs = hiplib.ba_obs_array(prob)
    cam = prob["cam"]
    with open(path, "wb") as f:
        f.write(np.array([len(prob["poses"]), len(prob["points"]), len(obs), robust], np.int32).tobytes())
        f.write(np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fxb"], np.sqrt(5.991), np.sqrt(7.815)], np.float64).tobytes())
        f.write(np.ascontiguousarray(prob["poses"], np.float64).tobytes())
        fx = np.zeros((len(prob["poses"]) + 7) // 8 * 8, np.uint8); fx[:len(prob["poses"])] = prob["fixed"]
        f.write(fx.tobytes())
        f.write(np.ascontiguousarray(prob["points"], np.float64).tobytes())
        f.write(obs.tobytes())


def _build_cxx(tmp_path, name, source, extra=()):
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not present on this box")
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "lpslam_amd")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", source), "-L" + libdir, "-llpslam_hip"] + list(extra) +
                       ["-lpthread", "-Wl,-rpath," + libdir + ":/opt/rocm/lib"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


_SHARED_CASES = {
    # name: (generator arguments, ranks, iterations)
    "12kf": (dict(n_kf=12, n_pts=600, n_obs=4000, w=640, h=480, seq_id=51), 2, 8),
    "12kf_3ranks": (dict(n_kf=12, n_pts=600, n_obs=4000, w=640, h=480, seq_id=51), 3, 8),
    "rejected_trials": (dict(n_kf=6, n_pts=150, n_obs=800, w=640, h=480, seq_id=46, pose_noise=(0.5, 3.0), point_noise=3.0), 2, 8),
    "40kf": (dict(n_kf=40, n_pts=3000, n_obs=20000, w=1280, h=720, seq_id=11), 2, 8),
    "global_200kf": (dict(n_kf=200, n_pts=30000, n_obs=240000, w=1920, h=1080, seq_id=2, kf_stride=2), 2, 10),
    # contiguous tracks: the shards' reduced systems are block-banded (sparse); the partitioned driver all-reduces the dense buffer, so
    # a shard that qualifies for the band path (40 keyframes) is switched to the dense solver by the driver, and the unpartitioned
    # solve it is compared with below takes the band path
    "contiguous_40kf": (dict(n_kf=40, n_pts=3000, n_obs=20000, w=1280, h=720, seq_id=11, tracks="contiguous", top_up=True), 2, 8),
    "contiguous_global_200kf": (dict(n_kf=200, n_pts=30000, n_obs=240000, w=1920, h=1080, seq_id=2, kf_stride=2, tracks="contiguous"), 3, 10),
}


@pytest.fixture(scope="module")
def shared_device_exe(tmp_path_factory):
    return _build_cxx(tmp_path_factory.mktemp("cxx"), "partitioned_shared_device", "partitioned_shared_device_main.cpp")


@pytest.mark.parametrize("case", list(_SHARED_CASES))
def test_partitioned_driver_two_ranks_on_one_device(hiplib, oracle, tmp_path, shared_device_exe, case):
    """lpslam_hip_ba_optimize_partitioned_with -- the C++ driver behind lpslam_hip_ba_optimize_partitioned (RCCL is one callback of
    it) -- with R >= 2 ranks on the one GPU: host threads, one context / BA stream each, a stream-ordered in-process all-reduce
    (tests/cpp/partitioned_shared_device_main.cpp).  With one rank every all-reduce is the identity; here the in-place reduced
    tail (rhs | b_p | diag H_pp | chi2), the packed triangle, the lambda_0 max-reduce and the device-side accept / reject see real
    sums.  The chi2 trajectory, the trial counts and lambda must follow the oracle, the ranks must agree bit for bit."""
    from lpslam_amd import synth
    g, ranks, iters = _SHARED_CASES[case]
    g = dict(g)
    prob = synth.ba_problem(g.pop("n_kf"), g.pop("n_pts"), g.pop("n_obs"), g.pop("w"), g.pop("h"), **g)
    _write_problem(tmp_path / "problem.bin", hiplib, prob)
    r = subprocess.run([shared_device_exe, str(tmp_path / "problem.bin"), str(tmp_path / "result.bin"), str(ranks), str(iters)], capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    raw = open(tmp_path / "result.bin", "rb").read()
    n_ranks, done, calls = np.frombuffer(raw[:12], np.int32)
    assert n_ranks == ranks
    isz, off = hiplib.BA_LOG_DTYPE.itemsize, 12
    logs, poses = [], []
    for _ in range(ranks):
        logs.append(np.frombuffer(raw[off:off + done * isz], hiplib.BA_LOG_DTYPE)); off += done * isz
        poses.append(np.frombuffer(raw[off:off + len(prob["poses"]) * 56], np.float64).reshape(-1, 7)); off += len(prob["poses"]) * 56
    points = np.frombuffer(raw[off:], np.float64).reshape(-1, 3)
    op, ox, olog = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], oracle.ba_obs(prob), prob["cam"], True, iters)
    assert done == len(olog)
    if case == "rejected_trials":
        assert olog["trials"].max() > 1, "the case is meant to contain rejected trials"
    # per trial: packed system + (trial chi2, scale); the first trial adds the diagonal SUM and the MAX
    assert calls == 2 + 2 * int(olog["trials"].sum())
    for rk in range(ranks):
        assert np.array_equal(logs[rk]["trials"], olog["trials"])
        assert np.allclose(logs[rk]["chi2_before"], olog["chi2_before"], rtol=1e-9) and np.allclose(logs[rk]["chi2_after"], olog["chi2_after"], rtol=1e-9)
        assert np.allclose(logs[rk]["lambda"], olog["lambda"], rtol=1e-6)
        assert logs[rk].tobytes() == logs[0].tobytes() and np.array_equal(poses[rk], poses[0])      # identical decisions, identical reduced systems
        dq = 2 * np.arccos(np.clip(np.abs(np.sum(poses[rk][:, :4] * op[:, :4], axis=1)), 0, 1))
        assert dq.max() < 1e-4 and np.abs(poses[rk][:, 4:] - op[:, 4:]).max() < 1e-3
    assert np.abs(points - ox).max() < 1e-3
    # and the unpartitioned fused solve on the same GPU
    ctx = hiplib.Context(640, 480, 500, 1.2, 4, max_images=1)
    ba = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hiplib.ba_obs_array(prob), prob["cam"])
    slog = ba.optimize(True, iters)
    sp, sx = ba.state()
    assert np.array_equal(slog["trials"], logs[0]["trials"]) and np.allclose(slog["chi2_after"], logs[0]["chi2_after"], rtol=1e-9)
    assert np.abs(sp - poses[0]).max() < 1e-6 and np.abs(sx - points).max() < 1e-6
    ba.close(); ctx.close()


@pytest.mark.parametrize("n_kf,n_pts,n_obs,extra", [(12, 600, 4000, []), (40, 3000, 20000, []),
                                                     (6, 150, 800, ["640", "480", "46", "6", "0.5", "3.0", "3.0"])])      # the last: rejected trials
def test_partitioned_ba_equals_single_gpu(hiplib, tmp_path, n_kf, n_pts, n_obs, extra):
    iters = 6
    r = _torchrun([os.path.join("tests", "_dist_ba_worker.py"), str(tmp_path), str(n_kf), str(n_pts), str(n_obs), str(iters)] + extra,
                  {"LPSLAM_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    single = np.load(tmp_path / "single.npz")
    if extra:
        assert single["trials"].max() > 1, "the case is meant to contain rejected trials"
    ranks = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    for rk in ranks:
        assert int(rk["outer"]) == iters and int(rk["trials"]) == int(single["trials"].sum())
        assert np.isclose(float(rk["chi2"]), single["chi2_after"][-1], rtol=1e-9)
        assert np.isclose(float(rk["lam"]), single["lam"][-1], rtol=1e-6)
        dq = 2 * np.arccos(np.clip(np.abs(np.sum(rk["poses"][:, :4] * single["poses"][:, :4], axis=1)), 0, 1))
        assert dq.max() < 1e-4 and np.abs(rk["poses"][:, 4:] - single["poses"][:, 4:]).max() < 1e-3
        assert np.abs(rk["points"] - single["points"][rk["ids"]]).max() < 1e-3
        assert int(rk["reduces"]) == 2 + 2 * int(rk["trials"])
    assert np.array_equal(ranks[0]["poses"], ranks[1]["poses"])      # identical decisions and identical reduced systems


def test_partitioned_global_ba_full_size(hiplib, oracle, tmp_path):
    """BASELINE configs[4] at its stated size -- 200 keyframes / 30 000 landmarks / ~240 k observations, 10 LM iterations --
    partitioned over two ranks (gloo; both share the one GPU) against the single-GPU solve AND the oracle's chi2 trajectory."""
    from lpslam_amd import synth
    iters = 10
    r = _torchrun([os.path.join("tests", "_dist_ba_worker.py"), str(tmp_path), "200", "30000", "240000", str(iters), "1920", "1080", "2", "2"],
                  {"LPSLAM_DIST_BACKEND": "gloo"}, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    single = np.load(tmp_path / "single.npz")
    ranks = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    prob = synth.ba_problem(200, 30000, 240000, 1920, 1080, seq_id=2, kf_stride=2)
    assert abs(len(prob["obs_pose"]) - 240000) <= 0.05 * 240000
    op, ox, olog = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], oracle.ba_obs(prob), prob["cam"], True, iters)
    assert len(olog) == iters and np.allclose(single["chi2_after"], olog["chi2_after"], rtol=1e-9)
    for rk in ranks:
        assert int(rk["outer"]) == iters and int(rk["trials"]) == int(olog["trials"].sum())
        assert np.allclose(rk["chi2_after"], olog["chi2_after"], rtol=1e-9)          # the oracle's trajectory, iteration by iteration
        assert np.allclose(rk["chi2_after"], single["chi2_after"], rtol=1e-9)
        assert np.isclose(float(rk["lam"]), olog["lambda"][-1], rtol=1e-6)
        dq = 2 * np.arccos(np.clip(np.abs(np.sum(rk["poses"][:, :4] * op[:, :4], axis=1)), 0, 1))
        assert dq.max() < 1e-4 and np.abs(rk["poses"][:, 4:] - op[:, 4:]).max() < 1e-3
        assert np.abs(rk["points"] - ox[rk["ids"]]).max() < 1e-3
    assert np.array_equal(ranks[0]["poses"], ranks[1]["poses"])


def test_bench_two_replicas(hiplib):
    """bench.py under torch.distributed.run with 2 ranks (gloo stands in for RCCL when both ranks share one GPU)."""
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu"], {"LPSLAM_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["replicas"] == 2
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0


def test_bench_launches_its_own_ranks(hiplib):
    """plain `python bench.py --gpus 2` (no launcher around it): bench.py starts the two ranks itself as a child torch.distributed.run
    and relays the line and the exit code.  Two ranks on this box's one GPU need the gloo backend; with RCCL the same command must
    refuse (one rank per GPU) with a non-zero status instead of quietly running one rank."""
    env = dict(os.environ); env.update({"LPSLAM_BENCH_BACKEND": "gloo"})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["replicas"] == 2 and d["value"] > 0
    if hiplib.device_count() < 2:
        env.pop("LPSLAM_BENCH_BACKEND")
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-extras"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "only 1 HIP device" in (r.stdout + r.stderr)


def test_partitioned_driver_in_cxx_with_rccl_single_rank(hiplib, oracle):
    """lpslam_hip_ba_optimize_partitioned on a one-rank RCCL communicator (this box has one GPU; more ranks need more devices):
    the device-driven chain -- linearise, packed-triangle all-reduce on the problem's stream, redundant factorisation, 2-double
    all-reduce, decision -- must follow the oracle and agree with the fused single-GPU solve, rejected trials included."""
    from lpslam_amd import synth
    try:
        uid = hiplib.RcclComm.unique_id()
        comm = hiplib.RcclComm(uid, 1, 0)
    except hiplib.LpslamHipError as e:
        pytest.skip("RCCL unavailable: %s" % e)
    ctx = hiplib.Context(640, 480, 500, 1.2, 4, max_images=1)
    cases = [synth.ba_problem(12, 600, 4000, 640, 480, seq_id=51), synth.ba_problem(40, 3000, 20000, 1280, 720, seq_id=11),
             synth.ba_problem(6, 150, 800, 640, 480, seq_id=46, pose_noise=(0.5, 3.0), point_noise=3.0)]
    for prob in cases:
        iters = 8
        obs = hiplib.ba_obs_array(prob)
        ba = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"])
        log = ba.optimize_partitioned(comm, True, iters)
        gp, gx = ba.state()
        op, ox, olog = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], oracle.ba_obs(prob), prob["cam"], True, iters)
        assert len(log) == len(olog) and np.allclose(log["chi2_after"], olog["chi2_after"], rtol=1e-9) and np.array_equal(log["trials"], olog["trials"])
        assert np.allclose(log["lambda"], olog["lambda"], rtol=1e-6)
        dq = 2 * np.arccos(np.clip(np.abs(np.sum(gp[:, :4] * op[:, :4], axis=1)), 0, 1))
        assert dq.max() < 1e-4 and np.abs(gp[:, 4:] - op[:, 4:]).max() < 1e-3 and np.abs(gx - ox).max() < 1e-3
        ba.close()
    comm.close()
    ctx.close()


def test_partitioned_driver_cxx_binary_one_rank_per_device(hiplib, oracle, tmp_path):
    """tests/cpp/partitioned_rccl_main.cpp: a C++ host (hipcc, links RCCL) that runs lpslam_hip_ba_optimize_partitioned with one
    rank per visible device in one process (ncclCommInitAll, a host thread per rank).  With several devices the ranks must return
    identical poses and follow the oracle.  On a one-GPU box RCCL forms ONE rank, whose all-reduces are identities: the run is still
    checked (binding, launch chain) but the test then reports SKIPPED, so the records show that RCCL with N > 1 was not exercised
    (the multi-rank driver logic is covered by test_partitioned_driver_two_ranks_on_one_device)."""
    from lpslam_amd import synth
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("RCCL headers not present on this box")
    exe = _build_cxx(tmp_path, "partitioned_rccl", "partitioned_rccl_main.cpp", ["-L/opt/rocm/lib", "-lrccl"])
    prob = synth.ba_problem(40, 3000, 20000, 1280, 720, seq_id=11)
    iters = 8
    _write_problem(tmp_path / "problem.bin", hiplib, prob)
    r = subprocess.run([exe, str(tmp_path / "problem.bin"), str(tmp_path / "result.bin"), "8", str(iters)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    raw = open(tmp_path / "result.bin", "rb").read()
    ranks, done = np.frombuffer(raw[:8], np.int32)
    log = np.frombuffer(raw[8:8 + done * hiplib.BA_LOG_DTYPE.itemsize], hiplib.BA_LOG_DTYPE)
    poses = np.frombuffer(raw[8 + done * hiplib.BA_LOG_DTYPE.itemsize:], np.float64).reshape(ranks, -1, 7)
    op, ox, olog = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], oracle.ba_obs(prob), prob["cam"], True, iters)
    assert ranks >= 1 and done == len(olog) and np.allclose(log["chi2_after"], olog["chi2_after"], rtol=1e-9) and np.array_equal(log["trials"], olog["trials"])
    for rk in range(ranks):
        dq = 2 * np.arccos(np.clip(np.abs(np.sum(poses[rk][:, :4] * op[:, :4], axis=1)), 0, 1))
        assert dq.max() < 1e-4 and np.abs(poses[rk][:, 4:] - op[:, 4:]).max() < 1e-3
    if ranks == 1:
        pytest.skip("one device: the RCCL communicator has one rank, its all-reduces reduce nothing (checks above passed)")


def test_bench_distributed_path_with_rccl_one_rank(hiplib):
    """The N > 1 code path of bench.py on backend nccl (= RCCL) with the one rank this box can host: process group, timing
    reduction, and the partitioned global BA through the C++ RCCL driver (communicator id handed out over torch.distributed)."""
    env = dict(os.environ); env.update({"LPSLAM_BENCH_FORCE_DIST": "1"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    g = d["global_ba_partitioned"]
    assert "error" not in g, g           # the watchdog of that section also ends the run with a non-zero status
    assert g["ranks"] == 1 and g["iterations"] == 10 and g["chi2_last"] < 0.1 * g["chi2_first"] and g["ms_per_iter"] > 0
