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


@pytest.mark.parametrize("n_kf,n_pts,n_obs", [(12, 600, 4000), (40, 3000, 20000)])
def test_partitioned_ba_equals_single_gpu(hiplib, tmp_path, n_kf, n_pts, n_obs):
    iters = 6
    r = _torchrun([os.path.join("tests", "_dist_ba_worker.py"), str(tmp_path), str(n_kf), str(n_pts), str(n_obs), str(iters)],
                  {"LPSLAM_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    single = np.load(tmp_path / "single.npz")
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
    rank per visible device in one process (ncclCommInitAll, a host thread per rank).  On a one-GPU box that is one rank; with more
    devices the ranks must return identical poses.  Either way the result follows the oracle."""
    import shutil
    from lpslam_amd import synth
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("hipcc / RCCL headers not present on this box")
    exe = str(tmp_path / "partitioned_rccl")
    libdir = os.path.join(ROOT, "lpslam_amd")
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "partitioned_rccl_main.cpp"), "-L" + libdir, "-llpslam_hip",
                        "-L/opt/rocm/lib", "-lrccl", "-lpthread", "-Wl,-rpath," + libdir + ":/opt/rocm/lib"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    prob = synth.ba_problem(40, 3000, 20000, 1280, 720, seq_id=11)
    iters = 8
    obs = hiplib.ba_obs_array(prob)
    cam = prob["cam"]
    with open(tmp_path / "problem.bin", "wb") as f:
        f.write(np.array([len(prob["poses"]), len(prob["points"]), len(obs), 1], np.int32).tobytes())
        f.write(np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["fxb"], np.sqrt(5.991), np.sqrt(7.815)], np.float64).tobytes())
        f.write(np.ascontiguousarray(prob["poses"], np.float64).tobytes())
        fx = np.zeros((len(prob["poses"]) + 7) // 8 * 8, np.uint8); fx[:len(prob["poses"])] = prob["fixed"]
        f.write(fx.tobytes())
        f.write(np.ascontiguousarray(prob["points"], np.float64).tobytes())
        f.write(obs.tobytes())
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
    assert "error" not in g, g
    assert g["ranks"] == 1 and g["iterations"] == 10 and g["chi2_last"] < 0.1 * g["chi2_first"] and g["ms_per_iter"] > 0
