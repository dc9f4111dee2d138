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
