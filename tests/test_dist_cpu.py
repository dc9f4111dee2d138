"""world_size-2 gloo tests (CPU) of the multi-GPU path: landmark sharding and the partitioned Levenberg-Marquardt
protocol of lpslam_amd.dist_ba, exercised with the numpy phase double of tests/_fake_ba.py and checked against the
unpartitioned oracle.  Rendezvous on 127.0.0.1."""
import os
import socket

import numpy as np
import pytest

from lpslam_amd import synth
from lpslam_amd.dist_ba import shard_problem


def test_shards_cover_every_landmark_and_observation_once():
    prob = synth.ba_problem(5, 90, 400, 640, 480, seq_id=1)
    for world in (1, 2, 3, 8):
        ids, obs = [], []
        for r in range(world):
            s = shard_problem(prob, r, world)
            ids.append(s["landmark_ids"]); obs.append(s["obs_index"])
            assert len(s["poses"]) == len(prob["poses"])                        # poses are replicated
            assert (s["obs_point"] >= 0).all() and s["obs_point"].max() < len(s["points"])
            assert np.array_equal(s["points"], prob["points"][s["landmark_ids"]])
            assert np.array_equal(s["landmark_ids"][s["obs_point"]], prob["obs_point"][s["obs_index"]])
        assert sorted(np.concatenate(ids).tolist()) == list(range(len(prob["points"])))
        assert sorted(np.concatenate(obs).tolist()) == list(range(len(prob["obs_pose"])))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, iters, out):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _fake_ba import FakeStepBA, NumpyReducer
    from lpslam_amd.dist_ba import PartitionedBA
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synth.ba_problem(4, 40, 150, 640, 480, seq_id=3)
    shard = shard_problem(prob, rank, world)
    ba = FakeStepBA(shard)
    drv = PartitionedBA(ba, NumpyReducer())
    res = drv.optimize(True, iters)
    poses, pts = ba.state()
    np.savez(out % rank, poses=poses, points=pts, ids=shard["landmark_ids"], chi2=res["chi2"], lam=res["lam"], outer=res["outer"],
             trials=res["trials"], calls=len(ba.calls), reduces=drv.all_reduce_calls)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_partitioned_lm_matches_the_unpartitioned_oracle(oracle, tmp_path, world):
    import torch.multiprocessing as mp
    iters = 6
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), iters, out), nprocs=world, join=True)
    prob = synth.ba_problem(4, 40, 150, 640, 480, seq_id=3)
    op, ox, olog = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], oracle.ba_obs(prob), prob["cam"], True, iters)
    res = [np.load(out % r) for r in range(world)]
    for r in res:
        assert int(r["outer"]) == len(olog) == iters
        assert np.isclose(float(r["chi2"]), olog["chi2_after"][-1], rtol=1e-7)          # same LM trajectory as one solver
        assert np.isclose(float(r["lam"]), olog["lambda"][-1], rtol=1e-5)
        assert np.abs(r["poses"] - op).max() < 1e-6                                      # poses replicated and identical
        assert np.abs(r["points"] - ox[r["ids"]]).max() < 1e-6                           # every rank owns its landmarks
        assert int(r["trials"]) == int(olog["trials"].sum())
        # per trial: one all-reduce of the reduced system + one of the two scalars; plus 2 for lambda_0
        assert int(r["reduces"]) == 2 + 2 * int(r["trials"])
    assert np.array_equal(res[0]["poses"], res[1]["poses"])                              # lock step without a broadcast
