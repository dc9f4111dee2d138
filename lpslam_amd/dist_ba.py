"""Landmark-partitioned bundle adjustment over several GPUs (BASELINE configs[4]: global BA with shared poses).

Every rank holds all keyframe poses and the observations of its own landmarks (round-robin by landmark id).  One
Levenberg-Marquardt trial = local linearisation + partial Schur complement, ONE sum all-reduce of the contiguous
reduced system [S | rhs | b_p | diag H_pp | chi2], a redundant factorisation on every rank, local landmark back
substitution, and a 2-double sum all-reduce of (trial chi2, landmark scale term).  The accept / reject decision runs on
every rank's device on identical inputs, so the ranks stay in lock step without a broadcast.

The PRODUCT path for this solve is C++: lpslam_hip_ba_optimize_partitioned (csrc/ba.hip) takes an ncclComm_t, all-reduces the
packed lower triangle on the problem's own stream and drives the trials from the device control block -- bench.py and a host
application use that (hip.BundleAdjuster.optimize_partitioned, tests/cpp/partitioned_rccl_main.cpp).  This module is the
REHEARSAL of the same partition through the step-wise entry points with torch.distributed as the transport (backend "gloo":
CPU tests with world size 2, or two ranks sharing one GPU, buffers staged through host memory; its per-step host
synchronisations are why it is not the product path) -- and shard_problem / merge helpers that both paths share.
"""
import numpy as np


def shard_problem(prob, rank, world):
    """Sub-problem of `rank`: landmarks j with j % world == rank (renumbered), their observations, all poses."""
    mine = np.nonzero(np.arange(len(prob["points"])) % world == rank)[0]
    remap = -np.ones(len(prob["points"]), np.int64)
    remap[mine] = np.arange(len(mine))
    keep = remap[prob["obs_point"]] >= 0
    out = dict(prob)
    out["points"] = prob["points"][mine]
    out["obs_pose"] = prob["obs_pose"][keep]
    out["obs_point"] = remap[prob["obs_point"][keep]].astype(np.int32)
    out["obs_uvr"] = prob["obs_uvr"][keep]
    out["obs_inv_sigma2"] = prob["obs_inv_sigma2"][keep]
    out["landmark_ids"] = mine
    out["obs_index"] = np.nonzero(keep)[0]
    return out


class _DeviceArray:
    """Exposes a raw device pointer of float64 values to torch (zero copy) via __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchReducer:
    """all-reduce of device buffers through torch.distributed (nccl: in place on the device; gloo: staged through the host)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.device_collectives = dist.get_backend(group) == "nccl"

    def tensor(self, ptr, n):
        return self.torch.as_tensor(_DeviceArray(ptr, n), device="cuda")

    def all_reduce(self, t, op="sum"):
        o = self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX
        if self.device_collectives:
            self.dist.all_reduce(t, op=o, group=self.group)
            self.torch.cuda.synchronize()
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=o, group=self.group)
            t.copy_(h)
            self.torch.cuda.synchronize()


class PartitionedBA:
    """Drives lpslam_hip_ba_step_* on this rank's shard; `ba` is a lpslam_amd.hip.BundleAdjuster (or a test double with
    the same phase methods and numpy-backed buffers), `reducer` provides tensor(ptr, n) and all_reduce(t, op)."""

    def __init__(self, ba, reducer):
        self.ba, self.red = ba, reducer
        p, n = ba.reduced_buffer()
        self.t_red = reducer.tensor(p, n)
        p, n = ba.scalar_buffer()
        self.t_scal = reducer.tensor(p, n)
        self.all_reduce_calls = 0

    def _sum(self, t):
        self.red.all_reduce(t, "sum"); self.all_reduce_calls += 1

    def optimize(self, robust=True, iters=10):
        ba = self.ba
        ba.step_begin(robust, True)                      # linearise only
        self._sum(self.t_red)                            # diag H_pp, b_p, chi2 of every rank
        self.red.all_reduce(self.t_scal[4:5], "max"); self.all_reduce_calls += 1     # max diag H_ll
        ba.step_lambda0()
        outer, trials, chi_log = 0, 0, []
        while outer < iters:
            ba.step_begin(robust, False)                 # (re-linearise if the state moved) + partial Schur complement
            self._sum(self.t_red)
            ba.step_solve()
            self._sum(self.t_scal[1:3])                  # trial chi2, landmark part of the gain-ratio denominator
            accepted, finished = ba.step_end()
            trials += 1
            if finished:
                outer += 1
                st = ba.status()
                chi_log.append(st["chi2"])               # chi2 after this outer iteration (the g2o log's chi2_after)
                if st["stopped"]:
                    break
        return dict(outer=outer, trials=trials, chi2_after=np.array(chi_log), **ba.status())
