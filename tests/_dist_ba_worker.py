"""Worker of tests/test_dist_gpu.py: one rank of a landmark-partitioned bundle adjustment on real kernels.
Launched by torch.distributed.run; ranks may share one GPU (gloo backend, host-staged all-reduce)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from lpslam_amd import hip, synth
    from lpslam_amd.dist_ba import PartitionedBA, TorchReducer, shard_problem
    out_dir = sys.argv[1]
    n_kf, n_pts, n_obs, iters = (int(x) for x in sys.argv[2:6])
    # optional: image size, sequence id and keyframe stride of the generator (defaults: the small cases)
    gw, gh, seq_id, kf_stride = (int(x) for x in sys.argv[6:10]) if len(sys.argv) >= 10 else (1280, 720, 11, 1)
    # optional: pose noise (rotation, translation) and landmark noise of the generator (the rejected-trial case)
    noise = dict(pose_noise=(float(sys.argv[10]), float(sys.argv[11])), point_noise=float(sys.argv[12])) if len(sys.argv) >= 13 else {}
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(hip.device_count(), 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(os.environ.get("LPSLAM_DIST_BACKEND", "gloo"))
    prob = synth.ba_problem(n_kf, n_pts, n_obs, gw, gh, seq_id=seq_id, kf_stride=kf_stride, **noise)
    shard = shard_problem(prob, rank, world)
    ctx = hip.Context(640, 480, 500, 1.2, 4, max_images=1, device=dev)
    ba = hip.BundleAdjuster(ctx, shard["poses"], shard["fixed"], shard["points"], hip.ba_obs_array(shard), shard["cam"])
    drv = PartitionedBA(ba, TorchReducer())
    res = drv.optimize(True, iters)
    poses, pts = ba.state()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), poses=poses, points=pts, ids=shard["landmark_ids"], chi2=res["chi2"],
             lam=res["lam"], outer=res["outer"], trials=res["trials"], reduces=drv.all_reduce_calls, chi2_after=res["chi2_after"])
    if rank == 0:       # the same problem on one GPU, unpartitioned
        full = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
        log = full.optimize(True, iters)
        fp, fx = full.state()
        np.savez(os.path.join(out_dir, "single.npz"), poses=fp, points=fx, chi2_after=log["chi2_after"], lam=log["lambda"], trials=log["trials"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
