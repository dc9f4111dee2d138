"""Developer diagnostic for the partitioned BA (run under torch.distributed.run with 2 ranks on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
from lpslam_amd import hip, synth
from lpslam_amd.dist_ba import TorchReducer, shard_problem

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
prob = synth.ba_problem(12, 600, 4000, 1280, 720, seq_id=11)
shard = shard_problem(prob, rank, world)
ctx = hip.Context(640, 480, 500, 1.2, 4, max_images=1)
ba = hip.BundleAdjuster(ctx, shard["poses"], shard["fixed"], shard["points"], hip.ba_obs_array(shard), shard["cam"])
red = TorchReducer()
p, n = ba.reduced_buffer(); t_red = red.tensor(p, n)
p, n = ba.scalar_buffer(); t_scal = red.tensor(p, n)
log = open("gpurun_out/dist_rank%d.log" % rank, "w")
def P(*a):
    print(*a, file=log); log.flush()
dimp = int(round((-3 + (9 + 4 * (len(t_red) - 8)) ** 0.5) / 2))
P("dim_pad", dimp, "red", len(t_red))
ba.step_begin(True, True)
P("after lin: chi_loc", float(t_red[dimp * dimp + 3 * dimp]), "scal", t_scal.cpu().numpy())
red.all_reduce(t_red, "sum"); red.all_reduce(t_scal[4:5], "max")
P("after reduce: chi", float(t_red[dimp * dimp + 3 * dimp]), "scal4", float(t_scal[4]))
ba.step_lambda0()
P("status", ba.status())
for trial in range(14):
    ba.step_begin(True, False)
    S = t_red[:dimp * dimp].cpu().numpy().reshape(dimp, dimp)
    P("trial", trial, "S partial diag[:3]", np.diag(S)[:3], "rhs[:3]", t_red[dimp * dimp:dimp * dimp + 3].cpu().numpy())
    red.all_reduce(t_red, "sum")
    ba.step_solve()
    P("  scal after solve", t_scal.cpu().numpy())
    red.all_reduce(t_scal[1:3], "sum")
    a, f = ba.step_end()
    P("  accepted", a, "finished", f, ba.status())
    if ba.status()["stopped"]:
        break
dist.barrier()
