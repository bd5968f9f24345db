import os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
def log(*a):
    print(f"[r{rank} {time.time()%1000:.2f}]", *a, flush=True)
faulthandler.dump_traceback_later(40, exit=True)
dist.init_process_group("gloo")
log("pg up")
torch.cuda.set_device(0)
import sdfs_via_autodiff_amd as S
from sdfs_via_autodiff_amd import distributed as D
shapes = (5, 7, 6, 4)
m = S.SSY(); arr = S.discretize_ssy(m, shapes)
op = D.ShardedKoopmans("ssy", shapes, m.params, arr)
log("op created", op.a_sizes, op.b_sizes, "\n" + op.backend.describe_plan())
w = 400 + 500 * np.random.default_rng(0).random(shapes)
w_loc = op.scatter_from_full(torch.from_numpy(w)).cuda()
log("w_loc", tuple(w_loc.shape))
y = op.backend.run(0, 0, w_loc); torch.cuda.synchronize(); log("stage0 done", float(y.sum()))
z = op.a_to_b(y); torch.cuda.synchronize(); log("a_to_b done", tuple(z.shape))
t = op.backend.run(1, 0, z); torch.cuda.synchronize(); log("stage1 done", float(t.sum()))
Tw = op.b_to_a(t); log("b_to_a done")
full = op.gather_full(Tw); log("gather done")
T1 = S.ssy_operator(shapes, m.params, arr)
ref = T1(w)
log("max rel err", float(np.max(np.abs(full.cpu().numpy() - ref) / ref)))
dist.destroy_process_group()
