import os, sys, time, faulthandler
faulthandler.enable(); faulthandler.dump_traceback_later(50, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo"); rank = dist.get_rank()
def log(*a): print(f"[r{rank}]", *a, flush=True)
torch.cuda.set_device(0)
from sdfs_via_autodiff_amd import distributed as D
from oracle import models, ssy
shapes = (5, 7, 6, 4)
p = models.ssy_params(); arr = list(ssy.discretize_ssy(p, shapes))
if os.environ.get("PERTURB", "1") == "1":
    rng = np.random.default_rng(7); qq = rng.random(arr[7].shape) + 0.05; arr[7] = qq / qq.sum(axis=-1, keepdims=True)
T = lambda w: ssy.T_ssy_factorised(w, shapes, p, arr); J = lambda w, v: ssy.jvp_ssy(w, v, shapes, p, arr)
op = D.ShardedKoopmans("ssy", shapes, p, arr)
w = 400 + 500 * np.random.default_rng(0).random(shapes); v = np.random.default_rng(1).standard_normal(shapes)
w_loc = op.scatter_from_full(torch.from_numpy(w)).cuda(); v_loc = op.scatter_from_full(torch.from_numpy(v)).cuda()
def relerr(x_loc, ref):
    full = op.gather_full(x_loc).cpu().numpy(); return float(np.max(np.abs(full - ref)) / np.max(np.abs(ref))), bool(np.isnan(full).any())
log("T   ", relerr(op.apply_T(w_loc), T(w)))
log("T   ", relerr(op.apply_T(w_loc), T(w)))
log("Tlin", relerr(op.linearize(w_loc), T(w)))
log("T   ", relerr(op.apply_T(w_loc), T(w)))
log("jvp ", relerr(op.jvp(v_loc), J(w, v)))
log("Tlin", relerr(op.linearize(w_loc), T(w)))
log("jvp ", relerr(op.jvp(v_loc), J(w, v)))
dist.destroy_process_group()
