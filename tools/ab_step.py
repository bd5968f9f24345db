"""A/B of two builds of the package on one box: the GCY 20^6 headline step (T + fused residual, resident in HBM) with the
package found under argv[1] -- per-step time over 200 steps after spin-up, per-kernel HIP-event times of a second loop.
tools/ab_step.sh runs it alternately for the round-3 build (tools/probes/r3pkg, a worktree of the round-3 HEAD) and the
working tree, since boxes differ by more than the changes under test."""
import json, os, sys, time
import numpy as np
pkg = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sys.path.insert(0, pkg)
import torch
import sdfs_via_autodiff_amd as S
assert os.path.abspath(S.__file__).startswith(os.path.abspath(pkg)), (S.__file__, pkg)
g = S.GCY(); shp = (n,) * 6
arr = S.discretize_gcy(g, shp)
op = S.KoopmansOperator("gcy", shp, g.params, arr)
stream = torch.cuda.current_stream()
op.set_stream(stream.cuda_stream)
w = 400 + 500 * np.random.default_rng(0).random(shp)
bufs = [torch.from_numpy(w).cuda(), torch.empty(shp, dtype=torch.float64, device="cuda")]
resid = torch.zeros(1, dtype=torch.float64, device="cuda")
def step(i):
    op.apply_dev(bufs[i & 1].data_ptr(), bufs[(i + 1) & 1].data_ptr(), resid.data_ptr())
for i in range(150):
    step(i)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(200):
        step(i)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
op.set_profiling(True)
for i in range(8):
    step(i)
torch.cuda.synchronize()
op.reset_counters()
for i in range(200):
    step(i)
torch.cuda.synchronize()
ks = [(c["name"], c["total_ms"] / max(c["launches"], 1)) for c in op.counters() if c["launches"]]
print(json.dumps({"pkg": pkg, "grid": n, "ms_per_step": round(best, 4), "kernels_ms": [(k, round(v, 4)) for k, v in ks], "resid": float(resid.item())}), flush=True)
