"""Per-kernel times of successive approximation with fp64 and with fp32 intermediates (opts.t_f32) at GCY 20^6, and the
device-resident time to 1e-8 of both."""
import sys, time, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdfs_via_autodiff_amd as S
g = S.GCY(); shp=(20,)*6
T = S.gcy_operator(shp, g.params, S.discretize_gcy(g, shp))
op = T.op if hasattr(T, "op") else T
w = torch.full(shp, 800.0, dtype=torch.float64, device="cuda")
for t32 in (0, 1):
    op.set_profiling(True); op.reset_counters()
    x = w.clone()
    n, info = op.solve_dev(x.data_ptr(), "successive_approx", tol=1e-8, max_iter=120, t_f32=t32)
    print("t_f32", t32, "iterations", n)
    for c in op.counters():
        if c["launches"]:
            print(f"   {c['name']:52s} {c['launches']:5d} {c['total_ms']/c['launches']*1e3:8.1f} us")
    op.set_profiling(False)
    for rep in range(2):
        x = w.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        n, info = op.solve_dev(x.data_ptr(), "successive_approx", tol=1e-8, t_f32=t32); torch.cuda.synchronize()
        print(f"   SA to 1e-8: {n} iterations, {time.perf_counter()-t0:.4f} s")
