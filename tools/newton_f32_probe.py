"""fp32 Krylov storage (opts.krylov_f32) against fp64 Newton; from w = 800 and from a closer start."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S
g = S.GCY()
for shp in ((16,) * 6, (20,) * 6):
    T = S.gcy_operator(shp, g.params, S.discretize_gcy(g, shp))
    w800 = np.full(shp, 800.0)
    wsa, _, _ = T.solve(w800, "successive_approx", max_iter=60, tol=1e-12)
    for name, w0 in (("w=800", w800), ("60 SA steps", wsa)):
        for f32 in (0, 1):
            t0 = time.perf_counter()
            x, n, info = T.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=f32)
            dt = time.perf_counter() - t0
            print(f"GCY {shp[0]}^6 newton from {name:12s} krylov_f32={f32}: iterations {n:2d} applies {info['n_apply']:4d} "
                  f"{dt:.3f} s  {info['n_apply']/dt:6.0f} applies/s  max|Tw-w| {float(np.max(np.abs(T(x) - x))):.2e}", flush=True)
    T.close()
