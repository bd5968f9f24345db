import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S
g = S.GCY(); shp = (20,) * 6
T = S.gcy_operator(shp, g.params, S.discretize_gcy(g, shp))
w800 = np.full(shp, 800.0)
wsa, _, _ = T.solve(w800, "successive_approx", max_iter=60, tol=1e-12)
for f32 in (0, 1):
    T.set_profiling(True); T.reset_counters()
    t0 = time.perf_counter()
    x, n, info = T.solve(wsa, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=f32)
    dt = time.perf_counter() - t0
    print(f"krylov_f32={f32}: {dt:.3f} s, applies {info['n_apply']}")
    tot = 0
    for c in T.counters():
        if c["launches"]:
            print(f"   {c['name']:52s} launches {c['launches']:5d} total {c['total_ms']:8.2f} ms  avg {c['total_ms']/c['launches']:.4f} ms")
            tot += c["total_ms"]
    print(f"   sum of kernels {tot:.1f} ms")
    T.set_profiling(False)
