import time, numpy as np, sys
sys.path.insert(0, '.')
import sdfs_via_autodiff_amd as S
m = S.SSY(); shp = (15,)*4
T = S.ssy_operator(shp, m.params, S.discretize_ssy(m, shp))
print(T.describe_plan())
w0 = np.full(shp, 800.0)
T.solve(w0, "successive_approx", max_iter=64)
T.set_profiling(True)
T.reset_counters()
x, n, info = T.solve(w0, "successive_approx", tol=1e-8, max_iter=2000)
for c in T.counters():
    print(c["name"], c["launches"], "avg us %.2f" % (c["total_ms"] / max(c["launches"], 1) * 1e3))
T.set_profiling(False)
for ce in (8, 32, 128, 512):
    t0 = time.perf_counter(); x, n, info = T.solve(w0, "successive_approx", tol=1e-8, check_every=ce); dt = time.perf_counter() - t0
    print("check_every", ce, "iters", n, "%.1f ms" % (dt * 1e3), "%.0f it/s" % (n / dt))
