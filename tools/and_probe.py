import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S
g = S.GCY()
for shp in ((6,) * 6, (10,) * 6, (16,) * 6):
    T = S.gcy_operator(shp, g.params, S.discretize_gcy(g, shp))
    w0 = np.full(shp, 800.0)
    for kw in (dict(), dict(beta=1.0), dict(ridge=1e-6 * float(np.prod(shp)))):
        t0 = time.perf_counter()
        x, n, info = T.solve(w0, "anderson", tol=1e-6, max_iter=5000, **kw)
        print(shp[0], kw, "iters", n, "status", info["status"], "err", info["final_err"], "resid", float(np.max(np.abs(T(x) - x))), f"{time.perf_counter()-t0:.2f}s", flush=True)
    T.close()
