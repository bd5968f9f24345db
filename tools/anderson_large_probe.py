"""Anderson on large grids: the batched-Gram device loop (default) against the row-per-pass loop (SDFS_AND_FUSED=0) --
passes, seconds, ms per pass, the fixed points' distance and residuals:
    python tools/anderson_large_probe.py [12|15|16|20 ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S  # noqa: E402

g = S.GCY()
for n in [int(a) for a in sys.argv[1:]] or [12, 16, 20]:
    shp = (n,) * 6
    arr = S.discretize_gcy(g, shp)
    w0 = np.full(shp, 800.0)
    res = {}
    for name, env in (("batched Gram", "1"), ("row per pass", "0")):
        os.environ["SDFS_AND_FUSED"] = env
        T = S.gcy_operator(shp, g.params, arr)
        T.solve(w0, "anderson", tol=0.0, max_iter=40)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            x, it, info = T.solve(w0, "anderson", tol=1e-6, max_iter=5000, record_errors=True)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, x, it, info)
        dt, x, it, info = best
        r = float(np.max(np.abs(T(x) - x)))
        res[name] = x
        print(f"GCY {n}^6 {name:13s}: passes {it:5d} (trace {len(info['errors'])}) status {info['status']} final {info['final_err']:.3e} "
              f"{dt:7.3f} s  {dt / it * 1e3:6.3f} ms/pass  max|Tx - x| {r:.2e}", flush=True)
        T.close()
    print(f"   distance of the two results: {float(np.max(np.abs(res['batched Gram'] - res['row per pass']))):.2e}", flush=True)
