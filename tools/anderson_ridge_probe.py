"""Anderson to the metric's tolerance (1e-8) on the 6-D grids: jaxopt's absolute ridge 1e-6 (the reference's,
code/solvers.py:113) against the opt-in RELATIVE ridge (sdfs_opts.ridge < 0: |ridge| trace(G) / m) and successive
approximation, from w = 800 on the device-resident grid.  The error is Anderson's own, |T x - x|_2 (SA: the sup-norm step).
    python tools/anderson_ridge_probe.py [16|20 ...] > profiles/round4_anderson_ridge.txt"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402
import sdfs_via_autodiff_amd as S  # noqa: E402

g = S.GCY()
for n in [int(a) for a in sys.argv[1:]] or [16, 20]:
    shp = (n,) * 6
    T = S.gcy_operator(shp, g.params, S.discretize_gcy(g, shp))
    ws = torch.full(shp, 800.0, dtype=torch.float64, device="cuda")
    T.solve_dev(ws.data_ptr(), "anderson", tol=0.0, max_iter=40)

    def run(algo, **kw):
        ws.fill_(800.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it, info = T.solve_dev(ws.data_ptr(), algo, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        x = ws.clone()
        out = torch.empty_like(x)
        res = torch.zeros(1, dtype=torch.float64, device="cuda")
        T.apply_dev(x.data_ptr(), out.data_ptr(), res.data_ptr())
        torch.cuda.synchronize()
        return it, dt, info, float(res.item())

    for tol in (1e-6, 1e-8):
        it, dt, info, r = run("successive_approx", tol=tol, max_iter=100000)
        print(f"GCY {n}^6 tol {tol:.0e}  successive approximation      : {it:6d} iterations {dt:7.3f} s  status {info['status']}  sup|Tx - x| {r:.2e}", flush=True)
        for ridge in (1e-6, -1e-6, -1e-8, -1e-10, -1e-12):
            it, dt, info, r = run("anderson", tol=tol, max_iter=6000, ridge=ridge)
            kind = "absolute 1e-6 (reference)" if ridge > 0 else f"relative {-ridge:.0e}         "
            print(f"GCY {n}^6 tol {tol:.0e}  Anderson ridge {kind}: {it:6d} passes     {dt:7.3f} s  status {info['status']}  "
                  f"|Tx - x|_2 {info['final_err']:.2e}  sup|Tx - x| {r:.2e}", flush=True)
    T.close()
