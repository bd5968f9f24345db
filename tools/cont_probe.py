"""GPU exploration: which small GCY continuous configurations have a fixed point, and timings."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S
gcy = S.GCY()
for nsd, d, sizes in [(3.2, 3, (3, 3, 3, 3, 4, 4)), (2.0, 2, (3, 3, 3, 3, 4, 4)), (2.0, 3, (3, 3, 3, 3, 4, 4)),
                      (1.5, 2, (3, 3, 3, 3, 4, 4)), (3.2, 3, (4, 4, 4, 4, 6, 6)), (3.2, 5, (4, 4, 4, 4, 6, 6))]:
    grids = S.build_grid(gcy, *sizes, nsd)
    nodes, weights = S.qnwnorm([d] * 6)
    T = S.T_fun_factory((np.array(gcy.params), grids, nodes.T.copy(), weights), "quadrature", int(np.prod(sizes)))
    t0 = time.time()
    x, n, info = T.solve(np.ones(sizes), "successive_approx", tol=1e-6, max_iter=30000)
    print("gcy", nsd, d, sizes, "iters", n, "err", info["final_err"], "range", x.min(), x.max(), f"{time.time()-t0:.2f}s", flush=True)
ssy = S.SSY()
for sizes, d in [((10, 10, 10, 20), 5), ((15, 15, 15, 15), 5)]:
    grids = S.build_grid(ssy, *sizes)
    nodes, weights = S.qnwnorm([d] * 4)
    T = S.T_fun_factory((np.array(ssy.params), grids, nodes.T.copy(), weights), "quadrature", int(np.prod(sizes)))
    w = np.ones(sizes)
    T(w); t0 = time.time()
    for _ in range(20): T(w)
    print("ssy apply (host round trip)", sizes, d, (time.time() - t0) / 20 * 1e3, "ms")
    for algo, kw in (("successive_approx", dict(tol=1e-5)), ("newton", dict(tol=1e-7, inner_rtol=1e-6, inner_atol=0.0))):
        t0 = time.time()
        x, n, info = T.solve(w, algo, **kw)
        print("ssy", sizes, algo, "iters", n, "applies", info["n_apply"], "err", info["final_err"], f"{time.time()-t0:.3f}s",
              x.min(), x.max(), flush=True)
