import time, numpy as np, sys
sys.path.insert(0, '.')
import sdfs_via_autodiff_amd as S
m = S.SSY(); shp = (15,)*4
T = S.ssy_operator(shp, m.params, S.discretize_ssy(m, shp))
w0 = np.full(shp, 800.0)
for algo, kw in (("newton", dict(tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)), ("newton", dict(tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, use_graph=0)), ("successive_approx", dict(tol=1e-8)), ("anderson", dict(tol=1e-8))):
    T.solve(w0, algo, max_iter=2, **{k: v for k, v in kw.items() if k != "tol"})
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); x, n, info = T.solve(w0, algo, **kw); best = min(best, time.perf_counter() - t0)
    print(algo, kw.get("use_graph", 1), "iters", n, "applies", info["n_apply"], "best %.2f ms" % (best * 1e3), "err", info["final_err"])
