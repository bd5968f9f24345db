"""Time-to-converge versus tolerance (BASELINE.json configs 2-5 axis), fp64, one MI355X.
    python tools/tol_sweep.py > profiles/round1_tolerance_sweep.txt
Device-resident solves from the reference's start w = 800 (wall time includes the upload of w_init
and the download of the result)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S

def run(name, T, shapes, algos, tols):
    w0 = np.full(shapes, 800.0)
    T.solve(w0, "successive_approx", max_iter=8)          # buffers, graph capture
    for algo, kw in algos:
        for tol in tols:
            t0 = time.perf_counter()
            x, n, info = T.solve(w0, algo, tol=tol, **kw)
            dt = time.perf_counter() - t0
            res = float(np.max(np.abs(T(x) - x)))
            print(f"{name:22s} {algo:18s} tol {tol:7.0e}  iterations {n:6d}  applies {info['n_apply']:6d}  "
                  f"{dt:8.3f} s  {info['n_apply'] / dt:9.0f} applies/s  max|Tw-w| {res:9.2e}", flush=True)

tols = (1e-4, 1e-5, 1e-6, 1e-7, 1e-8)
newton = ("newton", dict(inner_rtol=1e-6, inner_atol=0.0))
m = S.SSY(); shp = (15,) * 4
run("SSY 15^4", S.ssy_operator(shp, m.params, S.discretize_ssy(m, shp)), shp,
    [("successive_approx", {}), newton, ("anderson", {})], tols)
g = S.GCY()
for shp in ((16,) * 6, (20,) * 6):
    T = S.gcy_operator(shp, g.params, S.discretize_gcy(g, shp))
    run(f"GCY {shp[0]}^6", T, shp, [newton, ("anderson", {})], tols)
    run(f"GCY {shp[0]}^6", T, shp, [("successive_approx", {})], (1e-4, 1e-8))
    T.close()
