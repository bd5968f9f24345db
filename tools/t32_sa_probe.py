import sys, time, numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S
g = S.GCY()
for n in (16, 20):
    shp = (n,) * 6
    T = S.gcy_operator(shp, g.params, S.discretize_gcy(g, shp))
    w0 = np.full(shp, 800.0)
    for tol in (1e-4, 1e-6, 1e-8):
        res = {}
        for t32 in (0, 1):
            best = None
            for rep in range(2):
                t0 = time.perf_counter(); x, it, info = T.solve(w0, "successive_approx", tol=tol, t_f32=t32); dt = time.perf_counter() - t0
                if best is None or dt < best[0]: best = (dt, x, it, info)
            res[t32] = best
        d = float(np.max(np.abs(res[0][1] - res[1][1])))
        print(f"GCY {n}^6 SA tol {tol:.0e}: fp64 {res[0][2]} it {res[0][0]:.3f} s | fp32 intermediates {res[1][2]} it {res[1][0]:.3f} s | max|x32 - x64| {d:.2e}  final steps {res[0][3]['final_err']:.2e} / {res[1][3]['final_err']:.2e}", flush=True)
    T.close()
