"""Sanity run of the fused Newton / Anderson loops on pair-plan extents other than 16 and 20: against the generic tiles."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdfs_via_autodiff_amd as S
for shapes in ((24, 24, 24, 24, 16, 16), (32, 32, 16, 16, 24, 24)):
    m = S.GCY(); arr = S.discretize_gcy(m, shapes)
    T = S.KoopmansOperator("gcy", shapes, m.params, arr)
    os.environ["SDFS_PLAN"] = "classic"
    Tc = S.KoopmansOperator("gcy", shapes, m.params, arr)
    del os.environ["SDFS_PLAN"]
    w0 = np.full(shapes, 800.0)
    res = {}
    for name, op, kw in (("pair fp64", T, {}), ("classic fp64", Tc, {}), ("pair f32m", T, dict(krylov_f32=3)), ("pair anderson rel", T, None)):
        if kw is None:
            x, n, info = op.solve(w0, "anderson", tol=1e-8, max_iter=3000, ridge=-1e-6)
        else:
            x, n, info = op.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, **kw)
        res[name] = x
        print(shapes, name, n, info["n_apply"], info["status"], info["final_err"], flush=True)
    ref = res["classic fp64"]
    for k, x in res.items():
        print("   ", k, "max|x - classic|", float(np.max(np.abs(x - ref))))
    T.close(); Tc.close()
