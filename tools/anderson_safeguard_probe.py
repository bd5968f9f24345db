import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S
def build(model, shapes, host, fused):
    m = S.SSY() if model == "ssy" else S.GCY()
    arr = (S.discretize_ssy if model == "ssy" else S.discretize_gcy)(m, shapes)
    os.environ["SDFS_AND_HOST"] = "1" if host else "0"
    os.environ["SDFS_AND_FUSED"] = "1" if fused else "0"
    return S.KoopmansOperator(model, shapes, m.params, arr)
for model, shapes in (("gcy", (6,) * 6), ("ssy", (15,) * 4), ("ssy", (8,)*4)):
    Tf, Tu, Th = build(model, shapes, False, True), build(model, shapes, False, False), build(model, shapes, True, False)
    for level, beta in ((800.0, 8.0), (5.0, 8.0), (5.0, 30.0), (1.0, 60.0), (3000.0, 40.0), (20000.0, 8.0), (0.5, 8.0)):
        w0 = np.full(shapes, level)
        out = []
        for nm, T in (("fused", Tf), ("unfused", Tu), ("host", Th)):
            x, n, info = T.solve(w0, "anderson", tol=1e-6, max_iter=20000, beta=beta, record_errors=True)
            out.append(f"{nm}: n {n} trace {len(info['errors'])} status {info['status']} final {info['final_err']:.2e} finite {bool(np.all(np.isfinite(x)))} resid {float(np.max(np.abs(T(x)-x))) if np.all(np.isfinite(x)) else -1:.2e}")
        print(model, shapes, level, beta, " | ".join(out), flush=True)
