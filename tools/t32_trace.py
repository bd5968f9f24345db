"""Where do the extra iterations of SA with fp32 intermediates (opts.t_f32) come from?  Error traces of the t_f32 solve and
of the fp64 solve on the shapes of tests/test_hip_pair_plan.py::test_sa_with_fp32_intermediates, the iteration of the
phase switch (step <= 64 * 2^-24 * w / |theta|) and the ratio of consecutive steps around it."""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S

for shapes in [(32, 32, 16, 16), (16, 16, 24, 24), (16,) * 6]:
    model = "gcy" if len(shapes) == 6 else "ssy"
    m = S.GCY() if model == "gcy" else S.SSY()
    arr = (S.discretize_gcy if model == "gcy" else S.discretize_ssy)(m, shapes)
    os.environ["SDFS_PLAN"] = "pair"          # (4-D grids take the small / padded plans by default: opts.t_f32 lives on the pair plan)
    try:
        T = S.KoopmansOperator(model, shapes, m.params, arr)
    finally:
        del os.environ["SDFS_PLAN"]
    w0 = np.full(shapes, 800.0)
    tol = 1e-7
    xa, na, ia = T.solve(w0, "successive_approx", tol=tol, t_f32=1, record_errors=True)
    xb, nb, ib = T.solve(w0, "successive_approx", tol=tol, record_errors=True)
    ea, eb = np.array(ia["errors"]), np.array(ib["errors"])
    theta = T.theta if hasattr(T, "theta") else None
    sw = 64 * 2.0 ** -24 * 800.0 / (16.02 if model == "ssy" else 36.03)
    ka = int(np.argmax(ea <= sw)) if np.any(ea <= sw) else -1
    kb = int(np.argmax(eb <= sw)) if np.any(eb <= sw) else -1
    print(f"{model} {shapes}: t_f32 {na} iterations, fp64 {nb}; switch level {sw:.3e} reached at iteration {ka} (t_f32) / {kb} (fp64)")
    for k in sorted(set([10, 50, 100, 200, max(ka - 20, 0), max(ka - 5, 0), ka, ka + 1, ka + 2, ka + 5, ka + 20, ka + 50, na - 1, nb - 1])):
        a = ea[k] if 0 <= k < len(ea) else float("nan")
        b = eb[k] if 0 <= k < len(eb) else float("nan")
        ra = ea[k] / ea[k - 1] if 0 < k < len(ea) else float("nan")
        rb = eb[k] / eb[k - 1] if 0 < k < len(eb) else float("nan")
        print(f"   it {k:5d}: step t_f32 {a:.4e} (ratio {ra:.5f})   fp64 {b:.4e} (ratio {rb:.5f})")
    print(f"   max|x32 - x64| = {np.max(np.abs(xa - xb)):.3e}")
    # where the two traces part: first iteration at which the steps differ by more than 1e-3 relative
    k = next((i for i in range(min(len(ea), len(eb))) if abs(ea[i] - eb[i]) > 1e-3 * eb[i]), None)
    print(f"   traces agree to 1e-3 relative up to iteration {k}")
    if k is not None:
        for i in range(max(k - 2, 0), min(k + 12, len(ea), len(eb))):
            print(f"      it {i:5d}: t_f32 {ea[i]:.6e}   fp64 {eb[i]:.6e}")
    T.close()
