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

for model, shapes in (("ssy", (3,)*4), ("ssy", (15,)*4), ("ssy", (7,16,5,9)), ("gcy", (3,4,2,3,2,4)), ("gcy", (6,)*6), ("gcy", (8,)*6)):
    Tf, Tu, Th = build(model, shapes, False, True), build(model, shapes, False, False), build(model, shapes, True, False)
    w0 = np.full(shapes, 800.0)
    print(model, shapes, flush=True)
    for k in (1, 5, 11, 12, 13, 20, 21, 41):
        xf, nf, i_f = Tf.solve(w0, "anderson", tol=0.0, max_iter=k, record_errors=True)
        xh, nh, ih = Th.solve(w0, "anderson", tol=0.0, max_iter=k, record_errors=True)
        ef, eh = np.asarray(i_f["errors"]), np.asarray(ih["errors"])
        n = min(len(ef), len(eh))
        print("  k", k, nf, nh, len(ef), len(eh), "max rel err-trace diff", float(np.max(np.abs(ef[:n]-eh[:n])/eh[:n])) if n else None,
              "x rel diff", float(np.max(np.abs(xf-xh)/np.abs(xh))), flush=True)
    for tol in (1e-6, 1e-8):
        res = {}
        for name, T in (("fused", Tf), ("unfused", Tu), ("host", Th)):
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); x, n, info = T.solve(w0, "anderson", tol=tol, max_iter=20000, record_errors=True); best = min(best, time.perf_counter()-t0)
            res[name] = (x, n, info, best)
            print(f"  tol {tol:g} {name:8s} passes {n:6d} trace {len(info['errors']):6d} status {info['status']} final {info['final_err']:.3e} {best*1e3:9.2f} ms  {best/n*1e6:7.2f} us/pass  resid {float(np.max(np.abs(T(x)-x))):.2e}", flush=True)
        print("   |xf - xh|", float(np.max(np.abs(res['fused'][0]-res['host'][0]))), "|xu - xh|", float(np.max(np.abs(res['unfused'][0]-res['host'][0]))))
        for kw in (dict(check_every=7), dict(use_graph=0, check_every=3), dict(check_every=1), dict(check_every=100)):
            x, n, info = Tf.solve(w0, "anderson", tol=tol, max_iter=20000, record_errors=True, **kw)
            print("   ", kw, n, np.array_equal(x, res['fused'][0]), flush=True)
    t0 = time.perf_counter(); x, n, info = Tf.solve(w0, "successive_approx", tol=1e-8, max_iter=100000); print("  SA 1e-8", n, (time.perf_counter()-t0)*1e3, "ms")
