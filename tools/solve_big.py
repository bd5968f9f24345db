"""Time-to-converge on a large grid: Newton-Krylov and successive approximation (device-resident)."""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdfs_via_autodiff_amd as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
algos = sys.argv[2].split(",") if len(sys.argv) > 2 else ["newton", "successive_approx"]
shapes = (n,) * 6
t0 = time.perf_counter()
m = S.GCY(); arr = S.discretize_gcy(m, shapes)
T = S.gcy_operator(shapes, m.params, arr)
print("setup", round(time.perf_counter() - t0, 2), "s", flush=True)
w0 = np.full(shapes, 800.0)
out = {}
for algo in algos:
    kw = dict(tol=1e-8, inner_rtol=1e-6, inner_atol=0.0) if algo == "newton" else dict(tol=1e-8, check_every=64)
    t0 = time.perf_counter()
    x, it, info = T.solve(w0, algo, record_errors=True, **kw)
    dt = time.perf_counter() - t0
    r = T(x); res = float(np.max(np.abs(r - x)))
    out[algo] = dict(seconds=dt, iterations=it, applies=info["n_apply"], final_err=info["final_err"],
                     residual=res, wmin=float(x.min()), wmax=float(x.max()), status=info["status"])
    print(algo, json.dumps(out[algo]), flush=True)
    if algo == "newton":
        xs = x
if "newton" in out and "successive_approx" in out:
    print("max |w_newton - w_sa| =", float(np.max(np.abs(xs - x))))
if os.environ.get("SOLVE_PROFILE"):
    T.set_profiling(True); T.reset_counters()
    t0 = time.perf_counter()
    x, it, info = T.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
    dt = time.perf_counter() - t0
    print("profiled newton", dt, it, info["n_apply"])
    tot = 0
    for c in T.counters():
        tot += c["total_ms"]
        print(f"  {c['name']:50s} n={c['launches']:5d} total_ms={c['total_ms']:9.2f} avg_ms={c['total_ms']/max(c['launches'],1):8.4f} GB/s={c['alg_bytes']/max(c['total_ms']/max(c['launches'],1),1e-9)/1e6:8.1f}")
    print("  sum of kernel ms", tot)
