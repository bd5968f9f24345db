"""BASELINE config 5: tolerance-versus-throughput sweep of Newton-Krylov at GCY 16^6 / 20^6, fp64 against fp32
Krylov storage (opts.krylov_f32: Krylov vectors, c1 / c2 and every J.v stream in fp32, arithmetic, reductions,
outer residual and iterate in fp64), with the ACHIEVED fixed-point error of every run.
    python tools/mixed_precision_sweep.py [16|20 ...] > profiles/round2_mixed_precision_sweep.txt
Reference fixed point: fp64 Newton to 1e-11 on the device, its residual |T(x) - x| confirmed with the C oracle
(oracle/c, test infrastructure) on the full grid.  Wall time includes the upload of w_init = 800 and the
download of the result."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 20]
    g = S.GCY()
    rows = []
    for n in sizes:
        shp = (n,) * 6
        arr = S.discretize_gcy(g, shp)
        T = S.gcy_operator(shp, g.params, arr)
        w0 = np.full(shp, 800.0)
        xs, ns, infos = T.solve(w0, "newton", tol=1e-11, inner_rtol=1e-8, inner_atol=0.0, max_iter=40)
        from oracle.c_oracle import COperator
        res_c = float(np.max(np.abs(COperator("gcy", shp, g.params, arr)(xs) - xs)))
        print(f"# GCY {n}^6 reference fixed point: fp64 Newton tol 1e-11, {ns} iterations, C-oracle max|T(x)-x| = {res_c:.2e}", flush=True)
        for tol in (1e-4, 1e-5, 1e-6, 1e-7, 1e-8):
            for f32 in (0, 1):
                best = None
                for rep in range(2):
                    t0 = time.perf_counter()
                    x, it, info = T.solve(w0, "newton", tol=tol, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=f32)
                    dt = time.perf_counter() - t0
                    if best is None or dt < best[0]:
                        best = (dt, x, it, info)
                dt, x, it, info = best
                err = float(np.max(np.abs(x - xs)))
                row = dict(grid=f"GCY {n}^6", tol=tol, krylov_f32=f32, iterations=it, applies=info["n_apply"], seconds=dt,
                           applies_per_s=info["n_apply"] / dt, final_step=info["final_err"], err_vs_fp64_fixed_point=err)
                rows.append(row)
                print(f"GCY {n}^6  tol {tol:7.0e}  {'fp32 Krylov' if f32 else 'fp64       '}  Newton steps {it:3d}  applies {info['n_apply']:5d}  "
                      f"{dt:7.3f} s  {info['n_apply'] / dt:8.0f} applies/s  last step {info['final_err']:9.2e}  "
                      f"|x - x*|_inf {err:9.2e}", flush=True)
                del x
        T.close()
    print("JSON " + json.dumps(rows))


if __name__ == "__main__":
    main()
