"""One-off fuzz over calibrations: random (beta, gamma, psi) for SSY / GCY on small and pair-plan grids, so that the
power routines see other exponents theta = (1-gamma)/(1-1/psi) and the fused SA kernels see both branches of
next_power_fast (wealth-consumption ratios below and above its range).  T, JVP and five iterations of the device SA
loop against the numpy oracle; Newton's fixed point against the oracle operator."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdfs_via_autodiff_amd as S  # noqa: E402
from oracle import models, ssy as ossy, gcy as ogcy  # noqa: E402

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
worst = 0.0
shapes_pool = {"ssy": [(15,) * 4, (5, 7, 6, 4), (16,) * 4, (20, 5, 3, 17)], "gcy": [(5, 4, 6, 3, 4, 5), (3,) * 6, (8,) * 6]}
for trial in range(int(os.environ.get("FUZZ_N", "24"))):
    model = "ssy" if trial % 2 == 0 else "gcy"
    shapes = shapes_pool[model][rng.integers(len(shapes_pool[model]))]
    beta = float(rng.choice([0.97, 0.99, 0.995, 0.999, 0.9987]))
    gamma = float(rng.uniform(2.0, 15.0))
    psi = float(rng.choice([1.5, 1.97, 2.5, 0.7]))            # psi < 1: theta > 0
    if model == "ssy":
        m = S.SSY(β=beta, γ=gamma, ψ=psi); p = models.ssy_params(beta=beta, gamma=gamma, psi=psi)
        arr = S.discretize_ssy(m, shapes)
        To = lambda w: ossy.T_ssy_factorised(w, shapes, p, arr)       # noqa: E731
        Jo = lambda w, v: ossy.jvp_ssy(w, v, shapes, p, arr)          # noqa: E731
    else:
        m = S.GCY(β=beta, γ=gamma, ψ=psi); p = models.gcy_params(beta=beta, gamma=gamma, psi=psi)
        arr = S.discretize_gcy(m, shapes)
        To = lambda w: ogcy.T_gcy_factorised(w, shapes, p, arr)       # noqa: E731
        Jo = lambda w, v: ogcy.jvp_gcy(w, v, shapes, p, arr)          # noqa: E731
    T = S.KoopmansOperator(model, shapes, m.params, arr)
    scale = float(rng.choice([5.0, 60.0, 300.0, 900.0]))              # start levels on both sides of next_power_fast's range
    w = scale * (0.6 + 0.8 * rng.random(shapes))
    v = rng.standard_normal(shapes)
    with np.errstate(all="ignore"):
        tw = To(w)
        e1 = np.max(np.abs(T(w) - tw) / np.abs(tw))
        jo = Jo(w, v)
        e2 = np.max(np.abs(T.jvp(w, v) - jo)) / np.max(np.abs(jo))
        x5, n5, _ = T.solve(w, "successive_approx", tol=0.0, max_iter=5)
        w5 = w
        for _ in range(5):
            w5 = To(w5)
        e3 = np.max(np.abs(x5 - w5) / np.abs(w5))
    ok = np.isfinite([e1, e2, e3]).all() and max(e1, e2, e3) < 1e-11
    worst = max(worst, e1, e2, e3)
    print(f"{trial:3d} {model} {shapes} beta {beta} gamma {gamma:.2f} psi {psi} theta {m.θ:8.2f} w~{scale:5.0f}: "
          f"T {e1:.1e} jvp {e2:.1e} sa5 {e3:.1e}{'' if ok else '  <-- FAIL'}", flush=True)
    T.close()
print("worst", worst)
sys.exit(0 if worst < 1e-11 else 1)
