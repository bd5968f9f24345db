"""One-off fuzz: random grid shapes (2..32 per axis; FUZZ_SMALL=1: 2..16, the small-grid plan) for SSY / GCY against
the numpy oracle; T, JVP, three successive-approximation iterations of the device loop (fused schedules) and the
linearised T, with the slice-merge planner path on and off and perturbed (slice-dependent) tensors."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import sdfs_via_autodiff_amd as S
from oracle import models, ssy as ossy, gcy as ogcy

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
worst = 0.0
for trial in range(int(os.environ.get("FUZZ_N", "40"))):
    model = "ssy" if trial % 2 == 0 else "gcy"
    nd = 4 if model == "ssy" else 6
    while True:
        pool = list(range(2, 17)) if os.environ.get("FUZZ_SMALL") else [2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 20, 21, 24, 25, 31, 32]
        shapes = tuple(int(x) for x in rng.choice(pool, nd))
        if np.prod(shapes) <= (60000 if model == "gcy" else 200000):
            break
    m = S.SSY() if model == "ssy" else S.GCY()
    arr = list((S.discretize_ssy if model == "ssy" else S.discretize_gcy)(m, shapes))
    perturb = trial % 3 == 0
    if perturb:                      # make every slice of the conditional tensors different (rows still sum to 1)
        for qi in ([7] if model == "ssy" else [1, 3]):
            q = arr[qi].copy()
            q *= 1 + 0.1 * rng.random(q.shape)
            q /= q.sum(-1, keepdims=True)
            arr[qi] = q
    p = models.ssy_params() if model == "ssy" else models.gcy_params()
    o = ossy if model == "ssy" else ogcy
    To = (lambda w: o.T_ssy_factorised(w, shapes, p, arr)) if model == "ssy" else (lambda w: o.T_gcy_factorised(w, shapes, p, arr))
    Jo = (lambda w, v: o.jvp_ssy(w, v, shapes, p, arr)) if model == "ssy" else (lambda w, v: o.jvp_gcy(w, v, shapes, p, arr))
    T = S.KoopmansOperator(model, shapes, m.params, arr)
    w = 300 + 600 * rng.random(shapes)
    v = rng.standard_normal(shapes)
    e1 = np.max(np.abs(T(w) - To(w)) / To(w))
    jo = Jo(w, v)
    e2 = np.max(np.abs(T.jvp(w, v) - jo)) / np.max(np.abs(jo))
    x3, n3, _ = T.solve(w, "successive_approx", tol=0.0, max_iter=3)
    w3 = To(To(To(w)))
    e3 = np.max(np.abs(x3 - w3) / w3)
    worst = max(worst, e1, e2, e3)
    flag = "" if max(e1, e2, e3) < 1e-11 else "  <-- FAIL"
    small = "small-grid plan" in T.describe_plan()
    print(f"{trial:3d} {model} {shapes} perturbed={perturb} small_plan={small}: T {e1:.1e} jvp {e2:.1e} sa3 {e3:.1e}{flag}", flush=True)
    T.close()
print("worst", worst)
sys.exit(0 if worst < 1e-11 else 1)
