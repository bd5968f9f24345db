"""Fuzz of the padded pair plan (csrc/pad_kernels.hpp): random 6-D shapes with extents 2..16 / 2..32 (alternating) and 4e5 < N <= FUZZ_MAX
points (default 3e6), GCY calibration, against the C oracle (oracle/c, test infrastructure): T with its residual, J.v,
the adjoint identity for J^T.v and three iterations of the device SA loop.  Exit code 1 beyond 1e-11 relative.
    python tools/fuzz_pad_plan.py > profiles/round3_fuzz_pad_plan.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
os.environ.setdefault("SDFS_PAD_PLAN", "2")          # also shapes that mix extents above and below 16 (not the default: no faster than the generic tiles)
import sdfs_via_autodiff_amd as S  # noqa: E402
from oracle.c_oracle import COperator  # noqa: E402

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "3")))
MAXP = float(os.environ.get("FUZZ_MAX", "3e6"))
g = S.GCY()
worst, fails, done = 0.0, 0, 0
fixed = [(16, 16, 16, 16, 3, 3), (3, 3, 16, 16, 16, 16), (16, 3, 16, 3, 16, 16), (3, 16, 15, 16, 16, 3), (16, 16, 3, 3, 16, 16), (16, 16, 16, 16, 2, 4)]
tries = 0
while done < int(os.environ.get("FUZZ_N", "24")) and tries < 100000:
    tries += 1
    shapes = fixed.pop(0) if fixed else tuple(int(x) for x in rng.integers(2, 33 if done % 2 else 17, 6))
    n = int(np.prod(shapes))
    if not (4e5 < n <= MAXP):
        continue
    arr = S.discretize_gcy(g, shapes)
    T = S.KoopmansOperator("gcy", shapes, g.params, arr)
    padded = "padded pair plan" in T.describe_plan()
    oc = COperator("gcy", shapes, g.params, arr)
    w = 300 + 600 * rng.random(shapes)
    v = rng.standard_normal(shapes)
    u = rng.standard_normal(shapes)
    want = oc(w)
    e_T = float(np.max(np.abs(T(w) - want) / want))
    rs = float(np.max(np.abs(want - w)))
    e_r = abs(T.residual() - rs) / rs
    jo = oc.jvp(w, v)
    jv = T.jvp(w, v)
    e_J = float(np.max(np.abs(jv - jo)) / np.max(np.abs(jo)))
    e_A = abs(float(np.vdot(u, jv) - np.vdot(T.vjp(w, u), v))) / (np.linalg.norm(u) * np.linalg.norm(jv))
    x3, _, _ = T.solve(w, "successive_approx", tol=0.0, max_iter=3)
    w3 = oc(oc(want))
    e_s = float(np.max(np.abs(x3 - w3) / w3))
    bad = max(e_T, e_r, e_J, e_A, e_s) >= 1e-11
    worst = max(worst, e_T, e_r, e_J, e_A, e_s)
    fails += bad
    print(f"{done:3d} {shapes} N={n} padded_plan={padded}: T {e_T:.1e} resid {e_r:.1e} jvp {e_J:.1e} adjoint {e_A:.1e} sa3 {e_s:.1e}{'  <-- FAIL' if bad else ''}", flush=True)
    T.close()
    done += 1
print("worst", worst, "failures", fails)
sys.exit(1 if fails else 0)
