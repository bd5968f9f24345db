"""Sweep of the pair plan's extent combinations (pairs of 16 / 20 / 24 / 32) with the streamed line kernels switched on
for EVERY extent (SDFS_LINE_STREAM=7; the default has them on 20-extent pairs only): T with its residual, the
linearising T + J.v, three iterations of the device SA loop (streamed and fused schedules) and one application with
fp32 intermediates, against the C oracle (oracle/c, test infrastructure).  4-D: all 16 combinations; 6-D: the
combinations below FUZZ_MAX points (default 7e7).  Exit code 1 if anything is off by more than 1e-11 relative.
    python tools/fuzz_pair_stream.py > profiles/round3_fuzz_pair_stream.txt"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, ".")
os.environ["SDFS_PLAN"] = "pair"
os.environ["SDFS_LINE_STREAM"] = "7"
import sdfs_via_autodiff_amd as S  # noqa: E402
from oracle.c_oracle import COperator  # noqa: E402

EXT = (16, 20, 24, 32)
MAXP = float(os.environ.get("FUZZ_MAX", "7e7"))
rng = np.random.default_rng(5)
worst = 0.0


def case(model, shapes):
    global worst
    m = S.SSY() if model == "ssy" else S.GCY()
    arr = (S.discretize_ssy if model == "ssy" else S.discretize_gcy)(m, shapes)
    T = S.KoopmansOperator(model, shapes, m.params, arr)
    os.environ["SDFS_SA_FUSED"] = "1"
    Tf = S.KoopmansOperator(model, shapes, m.params, arr)
    del os.environ["SDFS_SA_FUSED"]
    oc = COperator(model, shapes, m.params, arr)
    w = 300 + 600 * rng.random(shapes)
    v = rng.standard_normal(shapes)
    want = oc(w)
    e_T = float(np.max(np.abs(T(w) - want) / want))
    e_res = abs(T.residual() - float(np.max(np.abs(want - w)))) / float(np.max(np.abs(want - w)))
    jo = oc.jvp(w, v)
    e_J = float(np.max(np.abs(T.jvp(w, v) - jo)) / np.max(np.abs(jo)))
    w3 = oc(oc(want))
    x3, _, _ = T.solve(w, "successive_approx", tol=0.0, max_iter=3)
    e_sa = float(np.max(np.abs(x3 - w3) / w3))
    x3f, _, _ = Tf.solve(w, "successive_approx", tol=0.0, max_iter=3)
    e_saf = float(np.max(np.abs(x3f - w3) / w3))
    x32, _, _ = T.solve(w, "successive_approx", tol=0.0, max_iter=1, t_f32=1)
    e_32 = float(np.max(np.abs(x32 - want) / want))
    streamed = "streamed" in T.describe_plan()
    bad = max(e_T, e_res, e_J, e_sa, e_saf) >= 1e-11 or e_32 >= 2e-8
    worst = max(worst, e_T, e_res, e_J, e_sa, e_saf)
    print(f"{model} {shapes} streamed={streamed}: T {e_T:.1e} resid {e_res:.1e} jvp {e_J:.1e} sa3 {e_sa:.1e} sa3 fused {e_saf:.1e} "
          f"fp32-intermediate T {e_32:.1e}{'  <-- FAIL' if bad else ''}", flush=True)
    T.close(); Tf.close()
    return bad


fails = 0
for a, b in itertools.product(EXT, EXT):
    fails += case("ssy", (a, a, b, b))
for a, b, c in itertools.product(EXT, EXT, EXT):
    if float(a * a) * b * b * c * c <= MAXP:
        fails += case("gcy", (a, a, b, b, c, c))
print("worst", worst, "failures", fails)
sys.exit(1 if fails else 0)
