"""
GPU parity tests of the padded pair plan (csrc/pad_kernels.hpp): grids between the plans -- every extent <= 32, neither
the compile-time pair kernels' shapes nor the small-grid plan's sizes (10^6 ... 15^6, 17^6, ragged shapes, 4-D grids
with extents above 16) -- run the pair plan's passes on 16 / 24 / 32-wide tiles with run-time extents and the grid's
real strides.  Against the C oracle (oracle/c, a restatement of
code/gcy/discrete/gcy_wc_ratio.py:134-238 and code/ssy/discrete/ssy_wc_ratio.py:82-151) on the same inputs: T with its residual, the linearising T + J.v, the
adjoint identity for J^T.v, the device SA loop, Newton-Krylov; and against the generic tiles (SDFS_PAD_PLAN=0), which
these grids ran on before.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(10,) * 6, (12, 11, 9, 13, 10, 7), (15,) * 6, (16, 15, 14, 13, 12, 11), (9, 16, 5, 16, 16, 15), (13, 13, 16, 16, 3, 4),
          # 24- and 32-wide tiles, mixed per pass; 4-D grids beyond the small-grid plan
          (17,) * 6, (25, 19, 9, 9, 12, 7), (32, 32, 5, 5, 6, 24), (3, 30, 22, 21, 17, 18), (20,) * 4, (25, 18, 32, 7), (32,) * 4,
          # round 4, 16-byte requests: even remainders behind a line pair (18^6, 14^6) against odd ones (21^4, 15^6 above);
          # 32-wide slices of an odd and of an even number of points (25 x 25, 26 x 31)
          (18,) * 6, (14,) * 6, (21,) * 4, (25,) * 4, (27, 29, 26, 31)]


@pytest.fixture(scope="module")
def S():
    import sdfs_via_autodiff_amd as S
    return S


def build(S, shapes, pad=True):
    model = "gcy" if len(shapes) == 6 else "ssy"
    g = S.GCY() if model == "gcy" else S.SSY()
    arr = (S.discretize_gcy if model == "gcy" else S.discretize_ssy)(g, shapes)
    old = os.environ.get("SDFS_PAD_PLAN")
    # (shapes that mix extents above and below 16 take the padded plan only under SDFS_PAD_PLAN=2 -- correct, measured
    # no faster than the generic tiles; all-small 6-D and all-wide shapes take it by default)
    wide = max(shapes) > 16 and min(shapes) <= 16
    os.environ["SDFS_PAD_PLAN"] = ("2" if wide else "1") if pad else "0"
    try:
        T = S.KoopmansOperator(model, shapes, g.params, arr)
    finally:
        if old is None:
            del os.environ["SDFS_PAD_PLAN"]
        else:
            os.environ["SDFS_PAD_PLAN"] = old
    return T, g.params, arr


@pytest.mark.parametrize("shapes", SHAPES)
def test_padded_pair_plan_vs_c_oracle(S, shapes):
    from oracle.c_oracle import COperator
    T, params, arr = build(S, shapes)
    assert T.describe_plan().count("padded pair plan pass") == len(shapes) // 2, T.describe_plan()
    oc = COperator("gcy" if len(shapes) == 6 else "ssy", shapes, params, arr)
    rng = np.random.default_rng(sum(shapes))
    w = 300 + 600 * rng.random(shapes)
    v = rng.standard_normal(shapes)
    want = oc(w)
    got = T(w)
    assert np.max(np.abs(got - want) / want) < 1e-12
    assert T.residual() == pytest.approx(float(np.max(np.abs(want - w))), rel=1e-12)
    jw = oc.jvp(w, v)
    jv = T.jvp(w, v)
    assert np.max(np.abs(jv - jw)) <= 1e-11 * np.max(np.abs(jw))
    # adjoint: <u, J v> = <J^T u, v>
    u = rng.standard_normal(shapes)
    jtu = T.vjp(w, u)
    assert abs(np.vdot(u, jv) - np.vdot(jtu, v)) <= 1e-10 * (np.linalg.norm(u) * np.linalg.norm(jv))
    # the device SA loop: k iterates and the error of the last
    x, prev = w, None
    for k in (1, 2, 3):
        prev, x = x, oc(x)
        xk, n, info = T.solve(w, "successive_approx", tol=0.0, max_iter=k)
        assert n == k
        assert np.max(np.abs(xk - x) / x) < 1e-11
        err = float(np.max(np.abs(x - prev)))
        assert abs(info["final_err"] - err) <= 1e-9 * err
    T.close()


@pytest.mark.parametrize("shapes", [(10,) * 6, (12, 11, 9, 13, 10, 7), (20, 17, 9, 9, 6, 5), (20,) * 4])
def test_padded_pair_plan_solvers_match_generic_tiles(S, shapes):
    """Newton-Krylov (fused dots out of the last J.v pass), Anderson and SA to a tolerance: the same counts and fixed
    point as on the generic tiles."""
    Tp, _, _ = build(S, shapes)
    Tg, _, _ = build(S, shapes, pad=False)
    assert "padded pair plan" in Tp.describe_plan() and "padded pair plan" not in Tg.describe_plan()
    w0 = np.full(shapes, 800.0)
    xp, n_p, ip = Tp.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
    xg, n_g, ig = Tg.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
    assert ip["status"] == 0 and n_p == n_g and abs(ip["n_apply"] - ig["n_apply"]) <= max(4, ig["n_apply"] // 20)
    np.testing.assert_allclose(xp, xg, rtol=0, atol=1e-8)
    assert np.max(np.abs(Tp(xp) - xp)) < 1e-9
    xs, n_s, _ = Tp.solve(w0, "successive_approx", tol=1e-5, max_iter=20000)
    xs2, n_s2, _ = Tg.solve(w0, "successive_approx", tol=1e-5, max_iter=20000)
    assert n_s == n_s2
    np.testing.assert_allclose(xs, xs2, rtol=1e-12)
    xa, n_a, ia = Tp.solve(w0, "anderson", tol=1e-6, max_iter=5000)
    assert ia["status"] == 0 and ia["final_err"] <= 1e-6
    np.testing.assert_allclose(xa, xg, rtol=0, atol=1e-3)
    # fp32 Krylov storage keeps the generic J.v kernels (the padded plan has no fp32 forms) and the padded T
    x32, n32, i32 = Tp.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, krylov_f32=1)
    assert i32["status"] == 0
    np.testing.assert_allclose(x32, xg, rtol=0, atol=1e-8)
    Tp.close(); Tg.close()


def test_padded_plan_is_not_taken_where_other_plans_are(S):
    for shapes in ((8,) * 6, (16,) * 6, (5, 4, 6, 3, 4, 5)):
        T, _, _ = build(S, shapes)
        assert "padded pair plan" not in T.describe_plan()
        T.close()
    # by default not for shapes that mix extents above and below 16 ...
    g = S.GCY(); shapes = (17, 17, 9, 9, 12, 7)
    T = S.KoopmansOperator("gcy", shapes, g.params, S.discretize_gcy(g, shapes))
    assert "padded pair plan" not in T.describe_plan()
    T.close()
    # ... but for every extent in 17 .. 32, 4-D and 6-D (20- / 24- / 32-wide tiles)
    m = S.SSY(); shapes = (20, 25, 18, 32)
    T = S.KoopmansOperator("ssy", shapes, m.params, S.discretize_ssy(m, shapes))
    assert T.describe_plan().count("padded pair plan pass") == 2
    T.close()
