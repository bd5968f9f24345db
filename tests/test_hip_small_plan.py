"""
GPU parity tests of the small-grid plan (csrc/fast_kernels.hpp: small_tile_kernel), the kernels SSY 15^4
(BASELINE configs 1 and 2) and every other unconditional 4-D / 6-D grid with extents <= 16 run on by default.
Checked against
 (1) the oracle on the same seeded inputs (numpy restatement; T, J.v and J^T.v),
 (2) the generic-tile kernels on the same inputs (SDFS_PLAN=classic), an independent HIP route,
 (3) solver-level behaviour (iteration counts and fixed points of SA / Newton-Krylov / Anderson).
SDFS_PLAN / SDFS_SMALL_R are create-time knobs, so each operator is built under the environment it is meant for.
Tolerance: 1e-12 relative per application (sums in a different order, powers within a few ulp).
"""
import contextlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

APPLY_RTOL = 1e-12


@pytest.fixture(scope="module")
def S():
    import sdfs_via_autodiff_amd as S
    return S


@contextlib.contextmanager
def env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def build(S, model, shapes, **knobs):
    m = S.SSY() if model == "ssy" else S.GCY()
    arr = (S.discretize_ssy if model == "ssy" else S.discretize_gcy)(m, shapes)
    with env(**knobs):
        return S.KoopmansOperator(model, shapes, m.params, arr)


def oracle_ops(model, shapes):
    from oracle import models, ssy, gcy
    if model == "ssy":
        p = models.ssy_params(); arr = ssy.discretize_ssy(p, shapes)
        return (lambda w: ssy.T_ssy_factorised(w, shapes, p, arr),
                lambda w, v: ssy.jvp_ssy(w, v, shapes, p, arr))
    p = models.gcy_params(); arr = gcy.discretize_gcy(p, shapes)
    return (lambda w: gcy.T_gcy_factorised(w, shapes, p, arr),
            lambda w, v: gcy.jvp_gcy(w, v, shapes, p, arr))


def wbench(shapes, seed=0):
    return 400 + 500 * np.random.default_rng(seed).random(shapes)


CASES = [("ssy", (15, 15, 15, 15)), ("ssy", (16, 16, 16, 16)), ("ssy", (2, 3, 4, 5)), ("ssy", (5, 4, 3, 2)),
         ("ssy", (3, 3, 3, 3)), ("ssy", (16, 2, 2, 16)), ("ssy", (7, 13, 11, 9)), ("ssy", (2, 2, 2, 2)),
         ("gcy", (3, 4, 5, 2, 3, 4)), ("gcy", (6,) * 6), ("gcy", (2,) * 6), ("gcy", (9, 3, 2, 16, 5, 7)),
         ("gcy", (3, 3, 12, 12, 13, 13))]


@pytest.mark.parametrize("wpt", [1, 4])
@pytest.mark.parametrize("run", [0, 1, 4])
@pytest.mark.parametrize("model,shapes", CASES)
def test_small_plan_T_jvp_vjp_vs_oracle(S, model, shapes, run, wpt):
    """run = 0: the planner's own choice of run length; 1 / 4: forced (strided 2-D slices / 32-byte runs with
    partial trailing chunks whenever the remainder is not a multiple of four).  wpt: waves per tile (1: four
    wave-private tiles per workgroup; 4: one tile per workgroup)."""
    Ts = build(S, model, shapes, SDFS_PLAN=None, SDFS_SMALL_R=run if run else None, SDFS_SMALL_WPT=wpt)
    assert ("%d wave%s per tile" % (wpt, "" if wpt == 1 else "s")) in Ts.describe_plan()
    Tc = build(S, model, shapes, SDFS_PLAN="classic")
    assert "small-grid plan pass" in Ts.describe_plan(), Ts.describe_plan()
    assert "small-grid plan pass" not in Tc.describe_plan()
    if run:
        assert all(("run %d," % run) in ln for ln in Ts.describe_plan().splitlines() if "lines[" in ln and "small-grid" in ln)
    oT, oJ = oracle_ops(model, shapes)
    w = wbench(shapes)
    want = oT(w)
    got = Ts(w)
    np.testing.assert_allclose(got, want, rtol=APPLY_RTOL)
    r = np.max(np.abs(want - w))
    assert abs(Ts.residual() - r) <= 1e-12 * r + 1e-9
    np.testing.assert_allclose(got, Tc(w), rtol=APPLY_RTOL)
    v = np.random.default_rng(1).standard_normal(shapes)
    jw = oJ(w, v)
    np.testing.assert_allclose(Ts.jvp(w, v), jw, rtol=1e-11, atol=1e-12 * np.max(np.abs(jw)))
    # J^T u against <u, J v> = <J^T u, v> and against the generic kernels
    u = np.random.default_rng(2).standard_normal(shapes)
    jtu = Ts.vjp(w, u)
    np.testing.assert_allclose(np.vdot(jtu, v), np.vdot(u, jw), rtol=1e-10)
    jc = Tc.vjp(w, u)
    np.testing.assert_allclose(jtu, jc, rtol=1e-10, atol=1e-12 * np.max(np.abs(jc)))
    np.testing.assert_array_equal(w, wbench(shapes))                    # inputs never mutated


def test_small_plan_scope(S):
    """Extents above 16, conditional tensors and SDFS_SMALL_PLAN=0 keep the generic kernels."""
    for model, shapes in (("ssy", (17, 5, 20, 3)), ("ssy", (20, 20, 20, 20)), ("gcy", (3, 3, 3, 3, 3, 17))):
        assert "small-grid plan pass" not in build(S, model, shapes, SDFS_PLAN=None).describe_plan()
    assert "small-grid plan pass" not in build(S, "ssy", (15,) * 4, SDFS_PLAN=None, SDFS_SMALL_PLAN=0).describe_plan()
    from oracle import models, ssy
    shapes = (6, 5, 4, 3)
    p = models.ssy_params(); arr = list(ssy.discretize_ssy(p, shapes))
    q = np.random.default_rng(3).random(arr[7].shape) + 0.05
    arr[7] = q / q.sum(axis=-1, keepdims=True)
    T = S.ssy_operator(shapes, p, arr)
    assert "small-grid plan pass" not in T.describe_plan()
    w = wbench(shapes)
    np.testing.assert_allclose(T(w), ssy.T_ssy_factorised(w, shapes, p, arr), rtol=APPLY_RTOL)


def test_small_plan_full_range_power_path(S):
    shapes = (15, 15, 15, 15)
    Ts = build(S, "ssy", shapes, SDFS_PLAN=None)
    oT, _ = oracle_ops("ssy", shapes)
    w = wbench(shapes)
    w[3, 4, 5, 6] = -1.0
    w[0, 0, 0, 0] = 0.0
    w[9, 9, 9, 9] = 1e300
    with np.errstate(all="ignore"):
        want = oT(w)
    got = Ts(w)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=APPLY_RTOL)
    assert Ts.residual() == np.inf


@pytest.mark.parametrize("model,shapes", [("ssy", (15, 15, 15, 15)), ("gcy", (5, 4, 6, 3, 4, 5))])
def test_small_plan_solvers_match_generic_plan(S, model, shapes):
    Ts = build(S, model, shapes, SDFS_PLAN=None)
    Tc = build(S, model, shapes, SDFS_PLAN="classic")
    w0 = np.full(shapes, 800.0)
    xs, n_s, is_ = Ts.solve(w0, "successive_approx", tol=1e-6, record_errors=True)
    xc, n_c, ic = Tc.solve(w0, "successive_approx", tol=1e-6, record_errors=True)
    assert n_s == n_c
    np.testing.assert_allclose(is_["errors"], ic["errors"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(xs, xc, rtol=0, atol=1e-9)
    xs, n_s, _ = Ts.solve(w0, "newton", tol=1e-10, inner_rtol=1e-8, inner_atol=0.0)
    xc, n_c, _ = Tc.solve(w0, "newton", tol=1e-10, inner_rtol=1e-8, inner_atol=0.0)
    assert n_s == n_c
    np.testing.assert_allclose(xs, xc, rtol=0, atol=1e-8)
    oT, oJ = oracle_ops(model, shapes)
    from oracle import solvers as osol
    xo = osol.newton_polish(oT, oJ, xs.copy())
    assert np.max(np.abs(xs - xo)) < 1e-8
    xa, n_a, _ = Ts.solve(w0, "anderson", tol=1e-7)
    assert np.max(np.abs(xa - xo)) < 1e-5
    # fp32 Krylov storage runs its linearised part on the generic kernels, same fixed point
    xf, n_f, _ = Ts.solve(w0, "newton", tol=1e-10, inner_rtol=1e-8, inner_atol=0.0, krylov_f32=1)
    assert np.max(np.abs(xf - xo)) < 1e-8


@pytest.mark.parametrize("wpt", [0, 1, 4])
@pytest.mark.parametrize("model,shapes,run", [("ssy", (15, 15, 15, 15), 0), ("ssy", (3, 3, 3, 3), 0), ("ssy", (7, 13, 11, 9), 4),
                                              ("gcy", (5, 4, 6, 3, 4, 5), 0), ("gcy", (6,) * 6, 4), ("gcy", (3, 4, 5, 2, 3, 4), 1)])
def test_fused_successive_approximation(S, model, shapes, run, wpt):
    """SA on the small-grid plan reverses the pair order every iteration and runs the last pass of one application
    with the first pass of the next as one kernel; the sup-norm error travels as per-workgroup maxima that the next
    iteration's kernels reduce (no atomics).  Same iteration count, error trace and iterate as one launch per
    pass (SDFS_SA_FUSED=0) and as the generic kernels, with and without hipGraph chunks, for odd chunk lengths and
    when max_iter cuts a chunk short."""
    knobs = dict(SDFS_PLAN=None, SDFS_SMALL_R=run if run else None, SDFS_SMALL_WPT=wpt if wpt else None)
    Tf = build(S, model, shapes, **knobs)
    Tu = build(S, model, shapes, SDFS_SA_FUSED=0, **knobs)
    Ta = build(S, model, shapes, SDFS_SA_FUSED=2, **knobs)            # fused kernels, residual through the atomic word
    Tc = build(S, model, shapes, SDFS_PLAN="classic")
    w0 = np.full(shapes, 800.0)
    xc, n_c, ic = Tc.solve(w0, "successive_approx", tol=1e-6, record_errors=True)
    for T, kw in ((Tf, {}), (Tu, {}), (Ta, {}), (Ta, dict(check_every=3)), (Tf, dict(check_every=7)), (Tf, dict(use_graph=0, check_every=5)), (Tf, dict(check_every=1))):
        x, n, info = T.solve(w0, "successive_approx", tol=1e-6, record_errors=True, **kw)
        assert n == n_c and info["n_apply"] == n
        np.testing.assert_allclose(info["errors"], ic["errors"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(x, xc, rtol=0, atol=1e-9)
    # max_iter in the middle of a chunk, odd and even: the iterate is T^k(w0)
    oT, _ = oracle_ops(model, shapes)
    for k in (1, 2, 5):
        x, n, info = Tf.solve(w0, "successive_approx", tol=0.0, max_iter=k)
        want = w0
        for _ in range(k):
            want = oT(want)
        assert n == k
        np.testing.assert_allclose(x, want, rtol=1e-11)
    # the operator itself is untouched by the loop's private intermediate
    w = wbench(shapes)
    np.testing.assert_allclose(Tf(w), oT(w), rtol=APPLY_RTOL)
