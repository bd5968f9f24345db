"""
GPU parity tests of the pair plan (csrc/fast_kernels.hpp: slice_kernel + line_kernel), the kernels the
headline grids (GCY 16^6, 20^6) run on.  Checked against
 (1) the oracle (numpy restatement; the C twin at 6-D sizes) on the same seeded inputs,
 (2) the generic-tile kernels on the same handle inputs (SDFS_PLAN=classic), an independent HIP route,
 (3) solver-level behaviour: iteration counts, fixed points, the full-range power path.
SDFS_PLAN is a create-time knob, so each operator is built under the environment it is meant for.
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
def plan_env(which):
    old = os.environ.get("SDFS_PLAN")
    os.environ["SDFS_PLAN"] = which
    try:
        yield
    finally:
        if old is None:
            del os.environ["SDFS_PLAN"]
        else:
            os.environ["SDFS_PLAN"] = old


def make_ops(S, model, shapes):
    """(pair-plan operator, generic-plan operator, params, arrays)"""
    if model == "ssy":
        m = S.SSY(); arr = S.discretize_ssy(m, shapes)
    else:
        m = S.GCY(); arr = S.discretize_gcy(m, shapes)
    with plan_env("pair"):
        Tp = S.KoopmansOperator(model, shapes, m.params, arr)
    with plan_env("classic"):
        Tc = S.KoopmansOperator(model, shapes, m.params, arr)
    assert "pair plan pass" in Tp.describe_plan(), Tp.describe_plan()
    assert "pair plan pass" not in Tc.describe_plan()
    return Tp, Tc, m.params, arr


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


# every extent class of the kernels (16 / 20 / 24 / 32) as slice pair and as line pair
SSY_SHAPES = [(16, 16, 16, 16), (20, 20, 20, 20), (24, 24, 24, 24), (32, 32, 32, 32),
              (16, 16, 20, 20), (20, 20, 16, 16), (24, 24, 32, 32), (32, 32, 24, 24)]


@pytest.mark.parametrize("shapes", SSY_SHAPES)
def test_pair_plan_T_and_jvp_vs_oracle_4d(S, shapes):
    Tp, Tc, _, _ = make_ops(S, "ssy", shapes)
    oT, oJ = oracle_ops("ssy", shapes)
    w = wbench(shapes)
    want = oT(w)
    got = Tp(w)
    np.testing.assert_allclose(got, want, rtol=APPLY_RTOL)
    r = np.max(np.abs(want - w))
    assert abs(Tp.residual() - r) <= 1e-12 * r + 1e-9
    np.testing.assert_allclose(got, Tc(w), rtol=APPLY_RTOL)            # independent HIP route
    v = np.random.default_rng(1).standard_normal(shapes)
    jw = oJ(w, v)
    np.testing.assert_allclose(Tp.jvp(w, v), jw, rtol=1e-11, atol=1e-12 * np.max(np.abs(jw)))
    np.testing.assert_array_equal(w, wbench(shapes))                    # inputs never mutated


def test_pair_plan_not_chosen_where_it_is_illegal(S):
    with plan_env("pair"):
        for model, shapes in (("ssy", (15, 15, 15, 15)), ("ssy", (16, 16, 16, 20)), ("gcy", (6,) * 6)):
            m = S.SSY() if model == "ssy" else S.GCY()
            arr = (S.discretize_ssy if model == "ssy" else S.discretize_gcy)(m, shapes)
            T = S.KoopmansOperator(model, shapes, m.params, arr)
            assert "pair plan pass" not in T.describe_plan()
        # conditional tensors whose slices differ keep the generic kernels
        from oracle import models, ssy
        shapes = (16, 16, 16, 16)
        p = models.ssy_params(); arr = list(ssy.discretize_ssy(p, shapes))
        q = np.random.default_rng(3).random(arr[7].shape) + 0.05
        arr[7] = q / q.sum(axis=-1, keepdims=True)
        T = S.ssy_operator(shapes, p, arr)
        assert "pair plan pass" not in T.describe_plan()
        w = wbench(shapes)
        np.testing.assert_allclose(T(w), ssy.T_ssy_factorised(w, shapes, p, arr), rtol=APPLY_RTOL)


def test_automatic_choice(S):
    """Small 4-D grids have too few line tiles to fill the chip: not the compile-time pair kernels unless forced (the
    padded pair plan's 20-wide tiles with 64-byte rows take SSY 20^4: tests/test_hip_pad_plan.py)."""
    os.environ.pop("SDFS_PLAN", None)
    m = S.SSY(); shp = (20, 20, 20, 20)
    T = S.ssy_operator(shp, m.params, S.discretize_ssy(m, shp))
    assert not any(line.startswith("pair plan pass") for line in T.describe_plan().splitlines())
    assert T.describe_plan().count("padded pair plan pass") == 2
    g = S.GCY(); shp = (16,) * 6
    T = S.gcy_operator(shp, g.params, S.discretize_gcy(g, shp))
    assert "pair plan pass" in T.describe_plan()


@pytest.mark.parametrize("shapes", [(16,) * 6, (16, 16, 20, 20, 16, 16)])
def test_pair_plan_gcy_6d_vs_c_oracle(S, shapes):
    """Three passes (slices, middle lines, last lines with a3 index tables over four remainder axes)."""
    from oracle.c_oracle import COperator
    from oracle import models, gcy
    Tp, Tc, _, _ = make_ops(S, "gcy", shapes)
    p = models.gcy_params()
    oc = COperator("gcy", shapes, p, gcy.discretize_gcy(p, shapes))
    w = wbench(shapes)
    want = oc(w)
    got = Tp(w)
    np.testing.assert_allclose(got, want, rtol=APPLY_RTOL)
    r = np.max(np.abs(want - w))
    assert abs(Tp.residual() - r) <= 1e-12 * r + 1e-9
    v = np.random.default_rng(1).standard_normal(shapes)
    jw = oc.jvp(w, v)
    jp = Tp.jvp(w, v)
    np.testing.assert_allclose(jp, jw, rtol=1e-11, atol=1e-12 * np.max(np.abs(jw)))
    np.testing.assert_allclose(jp, Tc.jvp(w, v), rtol=1e-11, atol=1e-12 * np.max(np.abs(jw)))


def test_pair_plan_full_range_power_path(S):
    """w <= 0, NaN and huge values leave the straight-line power routine: the wave redoes its units with the
    full routine and the results equal numpy's (NaN where the reference gives NaN, residual +inf)."""
    shapes = (16, 16, 16, 16)
    Tp, Tc, _, _ = make_ops(S, "ssy", shapes)
    oT, _ = oracle_ops("ssy", shapes)
    w = wbench(shapes)
    w[3, 4, 5, 6] = -1.0            # negative base -> NaN in every point it reaches
    w[0, 0, 0, 0] = 0.0             # 0^theta = inf (theta < 0)
    w[9, 9, 9, 9] = 1e300           # y log2 x beyond the fast range: underflows to 0
    with np.errstate(all="ignore"):
        want = oT(w)
    got = Tp(w)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=APPLY_RTOL)
    assert Tp.residual() == np.inf
    gc = Tc(w)
    assert np.array_equal(np.isnan(got), np.isnan(gc))
    # tiny positive inputs (subnormal w^theta range is not reached, but w itself tiny is legal)
    w2 = wbench(shapes) * 1e-3
    np.testing.assert_allclose(Tp(w2), oT(w2), rtol=APPLY_RTOL)


def test_pair_plan_solvers_match_generic_plan(S):
    shapes = (16, 16, 16, 16)
    Tp, Tc, _, _ = make_ops(S, "ssy", shapes)
    w0 = np.full(shapes, 800.0)
    xp, n_p, ip = Tp.solve(w0, "successive_approx", tol=1e-6, record_errors=True)
    xc, n_c, ic = Tc.solve(w0, "successive_approx", tol=1e-6, record_errors=True)
    assert n_p == n_c
    np.testing.assert_allclose(ip["errors"], ic["errors"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(xp, xc, rtol=0, atol=1e-9)
    # Newton-Krylov (linearising T, J.v with the fused dots of the last line pass) and Anderson
    xp, n_p, ip = Tp.solve(w0, "newton", tol=1e-10, inner_rtol=1e-8, inner_atol=0.0)
    xc, n_c, ic = Tc.solve(w0, "newton", tol=1e-10, inner_rtol=1e-8, inner_atol=0.0)
    assert n_p == n_c
    np.testing.assert_allclose(xp, xc, rtol=0, atol=1e-8)
    oT, oJ = oracle_ops("ssy", shapes)
    from oracle import solvers as osol
    xs = osol.newton_polish(oT, oJ, xp.copy())
    assert np.max(np.abs(xp - xs)) < 1e-8
    xa, n_a, _ = Tp.solve(w0, "anderson", tol=1e-7)
    assert np.max(np.abs(xa - xs)) < 1e-5
    # fp32 Krylov storage falls back to the generic kernels for the linearised part, same fixed point
    xf, n_f, _ = Tp.solve(w0, "newton", tol=1e-10, inner_rtol=1e-8, inner_atol=0.0, krylov_f32=1)
    assert np.max(np.abs(xf - xs)) < 1e-8


def test_pair_plan_gcy16_newton_fixed_point(S):
    """GCY 16^6 through the default (automatic) plan: Newton-Krylov to 1e-8, fixed point checked against the
    C oracle: |T(x) - x| small on the full grid."""
    from oracle.c_oracle import COperator
    from oracle import models, gcy
    os.environ.pop("SDFS_PLAN", None)
    shapes = (16,) * 6
    g = S.GCY()
    T = S.gcy_operator(shapes, g.params, S.discretize_gcy(g, shapes))
    assert "pair plan pass" in T.describe_plan()
    x, n, info = T.solve(np.full(shapes, 800.0), "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
    assert info["final_err"] <= 1e-8 and n < 25
    p = models.gcy_params()
    oc = COperator("gcy", shapes, p, gcy.discretize_gcy(p, shapes))
    assert np.max(np.abs(oc(x) - x)) < 1e-8


# (24 and 32: the fused line kernel at two workgroups per CU, the plain slice pass at N = 32)
@pytest.mark.parametrize("shapes", [(16,) * 6, (16, 16, 20, 20, 16, 16), (24, 24, 16, 16, 16, 16), (16, 16, 16, 16, 32, 32)])
def test_fused_sa_6d(S, shapes):
    """Successive approximation on the 6-D pair plan runs [slices, plain contraction] [lines, fused: end of one
    application + start of the next] per iteration, the fused pass alternating between the two line pairs.
    Same iterates as repeated application of T (one launch per pass) and as SDFS_SA_FUSED=0."""
    os.environ.pop("SDFS_PLAN", None)
    g = S.GCY(); arr = S.discretize_gcy(g, shapes)
    with plan_env("pair"):
        Tf = S.KoopmansOperator("gcy", shapes, g.params, arr)
        os.environ["SDFS_SA_FUSED"] = "0"
        try:
            Tu = S.KoopmansOperator("gcy", shapes, g.params, arr)
        finally:
            del os.environ["SDFS_SA_FUSED"]
    w0 = np.full(shapes, 800.0)
    want = w0
    done = 0
    for k in (1, 2, 3, 6):
        while done < k:
            want = Tu(want); done += 1
        x, n, info = Tf.solve(w0, "successive_approx", tol=0.0, max_iter=k)
        assert n == k and info["n_apply"] == k
        np.testing.assert_allclose(x, want, rtol=1e-11)
        assert abs(info["final_err"] - Tu.residual()) <= 1e-9 * Tu.residual()
    xf, n_f, i_f = Tf.solve(w0, "successive_approx", tol=1e-3, record_errors=True, check_every=9)
    xu, n_u, i_u = Tu.solve(w0, "successive_approx", tol=1e-3, record_errors=True)
    assert n_f == n_u and n_f > 20
    np.testing.assert_allclose(i_f["errors"], i_u["errors"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(xf, xu, rtol=0, atol=1e-9)
    # gate: once converged the remaining launches of the chunk leave the iterate alone
    xg, n_g, _ = Tf.solve(w0, "successive_approx", tol=1e-3, check_every=64)
    assert n_g == n_f
    np.testing.assert_array_equal(xg, xf)
    w = wbench(shapes)
    np.testing.assert_allclose(Tf(w), Tu(w), rtol=APPLY_RTOL)           # the operator itself is untouched


# the streamed forms of the line passes (csrc/stream_kernels.hpp): persistent middle pass with the next tile in
# flight (tickets), last pass with the residual's w loaded for the whole tile before the contractions.  Default for
# 20-extent pairs; SDFS_LINE_STREAM is a bit mask (1: middle pass, 2: T's last pass, 4: every extent), 0 = off.
@contextlib.contextmanager
def stream_env(v):
    old = os.environ.get("SDFS_LINE_STREAM")
    os.environ["SDFS_LINE_STREAM"] = str(v)
    try:
        yield
    finally:
        if old is None:
            del os.environ["SDFS_LINE_STREAM"]
        else:
            os.environ["SDFS_LINE_STREAM"] = old


@pytest.mark.parametrize("shapes", [(16, 16, 16, 16, 16, 16), (16, 16, 20, 20, 16, 16), (24, 24, 16, 16, 16, 16),
                                    (16, 16, 32, 32, 16, 16), (20, 20, 20, 20, 16, 16)])
def test_streamed_line_passes_match_plain_ones(S, shapes):
    """6-D grids (a middle and a last line pass): T with its residual, the linearising T and J.v through the streamed
    forms against the one-tile-per-workgroup forms and the C oracle; twice in a row (the ticket words must be back at
    zero after a launch); a gated launch must leave them alone too."""
    from oracle.c_oracle import COperator
    g = S.GCY(); arr = S.discretize_gcy(g, shapes)
    with plan_env("pair"):
        with stream_env(7):
            Ts = S.KoopmansOperator("gcy", shapes, g.params, arr)
        with stream_env(0):
            Tp = S.KoopmansOperator("gcy", shapes, g.params, arr)
    assert "streamed" in Ts.describe_plan() and "streamed" not in Tp.describe_plan()
    oc = COperator("gcy", shapes, g.params, arr)
    w = wbench(shapes, seed=21)
    want = oc(w)
    for _ in range(2):
        got = Ts(w)
        assert np.max(np.abs(got - want) / want) < APPLY_RTOL
        assert Ts.residual() == pytest.approx(float(np.max(np.abs(want - w))), rel=1e-12)
    np.testing.assert_allclose(Ts(w), Tp(w), rtol=1e-13)
    v = np.random.default_rng(22).standard_normal(shapes)
    js, jp = Ts.jvp(w, v), Tp.jvp(w, v)
    assert np.max(np.abs(js - jp)) <= 1e-12 * np.max(np.abs(jp))
    assert np.max(np.abs(js - oc.jvp(w, v))) <= 1e-11 * np.max(np.abs(jp))
    # solver loops on the streamed kernels: same iterates as on the plain ones (unfused loop: T applications)
    w0 = np.full(shapes, 800.0)
    os.environ["SDFS_SA_FUSED"] = "0"
    try:
        with plan_env("pair"):
            with stream_env(7):
                Tsu = S.KoopmansOperator("gcy", shapes, g.params, arr)
            with stream_env(0):
                Tpu = S.KoopmansOperator("gcy", shapes, g.params, arr)
    finally:
        del os.environ["SDFS_SA_FUSED"]
    xs, ns, i_s = Tsu.solve(w0, "successive_approx", tol=1e-2, record_errors=True, check_every=7)
    xp, n_p, i_p = Tpu.solve(w0, "successive_approx", tol=1e-2, record_errors=True)
    assert ns == n_p and ns > 10
    np.testing.assert_allclose(i_s["errors"], i_p["errors"], rtol=1e-11)
    np.testing.assert_allclose(xs, xp, rtol=1e-12)
    xs, ns, _ = Ts.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
    xp, n_p, _ = Tp.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0)
    assert ns == n_p
    assert np.max(np.abs(xs - xp)) < 1e-9
    for T in (Ts, Tp, Tsu, Tpu):
        T.close()


@pytest.mark.parametrize("shapes", [(16,) * 6, (20, 20, 16, 16, 16, 16), (16, 16, 24, 24), (32, 32, 16, 16)])
def test_sa_with_fp32_intermediates(S, shapes):
    """BASELINE config 5 for the T passes (opts.t_f32; new work, the reference is fp64 only): phase A applies T with
    the intermediates between its passes stored as scaled floats, phase B finishes in fp64.  One application differs
    from the fp64 one by the rounding of two stored floats divided by |theta| (tolerance here: 2e-8 relative); the
    converged iterate satisfies the fp64 stopping rule and sits within tol / (1 - modulus) of the all-fp64 result."""
    model = "gcy" if len(shapes) == 6 else "ssy"
    m = S.GCY() if model == "gcy" else S.SSY()
    arr = (S.discretize_gcy if model == "gcy" else S.discretize_ssy)(m, shapes)
    with plan_env("pair"):
        T = S.KoopmansOperator(model, shapes, m.params, arr)
    w = wbench(shapes, seed=31)
    want = T(w)
    x1, n1, i1 = T.solve(w, "successive_approx", tol=0.0, max_iter=1, t_f32=1)
    assert n1 == 1
    rel = np.max(np.abs(x1 - want) / want)
    assert 0.0 < rel < 2e-8, rel
    assert abs(i1["final_err"] - T.residual()) <= 1e-6 * T.residual()
    w0 = np.full(shapes, 800.0)
    tol = 1e-7
    xa, na, ia = T.solve(w0, "successive_approx", tol=tol, t_f32=1, record_errors=True)
    xb, nb, ib = T.solve(w0, "successive_approx", tol=tol)
    assert ia["status"] == 0 and ia["final_err"] <= tol
    assert len(ia["errors"]) == na
    assert np.max(np.abs(T(xa) - xa)) <= tol * 1.001                      # the fp64 rule holds at the result
    # (phase A's rounding noise, ~w 2^-24 / |theta| per application, lands on every mode of dT -- also on the slowest one,
    # which the fp64 path from w = 800 does not contain: at SSY 32x32x16x16 the fp64 steps shrink by 0.933 per iteration to
    # the end, the t_f32 solve's last 60 by 0.975 (tools/t32_trace.py -> profiles/round4_t32_trace.txt).  Phase B has to
    # damp that mode from the noise level to the tolerance: ln(noise / tol) / ln(1 / 0.975) = 145 iterations where the
    # smooth part needs 116 -- the 30 extra of 529 against 499.  None at GCY 16^6, one at SSY 16x16x24x24: nb // 12 bounds
    # what a spectrum with a slower hidden mode can cost; the result satisfies the fp64 rule either way.)
    assert abs(na - nb) <= max(20, nb // 12), (na, nb)
    assert np.max(np.abs(xa - xb)) < 1e-3 * 1.0, np.max(np.abs(xa - xb))      # both within tol / (1 - modulus) ~ 1e-4 of the fixed point
    T.close()


@pytest.mark.parametrize("shapes", [(16,) * 6, (20, 20, 16, 16, 16, 16), (24, 24, 20, 20, 16, 16), (32, 32, 16, 16, 16, 16)])
def test_newton_with_fp32_mfma_jvp(S, shapes):
    """BASELINE config 5 "on MFMA" (opts.krylov_f32 = 3; new work, the reference is fp64 only): the J.v passes of the inner
    solve on an fp32 LDS tile with v_mfma_f32_16x16x4_f32 (csrc/f32_kernels.hpp), everything the fixed point depends on --
    outer residual, iterate, reductions -- fp64.  The solve must reach the fp64 fixed point to 1e-8 like the fp32-storage
    form, in a comparable number of steps and applications, and the counters must show that the fp32-MFMA kernels ran."""
    m = S.GCY()
    arr = S.discretize_gcy(m, shapes)
    T = S.KoopmansOperator("gcy", shapes, m.params, arr)
    w0 = np.full(shapes, 800.0)
    xs, _, info = T.solve(w0, "newton", tol=1e-11, inner_rtol=1e-9, inner_atol=0.0, max_iter=40)
    assert info["status"] == 0
    res = {}
    for mode in (1, 3):
        T.set_profiling(True); T.reset_counters()
        x, n, info = T.solve(w0, "newton", tol=1e-8, inner_rtol=1e-4, inner_atol=0.0, krylov_f32=mode, max_iter=40)
        names = [c["name"] for c in T.counters() if c["launches"]]
        T.set_profiling(False)
        assert info["status"] == 0 and n < 30, (mode, info)
        assert np.max(np.abs(x - xs)) < 1e-8, (mode, np.max(np.abs(x - xs)))
        res[mode] = (n, info["n_apply"], names)
    assert any(nm.startswith("jvpm32:") for nm in res[3][2]), res[3][2]
    assert not any(nm.startswith("jvpm32:") for nm in res[1][2])
    assert not any(nm.startswith("jvp32:") for nm in res[3][2]), res[3][2]          # every J.v pass ran on the fp32-MFMA kernels
    # ... and BiCGSTAB's p / s updates rode on their first pass (csrc/krylov_kernels.hpp, slice32_jfused_kernel; the
    # fp32-storage loop of mode 1 keeps the separate BLAS-1 kernels: the comparison above is fused against unfused)
    assert any(nm.startswith("jvpm32+p:") for nm in res[3][2]) and any(nm.startswith("jvpm32+s:") for nm in res[3][2]), res[3][2]
    assert not any(nm.startswith("jvpm32:slices") for nm in res[3][2]), res[3][2]
    assert res[3][0] <= res[1][0] + 2 and res[3][1] <= 1.3 * res[1][1], (res[1][:2], res[3][:2])
    T.close()


@pytest.mark.parametrize("shapes", [(16,) * 6, (20, 20, 16, 16, 16, 16)])
def test_bicgstab_vector_updates_in_the_first_jvp_pass(S, shapes):
    """Large grids, fp64 Krylov storage on the pair plan: BiCGSTAB's p = r + beta (p - omega q) and s = r - alpha q are
    formed on the registers of J.v's first pass and <rhat, q> is summed by its last (csrc/krylov_kernels.hpp; the inner
    solve is jax.scipy.sparse.linalg.bicgstab at code/solvers.py:91-93) -- 31 grid streams per iteration instead of 34.
    Held to the loop with separate BLAS-1 kernels on the generic tiles (SDFS_PLAN=classic: another kernel family
    altogether): the same Newton steps, the same number of J.v applications to a few per cent, the same fixed point; the
    counters show which first passes ran."""
    m = S.GCY()
    arr = S.discretize_gcy(m, shapes)
    Tf = S.KoopmansOperator("gcy", shapes, m.params, arr)
    with plan_env("classic"):
        Tp = S.KoopmansOperator("gcy", shapes, m.params, arr)
    w0 = np.full(shapes, 800.0)
    out = {}
    for name, T in (("fused", Tf), ("plain", Tp)):
        T.set_profiling(True); T.reset_counters()
        x, n, info = T.solve(w0, "newton", tol=1e-8, inner_rtol=1e-6, inner_atol=0.0, max_iter=40, record_errors=True)
        names = [c["name"] for c in T.counters() if c["launches"]]
        T.set_profiling(False)
        assert info["status"] == 0
        out[name] = (x, n, info["n_apply"], info["errors"], names)
    assert any(nm.startswith("jvp+p:") for nm in out["fused"][4]) and any(nm.startswith("jvp+s:") for nm in out["fused"][4]), out["fused"][4]
    assert not any(nm.startswith("jvp+") for nm in out["plain"][4])
    assert out["fused"][1] == out["plain"][1], (out["fused"][1:3], out["plain"][1:3])
    # (BiCGSTAB's iteration count to a relative tolerance moves by a few per cent with the summation order of its inner
    # products -- 537 against 549 applications at 16^6 between these two kernel families)
    assert abs(out["fused"][2] - out["plain"][2]) <= max(6, out["plain"][2] // 20), (out["fused"][2], out["plain"][2])
    # the first Newton steps: both inner solves stop at a relative residual of 1e-6, at different iterates of that size
    np.testing.assert_allclose(out["fused"][3][:3], out["plain"][3][:3], rtol=3e-5)
    assert np.max(np.abs(out["fused"][0] - out["plain"][0])) < 1e-8
    Tf.close(); Tp.close()


@pytest.mark.parametrize("shapes", [(16,) * 6, (20, 20, 16, 16, 20, 20)])
def test_anderson_push_on_the_last_pass_of_T(S, shapes):
    """Large grids, Anderson (code/solvers.py:98-124): the pass's push -- r = T x - x, y = x + beta r, <r, r> -- rides on
    the streamed last pass of T (stream_kernels.hpp, LineIO::and_*) instead of a kernel that reads x and T x again.  Held
    to the loop with the separate push (SDFS_NO_AND_PUSH_FUSION=1) pass by pass on a well-conditioned system (a ridge
    that dwarfs the Gram matrix: rounding is not amplified) -- every kind of pass, chunk boundaries -- and to the same
    fixed point with the opt-in relative ridge."""
    m = S.GCY()
    arr = S.discretize_gcy(m, shapes)
    Tf = S.KoopmansOperator("gcy", shapes, m.params, arr)
    os.environ["SDFS_NO_AND_PUSH_FUSION"] = "1"
    try:
        Tu = S.KoopmansOperator("gcy", shapes, m.params, arr)
    finally:
        del os.environ["SDFS_NO_AND_PUSH_FUSION"]
    w0 = np.full(shapes, 800.0)
    for k in (3, 9, 18):
        xf, nf, inf_ = Tf.solve(w0, "anderson", tol=0.0, max_iter=k, ridge=1e9, record_errors=True)
        xu, nu, inu = Tu.solve(w0, "anderson", tol=0.0, max_iter=k, ridge=1e9, record_errors=True)
        assert nf == nu == k
        np.testing.assert_allclose(inf_["errors"], inu["errors"], rtol=1e-10)
        assert np.max(np.abs(xf - xu)) < 1e-9 * np.max(np.abs(xu))
    xf, nf, inf_ = Tf.solve(w0, "anderson", tol=1e-8, max_iter=3000, ridge=-1e-6)
    xu, nu, inu = Tu.solve(w0, "anderson", tol=1e-8, max_iter=3000, ridge=-1e-6)
    assert inf_["status"] == 0 and inu["status"] == 0
    assert np.max(np.abs(xf - xu)) < 1e-4                      # both within tol / (1 - modulus) of the fixed point
    assert nf <= 1.5 * nu + 10 and nu <= 1.5 * nf + 10, (nf, nu)
    Tf.close(); Tu.close()
