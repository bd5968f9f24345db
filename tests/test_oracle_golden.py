"""
Pin the oracle (CPU restatement) against the golden vectors captured from the
reference's own modules (tests/golden/make_golden.py) and against the one
recorded output in the reference (sandpit.ipynb:41-44).
"""
import numpy as np
import pytest

from conftest import load_golden, golden_arrays
from oracle import models, ssy, gcy, solvers

SSY_SHAPES = [(3, 3, 3, 3), (2, 3, 4, 5), (4, 7, 6, 5), (10, 10, 10, 10)]
GCY_SHAPES = [(2, 3, 2, 3, 2, 3), (3,) * 6, (2, 3, 4, 5, 6, 7)]


def tag(s):
    return "x".join(map(str, s))


def test_params_match_reference():
    g = load_golden("ssy_3x3x3x3.npz")
    assert np.array_equal(g["params"], np.array(models.ssy_params()))
    g = load_golden("gcy_3x3x3x3x3x3.npz")
    assert np.array_equal(g["params"], np.array(models.gcy_params()))


@pytest.mark.parametrize("shapes", SSY_SHAPES)
def test_discretize_ssy(shapes):
    g = load_golden(f"ssy_{tag(shapes)}.npz")
    got = ssy.discretize_ssy(models.ssy_params(), shapes)
    for a, b in zip(got, golden_arrays(g, "ssy")):
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=1e-14, atol=1e-300)


@pytest.mark.parametrize("shapes", GCY_SHAPES)
def test_discretize_gcy(shapes):
    g = load_golden(f"gcy_{tag(shapes)}.npz")
    got = gcy.discretize_gcy(models.gcy_params(), shapes)
    for a, b in zip(got, golden_arrays(g, "gcy")):
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=1e-14, atol=1e-300)


@pytest.mark.parametrize("shapes", SSY_SHAPES)
def test_T_ssy_factorised_vs_reference(shapes):
    g = load_golden(f"ssy_{tag(shapes)}.npz")
    p = tuple(g["params"])
    arr = golden_arrays(g, "ssy")
    for wname, tname in [("w_rand", "T_rand"), (None, "T_800")]:
        w = g[wname] if wname else np.full(shapes, 800.0)
        np.testing.assert_allclose(ssy.T_ssy_factorised(w, shapes, p, arr), g[tname], rtol=2e-14)
    if "Tloops_rand" in g:
        np.testing.assert_allclose(ssy.T_ssy_factorised(g["w_rand"], shapes, p, arr),
                                   g["Tloops_rand"], rtol=2e-14)


@pytest.mark.parametrize("shapes", [(3, 3, 3, 3), (2, 3, 4, 5)])
def test_T_ssy_literal_and_loops(shapes):
    g = load_golden(f"ssy_{tag(shapes)}.npz")
    p = tuple(g["params"])
    arr = golden_arrays(g, "ssy")
    np.testing.assert_allclose(ssy.T_ssy(g["w_rand"], shapes, p, arr), g["T_rand"], rtol=2e-14)
    np.testing.assert_allclose(ssy.T_ssy_loops(g["w_rand"], shapes, p, arr), g["T_rand"], rtol=2e-14)


@pytest.mark.parametrize("shapes", GCY_SHAPES)
def test_T_gcy_factorised_vs_reference(shapes):
    g = load_golden(f"gcy_{tag(shapes)}.npz")
    p = tuple(g["params"])
    arr = golden_arrays(g, "gcy")
    for wname, tname in [("w_rand", "T_rand"), (None, "T_800")]:
        w = g[wname] if wname else np.full(shapes, 800.0)
        np.testing.assert_allclose(gcy.T_gcy_factorised(w, shapes, p, arr), g[tname], rtol=2e-14)
    if "Tloops_rand" in g:
        np.testing.assert_allclose(gcy.T_gcy_factorised(g["w_rand"], shapes, p, arr),
                                   g["Tloops_rand"], rtol=2e-14)


def test_T_gcy_literal_and_loops():
    shapes = (2, 3, 2, 3, 2, 3)
    g = load_golden(f"gcy_{tag(shapes)}.npz")
    p = tuple(g["params"])
    arr = golden_arrays(g, "gcy")
    np.testing.assert_allclose(gcy.T_gcy(g["w_rand"], shapes, p, arr), g["T_rand"], rtol=2e-14)
    np.testing.assert_allclose(gcy.T_gcy_loops(g["w_rand"], shapes, p, arr), g["T_rand"], rtol=2e-14)


@pytest.mark.parametrize("model,shapes", [("ssy", (3, 4, 2, 5)), ("gcy", (2, 3, 2, 3, 2, 3))])
def test_jvp_matches_finite_difference(model, shapes):
    rng = np.random.default_rng(1)
    if model == "ssy":
        p = models.ssy_params(); arr = ssy.discretize_ssy(p, shapes)
        T, J = ssy.T_ssy_factorised, ssy.jvp_ssy
    else:
        p = models.gcy_params(); arr = gcy.discretize_gcy(p, shapes)
        T, J = gcy.T_gcy_factorised, gcy.jvp_gcy
    w = 400 + 500 * rng.random(shapes)
    v = rng.standard_normal(shapes)
    h = 1e-4
    fd = (T(w + h * v, shapes, p, arr) - T(w - h * v, shapes, p, arr)) / (2 * h)
    np.testing.assert_allclose(J(w, v, shapes, p, arr), fd, rtol=1e-6, atol=1e-9)


def test_sa_ssy_iteration_counts_and_fixed_point():
    shapes = (3, 3, 3, 3)
    g = load_golden("sa_ssy_3x3x3x3.npz")
    p = models.ssy_params(); arr = ssy.discretize_ssy(p, shapes)
    T = lambda w: ssy.T_ssy_factorised(w, shapes, p, arr)
    errs = []
    x, n = solvers.successive_approx(T, np.full(shapes, 800.0), tol=1e-8, verbose=False, errors=errs)
    assert n == int(g["n_1e8"]) == 12253
    np.testing.assert_allclose(x, g["w_1e8"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(errs[:50], g["errors"][:50], rtol=1e-10)
    assert abs(x[0, 0, 0, 0] - 829.829923729656) < 1e-9


def test_sa_gcy_fixed_point():
    shapes = (3,) * 6
    g = load_golden("sa_gcy_3x3x3x3x3x3.npz")
    p = models.gcy_params(); arr = gcy.discretize_gcy(p, shapes)
    T = lambda w: gcy.T_gcy_factorised(w, shapes, p, arr)
    x, n = solvers.successive_approx(T, np.full(shapes, 800.0), tol=1e-7, verbose=False)
    assert n == int(g["n_1e7"]) == 7520
    np.testing.assert_allclose(x, g["w_1e7"], rtol=0, atol=1e-9)


def test_newton_reproduces_sandpit_trace():
    """sandpit.ipynb:41-44 -- loose: the reference's inner solves are inexact."""
    g = load_golden("sandpit_trace.npz")
    shapes = tuple(int(s) for s in g["shapes"])
    p = models.ssy_params(); arr = ssy.discretize_ssy(p, shapes)
    T = lambda w: ssy.T_ssy_factorised(w, shapes, p, arr)
    J = lambda w, v: ssy.jvp_ssy(w, v, shapes, p, arr)
    errs = []
    x, n = solvers.newton_solver(T, np.full(shapes, 800.0), verbose=False, jvp=J, errors=errs)
    # inner solves stop at |r| <= 1e-5 |b| in both codes but along different
    # rounding paths, and each Newton step amplifies the previous step's slack
    for got, want, rtol in zip(errs[:4], g["errors"], (2e-6, 5e-6, 5e-5, 2e-3)):
        assert abs(got - want) <= rtol * want, (got, want)
    # reference quirk: stops once |g|_2 <= bicgstab_atol, residual ~1e-5
    assert np.max(np.abs(T(x) - x)) < 1e-3


def test_anderson_and_newton_reach_the_same_fixed_point():
    shapes = (3, 3, 3, 3)
    p = models.ssy_params(); arr = ssy.discretize_ssy(p, shapes)
    T = lambda w: ssy.T_ssy_factorised(w, shapes, p, arr)
    J = lambda w, v: ssy.jvp_ssy(w, v, shapes, p, arr)
    xa, na = solvers.anderson_solver(T, np.full(shapes, 800.0), tol=1e-8, verbose=False)
    xs = solvers.newton_polish(T, J, xa.copy())
    assert na < 10000
    # distance to the fixed point ~ residual / (1 - modulus), modulus ~ 0.9988
    np.testing.assert_allclose(xa, xs, rtol=0, atol=1e-5)
    assert np.max(np.abs(T(xs) - xs)) < 1e-11


def test_solver_front_end_matches_reference_default():
    g = load_golden("solver_front_ssy_2x3x4x5.npz")
    shapes = (2, 3, 4, 5)
    p = models.ssy_params(); arr = ssy.discretize_ssy(p, shapes)
    T = lambda w: ssy.T_ssy_factorised(w, shapes, p, arr)
    x = solvers.solver(T, np.full(shapes, 800.0), algorithm="successive_approx", verbose=False)
    np.testing.assert_allclose(x, g["w"], rtol=0, atol=1e-9)
