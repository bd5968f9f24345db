"""GPU parity of the continuous-state operator (csrc/cont_kernel.hpp) through the C ABI:
golden vectors made by the reference's own modules, the numpy oracle, and the solver loops."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "cont_*_sd*.npz")))
RTOL = 1e-12          # one application, fp64; measured ~1e-15 (different summation order only)


@pytest.fixture(scope="module")
def S():
    import sdfs_via_autodiff_amd as S
    return S


def load(fn):
    z = np.load(fn)
    grids = tuple(z[f"grid{i}"] for i in range(len(z["sizes"])))
    return z, grids


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_T_quadrature_and_monte_carlo_vs_reference_golden(S, fn):
    z, grids = load(fn)
    Tq = S.T_fun_factory((z["params"], grids, z["nodes"], z["weights"]), "quadrature", z["w"].size)
    np.testing.assert_allclose(Tq(z["w"]), z["T_quad"], rtol=RTOL)
    Tm = S.T_fun_factory((z["params"], grids, z["mc_draws"]), "monte_carlo", z["w"].size)
    np.testing.assert_allclose(Tm(z["w"]), z["T_mc"], rtol=RTOL)
    assert Tq.residual() == pytest.approx(np.max(np.abs(z["T_quad"] - z["w"])), rel=1e-12)


SSY_FILES = [f for f in FILES if "cont_ssy" in f]
GCY_FILES = [f for f in FILES if "cont_gcy" in f]


@pytest.mark.parametrize("fn", SSY_FILES[:1] + GCY_FILES[:1], ids=["ssy", "gcy"])
def test_lin_interp_vs_reference_golden(S, fn):
    z, grids = load(fn)
    np.testing.assert_allclose(S.lin_interp(z["x_query"], z["w"], grids), z["interp"], rtol=1e-14)
    f = S.construct_wstar_callable(z["w"], grids)
    np.testing.assert_allclose(f(z["x_query"]), z["interp"], rtol=1e-14)
    # on the grid points themselves interpolation is the identity
    mesh = np.stack([m.ravel() for m in np.meshgrid(*grids, indexing="ij")])
    np.testing.assert_allclose(S.lin_interp(mesh, z["w"], grids), z["w"].ravel(), rtol=1e-13)


def test_jvp_vs_oracle_and_finite_differences(S):
    from oracle import continuous as OC
    for fn, model in ((SSY_FILES[2], "ssy"), (GCY_FILES[2], "gcy")):
        z, grids = load(fn)
        T = S.T_fun_factory((z["params"], grids, z["nodes"], z["weights"]), "quadrature", z["w"].size)
        v = np.random.default_rng(5).standard_normal(z["w"].shape)
        got = T.jvp(z["w"], v)
        want = OC.jvp_factory(model, z["params"], grids, z["nodes"], z["weights"])(z["w"], v)
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-13)
        eps = 1e-6
        fd = (T(z["w"] + eps * v) - T(z["w"] - eps * v)) / (2 * eps)
        np.testing.assert_allclose(got, fd, rtol=5e-7, atol=1e-8)


def test_driver_successive_approx_matches_reference_run(S, tmp_path):
    """wc_ratio_continuous end to end with the reference's defaults on a 3x3x3x4 grid; the expected
    fixed point comes from the reference's own driver + solver (make_golden.py)."""
    z = np.load(os.path.join(GOLD, "cont_ssy_driver_3x3x3x4.npz"))
    fn = str(tmp_path / "w_star_data.npy")
    grids, w_star = S.wc_ratio_continuous(S.SSY(), 3, 3, 3, 4, num_std_devs=3.2, d=3,
                                          algorithm="successive_approx", verbose=False, filename=fn)
    for i, g in enumerate(grids):
        np.testing.assert_array_equal(g, z[f"grid{i}"])
    np.testing.assert_allclose(w_star, z["w_star"], rtol=0, atol=1e-7)
    g2, w2 = S.load_wstar(fn)
    np.testing.assert_array_equal(w2, w_star)
    f = S.construct_wstar_callable(datafile=fn)
    x0 = np.array([[g[1]] for g in grids])
    assert f(x0)[0] == pytest.approx(w_star[1, 1, 1, 1], rel=1e-13)


def test_newton_and_anderson_reach_the_same_fixed_point(S):
    ssy = S.SSY()
    grids = S.build_grid(ssy, 4, 4, 4, 6)
    nodes, weights = S.qnwnorm([3] * 4)
    T = S.T_fun_factory((np.array(ssy.params), grids, nodes.T.copy(), weights), "quadrature", 4 * 4 * 4 * 6)
    w0 = np.ones((4, 4, 4, 6))
    xs, ns, _ = T.solve(w0, "successive_approx", tol=1e-9)
    xn, nn, info = T.solve(w0, "newton", tol=1e-9, inner_rtol=1e-8, inner_atol=0.0)
    xa, na, _ = T.solve(w0, "anderson", tol=1e-9)
    assert nn < 30 and na < ns
    np.testing.assert_allclose(xn, xs, rtol=0, atol=2e-5)       # SA stops at step 1e-9 with rate ~0.999
    np.testing.assert_allclose(xa, xn, rtol=0, atol=1e-5)
    assert np.max(np.abs(T(xn) - xn)) < 1e-8


def test_gcy_continuous_solve_and_errors(S):
    gcy = S.GCY()
    # with num_std_devs = 3.2 this coarse grid has no fixed point (the iteration grows without bound,
    # in the oracle as well); 2.0 converges
    grids = S.build_grid(gcy, 3, 3, 3, 3, 4, 4, num_std_devs=2.0)
    nodes, weights = S.qnwnorm([2] * 6)
    T = S.T_fun_factory((np.array(gcy.params), grids, nodes.T.copy(), weights), "quadrature", 3 ** 4 * 16)
    # Newton from w = 1 overshoots into w < 0 on this coarse grid (so would the reference's): take
    # successive-approximation steps first, as the reference's drivers do, then polish
    xs, ns, _ = T.solve(np.ones(T.shapes), "successive_approx", tol=1e-2, max_iter=50000)
    x, n, info = T.solve(xs, "newton", tol=1e-8, inner_rtol=1e-8, inner_atol=0.0)
    assert np.max(np.abs(T(x) - x)) < 1e-7 and np.all(x > 1) and n <= 10
    with pytest.raises(ValueError):
        S.T_fun_factory((np.array(gcy.params), grids, nodes.T.copy(), weights), "quadrature", 7)
    with pytest.raises(KeyError):
        S.T_fun_factory((np.array(gcy.params), grids, nodes.T.copy(), weights), "simpson", 1)
    with pytest.raises(S.SdfsError):
        S.ContinuousOperator(np.array(gcy.params), [np.array([0.0])] * 6, nodes.T.copy(), weights)
