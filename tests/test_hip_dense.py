"""GPU parity of the single-index dense form (csrc/dense_kernel.hpp) through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def S():
    import sdfs_via_autodiff_amd as S
    return S


@pytest.mark.parametrize("name", ["dense_ssy_2x3x2x3", "dense_ssy_3x2x4x3"])
def test_single_index_T_vs_reference_golden(S, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    np.testing.assert_allclose(S.single_index_T(z["w"], z["H"], tuple(z["params"])), z["T_single"], rtol=1e-13)
    shapes = tuple(int(s) for s in z["shapes"])
    H = S.compute_H_single_index(S.SSY(), shapes)
    np.testing.assert_allclose(S.single_index_T(z["w"], H, tuple(z["params"])), z["T_multi"].ravel(), rtol=1e-12)


def test_dense_and_factorised_operators_agree(S):
    """The cross-check the dense form exists for: T, JVP and the fixed point by two independent kernels."""
    for model, shapes in (("ssy", (5, 4, 6, 7)), ("gcy", (3, 2, 3, 2, 3, 4))):
        m = S.SSY() if model == "ssy" else S.GCY()
        arr = (S.discretize_ssy if model == "ssy" else S.discretize_gcy)(m, shapes)
        Tf = S.KoopmansOperator(model, shapes, m.params, arr)
        H = S.compute_H_single_index(m, shapes)
        p = m.params
        beta = p[0]
        theta = (1 - p[1]) / (1 - 1 / p[2]) if model == "ssy" else (1 - p[2]) / (1 - 1 / p[1])
        Td = S.DenseOperator(H, beta, theta)
        rng = np.random.default_rng(3)
        w = 400 + 500 * rng.random(shapes)
        v = rng.standard_normal(shapes)
        np.testing.assert_allclose(Td(w).reshape(shapes), Tf(w), rtol=1e-12)
        np.testing.assert_allclose(Td.jvp(w, v).reshape(shapes), Tf.jvp(w, v), rtol=1e-10, atol=1e-12)
        assert Td.residual() == pytest.approx(np.max(np.abs(Tf(w) - w)), rel=1e-10)
        xd, nd, _ = Td.solve(np.full(Td.shapes, 800.0), "newton", tol=1e-9, inner_rtol=1e-8, inner_atol=0.0)
        xf, nf, _ = Tf.solve(np.full(shapes, 800.0), "newton", tol=1e-9, inner_rtol=1e-8, inner_atol=0.0)
        np.testing.assert_allclose(xd.reshape(shapes), xf, rtol=0, atol=1e-8)
        assert nd == nf


def test_dense_driver_and_errors(S):
    from sdfs_via_autodiff_amd.single_index import test_compute_wc_ratio_single_index as drive
    w = drive(3, 3, 3, 3)
    g = np.load(os.path.join(GOLD, "sa_ssy_3x3x3x3.npz"))
    # the reference's Newton defaults (inner atol 1e-4) stop once the inner solve returns a zero step:
    # a few 1e-4 from the fixed point, like the reference's own newton_solver on this grid
    np.testing.assert_allclose(w, g["w_1e8"], rtol=0, atol=3e-3)
    with pytest.raises(ValueError):
        S.DenseOperator(np.ones((3, 4)), 0.99, -16.0)
    with pytest.raises(S.SdfsError):
        S.DenseOperator(np.eye(3), 0.99, 0.0)
    Td = S.DenseOperator(np.eye(4) * 1e-40, 0.99, -16.0)
    with pytest.raises(ValueError):
        Td(np.ones(5))
