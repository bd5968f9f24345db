"""The C/OpenMP oracle (CPU baseline) agrees with the numpy oracle that is pinned to the reference."""
import numpy as np
import pytest

from oracle import models, ssy, gcy
from oracle.c_oracle import COperator, num_threads


@pytest.mark.parametrize("model,shapes", [("ssy", (4, 7, 6, 5)), ("ssy", (10, 10, 10, 10)),
                                          ("gcy", (2, 3, 4, 5, 6, 7)), ("gcy", (5,) * 6)])
def test_c_oracle_matches_numpy_oracle(model, shapes):
    rng = np.random.default_rng(3)
    if model == "ssy":
        p = models.ssy_params(); arr = ssy.discretize_ssy(p, shapes)
        T, J = ssy.T_ssy_factorised, ssy.jvp_ssy
    else:
        p = models.gcy_params(); arr = gcy.discretize_gcy(p, shapes)
        T, J = gcy.T_gcy_factorised, gcy.jvp_gcy
    arr = list(arr)
    # make the conditional tensors genuinely conditional
    for i in ((7,) if model == "ssy" else (1, 3)):
        q = rng.random(arr[i].shape) + 0.05
        arr[i] = q / q.sum(axis=-1, keepdims=True)
    op = COperator(model, shapes, p, arr)
    w = 400 + 500 * rng.random(shapes)
    v = rng.standard_normal(shapes)
    np.testing.assert_allclose(op(w), T(w, shapes, p, arr), rtol=1e-13)
    np.testing.assert_allclose(op.jvp(w, v), J(w, v, shapes, p, arr), rtol=1e-11, atol=1e-13)
    assert num_threads() >= 1
