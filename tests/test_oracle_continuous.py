"""Oracle (oracle/continuous.py) against the golden vectors made by the reference's own
continuous-state modules (tests/golden/make_golden.py: continuous_fixtures)."""
import glob
import os

import numpy as np
import pytest

from oracle import continuous as OC

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "cont_*_sd*.npz")))


def load(fn):
    z = np.load(fn)
    model = "ssy" if "cont_ssy" in fn else "gcy"
    grids = tuple(z[f"grid{i}"] for i in range(len(z["sizes"])))
    return model, z, grids


def test_fixture_inventory():
    assert len(FILES) == 8


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_grids_interp_and_T(fn):
    model, z, grids = load(fn)
    build = OC.build_grid_ssy if model == "ssy" else OC.build_grid_gcy
    mine = build(tuple(z["params"]), tuple(int(s) for s in z["sizes"]), float(z["num_std_devs"]))
    for a, b in zip(mine, grids):
        np.testing.assert_array_equal(a, b)
    nodes, weights = OC.qnwnorm([int(z["d"])] * len(grids))
    np.testing.assert_array_equal(nodes.T, z["nodes"])
    np.testing.assert_allclose(weights.sum(), 1.0, rtol=1e-14)
    np.testing.assert_allclose(OC.lin_interp(z["x_query"], z["w"], grids), z["interp"], rtol=1e-15)
    Tq = OC.T_fun_factory(model, z["params"], grids, z["nodes"], z["weights"])
    np.testing.assert_allclose(Tq(z["w"]), z["T_quad"], rtol=1e-13)
    Tm = OC.T_fun_factory(model, z["params"], grids, z["mc_draws"], None)
    np.testing.assert_allclose(Tm(z["w"]), z["T_mc"], rtol=1e-13)


def test_jvp_matches_finite_difference():
    model, z, grids = load(FILES[0])
    T = OC.T_fun_factory(model, z["params"], grids, z["nodes"], z["weights"])
    J = OC.jvp_factory(model, z["params"], grids, z["nodes"], z["weights"])
    w = z["w"]
    v = np.random.default_rng(2).standard_normal(w.shape)
    eps = 1e-6
    fd = (T(w + eps * v) - T(w - eps * v)) / (2 * eps)
    np.testing.assert_allclose(J(w, v), fd, rtol=2e-7, atol=1e-9)


def test_hermite_rule_moments():
    nodes, weights = OC.qnwnorm([5])
    for k, want in [(0, 1.0), (1, 0.0), (2, 1.0), (4, 3.0), (6, 15.0), (8, 105.0)]:
        assert abs(np.dot(weights, nodes[:, 0] ** k) - want) < 1e-12
