"""
The reference's discrete drivers, call for call
(code/ssy/discrete/ssy_wc_ratio.py:216-240, code/gcy/discrete/gcy_wc_ratio.py:319-340):

    SSY() / GCY()  ->  discretize_*  ->  T = lambda w: T_*(w, shapes, params, arrays)
                   ->  w_init = 800 * ones(shapes)  ->  solver(T, w_init, algorithm=algo)

``T`` is the reference's plain closure, not an operator object: ``solver`` recognises that it is
one device operator (solvers._resolve_operator) and runs the whole iteration on the GPU.
The reference moves the arrays to the device with ``jax.device_put``; here the operator uploads
them once when the closure is first called.
"""
import time

import numpy as np

from .models import SSY, GCY
from .discretize import discretize_ssy, discretize_gcy
from .operators import T_ssy, T_gcy
from .solvers import solver


def test_compute_wc_ratio_ssy(shapes=(2, 3, 4, 5), algo="successive_approx"):
    """Solve a small version of the model using T_ssy."""
    ssy = SSY()

    # Build discrete rep of SSY
    params = ssy.params
    arrays = discretize_ssy(ssy, shapes)

    # Marginalize T
    T = lambda w: T_ssy(w, shapes, params, arrays)

    # Call the solver
    init_val = 800.0
    w_init = np.ones(shapes) * init_val
    t0 = time.time()
    w_star = solver(T, w_init, algorithm=algo)
    t = time.time() - t0
    print(f"Computed solution in {t} seconds.")

    return w_star


def test_compute_wc_ratio_gcy(shapes=(3, 3, 3, 3, 3, 3), algo="successive_approx"):
    """Solve a small version of the model using T_gcy."""
    gcy = GCY()

    # Build discrete rep of GCY
    params = gcy.params
    arrays = discretize_gcy(gcy, shapes)

    # Marginalize T
    T = lambda w: T_gcy(w, shapes, params, arrays)

    # Call the solver
    init_val = 800.0
    w_init = np.ones(shapes) * init_val
    w_star = solver(T, w_init, algorithm=algo)

    return w_star


# pytest must not collect the reference-named drivers when this module is imported into a test file
test_compute_wc_ratio_ssy.__test__ = False
test_compute_wc_ratio_gcy.__test__ = False
