"""
The reference's two discrete drivers (code/ssy/discrete/ssy_wc_ratio.py:216-240,
code/gcy/discrete/gcy_wc_ratio.py:319-340) under their own names and signatures.

Both do the same five things -- calibration object, discretisation, the closure
``T = lambda w: T_*(w, shapes, params, arrays)``, a constant start at 800 and ``solver(T, w_init, algorithm=algo)`` --
so one helper carries them.  The closure is deliberately the reference's plain lambda, not an operator object:
``solver`` finds the device operator behind it (solvers._resolve_operator) and runs the whole iteration on the GPU.
Where the reference calls ``jax.device_put`` on the arrays, the operator here uploads them once, on first use.
"""
import time

import numpy as np

from .models import SSY, GCY
from .discretize import discretize_ssy, discretize_gcy
from .operators import T_ssy, T_gcy
from .solvers import solver

START_LEVEL = 800.0      # ssy_wc_ratio.py:233, gcy_wc_ratio.py:336


def _solve_discrete(model_cls, discretize, T_model, shapes, algo, timed):
    model = model_cls()
    params, arrays = model.params, discretize(model, shapes)
    T = lambda w: T_model(w, shapes, params, arrays)           # what the reference hands to its solvers
    w_init = np.full(shapes, START_LEVEL)
    t0 = time.time()
    w_star = solver(T, w_init, algorithm=algo)
    if timed:                                                   # only the SSY driver reports its time (qe.tic / qe.toc there)
        print(f"Computed solution in {time.time() - t0} seconds.")
    return w_star


def test_compute_wc_ratio_ssy(shapes=(2, 3, 4, 5), algo="successive_approx"):
    """Wealth-consumption ratio of the SSY model on a small grid (the reference's smoke solve)."""
    return _solve_discrete(SSY, discretize_ssy, T_ssy, shapes, algo, timed=True)


def test_compute_wc_ratio_gcy(shapes=(3, 3, 3, 3, 3, 3), algo="successive_approx"):
    """Wealth-consumption ratio of the GCY model on a small grid."""
    return _solve_discrete(GCY, discretize_gcy, T_gcy, shapes, algo, timed=False)


# pytest must not collect the reference-named drivers when this module is imported into a test file
test_compute_wc_ratio_ssy.__test__ = False
test_compute_wc_ratio_gcy.__test__ = False
