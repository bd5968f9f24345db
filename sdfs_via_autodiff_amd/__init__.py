"""
sdfs_via_autodiff_amd -- MI355X-native solver for the wealth-consumption-ratio
fixed point of the SSY / GCY long-run-risk models (the hot path of
jstac/sdfs_via_autodiff): hand-written HIP for gfx950 behind the reference's own
``T(w)`` / ``solver(f, x_init, algorithm)`` call shapes.

    from sdfs_via_autodiff_amd import SSY, discretize_ssy, ssy_operator, solver
    ssy = SSY(); shapes = (15, 15, 15, 15)
    T = ssy_operator(shapes, ssy.params, discretize_ssy(ssy, shapes))
    w_star = solver(T, 800.0 * np.ones(shapes), algorithm="newton")

Importing the package loads libsdfs_hip.so and fails loudly if it has not been
built; there is no CPU fallback for the operator.
"""
from .models import SSY, GCY
from .discretize import rouwenhorst, tauchen, discretize_ssy, discretize_gcy
from .operators import (KoopmansOperator, ssy_operator, gcy_operator, T_ssy, T_gcy)
from .solvers import (successive_approx, newton_solver, anderson_solver,
                      fixed_point_via_gradient_decent, solvers, solver,
                      default_tolerance, default_max_iter)
from .drivers import test_compute_wc_ratio_ssy, test_compute_wc_ratio_gcy
from .loglinear import wc_loglinear_factory, loglinear_guess
from .continuous import (ContinuousOperator, build_grid, T_fun_factory, wc_ratio_continuous, qnwnorm,
                         lin_interp, vals_to_coords, construct_wstar_callable, save_wstar, load_wstar)
from .single_index import (DenseOperator, compute_H_single_index, discretize_single_index, single_index_T,
                           single_to_multi, multi_to_single)
from ._lib import SdfsError, LIB_PATH

__all__ = ["SSY", "GCY", "rouwenhorst", "tauchen", "discretize_ssy", "discretize_gcy",
           "KoopmansOperator", "ssy_operator", "gcy_operator", "T_ssy", "T_gcy",
           "successive_approx", "newton_solver", "anderson_solver",
           "fixed_point_via_gradient_decent", "solvers", "solver",
           "default_tolerance", "default_max_iter", "test_compute_wc_ratio_ssy", "test_compute_wc_ratio_gcy",
           "wc_loglinear_factory", "loglinear_guess",
           "ContinuousOperator", "build_grid", "T_fun_factory", "wc_ratio_continuous", "qnwnorm",
           "lin_interp", "vals_to_coords", "construct_wstar_callable", "save_wstar", "load_wstar",
           "DenseOperator", "compute_H_single_index", "discretize_single_index", "single_index_T",
           "single_to_multi", "multi_to_single",
           "SdfsError", "LIB_PATH"]
