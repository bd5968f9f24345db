"""
Continuous-state wealth-consumption ratio on MI355X: mirror of
  code/ssy/continuous_junnan/ssy_wc_ratio_continuous.py  (build_grid :20-59, T_fun_factory :156-226,
      wc_ratio_continuous :229-299, construct_wstar_callable :304-326)
  code/gcy/continuous/gcy_wc_ratio_continuous.py         (build_grid :23-71, T_fun_factory :190-260,
      wc_ratio_continuous :264-340, construct_wstar_callable :342-364)
  code/utils.py:6-23                                      (vals_to_coords, lin_interp)

The two reference modules define the same function names for the two models; here one function
serves both and looks at the model object (``SSY`` / ``GCY``) or at the number of grids (4 / 6).
The expectation (quadrature or Monte Carlo over multilinear interpolation), the aggregator, the
JVP and the fixed-point loops run in libsdfs_hip.so (csrc/cont_kernel.hpp); the host builds grids,
Gauss-Hermite nodes and files.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import lib, check
from .models import SSY, GCY
from .operators import KoopmansOperator, _as_f64
from .solvers import solver


# -- quantecon.quad.qnwnorm ([d]*dim, N(0, I)): third-party in the reference, restated ----------
def qnwnorm(n):
    """Gauss-Hermite nodes (prod(n), dim) and weights (prod(n),) for independent N(0, 1) shocks;
    first dimension varies fastest (quantecon's gridmake / ckron(*weights[::-1]) order)."""
    n = [int(k) for k in np.atleast_1d(n)]
    M = int(np.prod(n))
    nodes = np.empty((M, len(n)))
    weights = np.ones(M)
    rep = 1
    for d, k in enumerate(n):
        x, w = np.polynomial.hermite.hermgauss(k)
        idx = (np.arange(M) // rep) % k
        nodes[:, d] = (x * np.sqrt(2.0))[idx]
        weights *= (w / np.sqrt(np.pi))[idx]
        rep *= k
    return nodes, weights


def _model_name(model_or_params, ngrids=None):
    if isinstance(model_or_params, SSY):
        return "ssy"
    if isinstance(model_or_params, GCY):
        return "gcy"
    n = ngrids if ngrids is not None else {13: 4, 18: 6}.get(len(model_or_params))
    if n == 4:
        return "ssy"
    if n == 6:
        return "gcy"
    raise ValueError("cannot tell SSY (4 grids, 13 params) from GCY (6 grids, 18 params)")


def build_grid(model, *sizes, num_std_devs=3.2):
    """Grids for linear interpolation: SSY (h_λ, h_c, h_z, z) -- ssy_wc_ratio_continuous.py:20-59;
    GCY (h_λ, h_c, h_z, h_zπ, z, z_π) -- gcy_wc_ratio_continuous.py:23-71.  A trailing positional
    value beyond the grid sizes is num_std_devs, as in the reference's positional calls."""
    name = _model_name(model)
    nd = 4 if name == "ssy" else 6
    if len(sizes) == nd + 1:
        num_std_devs = sizes[-1]
        sizes = sizes[:-1]
    if len(sizes) != nd:
        raise TypeError(f"build_grid needs {nd} grid sizes for {name.upper()}")
    sizes = [int(s) for s in sizes]
    p = model.params
    if name == "ssy":
        (β, γ, ψ, μ_c, ρ, ϕ_z, ϕ_c, ρ_z, ρ_c, ρ_λ, s_z, s_c, s_λ) = p
        grids = []
        for s, r, n in zip((s_λ, s_c, s_z), (ρ_λ, ρ_c, ρ_z), sizes[:3]):
            g_max = num_std_devs * np.sqrt(s ** 2 / (1 - r ** 2))
            grids.append(np.linspace(-g_max, g_max, n))
        h_z_max = num_std_devs * np.sqrt(s_z ** 2 / (1 - ρ_z ** 2))
        z_max = num_std_devs * (ϕ_z * np.exp(h_z_max))
        grids.append(np.linspace(-z_max, z_max, sizes[3]))
        return tuple(grids)
    (β, ψ, γ, ρ_λ, s_λ, μ_c, φ_c, ρ, ρ_π, φ_z, ρ_c, s_c, ρ_z, s_z, ρ_ππ, φ_zπ, ρ_zπ, s_zπ) = p
    grids = []
    for s, r, n in zip((s_λ, s_c, s_z, s_zπ), (ρ_λ, ρ_c, ρ_z, ρ_zπ), sizes[:4]):
        g_max = num_std_devs * np.sqrt(s ** 2 / (1 - r ** 2))
        grids.append(np.linspace(-g_max, g_max, n))
    h_zπ_max = num_std_devs * np.sqrt(s_zπ ** 2 / (1 - ρ_zπ ** 2))
    σ_zπ_max = φ_zπ * np.exp(h_zπ_max)
    zπ_max = num_std_devs * np.sqrt(σ_zπ_max ** 2 / (1 - ρ_ππ ** 2))
    zπ_grid = np.linspace(-zπ_max, zπ_max, sizes[5])
    h_z_max = num_std_devs * np.sqrt(s_z ** 2 / (1 - ρ_z ** 2))
    σ_z_max = φ_z * np.exp(h_z_max)
    # the reference's loop variable shadows ρ (gcy_wc_ratio_continuous.py:47): by :68-69 it is ρ_zπ.
    # Kept, so that grids -- and therefore results and saved files -- agree with the reference's.
    ρ_as_in_reference = ρ_zπ
    z_max = (ρ_π * zπ_grid[-1] + num_std_devs * σ_z_max) / (1 - ρ_as_in_reference)
    z_min = (ρ_π * zπ_grid[0] - num_std_devs * σ_z_max) / (1 - ρ_as_in_reference)
    grids.append(np.linspace(z_min, z_max, sizes[4]))
    grids.append(zπ_grid)
    return tuple(grids)


# -- code/utils.py --------------------------------------------------------------------------
def vals_to_coords(grids, x_vals):
    intervals = np.asarray([g[1] - g[0] for g in grids]).reshape(-1, 1)
    low_bounds = np.asarray([g[0] for g in grids]).reshape(-1, 1)
    return (np.asarray(x_vals) - low_bounds) / intervals


def lin_interp(x, fun_vals, grids, device=0):
    """Multilinear interpolation of ``fun_vals`` (on ``grids``) at the columns of x (dim, N) -- on the GPU."""
    grids = [_as_f64(g) for g in grids]
    fun_vals = _as_f64(fun_vals)
    x = _as_f64(x)
    squeeze = x.ndim == 1
    if squeeze:
        x = x.reshape(-1, 1)
    nd = len(grids)
    if x.shape[0] != nd or fun_vals.shape != tuple(len(g) for g in grids):
        raise ValueError("x must be (dim, N) and fun_vals must live on the grids")
    out = np.empty(x.shape[1])
    shp = (C.c_int64 * nd)(*[len(g) for g in grids])
    gp = (C.POINTER(C.c_double) * nd)(*[g.ctypes.data_as(C.POINTER(C.c_double)) for g in grids])
    rc = lib.sdfs_lin_interp(int(device), nd, shp, gp, fun_vals.ctypes.data_as(C.POINTER(C.c_double)),
                             x.ctypes.data_as(C.POINTER(C.c_double)), x.shape[1],
                             out.ctypes.data_as(C.POINTER(C.c_double)))
    check(rc)
    return out[0] if squeeze else out


# -- the operator ----------------------------------------------------------------------------
class ContinuousOperator(KoopmansOperator):
    """T of T_fun_factory as a device-resident operator (same methods as KoopmansOperator)."""

    def __init__(self, model_params, grids, nodes, weights=None, device=0):
        self.params = tuple(float(p) for p in np.asarray(model_params, dtype=np.float64).ravel())
        self.grids = tuple(_as_f64(g) for g in grids)
        self.model_name = _model_name(self.params, len(self.grids))
        self.model = {"ssy": _lib.SDFS_MODEL_SSY, "gcy": _lib.SDFS_MODEL_GCY}[self.model_name]
        self.shapes = tuple(len(g) for g in self.grids)
        nd = len(self.grids)
        self.nodes = _as_f64(nodes)
        if self.nodes.ndim != 2 or self.nodes.shape[0] != nd:
            raise ValueError(f"nodes / mc_draws must have shape ({nd}, M)")
        M = self.nodes.shape[1]
        self.weights = None if weights is None else _as_f64(weights).ravel()
        if self.weights is not None and self.weights.size != M:
            raise ValueError("weights must have one entry per node")
        self.device = int(device)
        shp = (C.c_int64 * nd)(*self.shapes)
        par = (C.c_double * len(self.params))(*self.params)
        gp = (C.POINTER(C.c_double) * nd)(*[g.ctypes.data_as(C.POINTER(C.c_double)) for g in self.grids])
        h = C.c_void_p()
        rc = lib.sdfs_create_continuous(
            self.model, nd, shp, par, len(self.params), gp, self.nodes.ctypes.data_as(C.POINTER(C.c_double)),
            None if self.weights is None else self.weights.ctypes.data_as(C.POINTER(C.c_double)), M,
            self.device, C.byref(h))
        if rc != 0:
            raise _lib.SdfsError(f"sdfs_create_continuous failed ({rc}): {_lib.last_error(None)}")
        self._h = h
        self._finalizer = weakref.finalize(self, lib.sdfs_destroy, h)
        self.size = int(lib.sdfs_grid_size(h))


def T_fun_factory(params, method="quadrature", batch_size=10000, device=0):
    """Function factory for the operator T; ``params`` as in the reference:
    (model_params, grids, nodes, weights) for "quadrature", (model_params, grids, mc_draws) for
    "monte_carlo".  ``batch_size`` is accepted and checked like the reference's (it must divide the
    state space) but unused: the kernel needs no host-side batching."""
    grids = params[1]
    total_size = int(np.prod([len(g) for g in grids]))
    if total_size % int(batch_size) != 0:
        raise ValueError("""Size of the state space cannot be evenly divided
        by batch_size.""")
    if method == "quadrature":
        model_params, grids, nodes, weights = params
        return ContinuousOperator(model_params, grids, nodes, weights, device=device)
    elif method == "monte_carlo":
        model_params, grids, mc_draws = params
        return ContinuousOperator(model_params, grids, mc_draws, None, device=device)
    raise KeyError("Method not found.")


def wc_ratio_continuous(model, *grid_sizes, num_std_devs=3.2, d=5, mc_draw_size=2000, seed=1234,
                        w_init=None, ram_free=20, tol=1e-5, method='quadrature',
                        algorithm="successive_approx", verbose=True, write_to_file=True,
                        filename='w_star_data.npy', **size_kw):
    """Iterate to convergence on the Koopmans operator of the continuous-state model and return
    (grids, w_star); defaults as ssy_wc_ratio_continuous.py:229-235 / gcy_wc_ratio_continuous.py:264-270
    (grid sizes 10,10,10,20 resp. 10,10,10,10,20,20; keyword names h_λ_grid_size … accepted).
    ``tol`` is accepted and -- exactly as in the reference, which never forwards it -- not used:
    the solver runs with its own default tolerance."""
    name = _model_name(model)
    names = (["h_λ_grid_size", "h_c_grid_size", "h_z_grid_size", "z_grid_size"] if name == "ssy" else
             ["h_λ_grid_size", "h_c_grid_size", "h_z_grid_size", "h_zπ_grid_size", "z_grid_size", "z_π_grid_size"])
    defaults = [10, 10, 10, 20] if name == "ssy" else [10, 10, 10, 10, 20, 20]
    sizes = list(grid_sizes) + defaults[len(grid_sizes):]
    for i, nm in enumerate(names):
        if nm in size_kw:
            sizes[i] = size_kw.pop(nm)
    if size_kw:
        raise TypeError(f"unexpected arguments {sorted(size_kw)}")
    grids = build_grid(model, *sizes, num_std_devs=num_std_devs)
    dim = len(grids)
    if w_init is None:
        w_init = np.ones(shape=tuple(int(s) for s in sizes))
    if method == 'quadrature':
        nodes, weights = qnwnorm([d] * dim)
        params = np.array(model.params), grids, np.ascontiguousarray(nodes.T), weights
        sim_size = weights.size
    elif method == 'monte_carlo':
        # jax.random.normal(PRNGKey(seed)) cannot be reproduced without jax; numpy's generator
        # with the same seed gives an equally valid draw
        mc_draws = np.random.default_rng(seed).standard_normal((dim, mc_draw_size))
        params = np.array(model.params), grids, mc_draws
        sim_size = mc_draw_size
    else:
        raise KeyError("Approximation method not found.")
    # The reference sizes a vmap batch from `ram_free` and prints it.  The kernel needs no batching;
    # the figure is still reported (largest divisor of the state space that fits the reference's
    # memory estimate) so that logs of the two codes line up.
    state_size = int(np.prod(sizes))
    cap = (ram_free * 1024 ** 3 // 14) // (dim * sim_size * (8 if algorithm == 'newton' else 1))
    batch_size = max((q for q in range(1, int(np.sqrt(state_size)) + 1)
                      for q in (q, state_size // q) if state_size % q == 0 and q <= max(cap, 1)), default=1)
    print("batch_size =", batch_size)

    T = T_fun_factory(params, method, batch_size)
    w_star = solver(T, w_init, algorithm=algorithm, verbose=verbose)

    if write_to_file:
        save_wstar(filename, grids, w_star)
    return grids, w_star


# -- result file + interpolated callable ----------------------------------------------------------
def save_wstar(filename, grids, w_star):
    """The reference's file: ``np.save(f, grids); np.save(f, w_star)`` in one stream
    (ssy_wc_ratio_continuous.py:291-295).  Equal-length grids give a 2-D float array exactly as the
    reference writes.  Grids of different lengths (its defaults!) cannot be one float array; they are
    written without pickle as an int64 array of lengths followed by the concatenated grid values, so
    the file never needs ``allow_pickle`` to be read back."""
    with open(filename, 'wb') as f:
        if len({len(g) for g in grids}) == 1:
            np.save(f, np.asarray(grids, dtype=np.float64))
        else:
            np.save(f, np.asarray([len(g) for g in grids], dtype=np.int64))
            np.save(f, np.concatenate([np.asarray(g, dtype=np.float64).ravel() for g in grids]))
        np.save(f, np.asarray(w_star))


def load_wstar(datafile='w_star_data.npy', allow_pickle=False):
    """Reads a result file written by ``save_wstar`` or by the reference.  ``allow_pickle=True`` is only
    for files whose ragged grids were stored as a pickled object array (what old numpy made of the
    reference's ``np.save(f, grids)``); it executes whatever the pickle holds, so it is opt-in."""
    with open(datafile, 'rb') as f:
        first = np.load(f, allow_pickle=allow_pickle)
        if first.dtype.kind in "iu" and first.ndim == 1:          # lengths + flat values
            flat = np.load(f, allow_pickle=False)
            cuts = np.cumsum(first)[:-1]
            grids = np.split(np.asarray(flat, dtype=np.float64), cuts)
        else:
            grids = first
        w_star_vals = np.load(f, allow_pickle=False)
    return tuple(np.asarray(g, dtype=np.float64) for g in grids), w_star_vals


def construct_wstar_callable(w_star_vals=None, grids=None, datafile='w_star_data.npy', device=0):
    """Callable x (dim, N) -> w*(x) by linear interpolation over the grid; data read from disk when
    not given (ssy_wc_ratio_continuous.py:304-326)."""
    if w_star_vals is None or grids is None:
        grids, w_star_vals = load_wstar(datafile)
    grids = tuple(_as_f64(g) for g in grids)
    w_star_vals = _as_f64(w_star_vals)

    def w_star_func(x):
        return lin_interp(x, w_star_vals, grids, device=device)

    return w_star_func
