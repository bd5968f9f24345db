"""
TEST INFRASTRUCTURE -- CPU restatement of the reference's continuous-state Koopmans operator.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (reference file:line)
  code/utils.py:6-23                                     vals_to_coords / lin_interp
  code/ssy/continuous_junnan/ssy_wc_ratio_continuous.py  build_grid :20-59, next_state :66-87,
      Kg_vmap_mc :94-119, Kg_vmap_quad :125-150, T_fun_factory :156-226
  code/gcy/continuous/gcy_wc_ratio_continuous.py         build_grid :23-71, next_state :78-116,
      Kg_vmap_mc :123-149, Kg_vmap_quad :159-184, T_fun_factory :190-260

Third-party arithmetic absent from /root/reference (versions unpinned there):
  quantecon.quad.qnwnorm  -- Gauss-Hermite rule for N(0, 1) shocks on a tensor grid.  Published
      algorithm (Miranda-Fackler CompEcon `qnwnorm`): 1-D Hermite nodes x_i / weights w_i, nodes
      sqrt(2) x_i, weights w_i / sqrt(pi); `gridmake` order (first dimension fastest) and
      `ckron(*weights[::-1])`.  numpy.polynomial.hermite.hermgauss supplies x_i, w_i.
  jax.scipy.ndimage.map_coordinates(order=1, mode='nearest') -- multilinear interpolation with the
      indices clipped to the grid; scipy.ndimage.map_coordinates has the same definition.

Pinned by tests/golden/cont_*.npz (the reference's own functions run under the shims of
tests/golden/make_golden.py).
"""
import numpy as np
from scipy.ndimage import map_coordinates


def qnwnorm(n):
    """Nodes (prod(n), dim) and weights (prod(n),) of the tensor Gauss-Hermite rule for N(0, I)."""
    n = [int(k) for k in np.atleast_1d(n)]
    nodes1, weights1 = [], []
    for k in n:
        x, w = np.polynomial.hermite.hermgauss(k)
        nodes1.append(x * np.sqrt(2.0))
        weights1.append(w / np.sqrt(np.pi))
    M = int(np.prod(n))
    nodes = np.empty((M, len(n)))
    weights = np.ones(M)
    rep = 1
    for d, k in enumerate(n):                 # gridmake: dimension 0 varies fastest
        idx = (np.arange(M) // rep) % k
        nodes[:, d] = nodes1[d][idx]
        weights *= weights1[d][idx]
        rep *= k
    return nodes, weights


def vals_to_coords(grids, x_vals):
    """utils.py:6-15 (uniform grids: first point and first spacing)."""
    intervals = np.asarray([g[1] - g[0] for g in grids]).reshape(-1, 1)
    low = np.asarray([g[0] for g in grids]).reshape(-1, 1)
    return (x_vals - low) / intervals


def lin_interp(x, fun_vals, grids):
    """utils.py:18-23."""
    return map_coordinates(fun_vals, vals_to_coords(grids, x), order=1, mode="nearest")


# -- SSY --------------------------------------------------------------------------------
def build_grid_ssy(params, sizes, num_std_devs=3.2):
    (β, γ, ψ, μ_c, ρ, ϕ_z, ϕ_c, ρ_z, ρ_c, ρ_λ, s_z, s_c, s_λ) = params
    grids = []
    for s, r, n in zip((s_λ, s_c, s_z), (ρ_λ, ρ_c, ρ_z), sizes[:3]):
        g_max = num_std_devs * np.sqrt(s ** 2 / (1 - r ** 2))
        grids.append(np.linspace(-g_max, g_max, n))
    h_z_max = num_std_devs * np.sqrt(s_z ** 2 / (1 - ρ_z ** 2))
    σ_z_max = ϕ_z * np.exp(h_z_max)
    z_max = num_std_devs * σ_z_max
    grids.append(np.linspace(-z_max, z_max, sizes[3]))
    return tuple(grids)


def next_state_ssy(params, x, eta):
    (β, γ, ψ, μ_c, ρ, ϕ_z, ϕ_c, ρ_z, ρ_c, ρ_λ, s_z, s_c, s_λ) = params
    h_λ, h_c, h_z, z = x
    σ_z = ϕ_z * np.exp(h_z)
    return np.array([ρ_λ * h_λ + s_λ * eta[0], ρ_c * h_c + s_c * eta[1],
                     ρ_z * h_z + s_z * eta[2], ρ * z + σ_z * eta[3]])


def _const_ssy(params, x):
    (β, γ, ψ, μ_c, ρ, ϕ_z, ϕ_c, ρ_z, ρ_c, ρ_λ, s_z, s_c, s_λ) = params
    σ_c = ϕ_c * np.exp(x[1])
    return np.exp((1 - γ) * (μ_c + x[3]) + 0.5 * (1 - γ) ** 2 * σ_c ** 2)


# -- GCY --------------------------------------------------------------------------------
def build_grid_gcy(params, sizes, num_std_devs=3.2):
    (β, ψ, γ, ρ_λ, s_λ, μ_c, φ_c, ρ, ρ_π, φ_z, ρ_c, s_c, ρ_z, s_z, ρ_ππ, φ_zπ, ρ_zπ, s_zπ) = params
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
    # gcy_wc_ratio_continuous.py:47 reuses the name ρ as its loop variable, so by :68-69 it holds the
    # last entry of rho_vals (ρ_zπ), not the persistence of z: the reference's grid is reproduced
    ρ_leaked = ρ_zπ
    z_max = (ρ_π * zπ_grid[-1] + num_std_devs * σ_z_max) / (1 - ρ_leaked)
    z_min = (ρ_π * zπ_grid[0] - num_std_devs * σ_z_max) / (1 - ρ_leaked)
    grids.append(np.linspace(z_min, z_max, sizes[4]))
    grids.append(zπ_grid)
    return tuple(grids)


def next_state_gcy(params, x, eta):
    (β, ψ, γ, ρ_λ, s_λ, μ_c, φ_c, ρ, ρ_π, φ_z, ρ_c, s_c, ρ_z, s_z, ρ_ππ, φ_zπ, ρ_zπ, s_zπ) = params
    h_λ, h_c, h_z, h_zπ, z, z_π = x
    σ_z = φ_z * np.exp(h_z)
    σ_zπ = φ_zπ * np.exp(h_zπ)
    return np.array([ρ_λ * h_λ + s_λ * eta[0], ρ_c * h_c + s_c * eta[1], ρ_z * h_z + s_z * eta[2],
                     ρ_zπ * h_zπ + s_zπ * eta[3], ρ * z + ρ_π * z_π + σ_z * eta[4],
                     ρ_ππ * z_π + σ_zπ * eta[5]])


def _const_gcy(params, x):
    (β, ψ, γ, ρ_λ, s_λ, μ_c, φ_c, ρ, ρ_π, φ_z, ρ_c, s_c, ρ_z, s_z, ρ_ππ, φ_zπ, ρ_zπ, s_zπ) = params
    σ_c = φ_c * np.exp(x[1])
    return np.exp((1 - γ) * (μ_c + x[4]) + 0.5 * (1 - γ) ** 2 * σ_c ** 2)


def _theta_beta(model, params):
    if model == "ssy":
        β, γ, ψ = params[0], params[1], params[2]
    else:
        β, ψ, γ = params[0], params[1], params[2]
    return (1 - γ) / (1 - 1 / ψ), β


def T_fun_factory(model, params, grids, nodes, weights=None):
    """T(w) of the reference's T_fun_factory; ``weights is None`` = Monte Carlo (plain mean).
    nodes has shape (dim, M) as in the reference."""
    params = tuple(float(p) for p in params)
    θ, β = _theta_beta(model, params)
    nxt, cst = (next_state_ssy, _const_ssy) if model == "ssy" else (next_state_gcy, _const_gcy)
    shape = tuple(len(g) for g in grids)
    mesh = np.meshgrid(*grids, indexing="ij")
    X = np.stack([m.ravel() for m in mesh], axis=1)

    def T(w):
        w = np.asarray(w, dtype=np.float64)
        Kg = np.empty(X.shape[0])
        for p, x in enumerate(X):
            nx = nxt(params, x, nodes)
            pf = np.exp(nx[0] * θ)
            g = lin_interp(nx, w, grids) ** θ
            e = np.dot(g * pf, weights) if weights is not None else np.mean(g * pf)
            Kg[p] = cst(params, x) * e
        return 1 + β * Kg.reshape(shape) ** (1 / θ)

    return T


def jvp_factory(model, params, grids, nodes, weights=None):
    """Analytic dT(w)[v] of the operator above (what jax.jvp gives the reference, solvers.py:87)."""
    params = tuple(float(p) for p in params)
    θ, β = _theta_beta(model, params)
    nxt, cst = (next_state_ssy, _const_ssy) if model == "ssy" else (next_state_gcy, _const_gcy)
    shape = tuple(len(g) for g in grids)
    mesh = np.meshgrid(*grids, indexing="ij")
    X = np.stack([m.ravel() for m in mesh], axis=1)
    M = nodes.shape[1]
    wts = weights if weights is not None else np.full(M, 1.0 / M)

    def jvp(w, v):
        out = np.empty(X.shape[0])
        for p, x in enumerate(X):
            nx = nxt(params, x, nodes)
            pf = np.exp(nx[0] * θ)
            g = lin_interp(nx, w, grids)
            iv = lin_interp(nx, v, grids)
            C = cst(params, x)
            Kg = C * np.dot(g ** θ * pf, wts)
            out[p] = β * Kg ** (1 / θ - 1) * C * np.dot(g ** (θ - 1) * iv * pf, wts)
        return out.reshape(shape)

    return jvp
