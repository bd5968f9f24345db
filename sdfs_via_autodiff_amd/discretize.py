"""
Host-side discretisation of the SSY / GCY state processes (NumPy, float64).

Mirrors, with the same names, argument meaning and return layout:
  discretize_ssy(ssy, shapes)  -- code/ssy/discrete/ssy_wc_ratio.py:23-79  (10-tuple)
  discretize_gcy(gcy, shapes)  -- code/gcy/discrete/gcy_wc_ratio.py:31-131 (15-tuple)
  rouwenhorst(n, rho, sigma, mu) -- quantecon.markov.approximation.rouwenhorst, the
      third-party routine both call (the reference's SSY module forgets to import it;
      here it is a plain function of this module).

A Rouwenhorst transition matrix depends only on (n, rho) -- the innovation scale
and the drift move the grid, not the probabilities -- so the conditional chains
(one per volatility state) share one matrix and only their grids are rebuilt.
The tensors are still returned in the reference's conditional layout
(z_Q[i, j, J], z_Q[b, c, e, a, A], ...), which is what the operator consumes.

``method="tauchen"`` (not in the reference; BASELINE.json's configurations name Tauchen grids) swaps
the chain builder for Tauchen's (1986) method in the same tuple layout.  Its matrices, too, depend
only on (n, rho, n_std): the grid scales with the innovation scale.
"""
from collections import namedtuple
from functools import lru_cache

import numpy as np

MarkovChain = namedtuple("MarkovChain", ["P", "state_values"])


@lru_cache(maxsize=64)
def _rouwenhorst_P(n, rho):
    p = (1.0 + rho) / 2.0
    theta = np.array([[p, 1.0 - p], [1.0 - p, p]])
    for m in range(3, n + 1):
        new = np.zeros((m, m))
        new[:-1, :-1] += p * theta
        new[:-1, 1:] += (1.0 - p) * theta
        new[1:, :-1] += (1.0 - p) * theta
        new[1:, 1:] += p * theta
        new[1:-1, :] /= 2.0
        theta = new
    theta.setflags(write=False)
    return theta


def _rouwenhorst_grid(n, rho, sigma, mu=0.0):
    """State values for (arrays of) sigma / mu, broadcast; trailing axis = state index."""
    sigma = np.asarray(sigma, dtype=np.float64)
    mu = np.asarray(mu, dtype=np.float64)
    psi = sigma * np.sqrt((n - 1) / (1.0 - rho**2))
    # np.linspace(-psi, psi, n): start + k*step with step = 2*psi/(n-1), last point exact
    k = np.arange(n, dtype=np.float64)
    step = (2.0 * psi / (n - 1))[..., None]
    grid = -psi[..., None] + k * step
    grid[..., -1] = psi
    return grid + (mu / (1.0 - rho))[..., None]


def rouwenhorst(n, rho, sigma, mu=0.0):
    """Discretise y' = mu + rho*y + sigma*eps; returns MarkovChain(P, state_values)."""
    if n < 2:
        raise ValueError("rouwenhorst: n must be >= 2")
    return MarkovChain(P=np.array(_rouwenhorst_P(int(n), float(rho))),
                       state_values=_rouwenhorst_grid(int(n), float(rho), float(sigma), float(mu)))


@lru_cache(maxsize=64)
def _tauchen_P(n, rho, n_std):
    """Tauchen (1986) for a unit innovation scale: grid ±n_std/sqrt(1-rho²), bin-mass probabilities."""
    from math import erf, sqrt
    Phi = np.vectorize(lambda v: 0.5 * (1.0 + erf(v / sqrt(2.0))))
    y = np.linspace(-n_std / np.sqrt(1.0 - rho ** 2), n_std / np.sqrt(1.0 - rho ** 2), n)
    h = y[1] - y[0]
    z = y[None, :] - rho * y[:, None]
    P = Phi(z + h / 2) - Phi(z - h / 2)
    P[:, 0] = Phi(z[:, 0] + h / 2)
    P[:, -1] = 1.0 - Phi(z[:, -1] - h / 2)
    P.setflags(write=False)
    return P


def _tauchen_grid(n, rho, sigma, mu=0.0, n_std=3):
    sigma = np.asarray(sigma, dtype=np.float64)
    mu = np.asarray(mu, dtype=np.float64)
    ymax = n_std * sigma / np.sqrt(1.0 - rho ** 2)
    k = np.arange(n, dtype=np.float64)
    grid = -ymax[..., None] + k * (2.0 * ymax / (n - 1))[..., None]
    grid[..., -1] = ymax
    return grid + (mu / (1.0 - rho))[..., None]


def tauchen(n, rho, sigma, mu=0.0, n_std=3):
    """Discretise y' = mu + rho*y + sigma*eps by Tauchen's method; MarkovChain(P, state_values)."""
    if n < 2:
        raise ValueError("tauchen: n must be >= 2")
    return MarkovChain(P=np.array(_tauchen_P(int(n), float(rho), float(n_std))),
                       state_values=_tauchen_grid(int(n), float(rho), float(sigma), float(mu), n_std))


def _chain_builders(method):
    if method == "rouwenhorst":
        return rouwenhorst, _rouwenhorst_grid, lambda n, rho: _rouwenhorst_P(n, float(rho))
    if method == "tauchen":
        return tauchen, _tauchen_grid, lambda n, rho: _tauchen_P(n, float(rho), 3.0)
    raise ValueError(f"unknown discretisation method {method!r} (rouwenhorst, tauchen)")


def discretize_ssy(ssy, shapes, method="rouwenhorst"):
    """Multi-index discretisation of SSY: states (h_λ, h_c, h_z, z) = indices (l, k, i, j)."""
    n_h_λ, n_h_c, n_h_z, n_z = (int(s) for s in shapes)
    β, γ, ψ, μ_c, ρ, φ_z, φ_c, ρ_z, ρ_c, ρ_λ, s_z, s_c, s_λ = ssy.params
    rouwenhorst, _rouwenhorst_grid, _rouwenhorst_P = _chain_builders(method)

    h_λ_mc = rouwenhorst(n_h_λ, ρ_λ, s_λ, 0)
    h_c_mc = rouwenhorst(n_h_c, ρ_c, s_c, 0)
    h_z_mc = rouwenhorst(n_h_z, ρ_z, s_z, 0)

    h_λ_states, h_c_states, h_z_states = (h_λ_mc.state_values, h_c_mc.state_values,
                                          h_z_mc.state_values)
    σ_z_states = φ_z * np.exp(h_z_states)
    σ_c_states = φ_c * np.exp(h_c_states)

    # z chain for every volatility state i: z_states[i, j], z_Q[i, j, jp]
    z_states = _rouwenhorst_grid(n_z, ρ, σ_z_states, 0.0)
    z_Q = np.broadcast_to(_rouwenhorst_P(n_z, float(ρ)), (n_h_z, n_z, n_z)).copy()

    return (h_λ_states, h_λ_mc.P,
            h_c_states, h_c_mc.P,
            h_z_states, h_z_mc.P,
            z_states, z_Q,
            σ_c_states, σ_z_states)


def discretize_gcy(gcy, shapes, method="rouwenhorst"):
    """Multi-index discretisation of GCY: states (z, z_π, h_z, h_c, h_zπ, h_λ)."""
    n_z, n_z_π, n_h_z, n_h_c, n_h_zπ, n_h_λ = (int(s) for s in shapes)
    (β, ψ, γ, ρ_λ, s_λ, μ_c, φ_c, ρ, ρ_π, φ_z, ρ_c, s_c, ρ_z, s_z,
     ρ_ππ, φ_zπ, ρ_zπ, s_zπ) = gcy.params
    rouwenhorst, _rouwenhorst_grid, _rouwenhorst_P = _chain_builders(method)

    h_z_mc = rouwenhorst(n_h_z, ρ_z, s_z)
    h_c_mc = rouwenhorst(n_h_c, ρ_c, s_c)
    h_zπ_mc = rouwenhorst(n_h_zπ, ρ_zπ, s_zπ)
    h_λ_mc = rouwenhorst(n_h_λ, ρ_λ, s_λ)

    σ_z_states = φ_z * np.exp(h_z_mc.state_values)
    σ_c_states = φ_c * np.exp(h_c_mc.state_values)
    σ_zπ_states = φ_zπ * np.exp(h_zπ_mc.state_values)

    # z_π' = ρ_ππ z_π + σ_zπ η: one grid per h_zπ state -> z_π_states[i_h_zπ, i_z_π]
    z_π_states = _rouwenhorst_grid(n_z_π, ρ_ππ, σ_zπ_states, 0.0)
    z_π_Q = np.broadcast_to(_rouwenhorst_P(n_z_π, float(ρ_ππ)), (n_h_zπ, n_z_π, n_z_π)).copy()

    # z' = ρ z + ρ_π z_π + σ_z η: one grid per (z_π, h_z, h_zπ) -> z_states[i_z_π, i_h_z, i_h_zπ, i_z]
    sig = np.broadcast_to(σ_z_states[None, :, None], (n_z_π, n_h_z, n_h_zπ))
    mu = np.broadcast_to((ρ_π * z_π_states.T)[:, None, :], (n_z_π, n_h_z, n_h_zπ))
    z_states = _rouwenhorst_grid(n_z, ρ, sig, mu)
    z_Q = np.broadcast_to(_rouwenhorst_P(n_z, float(ρ)), (n_z_π, n_h_z, n_h_zπ, n_z, n_z)).copy()

    return (z_states, z_Q,
            z_π_states, z_π_Q,
            h_z_mc.state_values, h_z_mc.P, σ_z_states,
            h_c_mc.state_values, h_c_mc.P, σ_c_states,
            h_zπ_mc.state_values, h_zπ_mc.P, σ_zπ_states,
            h_λ_mc.state_values, h_λ_mc.P)
