"""
Log-linear approximation of the wealth-consumption ratio (warm start for the solvers).

Mirrors ``wc_loglinear_factory`` of the reference -- code/ssy/ssy_model.py:88-156 (SSY) and
code/gcy/gcy_model.py:80-159 (GCY): a Campbell-Shiller style expansion around the mean log
ratio q_bar, which solves a scalar fixed-point equation (Brent's method on [-20, 20]).
``wc_loglinear_factory(model)`` returns a function of the state vector, as in the reference
(SSY: x = (h_λ, h_c, h_z, z); GCY: x = (h_λ, h_c, h_z, h_zπ, z, z_π)); it also accepts arrays
(broadcasting), which the reference's numba scalar version does not.

``loglinear_guess(model, shapes, arrays)`` evaluates it on a discretised grid and returns
w_init = exp(q) + 1 in the grid's axis order (the use sketched in
code/ssy/continuous_junnan/test_newton.md:63-66, :250).  Host-side only.
"""
import numpy as np
from scipy.optimize import brentq

from .models import SSY, GCY


def wc_loglinear_factory(model):
    if isinstance(model, SSY):
        return _ssy_factory(model)
    if isinstance(model, GCY):
        return _gcy_factory(model)
    raise TypeError("wc_loglinear_factory expects an SSY or GCY instance")


def _k1(x):
    return np.exp(x) / (1 + np.exp(x))


def _k0(x):
    return np.log(1 + np.exp(x)) - _k1(x) * x


def _ssy_factory(ssy):
    β, γ, ψ, μ_c, ρ, φ_z, φ_c, ρ_z, ρ_c, ρ_λ, s_z, s_c, s_λ = ssy.params
    θ = ssy.θ
    s_wc = 2 * φ_c**2 * s_c
    s_wx = 2 * φ_z**2 * s_z
    ies = 1 - 1 / ψ

    def A1(q): return ies / (1 - _k1(q) * ρ)
    def Aλ(q): return ρ_λ / (1 - _k1(q) * ρ_λ)
    def Az(q): return (θ / 2) * (_k1(q) * A1(q))**2 / (1 - _k1(q) * ρ_z)
    def Ac(q): return (θ / 2) * ies**2 / (1 - _k1(q) * ρ_c)

    def A0(q):
        k1 = _k1(q)
        num = (np.log(β) + _k0(q) + μ_c * ies
               + k1 * Az(q) * φ_z**2 * (1 - ρ_z)
               + k1 * Ac(q) * φ_c**2 * (1 - ρ_c)
               + (θ / 2) * ((k1 * Aλ(q) + 1)**2 * s_λ**2
                            + (k1 * Az(q) * s_wx)**2 + (k1 * Ac(q) * s_wc)**2))
        return num / (1 - k1)

    qbar = brentq(lambda q: q - A0(q) - Ac(q) * φ_c**2 - Az(q) * φ_z**2, -20, 20)
    c_z, c_λ, c_hz, c_hc, c_0 = A1(qbar), Aλ(qbar), Az(qbar), Ac(qbar), A0(qbar)

    def wc_loglinear(x):
        h_λ, h_c, h_z, z = x
        v_z = h_z * 2 * φ_z**2 + φ_z**2
        v_c = h_c * 2 * φ_c**2 + φ_c**2
        return c_0 + c_λ * h_λ + c_hc * v_c + c_hz * v_z + c_z * z

    wc_loglinear.qbar = qbar
    wc_loglinear.constants = dict(A0=c_0, Az=c_z, Ah_λ=c_λ, Ah_z=c_hz, Ah_c=c_hc)
    return wc_loglinear


def _gcy_factory(gcy):
    (β, ψ, γ, ρ_λ, s_λ, μ_c, φ_c, ρ, ρ_π, φ_z, ρ_c, s_c, ρ_z, s_z,
     ρ_ππ, φ_zπ, ρ_zπ, s_zπ) = gcy.params
    θ = gcy.θ
    s_wc = 2 * φ_c**2 * s_c
    s_wx = 2 * φ_z**2 * s_z
    s_wxπ = 2 * φ_zπ**2 * s_zπ
    ies = 1 - 1 / ψ

    def A1(q): return ies / (1 - _k1(q) * ρ)
    def Aλ(q): return ρ_λ / (1 - _k1(q) * ρ_λ)
    def Aπ(q): return _k1(q) * ies * ρ_π / ((1 - _k1(q) * ρ) * (1 - _k1(q) * ρ_ππ))
    def Az(q): return (θ / 2) * (_k1(q) * A1(q))**2 / (1 - _k1(q) * ρ_z)
    def Azπ(q): return (θ / 2) * (_k1(q) * Aπ(q))**2 / (1 - _k1(q) * ρ_zπ)
    def Ac(q): return (θ / 2) * ies**2 / (1 - _k1(q) * ρ_c)

    def A0(q):
        k1 = _k1(q)
        num = (np.log(β) + _k0(q) + μ_c * ies
               + k1 * Az(q) * φ_z**2 * (1 - ρ_z)
               + k1 * Ac(q) * φ_c**2 * (1 - ρ_c)
               + k1 * Azπ(q) * φ_zπ**2 * (1 - ρ_zπ)
               + (θ / 2) * ((k1 * Aλ(q) + 1)**2 * s_λ**2
                            + (k1 * Az(q) * s_wx)**2 + (k1 * Ac(q) * s_wc)**2
                            + (k1 * Azπ(q) * s_wxπ)**2))
        return num / (1 - k1)

    qbar = brentq(lambda q: q - A0(q) - Ac(q) * φ_c**2 - Az(q) * φ_z**2 - Azπ(q) * φ_zπ**2, -20, 20)
    c_z, c_π, c_λ = A1(qbar), Aπ(qbar), Aλ(qbar)
    c_hz, c_hc, c_hzπ, c_0 = Az(qbar), Ac(qbar), Azπ(qbar), A0(qbar)

    def wc_loglinear(x):
        h_λ, h_c, h_z, h_zπ, z, z_π = x
        v_z = h_z * 2 * φ_z**2 + φ_z**2
        v_c = h_c * 2 * φ_c**2 + φ_c**2
        v_zπ = h_zπ * 2 * φ_zπ**2 + φ_zπ**2
        return (c_0 + c_λ * h_λ + c_hc * v_c + c_hz * v_z + c_z * z + c_hzπ * v_zπ + c_π * z_π)

    wc_loglinear.qbar = qbar
    wc_loglinear.constants = dict(A0=c_0, Az=c_z, Az_π=c_π, Ah_λ=c_λ, Ah_z=c_hz, Ah_c=c_hc, Ah_zπ=c_hzπ)
    return wc_loglinear


def loglinear_guess(model, shapes, arrays):
    """w_init = exp(wc_loglinear(state)) + 1 on the discretised grid, in the grid's axis order."""
    f = wc_loglinear_factory(model)
    if isinstance(model, SSY):
        h_λ, _, h_c, _, h_z, _, z_states, *_ = arrays          # z_states[i, j]
        q = f((h_λ[:, None, None, None], h_c[None, :, None, None], h_z[None, None, :, None],
               z_states[None, None, :, :]))
        return np.broadcast_to(np.exp(q) + 1, tuple(shapes)).copy()
    (z_states, _, z_π_states, _, h_z, _, _, h_c, _, _, h_zπ, _, _, h_λ, _) = arrays
    # grid order (z, z_π, h_z, h_c, h_zπ, h_λ) = (a, b, c, d, e, f); z_states[b, c, e, a], z_π_states[e, b]
    z = np.transpose(z_states, (3, 0, 1, 2))[:, :, :, None, :, None]
    z_π = np.transpose(z_π_states, (1, 0))[None, :, None, None, :, None]
    q = f((h_λ[None, None, None, None, None, :], h_c[None, None, None, :, None, None],
           h_z[None, None, :, None, None, None], h_zπ[None, None, None, None, :, None], z, z_π))
    return np.broadcast_to(np.exp(q) + 1, tuple(shapes)).copy()
