"""
Oracle (test infrastructure): SSY discretisation and Koopmans operator T.

Follows code/ssy/discrete/ssy_wc_ratio.py of the reference:
  discretize_ssy  :23-79    (10-tuple ``arrays``)
  T_ssy           :82-149   (8-D broadcast product summed over the 4 next-state axes)
  T_ssy_loops     :159-199  (nested-loop twin)
State order (h_lam, h_c, h_z, z) = indices (l, k, i, j), z fastest in memory.

  Tw[l,k,i,j] = 1 + beta * ( a2[k] a3[i,j] sum_{L,K,I,J} Ql[l,L] Qc[k,K] Qz[i,I] zQ[i,j,J]
                                              a1[L] w[L,K,I,J]^theta )^(1/theta)
  a1 = exp(theta*h_lam) (:116), a2 = exp(0.5*((1-gamma)*sigma_c)^2) (:120),
  a3 = exp((1-gamma)*(mu_c + z_states)) (:124).

``T_ssy_factorised`` evaluates the same sum axis by axis (legal order: h_z before
z because zQ is conditioned on the *current* h_z index).  ``jvp_ssy`` is the
analytic directional derivative; the reference documents the same Jacobian in
dense form at code/ssy/discrete/temp_ssy.py:204-216.
"""
import numpy as np

from .models import theta_of
from .rouwenhorst import rouwenhorst


def discretize_ssy(params, shapes):
    n_hl, n_hc, n_hz, n_z = shapes
    (beta, gamma, psi, mu_c, rho, phi_z, phi_c,
     rho_z, rho_c, rho_lam, s_z, s_c, s_lam) = params

    mc_l = rouwenhorst(n_hl, rho_lam, s_lam, 0)
    mc_c = rouwenhorst(n_hc, rho_c, s_c, 0)
    mc_z = rouwenhorst(n_hz, rho_z, s_z, 0)

    h_l, h_c, h_z = mc_l.state_values, mc_c.state_values, mc_z.state_values
    sigma_z = phi_z * np.exp(h_z)
    sigma_c = phi_c * np.exp(h_c)

    z_states = np.zeros((n_hz, n_z))
    z_Q = np.zeros((n_hz, n_z, n_z))
    for i in range(n_hz):
        mc = rouwenhorst(n_z, rho, sigma_z[i], 0)
        z_states[i, :] = mc.state_values
        z_Q[i, :, :] = mc.P

    return (h_l, mc_l.P, h_c, mc_c.P, h_z, mc_z.P, z_states, z_Q, sigma_c, sigma_z)


def _pieces(params, arrays):
    (beta, gamma, psi, mu_c, *_rest) = params
    (h_l, Ql, h_c, Qc, h_z, Qz, z_states, zQ, sigma_c, sigma_z) = arrays
    theta = theta_of(gamma, psi)
    a1 = np.exp(theta * np.asarray(h_l))
    a2 = np.exp(0.5 * ((1 - gamma) * np.asarray(sigma_c)) ** 2)
    a3 = np.exp((1 - gamma) * (mu_c + np.asarray(z_states)))
    return beta, theta, a1, a2, a3, np.asarray(Ql), np.asarray(Qc), np.asarray(Qz), np.asarray(zQ)


def T_ssy(w, shapes, params, arrays):
    """Literal O(N^2) evaluation: the full 8-index kernel H is materialised."""
    beta, theta, a1, a2, a3, Ql, Qc, Qz, zQ = _pieces(params, arrays)
    w = np.asarray(w, dtype=np.float64)
    H = np.einsum("L,k,ij,lL,kK,iI,ijJ->lkijLKIJ", a1, a2, a3, Ql, Qc, Qz, zQ,
                  optimize=False)
    Hw = (H * (w ** theta)[None, None, None, None]).sum(axis=(4, 5, 6, 7))
    return 1 + beta * Hw ** (1 / theta)


def T_ssy_loops(w, shapes, params, arrays):
    """Scalar loops; tiny shapes only."""
    n_hl, n_hc, n_hz, n_z = shapes
    beta, theta, a1, a2, a3, Ql, Qc, Qz, zQ = _pieces(params, arrays)
    out = np.empty(shapes)
    for l in range(n_hl):
        for k in range(n_hc):
            for i in range(n_hz):
                for j in range(n_z):
                    acc = 0.0
                    for L in range(n_hl):
                        for K in range(n_hc):
                            for I in range(n_hz):
                                for J in range(n_z):
                                    acc += (w[L, K, I, J] ** theta * a1[L] * a2[k] * a3[i, j]
                                            * Ql[l, L] * Qc[k, K] * Qz[i, I] * zQ[i, j, J])
                    out[l, k, i, j] = 1 + beta * acc ** (1 / theta)
    return out


def expect_ssy(x, arrays_Q):
    """S = H0 x with H0 the pure transition kernel (no a1/a2/a3), axis by axis."""
    Ql, Qc, Qz, zQ = arrays_Q
    y = np.einsum("iI,LKIJ->LKiJ", Qz, x)      # h_z first (zQ needs current i)
    y = np.einsum("ijJ,LKiJ->LKij", zQ, y)     # z, conditioned on current h_z
    y = np.einsum("kK,LKij->Lkij", Qc, y)      # h_c
    y = np.einsum("lL,Lkij->lkij", Ql, y)      # h_lam
    return y


def T_ssy_factorised(w, shapes, params, arrays):
    beta, theta, a1, a2, a3, Ql, Qc, Qz, zQ = _pieces(params, arrays)
    w = np.asarray(w, dtype=np.float64)
    x = a1[:, None, None, None] * w ** theta
    S = expect_ssy(x, (Ql, Qc, Qz, zQ))
    K = a2[None, :, None, None] * a3[None, None, :, :]
    return 1 + beta * (K * S) ** (1 / theta)


def jvp_ssy(w, v, shapes, params, arrays):
    """dT(w)[v] = beta * (K S)^(1/theta - 1) * K * H0(a1 w^(theta-1) v),  S = H0(a1 w^theta)."""
    beta, theta, a1, a2, a3, Ql, Qc, Qz, zQ = _pieces(params, arrays)
    w = np.asarray(w, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    A1 = a1[:, None, None, None]
    K = a2[None, :, None, None] * a3[None, None, :, :]
    S = expect_ssy(A1 * w ** theta, (Ql, Qc, Qz, zQ))
    dS = expect_ssy(A1 * w ** (theta - 1) * v, (Ql, Qc, Qz, zQ))
    return beta * (K * S) ** (1 / theta - 1) * K * dS
