"""
Oracle (test infrastructure): GCY discretisation and Koopmans operator T.

Follows code/gcy/discrete/gcy_wc_ratio.py of the reference:
  discretize_gcy  :31-131   (15-tuple ``arrays``)
  T_gcy           :134-236  (12-D broadcast product summed over the 6 next-state axes)
  T_gcy_loops     :244-302  (nested-loop twin)
State order (z, z_pi, h_z, h_c, h_zpi, h_lam) = indices (a, b, c, d, e, f), h_lam
fastest in memory.  Conditional tensors (reference layout):
  z_pi_states[e, b], z_pi_Q[e, b, B]              (:93-101)
  z_states[b, c, e, a], z_Q[b, c, e, a, A]        (:112-120; drift mu = rho_pi * z_pi)

  Tw[a,b,c,d,e,f] = 1 + beta * ( a2[d] a3[b,c,e,a] * sum_{A..F} zQ[b,c,e,a,A] zpiQ[e,b,B]
        Qhz[c,C] Qhc[d,D] Qhzpi[e,E] Qhl[f,F] a1[F] w[A,B,C,D,E,F]^theta )^(1/theta)

Factorised order must respect the conditioning: (h_z, h_c, h_zpi, h_lam) any order,
then z_pi (needs current h_zpi), then z (needs current z_pi, h_z, h_zpi).
"""
import numpy as np

from .models import theta_of
from .rouwenhorst import rouwenhorst


def discretize_gcy(params, shapes):
    n_z, n_zp, n_hz, n_hc, n_hzp, n_hl = shapes
    (beta, psi, gamma, rho_lam, s_lam, mu_c, phi_c, rho, rho_pi, phi_z,
     rho_c, s_c, rho_z, s_z, rho_pipi, phi_zpi, rho_zpi, s_zpi) = params

    mc_hz = rouwenhorst(n_hz, rho_z, s_z)
    mc_hc = rouwenhorst(n_hc, rho_c, s_c)
    mc_hzp = rouwenhorst(n_hzp, rho_zpi, s_zpi)
    mc_hl = rouwenhorst(n_hl, rho_lam, s_lam)

    sigma_z = phi_z * np.exp(mc_hz.state_values)
    sigma_c = phi_c * np.exp(mc_hc.state_values)
    sigma_zp = phi_zpi * np.exp(mc_hzp.state_values)

    zp_states = np.zeros((n_hzp, n_zp))
    zp_Q = np.zeros((n_hzp, n_zp, n_zp))
    for e in range(n_hzp):
        mc = rouwenhorst(n_zp, rho_pipi, sigma_zp[e])
        zp_states[e, :] = mc.state_values
        zp_Q[e, :, :] = mc.P

    z_states = np.zeros((n_zp, n_hz, n_hzp, n_z))
    z_Q = np.zeros((n_zp, n_hz, n_hzp, n_z, n_z))
    for e in range(n_hzp):
        for c in range(n_hz):
            for b in range(n_zp):
                mc = rouwenhorst(n_z, rho, sigma_z[c], rho_pi * zp_states[e, b])
                z_states[b, c, e, :] = mc.state_values
                z_Q[b, c, e, :, :] = mc.P

    return (z_states, z_Q, zp_states, zp_Q,
            mc_hz.state_values, mc_hz.P, sigma_z,
            mc_hc.state_values, mc_hc.P, sigma_c,
            mc_hzp.state_values, mc_hzp.P, sigma_zp,
            mc_hl.state_values, mc_hl.P)


def _pieces(params, arrays):
    (beta, psi, gamma, rho_lam, s_lam, mu_c, *_rest) = params
    (z_states, zQ, zp_states, zpQ, h_z, Qhz, sigma_z, h_c, Qhc, sigma_c,
     h_zp, Qhzp, sigma_zp, h_l, Qhl) = [np.asarray(a) for a in arrays]
    theta = theta_of(gamma, psi)
    a1 = np.exp(theta * h_l)                                # [F]
    a2 = np.exp(0.5 * ((1 - gamma) * sigma_c) ** 2)         # [d]
    a3 = np.exp((1 - gamma) * (mu_c + z_states))            # [b, c, e, a]
    return beta, theta, a1, a2, a3, zQ, zpQ, Qhz, Qhc, Qhzp, Qhl


def T_gcy(w, shapes, params, arrays):
    """Literal O(N^2) evaluation: the 12-index kernel H is materialised (tiny shapes)."""
    beta, theta, a1, a2, a3, zQ, zpQ, Qhz, Qhc, Qhzp, Qhl = _pieces(params, arrays)
    w = np.asarray(w, dtype=np.float64)
    H = np.einsum("F,d,bcea,bceaA,ebB,cC,dD,eE,fF->abcdefABCDEF",
                  a1, a2, a3, zQ, zpQ, Qhz, Qhc, Qhzp, Qhl, optimize=False)
    Hw = (H * (w ** theta)[(None,) * 6]).sum(axis=(6, 7, 8, 9, 10, 11))
    return 1 + beta * Hw ** (1 / theta)


def T_gcy_loops(w, shapes, params, arrays):
    """Scalar loops; tiny shapes only."""
    n_z, n_zp, n_hz, n_hc, n_hzp, n_hl = shapes
    beta, theta, a1, a2, a3, zQ, zpQ, Qhz, Qhc, Qhzp, Qhl = _pieces(params, arrays)
    out = np.empty(shapes)
    for a in range(n_z):
        for b in range(n_zp):
            for c in range(n_hz):
                for d in range(n_hc):
                    for e in range(n_hzp):
                        for f in range(n_hl):
                            acc = 0.0
                            kk = a2[d] * a3[b, c, e, a]
                            for A in range(n_z):
                                pA = zQ[b, c, e, a, A]
                                for B in range(n_zp):
                                    pB = pA * zpQ[e, b, B]
                                    for C in range(n_hz):
                                        pC = pB * Qhz[c, C]
                                        for D in range(n_hc):
                                            pD = pC * Qhc[d, D]
                                            for E in range(n_hzp):
                                                pE = pD * Qhzp[e, E]
                                                for F in range(n_hl):
                                                    acc += (w[A, B, C, D, E, F] ** theta * a1[F]
                                                            * kk * pE * Qhl[f, F])
                            out[a, b, c, d, e, f] = 1 + beta * acc ** (1 / theta)
    return out


def expect_gcy(x, arrays_Q):
    """S = H0 x with H0 the pure transition kernel, axis by axis in a legal order."""
    zQ, zpQ, Qhz, Qhc, Qhzp, Qhl = arrays_Q
    y = np.einsum("fF,ABCDEF->ABCDEf", Qhl, x)
    y = np.einsum("eE,ABCDEf->ABCDef", Qhzp, y)
    y = np.einsum("dD,ABCDef->ABCdef", Qhc, y)
    y = np.einsum("cC,ABCdef->ABcdef", Qhz, y)
    y = np.einsum("ebB,ABcdef->Abcdef", zpQ, y)
    y = np.einsum("bceaA,Abcdef->abcdef", zQ, y)
    return y


def kfactor_gcy(a2, a3):
    """K[a,b,c,d,e,f] = a2[d] * a3[b,c,e,a] broadcast over f."""
    a3t = np.transpose(a3, (3, 0, 1, 2))                    # [a, b, c, e]
    return a2[None, None, None, :, None, None] * a3t[:, :, :, None, :, None]


def T_gcy_factorised(w, shapes, params, arrays):
    beta, theta, a1, a2, a3, zQ, zpQ, Qhz, Qhc, Qhzp, Qhl = _pieces(params, arrays)
    w = np.asarray(w, dtype=np.float64)
    S = expect_gcy(a1 * w ** theta, (zQ, zpQ, Qhz, Qhc, Qhzp, Qhl))
    return 1 + beta * (kfactor_gcy(a2, a3) * S) ** (1 / theta)


def jvp_gcy(w, v, shapes, params, arrays):
    """dT(w)[v] = beta * (K S)^(1/theta - 1) * K * H0(a1 w^(theta-1) v)."""
    beta, theta, a1, a2, a3, zQ, zpQ, Qhz, Qhc, Qhzp, Qhl = _pieces(params, arrays)
    w = np.asarray(w, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    Qs = (zQ, zpQ, Qhz, Qhc, Qhzp, Qhl)
    K = kfactor_gcy(a2, a3)
    S = expect_gcy(a1 * w ** theta, Qs)
    dS = expect_gcy(a1 * w ** (theta - 1) * v, Qs)
    return beta * (K * S) ** (1 / theta - 1) * K * dS
