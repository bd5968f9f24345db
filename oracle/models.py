"""
Oracle (test infrastructure): default SSY / GCY calibrations.

Values and ``params`` tuple order follow code/ssy/ssy_model.py:57-81 and
code/gcy/gcy_model.py:45-75 of the reference.  ASCII names are used here; the
product package (sdfs_via_autodiff_amd.models) keeps the reference's Greek
keyword names.
"""
import math

# SSY params order (ssy_model.py:81):
#   beta, gamma, psi, mu_c, rho, phi_z, phi_c, rho_z, rho_c, rho_lam, s_z, s_c, s_lam
SSY_FIELDS = ("beta", "gamma", "psi", "mu_c", "rho", "phi_z", "phi_c",
              "rho_z", "rho_c", "rho_lam", "s_z", "s_c", "s_lam")

# GCY params order (gcy_model.py:72-75):
#   beta, psi, gamma, rho_lam, s_lam, mu_c, phi_c, rho, rho_pi, phi_z, rho_c,
#   s_c, rho_z, s_z, rho_pipi, phi_zpi, rho_zpi, s_zpi
GCY_FIELDS = ("beta", "psi", "gamma", "rho_lam", "s_lam", "mu_c", "phi_c", "rho",
              "rho_pi", "phi_z", "rho_c", "s_c", "rho_z", "s_z",
              "rho_pipi", "phi_zpi", "rho_zpi", "s_zpi")


def ssy_params(**over):
    """SSY default calibration (ssy_model.py:57-71) as the 13-tuple of :81."""
    d = dict(beta=0.999, gamma=8.89, psi=1.97, rho=0.987, rho_z=0.992,
             rho_c=0.991, rho_lam=0.959, s_z=math.sqrt(0.0039),
             s_c=math.sqrt(0.0096), s_lam=0.0004, mu_c=0.0016,
             phi_z=0.215 * 0.0035 * math.sqrt(1 - 0.987**2),
             phi_c=1.00 * 0.0035)
    d.update(over)
    return tuple(float(d[k]) for k in SSY_FIELDS)


def gcy_params(**over):
    """GCY default calibration (gcy_model.py:45-63) as the 18-tuple of :72-75."""
    d = dict(beta=0.9987, psi=1.5, gamma=13.01, rho_lam=0.981,
             s_lam=0.12 * 0.0015, mu_c=0.0016, phi_c=0.0015, rho=0.983,
             rho_pi=-0.0075, phi_z=0.13 * 0.0015, rho_c=0.992, s_c=0.104,
             rho_z=0.980, s_z=0.09, rho_pipi=0.985, phi_zpi=0.08 * 0.0015,
             rho_zpi=0.970, s_zpi=0.271)
    d.update(over)
    return tuple(float(d[k]) for k in GCY_FIELDS)


def theta_of(gamma, psi):
    """theta = (1-gamma)/(1-1/psi)  (ssy_wc_ratio.py:99, gcy_wc_ratio.py:155)."""
    return (1.0 - gamma) / (1.0 - 1.0 / psi)
