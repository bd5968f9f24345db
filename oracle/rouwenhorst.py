"""
Oracle (test infrastructure): Rouwenhorst discretisation of an AR(1).

Restates quantecon.markov.approximation.rouwenhorst(n, rho, sigma, mu=0.),
which the reference calls but does not vendor (call sites
code/ssy/discrete/ssy_wc_ratio.py:48-50,63; code/gcy/discrete/gcy_wc_ratio.py:
65-68,97,115).  The reference reads exactly two attributes of the returned
chain, ``.P`` and ``.state_values`` (ssy_wc_ratio.py:53-55,65-66).

Published algorithm (Kopecky & Suen 2010, as implemented by QuantEcon.py):
  y' = mu + rho*y + sigma*eps;  p = q = (1+rho)/2;
  psi = sigma*sqrt((n-1)/(1-rho^2));  grid = linspace(-psi, psi, n) + mu/(1-rho);
  Theta_2 = [[p,1-p],[1-q,q]];
  Theta_n = p*[[T,0],[0,0]] + (1-p)*[[0,T],[0,0]] + (1-q)*[[0,0],[T,0]] + q*[[0,0],[0,T]]
  with T = Theta_{n-1}, then interior rows halved.
"""
from collections import namedtuple

import numpy as np

Chain = namedtuple("Chain", ["P", "state_values"])


def rouwenhorst_matrix(n, p, q):
    """n x n Rouwenhorst transition matrix, built bottom-up from the 2x2 case."""
    if n < 2:
        raise ValueError("rouwenhorst needs n >= 2")
    theta = np.array([[p, 1.0 - p], [1.0 - q, q]])
    for m in range(3, n + 1):
        prev = theta
        theta = np.zeros((m, m))
        theta[: m - 1, : m - 1] += p * prev
        theta[: m - 1, 1:] += (1.0 - p) * prev
        theta[1:, : m - 1] += (1.0 - q) * prev
        theta[1:, 1:] += q * prev
        theta[1 : m - 1, :] /= 2.0
    return theta


def rouwenhorst(n, rho, sigma, mu=0.0):
    """Chain(P, state_values) for y' = mu + rho*y + sigma*eps, eps ~ N(0,1)."""
    p = (1.0 + rho) / 2.0
    q = p
    psi = sigma * np.sqrt((n - 1) / (1.0 - rho**2))
    grid = np.linspace(-psi, psi, n) + mu / (1.0 - rho)
    return Chain(P=rouwenhorst_matrix(n, p, q), state_values=grid)
