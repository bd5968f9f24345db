"""
Single-index dense form of the SSY / GCY operator -- mirror of the cross-check code in
code/ssy/discrete/temp_ssy.py (split_index :23-27, single_to_multi :29-37, multi_to_single :39-41,
discretize_single_index :48-85, compute_H_single_index :109-113, single_index_T :147-153,
test_compute_wc_ratio_single_index :156-184).

    H[m, m'] = a1[l'] a2[k] a3[i, j] * h_λ_P[l, l'] h_c_P[k, k'] h_z_P[i, i'] z_Q[i, j, j']
    Tw       = 1 + β (H @ w^θ)^(1/θ)

H is assembled on the host with Kronecker products (the reference fills it with two nested Python
loops over N); the operator itself -- GEMV, powers, JVP, solver loops -- runs in libsdfs_hip.so
(csrc/dense_kernel.hpp).  Meant for N up to a few 10^4: an independent route to the numbers the
factorised kernels produce.  The GCY variant (not in the reference) follows the same construction.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import lib
from .discretize import discretize_ssy, discretize_gcy
from .operators import KoopmansOperator, _as_f64
from .solvers import solver


def split_index(i, M):
    return (i // M, i % M)


def single_to_multi(m, K, I, J):
    l, rem = split_index(m, K * I * J)
    k, rem = split_index(rem, I * J)
    i, j = split_index(rem, J)
    return (l, k, i, j)


def multi_to_single(l, k, i, j, K, I, J):
    return l * (K * I * J) + k * (I * J) + i * J + j


def _theta(model, params):
    if model == "ssy":
        β, γ, ψ = params[0], params[1], params[2]
    else:
        β, ψ, γ = params[0], params[1], params[2]
    return float(β), float((1 - γ) / (1 - 1 / ψ))


def discretize_single_index(ssy, shapes):
    """(params, arrays, x_states, P_x) with P_x[m, m'] the single-index transition matrix."""
    arrays = discretize_ssy(ssy, shapes)
    L, K, I, J = shapes
    (h_λ_states, h_λ_P, h_c_states, h_c_P, h_z_states, h_z_P, z_states, z_Q, σ_c_states, σ_z_states) = arrays
    # [i, j, i', j'] block of (h_z, z), then Kronecker with the two unconditional chains
    zz = (h_z_P[:, None, :, None] * z_Q[:, :, None, :]).reshape(I * J, I * J)
    P_x = np.kron(h_λ_P, np.kron(h_c_P, zz))
    mesh = np.meshgrid(np.arange(L), np.arange(K), np.arange(I), np.arange(J), indexing="ij")
    l, k, i, j = (m.ravel() for m in mesh)
    x_states = np.stack([h_λ_states[l], h_c_states[k], h_z_states[i], z_states[i, j]])
    return ssy.params, arrays, x_states, P_x


def compute_H_single_index(model, shapes):
    """Dense H (N x N) of the SSY model (temp_ssy.py:109-143) -- or of GCY, same construction."""
    shapes = tuple(int(s) for s in shapes)
    if len(shapes) == 4:
        params, arrays, x_states, P_x = discretize_single_index(model, shapes)
        β, γ, ψ, μ_c = params[:4]
        θ = (1 - γ) / (1 - 1 / ψ)
        L, K, I, J = shapes
        (h_λ_states, _, _, _, _, _, z_states, _, σ_c_states, _) = arrays
        a1 = np.exp(θ * h_λ_states)                                   # next-state h_λ'
        a2 = np.exp(0.5 * ((1 - γ) * σ_c_states) ** 2)                # current σ_c
        a3 = np.exp((1 - γ) * (μ_c + z_states))                       # current z[i, j]
        row = (np.ones(L)[:, None, None, None] * a2[None, :, None, None] * a3[None, None, :, :]).ravel()
        col = np.repeat(a1, K * I * J)
        return row[:, None] * P_x * col[None, :]
    arrays = discretize_gcy(model, shapes)
    params = model.params
    β, ψ, γ = params[0], params[1], params[2]
    μ_c = params[5]
    θ = (1 - γ) / (1 - 1 / ψ)
    na, nb, nc, nd, ne, nf = shapes
    (z_states, z_Q, z_π_states, z_π_Q, h_z_states, h_z_Q, σ_z_states, h_c_states, h_c_Q, σ_c_states,
     h_zπ_states, h_zπ_Q, σ_zπ_states, h_λ_states, h_λ_Q) = arrays
    # H[a,b,c,d,e,f ; A,B,C,D,E,F] (gcy_wc_ratio.py:161-230)
    H = np.einsum("bceaA,ebB,cC,dD,eE,fF->abcdefABCDEF", z_Q, z_π_Q, h_z_Q, h_c_Q, h_zπ_Q, h_λ_Q)
    a1 = np.exp(θ * h_λ_states)
    a2 = np.exp(0.5 * ((1 - γ) * σ_c_states) ** 2)
    a3 = np.exp((1 - γ) * (μ_c + z_states))                           # [b, c, e, a]
    H = H * a1[None, None, None, None, None, None, None, None, None, None, None, :]
    H = H * a2[None, None, None, :, None, None, None, None, None, None, None, None]
    H = H * np.transpose(a3, (3, 0, 1, 2))[:, :, :, None, :, None, None, None, None, None, None, None]
    N = int(np.prod(shapes))
    return np.ascontiguousarray(H.reshape(N, N))


class DenseOperator(KoopmansOperator):
    """T(w) = 1 + β (H @ w^θ)^(1/θ) on the GPU for a materialised H; same methods as KoopmansOperator
    (w is a vector of length N, or any array of N elements -- it is flattened)."""

    def __init__(self, H, beta, theta, device=0):
        H = _as_f64(H)
        if H.ndim != 2 or H.shape[0] != H.shape[1]:
            raise ValueError("H must be a square matrix")
        self.model_name = "dense"
        self.shapes = (H.shape[0],)
        self.device = int(device)
        self.params = (float(beta), float(theta))
        h = C.c_void_p()
        rc = lib.sdfs_create_dense(H.shape[0], H.ctypes.data_as(C.POINTER(C.c_double)), float(beta), float(theta),
                                   self.device, C.byref(h))
        if rc != 0:
            raise _lib.SdfsError(f"sdfs_create_dense failed ({rc}): {_lib.last_error(None)}")
        self._h = h
        self._finalizer = weakref.finalize(self, lib.sdfs_destroy, h)
        self.size = int(lib.sdfs_grid_size(h))

    def _host_in(self, w, name="w"):
        w = _as_f64(w)
        if w.size != self.size:
            raise ValueError(f"{name} has {w.size} elements, operator has N = {self.size}")
        return w.reshape(self.shapes)


_cache = {}


def single_index_T(w, H, params):
    """Drop-in for temp_ssy.py:147-153 (params: the SSY tuple, β first, θ from γ and ψ)."""
    key = (id(H), tuple(float(p) for p in params))
    op = _cache.get(key)
    if op is None:
        if len(_cache) >= 4:
            _cache.pop(next(iter(_cache))).close()
        β, θ = _theta("ssy" if len(params) == 13 else "gcy", params)
        op = DenseOperator(H, β, θ)
        op._keepalive = H
        _cache[key] = op
    w = np.asarray(w)
    return op(w).reshape(w.shape)


def test_compute_wc_ratio_single_index(L, K, I, J, single_index_output=False, algorithm="newton"):
    """Solve a small version of the model through the dense form (temp_ssy.py:156-184)."""
    from .models import SSY
    shapes = L, K, I, J
    ssy = SSY()
    H = compute_H_single_index(ssy, shapes)
    β, θ = _theta("ssy", ssy.params)
    T = DenseOperator(H, β, θ)
    N = L * K * I * J
    w_star = solver(T, np.ones(N) * 800.0, algorithm=algorithm)
    return w_star if single_index_output else np.reshape(w_star, (L, K, I, J))


test_compute_wc_ratio_single_index.__test__ = False      # a driver with the reference's name, not a pytest case
