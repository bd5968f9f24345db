"""
Oracle (test infrastructure): ctypes front end of oracle/c/wc_oracle.c, the plain C +
OpenMP twin of the factorised numpy operator (oracle/ssy.py, oracle/gcy.py).
Serves as the multi-core CPU baseline in bench.py ("kind": "port") and as a second
checker at grid sizes where numpy einsum is slow.  Built by ``make -C oracle/c``
(__graft_entry__.build() does it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .models import theta_of

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_SO = os.path.join(_DIR, "libwc_oracle.so")
_lib = None


def usable_cores():
    """Cores this process may use: the affinity mask, cut down to the cgroup CPU quota if there is one (v2:
    /sys/fs/cgroup/cpu.max "quota period"; v1: cpu.cfs_quota_us / cpu.cfs_period_us; "max" / -1 = no quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", _DIR], check=True)
        _lib = C.CDLL(_SO)
        _lib.wc_oracle_apply.restype = C.c_int
        _lib.wc_oracle_num_threads.restype = C.c_int
        if "OMP_NUM_THREADS" not in os.environ and hasattr(_lib, "wc_oracle_set_threads"):
            _lib.wc_oracle_set_threads(C.c_int(usable_cores()))
    return _lib


def num_threads():
    return int(_load().wc_oracle_num_threads())


class COperator:
    """T(w) and jvp(w, v) through the C oracle, for model in {"ssy", "gcy"}."""

    def __init__(self, model, shapes, params, arrays):
        self.shapes = tuple(int(s) for s in shapes)
        D = len(self.shapes)
        arr = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
        qs = np.zeros((D, D), dtype=np.int64)
        a1s = np.zeros(D, dtype=np.int64); a2s = np.zeros(D, dtype=np.int64); a3s = np.zeros(D, dtype=np.int64)
        if model == "ssy":
            beta, gamma, psi, mu_c = params[0], params[1], params[2], params[3]
            theta = theta_of(gamma, psi)
            nl, nc, nz, nj = self.shapes
            self.Q = [arr[1], arr[3], arr[5], arr[7]]
            qs[3, 2] = 1
            a1 = np.exp(theta * arr[0]); a1s[0] = 1
            a2 = np.exp(0.5 * ((1 - gamma) * arr[8]) ** 2); a2s[1] = 1
            a3 = np.exp((1 - gamma) * (mu_c + arr[6])); a3s[2] = nj; a3s[3] = 1
            order = [2, 3, 1, 0]
        else:
            beta, psi, gamma, mu_c = params[0], params[1], params[2], params[5]
            theta = theta_of(gamma, psi)
            na, nb, nc, nd, ne, nf = self.shapes
            self.Q = [arr[1], arr[3], arr[5], arr[8], arr[11], arr[14]]
            qs[0, 1] = nc * ne; qs[0, 2] = ne; qs[0, 4] = 1
            qs[1, 4] = 1
            a1 = np.exp(theta * arr[13]); a1s[5] = 1
            a2 = np.exp(0.5 * ((1 - gamma) * arr[9]) ** 2); a2s[3] = 1
            a3 = np.exp((1 - gamma) * (mu_c + arr[0]))          # [b, c, e, a]
            a3s[0] = 1; a3s[4] = na; a3s[2] = ne * na; a3s[1] = nc * ne * na
            order = [5, 4, 3, 2, 1, 0]
        self.theta, self.beta = float(theta), float(beta)
        self.a1, self.a2, self.a3 = (np.ascontiguousarray(a1), np.ascontiguousarray(a2),
                                     np.ascontiguousarray(a3))
        self.qs, self.a1s, self.a2s, self.a3s = qs, a1s, a2s, a3s
        self.order = np.array(order, dtype=np.int32)
        self.n = np.array(self.shapes, dtype=np.int64)
        self.N = int(np.prod(self.n))
        self._qptr = (C.c_void_p * D)(*[q.ctypes.data for q in self.Q])
        self._work = np.empty(4 * self.N)

    def _run(self, mode, w, v):
        lib = _load()
        w = np.ascontiguousarray(w, dtype=np.float64)
        out = np.empty_like(w)
        vp = np.ascontiguousarray(v, dtype=np.float64).ctypes.data if v is not None else None
        rc = lib.wc_oracle_apply(
            C.c_int(len(self.shapes)), C.c_void_p(self.n.ctypes.data), C.c_void_p(self.order.ctypes.data),
            self._qptr, C.c_void_p(self.qs.ctypes.data),
            C.c_void_p(self.a1.ctypes.data), C.c_void_p(self.a1s.ctypes.data),
            C.c_void_p(self.a2.ctypes.data), C.c_void_p(self.a2s.ctypes.data),
            C.c_void_p(self.a3.ctypes.data), C.c_void_p(self.a3s.ctypes.data),
            C.c_double(self.theta), C.c_double(self.beta), C.c_int(mode),
            C.c_void_p(w.ctypes.data), C.c_void_p(vp), C.c_void_p(out.ctypes.data),
            C.c_void_p(self._work.ctypes.data))
        if rc != 0:
            raise RuntimeError(f"wc_oracle_apply failed: {rc}")
        return out

    def __call__(self, w):
        return self._run(0, w, None)

    def jvp(self, w, v):
        return self._run(1, w, v)
