#!/usr/bin/env python3
"""
Generate golden vectors by importing the REFERENCE's own discrete modules.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite only ever read the committed ``*.npz`` fixtures this script writes.

The reference imports jax / jaxopt / numba / quantecon, none of which is
installed here (plain ModuleNotFoundError, nothing was denied).  Its discrete
path uses only ``jax.numpy`` functions that numpy provides under the same names
(exp, expand_dims, swapaxes, sum, max, abs, ones), ``jax.jit`` (a no-op for
results), ``jax.device_put`` (identity) and ``quantecon.rouwenhorst``.  So
before importing we register stand-in modules: numpy as ``jax.numpy``, identity
``jit`` / ``njit`` / ``device_put``, an empty ``jaxopt`` and a ``quantecon`` whose
``rouwenhorst`` is oracle/rouwenhorst.py (a restatement of the published
algorithm -- quantecon itself is third-party and absent).  The reference's
``T_ssy`` / ``T_gcy`` / ``*_loops`` / ``discretize_*`` / ``successive_approx`` /
``solver`` then run VERBATIM from /root/reference; no reference text is copied.

Fixtures written (tests/golden/):
  ssy_<shape>.npz, gcy_<shape>.npz : discretize_* outputs, T(w) and T_loops(w)
      at w = exp(default_rng(0).standard_normal(shapes)) and at w = 800
  sa_*.npz  : successive_approx fixed points + iteration counts + error traces
  sandpit_trace.npz : the recorded Newton trace of sandpit.ipynb:41-44 (typed in)
"""
import io
import os
import sys
import types
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle.rouwenhorst import rouwenhorst as _rouwenhorst  # noqa: E402


def install_shims():
    def passthrough(f=None, **kw):
        if f is None:
            return lambda g: g
        return f

    jnp = types.ModuleType("jax.numpy")
    for name in dir(np):
        if not name.startswith("_"):
            setattr(jnp, name, getattr(np, name))
    jax = types.ModuleType("jax")
    jax.numpy = jnp
    jax.jit = passthrough
    jax.device_put = lambda x: x
    cfg_mod = types.ModuleType("jax.config")
    cfg_mod.config = types.SimpleNamespace(update=lambda *a, **k: None)
    jax.config = cfg_mod
    numba = types.ModuleType("numba")
    numba.njit = passthrough
    jaxopt = types.ModuleType("jaxopt")
    qe = types.ModuleType("quantecon")
    qe.rouwenhorst = _rouwenhorst
    qe.tic = lambda: None
    qe.toc = lambda: 0.0
    for name, mod in [("jax", jax), ("jax.numpy", jnp), ("jax.config", cfg_mod),
                      ("numba", numba), ("jaxopt", jaxopt), ("quantecon", qe)]:
        sys.modules[name] = mod


def import_reference():
    install_shims()
    for p in (REF, REF + "/ssy", REF + "/ssy/discrete", REF + "/gcy", REF + "/gcy/discrete"):
        sys.path.insert(0, p)
    import ssy_wc_ratio as S
    S.rouwenhorst = _rouwenhorst          # the reference forgets this import (NameError at HEAD)
    import gcy_wc_ratio as G
    import solvers as SOL
    return S, G, SOL


def w_random(shapes):
    return np.exp(np.random.default_rng(0).standard_normal(shapes))


def tag(shapes):
    return "x".join(str(s) for s in shapes)


SSY_ARR = ["h_lam_states", "h_lam_Q", "h_c_states", "h_c_Q", "h_z_states", "h_z_Q",
           "z_states", "z_Q", "sigma_c_states", "sigma_z_states"]
GCY_ARR = ["z_states", "z_Q", "z_pi_states", "z_pi_Q", "h_z_states", "h_z_Q", "sigma_z_states",
           "h_c_states", "h_c_Q", "sigma_c_states", "h_zpi_states", "h_zpi_Q", "sigma_zpi_states",
           "h_lam_states", "h_lam_Q"]


def main():
    S, G, SOL = import_reference()
    ssy = S.SSY()
    gcy = G.GCY()

    # --- operator fixtures -------------------------------------------------
    for shapes, loops in [((3, 3, 3, 3), True), ((2, 3, 4, 5), True), ((4, 7, 6, 5), True),
                          ((10, 10, 10, 10), False)]:
        arrays = S.discretize_ssy(ssy, shapes)
        out = {"params": np.array(ssy.params), "shapes": np.array(shapes)}
        out.update({"arr_" + n: a for n, a in zip(SSY_ARR, arrays)})
        wr = w_random(shapes)
        w8 = np.full(shapes, 800.0)
        out["w_rand"] = wr
        out["T_rand"] = np.asarray(S.T_ssy(wr, shapes, ssy.params, arrays))
        out["T_800"] = np.asarray(S.T_ssy(w8, shapes, ssy.params, arrays))
        if loops:
            out["Tloops_rand"] = S.T_ssy_loops(wr, shapes, ssy.params, arrays)
        np.savez_compressed(os.path.join(HERE, f"ssy_{tag(shapes)}.npz"), **out)
        print("ssy", shapes, "ok")

    for shapes, loops in [((2, 3, 2, 3, 2, 3), True), ((3,) * 6, True), ((2, 3, 4, 5, 6, 7), False)]:
        arrays = G.discretize_gcy(gcy, shapes)
        out = {"params": np.array(gcy.params), "shapes": np.array(shapes)}
        out.update({"arr_" + n: a for n, a in zip(GCY_ARR, arrays)})
        wr = w_random(shapes)
        w8 = np.full(shapes, 800.0)
        out["w_rand"] = wr
        out["T_rand"] = np.asarray(G.T_gcy(wr, shapes, gcy.params, arrays))
        out["T_800"] = np.asarray(G.T_gcy(w8, shapes, gcy.params, arrays))
        if loops:
            out["Tloops_rand"] = G.T_gcy_loops(wr, shapes, gcy.params, arrays)
        np.savez_compressed(os.path.join(HERE, f"gcy_{tag(shapes)}.npz"), **out)
        print("gcy", shapes, "ok")

    # --- successive-approximation fixed points via the reference's own loop ---
    def run_sa(T, shapes, tol):
        errs = []
        x = np.ones(shapes) * 800.0
        # the reference's successive_approx, verbatim, wrapped to record errors
        def Trec(w):
            out = T(w)
            errs.append(float(np.max(np.abs(out - w))))
            return out
        with contextlib.redirect_stdout(io.StringIO()):
            x, n = SOL.successive_approx(Trec, x, tol=tol, verbose=False)
        return np.asarray(x), n, np.array(errs)

    for shapes in [(3, 3, 3, 3), (2, 3, 4, 5)]:
        arrays = S.discretize_ssy(ssy, shapes)
        T = lambda w: S.T_ssy(w, shapes, ssy.params, arrays)
        out = {"shapes": np.array(shapes)}
        for tol, nm in [(1e-7, "1e7"), (1e-8, "1e8")]:
            x, n, errs = run_sa(T, shapes, tol)
            out[f"w_{nm}"] = x
            out[f"n_{nm}"] = np.array(n)
            if nm == "1e8":
                out["errors"] = errs
            print("sa ssy", shapes, tol, n, x.flat[0])
        np.savez_compressed(os.path.join(HERE, f"sa_ssy_{tag(shapes)}.npz"), **out)

    shapes = (3,) * 6
    arrays = G.discretize_gcy(gcy, shapes)
    T = lambda w: G.T_gcy(w, shapes, gcy.params, arrays)
    x, n, errs = run_sa(T, shapes, 1e-7)
    np.savez_compressed(os.path.join(HERE, f"sa_gcy_{tag(shapes)}.npz"),
                        shapes=np.array(shapes), w_1e7=x, n_1e7=np.array(n), errors=errs)
    print("sa gcy", shapes, n, x.flat[0], x.min(), x.max())

    # solver() front end with the reference defaults (successive_approx, tol 1e-7)
    shapes = (2, 3, 4, 5)
    arrays = S.discretize_ssy(ssy, shapes)
    T = lambda w: S.T_ssy(w, shapes, ssy.params, arrays)
    with contextlib.redirect_stdout(io.StringIO()):
        xs = SOL.solver(T, np.ones(shapes) * 800.0, algorithm="successive_approx", verbose=False)
    np.savez_compressed(os.path.join(HERE, "solver_front_ssy_2x3x4x5.npz"), w=np.asarray(xs))

    # --- log-linear approximation (ssy_model.py:88-156, gcy_model.py:80-159), reference functions ---
    rng = np.random.default_rng(3)
    import ssy_model as SM
    import gcy_model as GM
    f_ssy = SM.wc_loglinear_factory(ssy)
    xs = rng.standard_normal((40, 4)) * np.array([0.002, 0.3, 0.3, 0.003])
    f_gcy = GM.wc_loglinear_factory(gcy)
    xg = rng.standard_normal((40, 6)) * np.array([0.002, 0.3, 0.3, 0.5, 0.003, 0.003])
    np.savez_compressed(os.path.join(HERE, "loglinear.npz"),
                        x_ssy=xs, q_ssy=np.array([f_ssy(x) for x in xs]),
                        x_gcy=xg, q_gcy=np.array([f_gcy(x) for x in xg]))
    print("loglinear ok", f_ssy(xs[0]), f_gcy(xg[0]))

    # --- recorded notebook output (sandpit.ipynb:41-44), typed in as data ---
    np.savez(os.path.join(HERE, "sandpit_trace.npz"),
             shapes=np.array((10, 10, 10, 10)),
             errors=np.array([4302.341800771495, 4074.9605304521597,
                              112.01772152357796, 3.834976201446807]))
    print("done")


if __name__ == "__main__":
    main()
