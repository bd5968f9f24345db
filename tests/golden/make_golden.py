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

    # the continuous modules also use jax.vmap (over axis 0 of the first argument only),
    # jax.lax.map and jax.scipy.ndimage.map_coordinates: plain loops and scipy's function
    def vmap(f, in_axes=0):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else None
        def g(*args):
            if axes is not None:
                assert axes[0] == 0 and all(a is None for a in axes[1:])
            return np.stack([f(a0, *args[1:]) for a0 in args[0]])
        return g
    jax.vmap = vmap
    lax = types.ModuleType("jax.lax")
    lax.map = lambda f, xs: np.stack([f(x) for x in xs])
    jax.lax = lax
    import scipy.ndimage as _ndi
    jsp = types.ModuleType("jax.scipy")
    jndi = types.ModuleType("jax.scipy.ndimage")
    jndi.map_coordinates = _ndi.map_coordinates
    jsp.ndimage = jndi
    jax.scipy = jsp
    qquad = types.ModuleType("quantecon.quad")
    from oracle.continuous import qnwnorm as _qnwnorm
    qquad.qnwnorm = _qnwnorm
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
    qe.quad = qquad
    for name, mod in [("jax", jax), ("jax.numpy", jnp), ("jax.config", cfg_mod), ("jax.lax", lax),
                      ("jax.scipy", jsp), ("jax.scipy.ndimage", jndi),
                      ("numba", numba), ("jaxopt", jaxopt), ("quantecon", qe), ("quantecon.quad", qquad)]:
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

    # --- continuous-state operator (code/ssy/continuous_junnan, code/gcy/continuous): the reference's
    #     build_grid / T_fun_factory / lin_interp run verbatim; quadrature and Monte-Carlo kernels ---
    continuous_fixtures(ssy, gcy)

    # --- single-index dense form (code/ssy/discrete/temp_ssy.py: a scratch file without imports) ---
    dense_fixtures(S, ssy)

    # --- recorded notebook output (sandpit.ipynb:41-44), typed in as data ---
    np.savez(os.path.join(HERE, "sandpit_trace.npz"),
             shapes=np.array((10, 10, 10, 10)),
             errors=np.array([4302.341800771495, 4074.9605304521597,
                              112.01772152357796, 3.834976201446807]))
    print("done")


def dense_fixtures(S, ssy):
    """temp_ssy.py has no import lines (it was cut out of a notebook): its text is executed as it
    stands in a namespace that supplies the names it expects -- numpy, the jax stand-ins, njit, and
    the reference's own discretize_ssy under the name the file uses (discretize_multi_index)."""
    import jax
    ns = {"np": np, "jax": jax, "jnp": jax.numpy, "njit": lambda f=None, **k: f if f is not None else (lambda g: g),
          "discretize_multi_index": S.discretize_ssy, "SSY": S.SSY, "solver": None}
    exec(compile(open(REF + "/ssy/discrete/temp_ssy.py").read(), "temp_ssy.py", "exec"), ns)
    rng = np.random.default_rng(9)
    for shapes in [(2, 3, 2, 3), (3, 2, 4, 3)]:
        H = ns["compute_H_single_index"](ssy, shapes)
        N = int(np.prod(shapes))
        w = 400 + 500 * rng.random(N)
        Tw = np.asarray(ns["single_index_T"](w, H, ssy.params))
        arrays = S.discretize_ssy(ssy, shapes)
        Tm = np.asarray(S.T_ssy(w.reshape(shapes), shapes, ssy.params, arrays))
        np.savez_compressed(os.path.join(HERE, f"dense_ssy_{tag(shapes)}.npz"), shapes=np.array(shapes),
                            params=np.array(ssy.params), H=H, w=w, T_single=Tw, T_multi=Tm)
        print("dense", shapes, float(np.max(np.abs(Tw - Tm.ravel()))))


def _load_by_path(name, path):
    """The SSY and GCY continuous modules define the same function names: load each under its own name."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def continuous_fixtures(ssy, gcy):
    from oracle.continuous import qnwnorm
    CS = _load_by_path("ref_ssy_cont", REF + "/ssy/continuous_junnan/ssy_wc_ratio_continuous.py")
    CG = _load_by_path("ref_gcy_cont", REF + "/gcy/continuous/gcy_wc_ratio_continuous.py")
    rng = np.random.default_rng(7)
    for tagname, C, model, sizes, d in [("ssy", CS, ssy, (3, 4, 3, 5), 3), ("ssy", CS, ssy, (5, 4, 6, 7), 4),
                                        ("gcy", CG, gcy, (2, 3, 2, 3, 4, 3), 2), ("gcy", CG, gcy, (3, 2, 3, 2, 3, 4), 3)]:
        for nsd in (3.2, 1.0):          # 1.0: many next states fall outside the grid (mode='nearest')
            grids = C.build_grid(model, *sizes, nsd)
            dim = len(grids)
            nodes, weights = qnwnorm([d] * dim)
            nodes = np.asarray(nodes.T)
            pars = np.array(model.params)
            w = 5.0 + 20.0 * rng.random(sizes)
            Tq = C.T_fun_factory((pars, grids, nodes, weights), "quadrature", int(np.prod(sizes)))
            draws = rng.standard_normal((dim, 50))
            Tm = C.T_fun_factory((pars, grids, draws), "monte_carlo", int(np.prod(sizes)))
            xq = np.stack([rng.uniform(g[0] - 0.3 * (g[-1] - g[0]), g[-1] + 0.3 * (g[-1] - g[0]), 40) for g in grids])
            out = {"params": pars, "sizes": np.array(sizes), "num_std_devs": np.array(nsd), "d": np.array(d),
                   "nodes": nodes, "weights": weights, "mc_draws": draws, "w": w,
                   "T_quad": np.asarray(Tq(w)), "T_mc": np.asarray(Tm(w)),
                   "x_query": xq, "interp": np.asarray(C.lin_interp(xq, w, grids))}
            out.update({f"grid{i}": np.asarray(g) for i, g in enumerate(grids)})
            fn = f"cont_{tagname}_{tag(sizes)}_sd{nsd}.npz"
            np.savez_compressed(os.path.join(HERE, fn), **out)
            print("continuous", fn, float(out["T_quad"].flat[0]), float(out["T_mc"].flat[0]))
    # the reference's driver end to end (successive approximation from w = 1, its defaults) on a tiny grid
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        grids, w_star = CS.wc_ratio_continuous(ssy, 3, 3, 3, 4, num_std_devs=3.2, d=3,
                                               algorithm="successive_approx", write_to_file=False)
    np.savez_compressed(os.path.join(HERE, "cont_ssy_driver_3x3x3x4.npz"), w_star=np.asarray(w_star),
                        **{f"grid{i}": np.asarray(g) for i, g in enumerate(grids)})
    print("continuous driver", float(np.asarray(w_star).flat[0]))


if __name__ == "__main__":
    main()
