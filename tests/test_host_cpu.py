"""
CPU-side checks (no GPU needed): the C-ABI library loads and exports every symbol
include/sdfs_hip.h declares, the host discretisation matches the golden vectors,
the model classes match the reference's params order, and the host solver loops
(the path taken for foreign callables) reproduce the reference's iteration counts.
No compute call goes through the HIP library here.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_golden, golden_arrays


def tag(s):
    return "x".join(map(str, s))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "sdfs_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sdfs_[a-z_A-Z0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    from sdfs_via_autodiff_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in sdfs_hip.h but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_opts_struct_layout_and_defaults():
    from sdfs_via_autodiff_amd import _lib
    o = _lib.default_opts()
    assert (o.tol, o.max_iter, o.inner_rtol, o.inner_atol) == (1e-7, 1000000, 1e-5, 1e-4)
    assert (o.history, o.mixing_freq, o.beta, o.ridge) == (10, 4, 8.0, 1e-6)
    assert (o.krylov_f32, o.t_f32) == (0, 0)
    assert ctypes.sizeof(_lib.sdfs_opts) == 88          # round 3: + t_f32 (and padding to 8)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sdfs_via_autodiff_amd as S
    with pytest.raises(S.SdfsError, match="no HIP device"):
        S.ssy_operator((3, 3, 3, 3), S.SSY().params, S.discretize_ssy(S.SSY(), (3, 3, 3, 3)))


def test_model_params_order():
    import sdfs_via_autodiff_amd as S
    assert np.array_equal(np.array(S.SSY().params), load_golden("ssy_3x3x3x3.npz")["params"])
    assert np.array_equal(np.array(S.GCY().params), load_golden("gcy_3x3x3x3x3x3.npz")["params"])
    assert S.SSY(γ=5.0).params[1] == 5.0 and S.GCY(ψ=1.2).params[1] == 1.2
    assert abs(S.SSY().θ - (1 - 8.89) / (1 - 1 / 1.97)) < 1e-15


@pytest.mark.parametrize("shapes", [(3, 3, 3, 3), (2, 3, 4, 5), (4, 7, 6, 5), (10, 10, 10, 10)])
def test_discretize_ssy_matches_reference(shapes):
    import sdfs_via_autodiff_amd as S
    g = load_golden(f"ssy_{tag(shapes)}.npz")
    got = S.discretize_ssy(S.SSY(), shapes)
    assert len(got) == 10
    for a, b in zip(got, golden_arrays(g, "ssy")):
        assert a.shape == b.shape and a.dtype == np.float64
        np.testing.assert_allclose(a, b, rtol=1e-14, atol=0)


@pytest.mark.parametrize("shapes", [(2, 3, 2, 3, 2, 3), (3,) * 6, (2, 3, 4, 5, 6, 7)])
def test_discretize_gcy_matches_reference(shapes):
    import sdfs_via_autodiff_amd as S
    g = load_golden(f"gcy_{tag(shapes)}.npz")
    got = S.discretize_gcy(S.GCY(), shapes)
    assert len(got) == 15
    for a, b in zip(got, golden_arrays(g, "gcy")):
        assert a.shape == b.shape and a.dtype == np.float64
        np.testing.assert_allclose(a, b, rtol=1e-14, atol=0)


def test_rouwenhorst_rows_sum_to_one_and_moments():
    import sdfs_via_autodiff_amd as S
    mc = S.rouwenhorst(9, 0.9, 0.3, 0.2)
    np.testing.assert_allclose(mc.P.sum(axis=1), 1.0, rtol=1e-14)
    assert abs(mc.state_values.mean() - 0.2 / (1 - 0.9)) < 1e-12
    with pytest.raises(ValueError):
        S.rouwenhorst(1, 0.9, 0.3)


def _oracle_T(shapes):
    from oracle import models, ssy
    p = models.ssy_params()
    arr = ssy.discretize_ssy(p, shapes)
    T = lambda w: ssy.T_ssy_factorised(w, shapes, p, arr)
    T.jvp = lambda w, v: ssy.jvp_ssy(w, v, shapes, p, arr)
    return T


def test_host_loop_successive_approx_foreign_callable(capsys):
    """Foreign callables go through the host loop; counts must equal the reference's."""
    import sdfs_via_autodiff_amd as S
    shapes = (3, 3, 3, 3)
    g = load_golden("sa_ssy_3x3x3x3.npz")
    T = _oracle_T(shapes)
    x, n = S.successive_approx(lambda w: T(w), np.full(shapes, 800.0), verbose=True)
    assert n == int(g["n_1e7"]) == 10385
    np.testing.assert_allclose(x, g["w_1e7"], atol=1e-9, rtol=0)
    out = capsys.readouterr().out
    assert out.startswith("Beginning iteration\n\n")
    assert "iter = 0, error = " in out and "iter = 10000, error = " in out
    assert "Iteration converged after 10385 iterations" in out


def test_host_loop_newton_and_anderson_foreign_callable():
    import sdfs_via_autodiff_amd as S
    from oracle import solvers as osol
    shapes = (3, 3, 3, 3)
    T = _oracle_T(shapes)
    x, n = S.newton_solver(T, np.full(shapes, 800.0), verbose=False)
    xo, no = osol.newton_solver(T, np.full(shapes, 800.0), verbose=False, jvp=T.jvp)
    assert n == no
    np.testing.assert_allclose(x, xo, atol=1e-8, rtol=0)
    xa, na = S.anderson_solver(lambda w: T(w), np.full(shapes, 800.0), tol=1e-6, verbose=False)
    xb, nb = osol.anderson_solver(T, np.full(shapes, 800.0), tol=1e-6, verbose=False)
    assert na == nb
    np.testing.assert_allclose(xa, xb, atol=1e-9, rtol=0)
    # a foreign callable without .jvp: forward-difference directional derivatives (the reference's call
    # shape -- a plain lambda with the default algorithm -- must never raise)
    xf, nf = S.newton_solver(lambda w: T(w), np.full(shapes, 800.0), verbose=False)
    assert abs(nf - no) <= 1
    np.testing.assert_allclose(xf, xo, atol=5e-3, rtol=0)
    assert np.max(np.abs(T(xf) - xf)) < 1e-3
    xs = S.solver(lambda w: T(w), np.full(shapes, 800.0))          # default algorithm = "newton"
    np.testing.assert_allclose(xs, xo, atol=5e-3, rtol=0)


class _StubOperator:
    """Stands in for a device operator on CPU: same tracing hook, a host loop as `solve`."""

    def __new__(cls, *a, **k):
        from sdfs_via_autodiff_amd.operators import KoopmansOperator

        class Stub(KoopmansOperator):
            def __init__(self, f, shapes):
                self.f, self.shapes, self.solves = f, shapes, 0

            def __call__(self, w):
                from sdfs_via_autodiff_amd.operators import _record_call
                out = self.f(np.asarray(w, dtype=np.float64))
                _record_call(self, w, out)
                return out

            def solve(self, x_init, algorithm="successive_approx", record_errors=False, **kw):
                self.solves += 1
                x, errs = np.asarray(x_init, dtype=np.float64), []
                for _ in range(int(kw.get("max_iter", 10 ** 6))):
                    xn = self.f(x); e = float(np.max(np.abs(xn - x))); errs.append(e); x = xn
                    if e <= kw.get("tol", 1e-7):
                        break
                return x, len(errs), dict(n_apply=len(errs), final_err=errs[-1], errors=np.array(errs), status=0)
        return Stub(*a, **k)


def test_closure_over_a_device_operator_is_resolved(capsys):
    """The reference's drivers pass `lambda w: T_ssy(w, shapes, params, arrays)`; the solvers must find the
    operator behind it (one traced call) and refuse closures that are not exactly that operator."""
    import sdfs_via_autodiff_amd as S
    from sdfs_via_autodiff_amd.solvers import _resolve_operator
    f = lambda x: 0.5 * x + 1.0
    op = _StubOperator(f, (3,))
    x0 = np.zeros(3)
    assert _resolve_operator(op, x0) is op
    assert _resolve_operator(lambda w: op(w), x0) is op
    # the operator must receive the very object: equal values are not enough (a clipping closure is
    # the identity on its argument at the start value and at the fixed point, and only there)
    assert _resolve_operator(lambda w: op(w * 1.0), x0) is None
    assert _resolve_operator(lambda w: op(np.clip(w, -1.0, 1.0)), x0) is None
    assert _resolve_operator(lambda w: op(np.asarray(w)), x0) is op
    assert _resolve_operator(lambda w: op(w) + 0.0, x0) is None          # result touched
    assert _resolve_operator(lambda w: op(w + 1.0), x0) is None          # argument changed
    assert _resolve_operator(lambda w: op(op(w)), x0) is None            # two applications
    assert _resolve_operator(f, x0) is None                              # foreign
    # solver front end: device path taken once, reference messages printed
    x = S.solver(lambda w: op(w), x0, algorithm="successive_approx")
    np.testing.assert_allclose(x, 2.0, atol=1e-6)
    assert op.solves == 1
    out = capsys.readouterr().out
    assert out.startswith("Beginning iteration") and "iter = 0, error = 1.0" in out and "Iteration converged after" in out
    for algo in ("newton", "anderson"):
        S.solver(lambda w: op(w), x0, algorithm=algo)
    assert op.solves == 3
    # a closure that changes behaviour after the probe is caught by the confirmation and redone on the host
    state = {"n": 0}

    def shifty(w):
        state["n"] += 1
        return op(w) if state["n"] == 1 else f(w)
    x, n = S.successive_approx(shifty, x0, verbose=False)
    np.testing.assert_allclose(x, 2.0, atol=1e-6)
    assert "not a pure closure" in capsys.readouterr().out


def test_solver_front_end_fallback_and_registry(capsys):
    import sdfs_via_autodiff_amd as S
    assert set(S.solvers) == {"newton", "anderson", "gd", "successive_approx"}
    f = lambda x: 0.5 * x + 1.0
    x = S.solver(f, np.zeros(3), algorithm="no-such-algorithm")
    np.testing.assert_allclose(x, 2.0, atol=1e-6)
    out = capsys.readouterr().out
    assert "Algorithm no-such-algorithm not found." in out
    assert "Falling back to successive approximation." in out
    with pytest.raises(TypeError, match="vjp"):                  # "gd" differentiates the loss: needs a VJP
        S.solver(f, np.zeros(3), algorithm="gd")


def test_gd_host_path_matches_oracle_restatement():
    """Registry entry "gd" (code/solvers.py:127-140) on a foreign callable that brings its own vjp: the host loop
    equals the oracle's restatement of jaxopt.GradientDescent (FISTA + backtracking; unpinned third party -- parity
    here is with that restatement of ProximalGradient's rules, NOT with jaxopt itself).  The state answers to the
    attribute names of jaxopt's ProxGradState."""
    import sdfs_via_autodiff_amd as S
    from oracle import solvers as osol
    A = np.array([[0.5, 0.2, 0.0], [0.1, 0.4, 0.1], [0.0, 0.3, 0.5]])
    b = np.array([1.0, 2.0, 3.0])

    class F:
        def __call__(self, x):
            return A @ x + b

        def vjp(self, x, u):
            return A.T @ u
    f = F()
    x, state = S.fixed_point_via_gradient_decent(f, np.zeros(3))
    xo, no = osol.fixed_point_via_gradient_decent(f, np.zeros(3), f.vjp)
    assert state["iter_num"] == state.iter_num == no and no < 1000
    assert state.error <= 1e-4 and state.error == state["errors"][-1] and state.stepsize > 0 and state.t > 1
    np.testing.assert_allclose(state["errors"], osol.fixed_point_via_gradient_decent.last_errors, rtol=1e-12)
    # the stopping error is the gradient of the loss at the extrapolated point: the loop ends at the first one <= tol
    assert np.all(state["errors"][:-1] > 1e-4)
    np.testing.assert_allclose(x, xo, rtol=0, atol=1e-12)
    np.testing.assert_allclose(x, np.linalg.solve(np.eye(3) - A, b), atol=1e-3)
    np.testing.assert_allclose(S.solver(f, np.zeros(3), algorithm="gd"), xo, atol=1e-12)


def test_max_iter_warning(capsys):
    import sdfs_via_autodiff_amd as S
    x, n = S.successive_approx(lambda x: x + 1.0, np.zeros(2), max_iter=5, verbose=False)
    assert n == 5
    assert "Warning: Hit maximum iteration number 5" in capsys.readouterr().out


def test_loglinear_matches_reference():
    """wc_loglinear_factory against values from the reference's own functions (tests/golden/loglinear.npz)."""
    import sdfs_via_autodiff_amd as S
    g = load_golden("loglinear.npz")
    f = S.wc_loglinear_factory(S.SSY())
    np.testing.assert_allclose([f(x) for x in g["x_ssy"]], g["q_ssy"], rtol=1e-12)
    np.testing.assert_allclose(f(tuple(g["x_ssy"].T)), g["q_ssy"], rtol=1e-12)      # array form
    f = S.wc_loglinear_factory(S.GCY())
    np.testing.assert_allclose([f(x) for x in g["x_gcy"]], g["q_gcy"], rtol=1e-12)
    with pytest.raises(TypeError):
        S.wc_loglinear_factory(object())


def test_loglinear_guess_shapes_and_values():
    import sdfs_via_autodiff_amd as S
    m = S.SSY(); shapes = (3, 4, 5, 6); arr = S.discretize_ssy(m, shapes)
    w0 = S.loglinear_guess(m, shapes, arr)
    f = S.wc_loglinear_factory(m)
    assert w0.shape == shapes
    l, k, i, j = 2, 1, 3, 4
    assert abs(w0[l, k, i, j] - (np.exp(f((arr[0][l], arr[2][k], arr[4][i], arr[6][i, j]))) + 1)) < 1e-9
    m = S.GCY(); shapes = (2, 3, 4, 2, 3, 2); arr = S.discretize_gcy(m, shapes)
    w0 = S.loglinear_guess(m, shapes, arr)
    f = S.wc_loglinear_factory(m)
    a, b, c, d, e, ff = 1, 2, 3, 0, 1, 1
    x = (arr[13][ff], arr[7][d], arr[4][c], arr[10][e], arr[0][b, c, e, a], arr[2][e, b])
    assert w0.shape == shapes and abs(w0[a, b, c, d, e, ff] - (np.exp(f(x)) + 1)) < 1e-9


def test_continuous_host_side_matches_reference_golden(tmp_path):
    """build_grid / qnwnorm / vals_to_coords and the result file of the continuous path (host code)."""
    import glob
    import sdfs_via_autodiff_amd as S
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cont_*_sd*.npz")))
    assert len(files) == 8
    for fn in files:
        z = np.load(fn)
        model = S.SSY() if "cont_ssy" in fn else S.GCY()
        grids = S.build_grid(model, *[int(s) for s in z["sizes"]], float(z["num_std_devs"]))
        for i, g in enumerate(grids):
            np.testing.assert_array_equal(g, z[f"grid{i}"])
        nodes, weights = S.qnwnorm([int(z["d"])] * len(grids))
        np.testing.assert_array_equal(nodes.T, z["nodes"])
        np.testing.assert_array_equal(weights, z["weights"])
        c = S.vals_to_coords(grids, np.stack([g[[0, -1]] for g in grids]))
        np.testing.assert_allclose(c[:, 0], 0.0, atol=1e-12)
        np.testing.assert_allclose(c[:, 1], z["sizes"] - 1, rtol=1e-12)
    # ragged (the reference's default sizes) and square grids both round-trip through the file
    for grids in (S.build_grid(S.SSY(), 3, 3, 3, 4), S.build_grid(S.SSY(), 3, 3, 3, 3)):
        w = np.random.default_rng(0).random(tuple(len(g) for g in grids))
        fn = str(tmp_path / "w.npy")
        S.save_wstar(fn, grids, w)
        g2, w2 = S.load_wstar(fn)
        np.testing.assert_array_equal(w2, w)
        for a, b in zip(grids, g2):
            np.testing.assert_array_equal(a, b)
    with pytest.raises(TypeError):
        S.build_grid(S.SSY(), 3, 3)


def test_tauchen_chains():
    import sdfs_via_autodiff_amd as S
    for n, rho, sig, mu in [(7, 0.9, 0.1, 0.0), (15, 0.987, 0.02, 0.003), (2, 0.5, 1.0, 0.0)]:
        mc = S.tauchen(n, rho, sig, mu)
        np.testing.assert_allclose(mc.P.sum(1), 1.0, atol=1e-14)
        assert np.all(mc.P >= 0) and np.all(np.diff(mc.state_values) > 0)
        np.testing.assert_allclose(mc.state_values.mean(), mu / (1 - rho), atol=1e-12)
        # conditional mean of the chain tracks mu + rho*y away from the edges
        if n >= 7:
            i = n // 2
            assert abs(mc.P[i] @ mc.state_values - (mu + rho * mc.state_values[i])) < 0.02 * sig + 1e-3 * abs(mu)
    a = S.discretize_ssy(S.SSY(), (3, 4, 5, 6), method="tauchen")
    b = S.discretize_ssy(S.SSY(), (3, 4, 5, 6))
    assert [x.shape for x in a] == [x.shape for x in b]
    with pytest.raises(ValueError):
        S.discretize_gcy(S.GCY(), (2,) * 6, method="simpson")


def test_single_index_H_matches_reference_golden():
    """compute_H_single_index (Kronecker assembly) against the H the reference's temp_ssy.py builds."""
    import sdfs_via_autodiff_amd as S
    for name in ("dense_ssy_2x3x2x3", "dense_ssy_3x2x4x3"):
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
        shapes = tuple(int(s) for s in z["shapes"])
        H = S.compute_H_single_index(S.SSY(), shapes)
        np.testing.assert_allclose(H, z["H"], rtol=1e-14, atol=0)
        params, arrays, x_states, P_x = S.discretize_single_index(S.SSY(), shapes)
        np.testing.assert_allclose(P_x.sum(1), 1.0, atol=1e-13)
        L, K, I, J = shapes
        m = S.multi_to_single(1, 1, 1, 2, K, I, J)
        assert S.single_to_multi(m, K, I, J) == (1, 1, 1, 2)
        assert x_states[3, m] == arrays[6][1, 2] and x_states[0, m] == arrays[0][1]
        # the reference's own numbers: dense and multi-index T agree
        np.testing.assert_allclose(z["T_single"], z["T_multi"].ravel(), rtol=1e-12)


def test_header_is_plain_c_and_binds_from_c(tmp_path):
    """include/sdfs_hip.h must be consumable by a C compiler (the boundary is a C ABI, no C++ or torch
    types): compile a C99 translation unit that takes the address of every declared entry point."""
    import subprocess
    hdr = open(os.path.join(REPO, "include", "sdfs_hip.h")).read()
    names = sorted(set(re.findall(r"\b(sdfs_[a-z_A-Z0-9]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S))))
    src = tmp_path / "bind.c"
    src.write_text('#include "sdfs_hip.h"\n#include <stddef.h>\n'
                   "const void* sdfs_entry_points[] = {\n" +
                   "".join(f"  (const void*)(size_t)&{n},\n" for n in names) + "};\n"
                   "int n_entry_points(void) { return (int)(sizeof sdfs_entry_points / sizeof sdfs_entry_points[0]); }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(REPO, "include"),
                        "-c", str(src), "-o", str(tmp_path / "bind.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_bench_gpus_flag_never_falls_back_to_one_gpu():
    """`python bench.py --gpus N` without a launcher spawns its own ranks (as a child process) and must exit
    non-zero -- never print an N = 1 line -- when fewer than N devices or ranks come up."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the rehearsal run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "needs 2 devices" in r.stderr and '"metric"' not in r.stdout
    # under a launcher whose world size disagrees with --gpus
    env2 = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env2)
    assert r.returncode != 0 and "WORLD_SIZE 2" in r.stderr and '"metric"' not in r.stdout
    # the spawn command line: one rank per GPU, rendezvous on 127.0.0.1, this script, the same arguments
    sys.path.insert(0, REPO)
    import bench
    captured = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None):
        captured["cmd"], captured["env"] = cmd, env
        return R()
    old_run, old_cnt, old_argv = subprocess.run, torch.cuda.device_count, sys.argv
    subprocess.run, torch.cuda.device_count, sys.argv = fake_run, (lambda: 8), ["bench.py", "--gpus", "4", "--steps", "3"]
    try:
        assert bench.spawn_ranks(4, "nccl") == 7
    finally:
        subprocess.run, torch.cuda.device_count, sys.argv = old_run, old_cnt, old_argv
    cmd = captured["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert captured["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
