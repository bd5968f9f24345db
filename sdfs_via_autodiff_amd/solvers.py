"""
Fixed-point solvers with the reference's names, signatures, defaults and messages
(code/solvers.py:16-177):

    successive_approx(f, x_init, tol=1e-7, max_iter=1e6, verbose=True, print_skip=1000)
    newton_solver(f, x_init, tol=1e-7, max_iter=1e6, bicgstab_atol=1e-4, verbose=True, print_skip=1)
    anderson_solver(f, x_init, tol=1e-7, max_iter=10000, verbose=True)
    solver(f, x_init, algorithm="newton", verbose=True)  -> x_star
    solvers = {"newton", "anderson", "gd", "successive_approx"}

When ``f`` is a ``KoopmansOperator`` the whole iteration runs on the GPU through
``sdfs_solve`` (device-resident iterates, fused residual reduction, hipGraph-replayed
iteration chunks, on-device BiCGSTAB) and only the error trace comes back for
printing.

The reference's drivers do not pass an operator object but a closure,
``T = lambda w: T_ssy(w, shapes, params, arrays)`` (ssy_wc_ratio.py:230,
gcy_wc_ratio.py:333), and rely on jax.jit / jax.jvp seeing through it.  Here the solvers
see through it by tracing one call (``_resolve_operator``): if ``f(x_init)`` turns out to be
exactly one application of one device operator to ``x_init`` whose result is returned
untouched, the loop runs on the device with that operator, and the same trace on the
result confirms the equivalence before it is returned.  Anything else (a genuinely
foreign callable) is driven by the same loops on the host, one ``f(x)`` call per
iteration exactly as the reference does; Newton then uses ``f.jvp`` when the callable has
one and a forward-difference directional derivative otherwise.
"""
from textwrap import dedent

import numpy as np

from .operators import KoopmansOperator, trace_calls

default_tolerance = 1e-7
default_max_iter = int(1e6)


def _traced_single_apply(f, x):
    """Call f(x) under a trace.  Returns (y, op) where op is the device operator when the call
    was exactly ``op(x)`` passed through unchanged -- the operator must have received the very object
    ``x`` (equal values are not enough: ``lambda w: T(np.clip(w, lo, hi))`` is the identity on its
    argument at the start value and at the fixed point, and only there) and its result must come back
    as the very object it returned -- else None."""
    with trace_calls() as calls:
        y = f(x)
    if len(calls) == 1:
        op, w_in, out = calls[0]
        if out is y and w_in is x:
            return y, op
    return y, None


def _resolve_operator(f, x_init):
    """The device operator behind a callable, or None.  A ``KoopmansOperator`` is itself; a closure
    over ``T_ssy`` / ``T_gcy`` / an operator object is recognised by tracing ``f(x_init)``."""
    if isinstance(f, KoopmansOperator):
        return f
    try:
        _, op = _traced_single_apply(f, x_init)
    except Exception:
        return None
    return op


def _confirm(f, op, x_star):
    """After a device solve through a traced closure: f must still be that one operator at x_star."""
    if f is op:
        return True
    _, op2 = _traced_single_apply(f, x_star)
    return op2 is op


def _on_device(f, x_init, algorithm, verbose, print_skip, max_iter, banner=False, **kw):
    """Device-resident solve when ``f`` is (a closure over) a device operator.  Returns
    (x, n) or None when f is foreign or failed the confirmation (the caller then runs its host loop)."""
    op = _resolve_operator(f, x_init)
    if op is None:
        return None
    if banner and verbose:
        print("Beginning iteration\n\n")
    x, n, info = op.solve(x_init, algorithm, record_errors=bool(verbose) and print_skip is not None,
                          max_iter=int(max_iter), **kw)
    if not _confirm(f, op, x):
        print("Warning: the callable is not a pure closure over one device operator; "
              "repeating the iteration on the host")
        return None
    if verbose and print_skip is not None:
        for it in range(0, n, print_skip):
            print("iter = {}, error = {}".format(it, info["errors"][it]))
    _report(verbose, n, max_iter)
    return x, n


def _report(verbose, current_iter, max_iter):
    if current_iter == max_iter:
        print(f"Warning: Hit maximum iteration number {max_iter}")
    elif verbose:
        print(f"Iteration converged after {current_iter} iterations")


def successive_approx(f, x_init, tol=default_tolerance, max_iter=default_max_iter,
                      verbose=True, print_skip=1000, _host_only=False, **device_opts):
    "Uses successive approximation on f."
    if verbose:
        print("Beginning iteration\n\n")
    # (_host_only: the Newton map of a foreign callable is known not to be a device operator -- tracing it would cost
    # one whole inner solve whose result is thrown away)
    done = None if _host_only else _on_device(f, x_init, "successive_approx", verbose, print_skip, max_iter, tol=tol, **device_opts)
    if done is not None:
        return done

    current_iter = 0
    x = x_init
    error = tol + 1
    while error > tol and current_iter < max_iter:
        x_new = f(x)
        error = np.max(np.abs(x_new - x))
        if verbose and current_iter % print_skip == 0:
            print("iter = {}, error = {}".format(current_iter, error))
        current_iter += 1
        x = x_new
    _report(verbose, current_iter, max_iter)
    return x, current_iter


def _host_bicgstab(A, b, tol=1e-5, atol=0.0, maxiter=None):
    """BiCGSTAB with jax.scipy.sparse.linalg.bicgstab's conventions (x0 = 0, stop on
    |r|^2 <= max(tol^2 |b|^2, atol^2)); used only for foreign callables."""
    shape = b.shape
    b = np.asarray(b, dtype=np.float64).ravel()
    mv = lambda u: np.asarray(A(u.reshape(shape))).ravel()
    maxiter = 10 * b.size if maxiter is None else maxiter
    atol2 = max(tol * tol * float(b @ b), atol * atol)
    x = np.zeros_like(b)
    r = b.copy(); rhat = b.copy(); p = b.copy(); q = b.copy()
    alpha = omega = rho = 1.0
    k = 0
    while float(r @ r) > atol2 and 0 <= k < maxiter:
        rho_new = float(rhat @ r)
        beta = rho_new / rho * alpha / omega
        p = r + beta * (p - omega * q)
        q = mv(p)
        alpha = rho_new / float(rhat @ q)
        s = r - alpha * q
        if float(s @ s) < atol2:
            x = x + alpha * p
            r = s
        else:
            t = mv(s)
            omega = float(t @ s) / float(t @ t)
            x = x + alpha * p + omega * s
            r = s - omega * t
        k = -1 if (rho_new == 0 or omega == 0 or alpha == 0) else k + 1
        rho = rho_new
    return x.reshape(shape)


def newton_solver(f, x_init, tol=default_tolerance, max_iter=default_max_iter,
                  bicgstab_atol=1e-4, verbose=True, print_skip=1, **device_opts):
    """
    Newton's method on g(x) = f(x) - x with a matrix-free Jacobian-vector product and
    BiCGSTAB for J(x)^{-1} g(x); the outer loop is successive approximation on
    q(x) = x - J(x)^{-1} g(x), as in the reference.
    """
    done = _on_device(f, x_init, "newton", verbose, print_skip, max_iter, banner=True, tol=tol,
                      inner_atol=bicgstab_atol, **device_opts)
    if done is not None:
        return done

    # foreign callable: the reference differentiates f with jax.jvp (code/solvers.py:87); here f.jvp when
    # the callable brings one, else a forward difference along v (the usual Jacobian-free Newton-Krylov step)
    jvp = getattr(f, "jvp", None)

    def q(x):
        x = np.asarray(x, dtype=np.float64)
        fx = np.asarray(f(x), dtype=np.float64)
        gx = fx - x
        if jvp is not None:
            mv = lambda v: np.asarray(jvp(x, v)) - v
        else:
            xn = float(np.linalg.norm(x.ravel()))

            def mv(v):
                vn = float(np.linalg.norm(v.ravel()))
                if vn == 0.0:
                    return np.zeros_like(v)
                eps = np.sqrt(np.finfo(np.float64).eps) * (1.0 + xn) / vn
                return (np.asarray(f(x + eps * v), dtype=np.float64) - fx) / eps - v
        return x - _host_bicgstab(mv, gx, atol=bicgstab_atol)

    return successive_approx(q, x_init, tol, max_iter, verbose, print_skip, _host_only=True)


def anderson_solver(f, x_init, tol=default_tolerance, max_iter=10000, verbose=True,
                    **device_opts):
    """Anderson acceleration with the reference's hard-coded jaxopt parameters
    (history 10, mixing frequency 4, beta 8, ridge 1e-6)."""
    m, mix, beta, ridge = 10, 4, 8.0, 1e-6
    done = _on_device(f, x_init, "anderson", verbose, None, max_iter, tol=tol,
                      history=m, mixing_freq=mix, beta=beta, ridge=ridge, **device_opts)
    if done is not None:
        return done

    x_init = np.asarray(x_init, dtype=np.float64)
    shape = x_init.shape
    x = x_init.ravel().copy()
    X = np.zeros((m, x.size)); R = np.zeros((m, x.size))
    it, error = 0, np.inf
    while error > tol and it < max_iter:
        fx = np.asarray(f(x.reshape(shape)), dtype=np.float64).ravel()
        r = fx - x
        X[it % m] = x; R[it % m] = r
        error = float(np.sqrt(r @ r))
        if it + 1 >= m and (it + 1) % mix == 0:
            Hm = np.zeros((m + 1, m + 1))
            Hm[0, 1:] = 1.0; Hm[1:, 0] = 1.0
            Hm[1:, 1:] = R @ R.T + ridge * np.eye(m)
            rhs = np.zeros(m + 1); rhs[0] = 1.0
            a = np.linalg.solve(Hm, rhs)[1:]
            x = a @ X + beta * (a @ R)
        else:
            x = fx
        it += 1
    _report(verbose, it, max_iter)
    return x.reshape(shape), it


class GDState(dict):
    """State returned by "gd": ``state.iter_num`` / ``state.error`` / ``state.stepsize`` / ``state.t`` as jaxopt's
    ProxGradState names them (attribute access, as the reference's callers would use it), also a dict (``state["errors"]``
    holds the error after every iteration)."""
    def __getattr__(self, name):
        # (AttributeError, not KeyError: hasattr, copy and pickle probe optional attributes and only catch the former)
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


def _gd_loop(value_and_grad, value, axpy, dot, x, maxiter, tol, maxls, decrease_factor):
    """FISTA + backtracking line search (see fixed_point_via_gradient_decent) on abstract vectors:
    value_and_grad(y) -> (f, g);  value(x) -> f;  axpy(a, u, v) -> a*u + v;  dot(u, v) -> float.
    Returns (x, GDState)."""
    eps = float(np.finfo(np.float64).eps)
    y, t, s, it, errs = x, 1.0, 1.0, 0, []
    err = float("inf")
    fy, g = None, None
    while err > tol and it < maxiter:
        if g is None:
            fy, g = value_and_grad(y)
        # line search from the carried step: halve while  s (f(x+) - f(y)) > s <x+ - y, g> + |x+ - y|^2 / 2 + eps
        xn = axpy(-s, g, y)
        for _ in range(maxls):
            d = axpy(-1.0, y, xn)
            if not (s * (value(xn) - fy) > s * dot(d, g) + 0.5 * dot(d, d) + eps):
                break
            s *= decrease_factor
            xn = axpy(-s, g, y)
        # a step that has become tiny restarts at 1, otherwise the next search tries a larger one
        s = 1.0 if s <= 1e-6 else s / decrease_factor
        tn = 0.5 * (1.0 + (1.0 + 4.0 * t * t) ** 0.5)
        y = axpy((t - 1.0) / tn, axpy(-1.0, x, xn), xn)
        x, t = xn, tn
        # the error is the gradient mapping at the NEW y with the NEW step; with the identity prox that is |grad(y)|
        fy, g = value_and_grad(y)
        err = dot(g, g) ** 0.5
        errs.append(err)
        it += 1
    return x, GDState(iter_num=it, error=err, stepsize=s, t=t, errors=np.array(errs))


def fixed_point_via_gradient_decent(f, x_init, maxiter=1000, tol=1e-4, maxls=15, decrease_factor=0.5):
    """Registry entry "gd" (code/solvers.py:127-140): ``jaxopt.GradientDescent(fun=loss, maxiter=1000,
    tol=0.0001, stepsize=0.0).run(x_init)`` with ``loss(x) = |f(x) - x|^2``; returns ``(solution, state)``.

    jaxopt is not vendored by the reference (unpinned): restated from ProximalGradient's update -- identity prox,
    FISTA acceleration (jaxopt's default) and, since ``stepsize <= 0``, a backtracking line search per iteration:
    the state's step starts at 1, is halved (at most ``maxls`` times) while the sufficient-decrease test
    ``s (f(x+) - f(y)) > s <x+ - y, g> + |x+ - y|^2 / 2 + eps`` fails, and the next search starts from twice the
    accepted step -- or from 1 again once a step has fallen to 1e-6.  The error that stops the loop is the gradient
    mapping at the NEW extrapolated point with the new step, which for the identity prox is ``|grad loss(y)|``
    (one gradient per iteration: it is the next iteration's).  **Parity is with `oracle/solvers.py`'s restatement of
    the same rules, not with jaxopt itself.**  The gradient
    ``2 (dT(x)^T r - r)`` needs the vector-Jacobian product: for (a closure over) a device operator the loop
    runs on the GPU (T, its linearisation and ``sdfs_apply_vjp_dev`` on device-resident vectors); a foreign
    callable must bring ``f.vjp(x, u)`` (the reference gets it from jax.grad).  ``state``: `GDState`
    (``state.iter_num``, ``state.error``, ``state.stepsize``, ``state.t``; ``state["errors"]`` the trace)."""
    op = _resolve_operator(f, x_init)
    if op is not None:
        import torch
        shape = op.shapes
        dev = torch.device("cuda", op.device)
        x0 = torch.from_numpy(np.ascontiguousarray(np.asarray(x_init, dtype=np.float64))).to(dev)
        op.set_stream(torch.cuda.current_stream(dev).cuda_stream)

        def residual(x):
            out = torch.empty_like(x)
            op.apply_dev(x.data_ptr(), out.data_ptr())
            return out.sub_(x)

        def value(x):
            r = residual(x)
            return float(torch.dot(r.view(-1), r.view(-1)))

        def value_and_grad(y):
            tw = torch.empty_like(y)
            op.linearize_dev(y.data_ptr(), tw.data_ptr())
            r = tw.sub_(y)
            g = torch.empty_like(y)
            op.vjp_dev(r.data_ptr(), g.data_ptr(), minus_identity=True)       # dT^T r - r
            return float(torch.dot(r.view(-1), r.view(-1))), g.mul_(2.0)

        try:
            x, state = _gd_loop(value_and_grad, value, lambda a, u, v: torch.add(v, u, alpha=a),
                                lambda u, v: float(torch.dot(u.view(-1), v.view(-1))), x0,
                                maxiter, tol, maxls, decrease_factor)
            x = x.cpu().numpy().reshape(shape)
        finally:
            op.set_stream(None, use_own=True)
        if _confirm(f, op, x):
            return x, state
        print("Warning: the callable is not a pure closure over one device operator; repeating on the host")
    vjp = getattr(f, "vjp", None)
    if vjp is None:
        raise TypeError("'gd' differentiates |f(x) - x|^2 (the reference uses jax.grad): it needs a device operator, "
                        "a closure over one, or a callable with a .vjp(x, u) method")
    x0 = np.asarray(x_init, dtype=np.float64)

    def value_and_grad(y):
        r = np.asarray(f(y), dtype=np.float64) - y
        return float(np.vdot(r, r)), 2.0 * (np.asarray(vjp(y, r), dtype=np.float64) - r)

    def value(x):
        r = np.asarray(f(x), dtype=np.float64) - x
        return float(np.vdot(r, r))

    return _gd_loop(value_and_grad, value, lambda a, u, v: a * u + v, lambda u, v: float(np.vdot(u, v)),
                    x0, maxiter, tol, maxls, decrease_factor)


# == List solvers for simple access == #
solvers = dict((("newton", newton_solver),
                ("anderson", anderson_solver),
                ("gd", fixed_point_via_gradient_decent),
                ("successive_approx", successive_approx)))


def solver(f, x_init, algorithm="newton", verbose=True):
    """A simple front end to the other solvers (code/solvers.py:154-177)."""
    try:
        fn = solvers[algorithm]
    except KeyError:
        msg = f"""\
                  Algorithm {algorithm} not found.
                  Falling back to successive approximation.
               """
        print(dedent(msg))
        fn = successive_approx
    x_star, num_iter = fn(f, x_init)
    return x_star
