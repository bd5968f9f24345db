"""
Oracle (test infrastructure): fixed-point solvers, restating code/solvers.py.

  successive_approx  solvers.py:19-48   x <- f(x) until max|x_new - x| <= tol
  newton_solver      solvers.py:51-95   Newton on g = f - id with a matrix-free
                                        J.v and BiCGSTAB (atol=1e-4, rtol 1e-5, x0 = 0),
                                        outer loop = successive_approx on q
  anderson_solver    solvers.py:98-124  jaxopt.AndersonAcceleration(m=10,
                                        mixing_frequency=4, beta=8, ridge=1e-6)
  solver             solvers.py:154-177 name -> function dispatch

Third-party pieces the reference calls but does not vendor (versions unpinned):
  * jax.jvp: replaced by an explicit ``jvp(x, v)`` callable (analytic derivative
    of T; oracle/ssy.py, oracle/gcy.py) -- same linear map up to rounding.
  * jax.scipy.sparse.linalg.bicgstab: restated below from its published
    algorithm (van der Vorst 1992 with JAX's stopping rule
    |r|^2 <= max(tol^2 |b|^2, atol^2), x0 = 0, maxiter = 10*size).  Pinned only
    loosely by the recorded Newton trace (sandpit.ipynb:41-44).
  * jaxopt.AndersonAcceleration: restated from its documented semantics.
    ITERATE-LEVEL PARITY UNPINNED (no recorded output in the reference); only
    the fixed point it converges to is a parity target.
"""
from textwrap import dedent

import numpy as np

default_tolerance = 1e-7
default_max_iter = int(1e6)


def successive_approx(f, x_init, tol=default_tolerance, max_iter=default_max_iter,
                      verbose=True, print_skip=1000, errors=None):
    if verbose:
        print("Beginning iteration\n\n")
    it = 0
    x = x_init
    error = tol + 1
    while error > tol and it < max_iter:
        x_new = f(x)
        error = np.max(np.abs(x_new - x))
        if errors is not None:
            errors.append(float(error))
        if verbose and it % print_skip == 0:
            print("iter = {}, error = {}".format(it, error))
        it += 1
        x = x_new
    if it == max_iter:
        print(f"Warning: Hit maximum iteration number {max_iter}")
    elif verbose:
        print(f"Iteration converged after {it} iterations")
    return x, it


def bicgstab(A, b, tol=1e-5, atol=0.0, maxiter=None, stats=None):
    """Unpreconditioned BiCGSTAB with x0 = 0 and JAX's stopping rule."""
    shape = b.shape
    b = b.ravel()
    mv = lambda u: A(u.reshape(shape)).ravel()
    if maxiter is None:
        maxiter = 10 * b.size
    atol2 = max(tol * tol * float(b @ b), atol * atol)
    x = np.zeros_like(b)
    r = b.copy()                      # b - A(0)
    rhat = r.copy()
    alpha = omega = rho = 1.0
    p = r.copy()
    q = r.copy()
    k = 0
    nmv = 0
    while float(r @ r) > atol2 and 0 <= k < maxiter:
        rho_new = float(rhat @ r)
        beta = rho_new / rho * alpha / omega
        p = r + beta * (p - omega * q)
        q = mv(p)
        nmv += 1
        alpha = rho_new / float(rhat @ q)
        s = r - alpha * q
        if float(s @ s) < atol2:
            x = x + alpha * p
            r = s
            omega_new = omega
            t = None
        else:
            t = mv(s)
            nmv += 1
            omega_new = float(t @ s) / float(t @ t)
            x = x + alpha * p + omega_new * s
            r = s - omega_new * t
        if rho_new == 0:
            k = -10
        elif omega_new == 0 or alpha == 0:
            k = -11
        else:
            k += 1
        omega, rho = omega_new, rho_new
    if stats is not None:
        stats["matvecs"] = stats.get("matvecs", 0) + nmv
        stats["iters"] = stats.get("iters", 0) + max(k, 0)
    return x.reshape(shape)


def newton_solver(f, x_init, tol=default_tolerance, max_iter=default_max_iter,
                  bicgstab_atol=1e-4, verbose=True, print_skip=1, jvp=None,
                  bicgstab_tol=1e-5, errors=None, stats=None):
    if jvp is None:
        jvp = getattr(f, "jvp", None)
    if jvp is None:
        raise ValueError("newton_solver needs jvp(x, v) (the reference uses jax.jvp)")

    def q(x):
        gx = f(x) - x
        step = bicgstab(lambda v: jvp(x, v) - v, gx, tol=bicgstab_tol,
                        atol=bicgstab_atol, stats=stats)
        return x - step

    return successive_approx(q, x_init, tol, max_iter, verbose, print_skip, errors=errors)


def anderson_solver(f, x_init, tol=default_tolerance, max_iter=10000, verbose=True,
                    history_size=10, mixing_frequency=4, beta=8.0, ridge=1e-6):
    """Anderson acceleration, type-II with ridge (jaxopt semantics; UNPINNED)."""
    m = history_size
    shape = x_init.shape
    x = np.asarray(x_init, dtype=np.float64).ravel().copy()
    X = np.zeros((m, x.size))
    R = np.zeros((m, x.size))
    it = 0
    error = np.inf
    while error > tol and it < max_iter:
        fx = np.asarray(f(x.reshape(shape))).ravel()
        r = fx - x
        pos = it % m
        X[pos] = x
        R[pos] = r
        error = float(np.sqrt(r @ r))
        if it + 1 >= m and (it + 1) % mixing_frequency == 0:
            G = R @ R.T
            # ridge < 0: the HIP library's opt-in relative ridge |ridge| trace(G) / m (include/sdfs_hip.h; not jaxopt's)
            G = G + (ridge if ridge >= 0 else -ridge * np.trace(G) / m) * np.eye(m)
            Hm = np.zeros((m + 1, m + 1))
            Hm[0, 1:] = 1.0
            Hm[1:, 0] = 1.0
            Hm[1:, 1:] = G
            rhs = np.zeros(m + 1)
            rhs[0] = 1.0
            alphas = np.linalg.solve(Hm, rhs)[1:]
            x = alphas @ X + beta * (alphas @ R)
        else:
            x = fx
        it += 1
    if it == max_iter:
        print(f"Warning: Hit maximum iteration number {max_iter}")
    elif verbose:
        print(f"Iteration converged after {it} iterations")
    return x.reshape(shape), it


def fixed_point_via_gradient_decent(f, x_init, vjp, maxiter=1000, tol=1e-4, maxls=15, decrease_factor=0.5):
    """code/solvers.py:127-140: jaxopt.GradientDescent(fun=|f(x) - x|^2, maxiter=1000, tol=1e-4, stepsize=0.0).
    jaxopt is not vendored (UNPINNED, restated from ProximalGradient's update): identity prox, FISTA acceleration
    (its default) and, because stepsize <= 0, a backtracking line search per iteration -- the state's step starts
    at 1, is halved (at most `maxls` times) while  s (f(x+) - f(y)) > s <x+ - y, g> + |x+ - y|^2 / 2 + eps,  and the
    next search starts from the accepted step / decrease_factor, or from 1 once a step has fallen to 1e-6; the loop
    runs while error > tol, the error being the gradient mapping at the new extrapolated point with the new step
    (identity prox: |grad loss(y)|).  `vjp(x, u)` = dT(x)^T u.
    Returns (x, iterations), the errors in ``fixed_point_via_gradient_decent.last_errors``."""
    shape = np.asarray(x_init).shape
    x = np.asarray(x_init, dtype=np.float64).copy()
    eps = float(np.finfo(np.float64).eps)

    def loss_and_grad(y):
        r = np.asarray(f(y), dtype=np.float64) - y
        return float(np.vdot(r, r)), 2.0 * (np.asarray(vjp(y, r), dtype=np.float64) - r)

    def loss(z):
        rz = np.asarray(f(z), dtype=np.float64) - z
        return float(np.vdot(rz, rz))

    y, t, s = x.copy(), 1.0, 1.0
    errs = []
    it = 0
    err = np.inf
    fy, g = loss_and_grad(y)
    while err > tol and it < maxiter:
        xn = y - s * g
        for _ in range(maxls):
            d = xn - y
            if not (s * (loss(xn) - fy) > s * float(np.vdot(d, g)) + 0.5 * float(np.vdot(d, d)) + eps):
                break
            s *= decrease_factor
            xn = y - s * g
        s = 1.0 if s <= 1e-6 else s / decrease_factor
        tn = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t * t))
        y = xn + (t - 1.0) / tn * (xn - x)
        x, t = xn, tn
        fy, g = loss_and_grad(y)
        err = float(np.sqrt(np.vdot(g, g)))
        errs.append(err)
        it += 1
    fixed_point_via_gradient_decent.last_errors = errs
    return x.reshape(shape), it


solvers = {"newton": newton_solver,
           "anderson": anderson_solver,
           "successive_approx": successive_approx}


def solver(f, x_init, algorithm="newton", verbose=True):
    try:
        fn = solvers[algorithm]
    except KeyError:
        print(dedent(f"""\
            Algorithm {algorithm} not found.
            Falling back to successive approximation.
            """))
        fn = successive_approx
    x_star, _ = fn(f, x_init)
    return x_star


def newton_polish(f, jvp, x, iters=6, inner_tol=1e-12):
    """Tight Newton iterations (relative inner tolerance) -> residual ~1e-12 fixed point."""
    for _ in range(iters):
        gx = f(x) - x
        if np.max(np.abs(gx)) < 1e-12:
            break
        x = x - bicgstab(lambda v: jvp(x, v) - v, gx, tol=inner_tol, atol=0.0)
    return x
