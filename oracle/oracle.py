"""ctypes front-end of the CPU oracle (oracle/pnnls_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by anything under pympc_amd/.

The functions return the same ``sol`` dict as the reference backend
(/root/reference/pympc/optimization/solvers/pnnls.py:128-134): every value
``None`` when the QP is infeasible / the LP infeasible or unbounded.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

OK, INFEASIBLE, MAXITER, NOT_PD = 0, 1, 3, -2

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    src = os.path.join(_HERE, "pnnls_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
        _lib.oracle_mpqp_solve_batch.argtypes = [_dp, _dp, _dp, _dp, _dp, ctypes.c_double, _dp, _dp, _dp, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, _dp, ctypes.c_long, ctypes.c_double, _dp,
                                                 _dp, _dp, _ip, _ip, _ip]
        _lib.oracle_point_location.argtypes = [_dp, _dp, _ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp,
                                               _dp, _dp, _dp, ctypes.c_long, ctypes.c_double, _ip, _dp, _dp]
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def nnls(A, b):
    """Lawson-Hanson NNLS; returns (x, rnorm, (outer, inner)); raises RuntimeError at 3*cols iterations like scipy."""
    A = _d(A); b = _d(b)
    rows, cols = A.shape
    x = np.zeros(cols); rn = ctypes.c_double(0.0); it = np.zeros(2, np.int32)
    st = lib().oracle_nnls(_p(A), ctypes.c_int(rows), ctypes.c_int(cols), _p(b), _p(x), ctypes.byref(rn), _pi(it))
    if st == MAXITER:
        raise RuntimeError("Maximum number of iterations reached.")
    return x, rn.value, (int(it[0]), int(it[1]))


def pnnls(A, B, c):
    """pnnls.py:5-39: (v, u, r)."""
    A = _d(A); B = _d(B); c = _d(c)
    rows, nv = A.shape
    nu = B.shape[1]
    v = np.zeros(nv); u = np.zeros(max(nu, 1)); rn = ctypes.c_double(0.0); it = np.zeros(2, np.int32)
    st = lib().oracle_pnnls(_p(A), _p(B), _p(c), ctypes.c_int(rows), ctypes.c_int(nv), ctypes.c_int(nu), _p(v), _p(u),
                            ctypes.byref(rn), _pi(it))
    if st == MAXITER:
        raise RuntimeError("Maximum number of iterations reached.")
    return v, u[:nu], rn.value


def _empty_sol():
    return {'min': None, 'argmin': None, 'active_set': None, 'multiplier_inequality': None, 'multiplier_equality': None}


def _finish(st, n_eq, xmin, arg, mi, me, act):
    sol = _empty_sol()
    if st == MAXITER:
        raise RuntimeError("Maximum number of iterations reached.")
    if st == NOT_PD:
        raise np.linalg.LinAlgError("Matrix is not positive definite")
    if st == OK:
        sol['min'] = float(xmin.value)
        sol['argmin'] = arg
        sol['multiplier_inequality'] = mi
        sol['active_set'] = [int(i) for i in np.where(act)[0]]
        if n_eq > 0:
            sol['multiplier_equality'] = me
    return sol


def quadratic_program(H, f, A, b, C=None, d=None, tol=1.e-7, info=None):
    if (C is None) != (d is None):
        raise ValueError('missing C or d.')
    H = _d(H); f = _d(f); A = _d(A); b = _d(b)
    n_ineq, n = A.shape
    n_eq = 0 if C is None else np.asarray(C).shape[0]
    Cc = _d(C) if n_eq else None
    dc = _d(d) if n_eq else None
    xmin = ctypes.c_double(0.0); rn = ctypes.c_double(0.0)
    arg = np.zeros(n); mi = np.zeros(n_ineq); me = np.zeros(max(n_eq, 1)); act = np.zeros(max(n_ineq, 1), np.int32)
    it = np.zeros(2, np.int32)
    st = lib().oracle_qp(_p(H), _p(f), _p(A), _p(b), _p(Cc), _p(dc), ctypes.c_int(n), ctypes.c_int(n_ineq),
                         ctypes.c_int(n_eq), ctypes.c_double(tol), ctypes.byref(xmin), _p(arg), _p(mi), _p(me), _pi(act),
                         ctypes.byref(rn), _pi(it))
    if info is not None:
        info.update(rnorm=rn.value, iters=(int(it[0]), int(it[1])), status=st)
    return _finish(st, n_eq, xmin, arg, mi, me[:n_eq], act[:n_ineq])


def linear_program(f, A, b, C=None, d=None, tol=1.e-7, info=None):
    if (C is None) != (d is None):
        raise ValueError('missing C or d.')
    f = _d(f); A = _d(A); b = _d(b)
    n_ineq, n = A.shape
    n_eq = 0 if C is None else np.asarray(C).shape[0]
    Cc = _d(C) if n_eq else None
    dc = _d(d) if n_eq else None
    xmin = ctypes.c_double(0.0); rn = ctypes.c_double(0.0)
    arg = np.zeros(n); mi = np.zeros(n_ineq); me = np.zeros(max(n_eq, 1)); act = np.zeros(max(n_ineq, 1), np.int32)
    it = np.zeros(2, np.int32)
    st = lib().oracle_lp(_p(f), _p(A), _p(b), _p(Cc), _p(dc), ctypes.c_int(n), ctypes.c_int(n_ineq), ctypes.c_int(n_eq),
                         ctypes.c_double(tol), ctypes.byref(xmin), _p(arg), _p(mi), _p(me), _pi(act), ctypes.byref(rn),
                         _pi(it))
    if info is not None:
        info.update(rnorm=rn.value, iters=(int(it[0]), int(it[1])), status=st)
    return _finish(st, n_eq, xmin, arg, mi, me[:n_eq], act[:n_ineq])


def mpqp_solve_batch(model, X, tol=1.e-7, want_mult=True):
    """Loop of MultiParametricQuadraticProgram.solve over the rows of X (reference-as-is: refactorises per state).

    model: mapping with Huu, Hux, Hxx, fu, fx, g, Au, Ax, b.  Returns dict of arrays:
    status (0 ok / 1 infeasible / 3 maxiter), argmin, min, multiplier_inequality, active (bool), iters.
    """
    Huu = _d(model['Huu']); Hux = _d(model['Hux']); Hxx = _d(model['Hxx']); fu = _d(model['fu']); fx = _d(model['fx'])
    Au = _d(model['Au']); Ax = _d(model['Ax']); b0 = _d(model['b']); g = float(model['g'])
    X = _d(X)
    B, p = X.shape
    m, n = Au.shape
    arg = np.zeros((B, n)); vmin = np.zeros(B); status = np.zeros(B, np.int32); iters = np.zeros((B, 2), np.int32)
    mult = np.zeros((B, m)) if want_mult else None
    act = np.zeros((B, m), np.int32) if want_mult else None
    lib().oracle_mpqp_solve_batch(_p(Huu), _p(Hux), _p(Hxx), _p(fu), _p(fx), g, _p(Au), _p(Ax), _p(b0), n, m, p, _p(X), B,
                                  tol, _p(arg), _p(vmin), _p(mult), _pi(act), _pi(status), _pi(iters))
    return dict(status=status, argmin=arg, min=vmin, multiplier_inequality=mult,
                active=None if act is None else act.astype(bool), iters=iters)


def point_location(table, X, tol=1.e-7):
    """First critical region containing each row of X, with u and V evaluated there (idx -1 / NaN when uncovered).

    table: mapping with A [R x p], b [R], row_ptr [nreg+1], ux [nreg x nu x p], u0 [nreg x nu],
    Vxx [nreg x p x p], Vx [nreg x p], V0 [nreg].
    """
    A = _d(table['A']); b = _d(table['b']); rp = np.ascontiguousarray(table['row_ptr'], np.int32)
    ux = _d(table['ux']); u0 = _d(table['u0']); Vxx = _d(table['Vxx']); Vx = _d(table['Vx']); V0 = _d(table['V0'])
    X = _d(X)
    B, p = X.shape
    nreg = rp.size - 1
    nu = u0.shape[1]
    idx = np.zeros(B, np.int32); u = np.zeros((B, nu)); V = np.zeros(B)
    lib().oracle_point_location(_p(A), _p(b), _pi(rp), nreg, p, nu, _p(ux), _p(u0), _p(Vxx), _p(Vx), _p(V0), _p(X), B, tol,
                                _pi(idx), _p(u), _p(V))
    return idx, u, V


def condense(systems, domains, Q, R, P, XN_A, XN_b, mode_sequence):
    """condense_optimal_control_problem for one mode sequence (controllers.py:682-749).

    systems: list of (A, B, c); domains: list of (D_A [r x (nx+nu)], D_b [r]).  Returns a dict with Abar, Bbar, cbar,
    Huu, Hux, Hxx, fu, fx, g, Au, Ax, b (compact rows, exactly the reference's ordering).
    """
    As = _d(np.array([s[0] for s in systems])); Bs = _d(np.array([s[1] for s in systems]))
    cs = _d(np.array([s[2] for s in systems]))
    nm, nx, nu = Bs.shape
    z = np.ascontiguousarray(mode_sequence, np.int32)
    N = z.size
    dptr = np.zeros(nm + 1, np.int32)
    dptr[1:] = np.cumsum([d[0].shape[0] for d in domains])
    DA = _d(np.vstack([d[0] for d in domains])); Db = _d(np.concatenate([d[1] for d in domains]))
    XA = _d(XN_A); Xb = _d(XN_b)
    rN = XA.shape[0]
    n = N * nu; nr = (N + 1) * nx
    m = rN + int(sum(dptr[k + 1] - dptr[k] for k in z))
    o = dict(Abar=np.zeros((nr, nx)), Bbar=np.zeros((nr, n)), cbar=np.zeros(nr), Huu=np.zeros((n, n)),
             Hux=np.zeros((n, nx)), Hxx=np.zeros((nx, nx)), fu=np.zeros(n), fx=np.zeros(nx), g=np.zeros(1),
             Au=np.zeros((m, n)), Ax=np.zeros((m, nx)), b=np.zeros(m))
    f = lib().oracle_condense
    f.argtypes = [ctypes.c_int] * 4 + [_ip] + [_dp] * 6 + [_ip, _dp, _dp, ctypes.c_int] + [_dp] * 14
    rc = f(nx, nu, N, nm, _pi(z), _p(As), _p(Bs), _p(cs), _p(_d(Q)), _p(_d(R)), _p(_d(P)), _pi(dptr), _p(DA), _p(Db), rN,
           _p(XA), _p(Xb), *[_p(o[k]) for k in ('Abar', 'Bbar', 'cbar', 'Huu', 'Hux', 'Hxx', 'fu', 'fx', 'g', 'Au', 'Ax', 'b')])
    if rc != m:
        raise ValueError('bad mode sequence')
    o['g'] = float(o['g'][0])
    return o
