"""LP side of the `hip` backend: same signature and `sol` dict as pnnls.linear_program, computed by the HIP LP kernel.

Mirrors /root/reference/pympc/optimization/solvers/pnnls.py:41-147:
    linear_program(f, A, b, C=None, d=None, tol=1e-7)
plus the batched form the reference does not have (its callers loop in Python, e.g. polyhedron.py:435-448):
    linear_program_batch(f, A, b, relax_idx=None, relax_val=0., ...)   B LPs of equal shape in one launch

and the two numerical cores as callables of the backend (module contract of pnnls.py: three public functions):
    pnnls(A, B, c)               pnnls.py:5-39    -> (v, u, r)
    nnls(A, b)                   scipy.optimize.nnls as called at pnnls.py:36,218 -> (x, rnorm)
with batched forms pnnls_batch / nnls_batch.

Result contract (pnnls.py:128-145): dict with keys min, argmin, active_set, multiplier_inequality,
multiplier_equality; every value None when the LP is infeasible or unbounded.
"""
import numpy as np
import torch

from ... import _hip
from .hip import BatchResult, TOL


def linear_program_batch(f, A, b, tol=TOL, n_eq=0, relax_idx=None, relax_val=0., want_argmin=True, want_mult=True,
                         want_active=True, want_iters=False):
    """B LPs  min f_k'x s.t. A_k x <= b_k  in one launch (one workgroup per LP).

    f: [B x n];  A: [m x n] (shared) or [B x m x n];  b: [m] (shared) or [B x m].
    relax_idx [B] (int): b_k[relax_idx[k]] += relax_val -- the relaxation of polyhedron.py:439-441 without
    materialising B right-hand sides.  Equalities must already be folded as rows (C, -C) (pnnls.py:101-106); n_eq
    tells BatchResult.sol how to unfold the multipliers.  Returns a hip.BatchResult (device tensors).
    """
    lib = _hip.load_library()
    dev = _hip.device()
    f = _hip.to_device(f); A = _hip.to_device(A); b = _hip.to_device(b)
    if f.dim() != 2:
        raise ValueError('f must be [B x n]')
    B, n = f.shape
    m = A.shape[-2]
    if A.shape[-1] != n or b.shape[-1] != m:
        raise ValueError('incoherent size of the inputs.')
    a_stride = 0 if A.dim() == 2 else m * n
    b_stride = 0 if b.dim() == 1 else m
    if (A.dim() == 3 and A.shape[0] != B) or (b.dim() == 2 and b.shape[0] != B):
        raise ValueError('incoherent batch size of the inputs.')
    nd = lib.pympc_lp_workspace_doubles(n, m, B, 1 if a_stride == 0 else 0)
    if nd == 0:
        raise NotImplementedError('LP with n=%d, m=%d is outside the LP kernel limits (n<=%d, m<=%d)'
                                  % ((n, m) + _hip.lp_limits()))
    ws = _hip.scratch(nd)
    rx = None if relax_idx is None else _hip.to_device(relax_idx, dtype=torch.int32)
    f64 = dict(dtype=torch.float64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    out = dict(status=torch.empty(B, **i32), min=torch.empty(B, **f64))
    out['argmin'] = torch.empty((B, n), **f64) if want_argmin else None
    out['multiplier_inequality'] = torch.empty((B, m), **f64) if want_mult else None
    out['active_bits'] = torch.empty((B, (m + 31) // 32), **i32) if want_active else None
    out['iters'] = torch.empty((B, 3), **i32) if want_iters else None
    _hip.check(lib.pympc_lp_solve_batch(n, m, B, _hip.ptr(f), _hip.ptr(A), a_stride, _hip.ptr(b), b_stride, _hip.ptr(rx),
                                        float(relax_val), float(tol), _hip.ptr(ws), _hip.ptr(out['argmin']),
                                        _hip.ptr(out['min']), _hip.ptr(out['multiplier_inequality']),
                                        _hip.ptr(out['active_bits']), _hip.ptr(out['status']), _hip.ptr(out['iters']),
                                        _hip.stream_ptr()))
    res = BatchResult(m, n_eq=n_eq, n_ineq=m - 2 * n_eq, **out)
    res._keepalive = (f, A, b, rx, ws)
    return res


def linear_program(f, A, b, C=None, d=None, tol=TOL):
    """Drop-in for pnnls.linear_program (pnnls.py:41-147): one LP on the GPU."""
    if (C is None) != (d is None):
        raise ValueError('missing C or d.')
    f = np.asarray(f, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    n_eq = 0 if C is None else np.asarray(C).shape[0]
    if n_eq > 0:   # state equalities as inequalities (pnnls.py:101-106)
        C = np.asarray(C, dtype=np.float64); d = np.asarray(d, dtype=np.float64)
        AC = np.vstack((A, C, -C))
        bd = np.concatenate((b, d, -d))
    else:
        AC, bd = A, b
    n_lp, m_lp = _hip.lp_limits()
    if f.size > n_lp and f.size <= 256 and 1 + AC.shape[0] + f.size <= 1024:
        # more variables than the active-set kernel's factor holds (a lane per row of it): the reference's own KKT form
        # on the workgroup-per-problem PNNLS kernels
        return linear_program_pnnls(f, A, b, C, d, tol=tol)
    res = linear_program_batch(f[None, :], AC, bd, tol=tol, n_eq=n_eq)
    if int(res.t['status'][0].item()) == _hip.MAXITER:
        # The active-set kernel can stall on a dual-degenerate LP (an objective that does not see most variables: the
        # Chebyshev LP, max x_i over a polytope in (x, u)): thousands of pivots through ties, or its iteration limit.
        # One more try with the objective tilted by 1e-9 (fixed seed), which breaks the ties; the value is reported for
        # the objective asked for.  An LP that fails again raises as before.
        ft = f + 1.e-9 * max(1., float(np.abs(f).max())) * np.random.RandomState(0).randn(f.size)
        again = linear_program_batch(ft[None, :], AC, bd, tol=tol, n_eq=n_eq)
        if int(again.t['status'][0].item()) == _hip.OPTIMAL:
            sol = again.sol(0)
            sol['min'] = float(f.dot(sol['argmin']))
            return sol
    return res.sol(0)


def retry_stalled(res, f, A, b, relax_idx=None, relax_val=0., tol=TOL):
    """For callers that read a batch's statuses anyway: the LPs of `res` that ended with the iteration-limit status (a
    dual-degenerate LP can stall the active-set kernel, see linear_program) are solved once more with their objectives
    tilted by 1e-9, and `status` / `min` of `res` (host copies) are patched with what comes out -- `min` evaluated with
    the objective asked for.  f, A, b, relax_idx as given to linear_program_batch.  Returns the number of LPs retried."""
    st = res.numpy('status')
    idx = np.where(st == _hip.MAXITER)[0]
    if idx.size == 0:
        return 0
    ft = _hip.to_device(f)[torch.as_tensor(idx, device=_hip.device())]
    scale = torch.clamp(ft.abs().max(dim=1, keepdim=True).values, min=1.)
    g = torch.Generator(device='cpu').manual_seed(0)
    tilt = torch.randn(ft.shape, generator=g, dtype=torch.float64).to(ft.device)
    A = _hip.to_device(A); b = _hip.to_device(b)
    ti = torch.as_tensor(idx, device=_hip.device())
    Ai = A if A.dim() == 2 else A[ti]
    bi = b if b.dim() == 1 else b[ti]
    rx = None if relax_idx is None else _hip.to_device(relax_idx, dtype=torch.int32)[ti]
    again = linear_program_batch(ft + 1.e-9 * scale * tilt, Ai, bi, tol=tol, relax_idx=rx, relax_val=relax_val,
                                 want_mult=False, want_active=False)
    st2 = again.numpy('status')
    val = (ft * again.t['argmin']).sum(dim=1).cpu().numpy()
    mn = res.numpy('min')
    st[idx] = st2
    mn[idx] = np.where(st2 == _hip.OPTIMAL, val, np.nan)
    return int(idx.size)


def nnls_batch(A, b):
    """B NNLS problems min |A_k x - b_k|, x >= 0 in one launch.  A: [rows x cols] (shared) or [B x rows x cols];
    b: [B x rows].  Returns device tensors (x [B x cols], rnorm [B], status [B])."""
    lib = _hip.load_library()
    dev = _hip.device()
    A = _hip.to_device(A); b = _hip.to_device(b)
    if b.dim() != 2 or A.dim() not in (2, 3) or A.shape[-2] != b.shape[1] or (A.dim() == 3 and A.shape[0] != b.shape[0]):
        raise ValueError('incoherent size of the inputs.')
    B, rows = b.shape
    cols = A.shape[-1]
    nd = lib.pympc_nnls_workspace_doubles(rows, cols, B)
    if nd == 0:
        raise NotImplementedError('NNLS with %d rows, %d columns is outside the kernel limits (1024 rows, 32768 columns)'
                                  % (rows, cols))
    ws = _hip.scratch(nd)
    x = torch.empty((B, cols), dtype=torch.float64, device=dev)
    rn = torch.empty(B, dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    _hip.check(lib.pympc_nnls_batch(rows, cols, B, _hip.ptr(A), 0 if A.dim() == 2 else rows * cols, _hip.ptr(b), _hip.ptr(ws),
                                    _hip.ptr(x), _hip.ptr(rn), _hip.ptr(st), None, _hip.stream_ptr()))
    return x, rn, st


def nnls(A, b):
    """scipy.optimize.nnls(A, b) on the GPU: (x, rnorm); RuntimeError at the iteration limit like scipy."""
    A = np.asarray(A, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    x, rn, st = nnls_batch(A, b[None, :])
    if int(st.item()) == _hip.MAXITER:
        raise RuntimeError('Maximum number of iterations reached.')
    return x[0].cpu().numpy(), float(rn.item())


def pnnls_batch(A, B, c):
    """B PNNLS problems min |A_k v + B_k u - c_k|, v >= 0 (pnnls.py:5-39) in one launch.  A [Bn x rows x cols],
    B [Bn x rows x nu], c [Bn x rows].  Returns device tensors (v, u, r, status)."""
    lib = _hip.load_library()
    dev = _hip.device()
    A = _hip.to_device(A); Bm = _hip.to_device(B); c = _hip.to_device(c)
    if A.dim() != 3 or Bm.dim() != 3 or c.dim() != 2 or A.shape[:2] != Bm.shape[:2] or A.shape[:2] != c.shape:
        raise ValueError('incoherent size of the inputs.')
    Bn, rows, cols = A.shape
    nu = Bm.shape[2]
    nd = lib.pympc_pnnls_workspace_doubles(rows, cols, nu, Bn)
    if nd == 0:
        raise NotImplementedError('PNNLS with %d rows, %d + %d variables is outside the kernel limits (1024 rows, 32768 + 256)'
                                  % (rows, cols, nu))
    ws = _hip.scratch(nd)
    v = torch.empty((Bn, cols), dtype=torch.float64, device=dev)
    u = torch.empty((Bn, nu), dtype=torch.float64, device=dev)
    rn = torch.empty(Bn, dtype=torch.float64, device=dev)
    st = torch.empty(Bn, dtype=torch.int32, device=dev)
    _hip.check(lib.pympc_pnnls_batch(rows, cols, nu, Bn, _hip.ptr(A), _hip.ptr(Bm), _hip.ptr(c), _hip.ptr(ws), _hip.ptr(v),
                                     _hip.ptr(u), _hip.ptr(rn), _hip.ptr(st), None, _hip.stream_ptr()))
    return v, u, rn, st


def pnnls(A, B, c):
    """Drop-in for pnnls.pnnls (pnnls.py:5-39): (v, u, r) with v >= 0 minimising |A v + B u - c|, r the residual."""
    A = np.asarray(A, dtype=np.float64); B = np.asarray(B, dtype=np.float64); c = np.asarray(c, dtype=np.float64)
    v, u, rn, st = pnnls_batch(A[None], B[None], c[None])
    if int(st.item()) == _hip.MAXITER:
        raise RuntimeError('Maximum number of iterations reached.')
    return v[0].cpu().numpy(), u[0].cpu().numpy(), float(rn.item())


def linear_program_pnnls(f, A, b, C=None, d=None, tol=TOL):
    """The reference's own LP formulation on the device, for LPs of 1 + m + n <= 1024 rows after folding equalities and
    n <= 256 (up to 64 rows and variables on the wave kernel, beyond on the workgroup-per-problem kernels of
    csrc/nnls_block.hpp):
    the KKT system of pnnls.py:109-124 solved by pnnls(), solvable iff the residual is below tol (pnnls.py:137).
    linear_program() reaches the same primal-dual pair with a dual active-set method in n-space and decides
    feasibility row by row (violation above 1e-9 max(1, |b_j|)); the two decisions differ only for LPs that are
    infeasible or unbounded by a margin between those tolerances -- this form reproduces the reference's decision there."""
    if (C is None) != (d is None):
        raise ValueError('missing C or d.')
    f = np.asarray(f, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    n_ineq, n_x = A.shape
    n_eq = 0 if C is None else np.asarray(C).shape[0]
    if n_eq > 0:
        C = np.asarray(C, dtype=np.float64); d = np.asarray(d, dtype=np.float64)
        AC = np.vstack((A, C, -C)); bd = np.concatenate((b, d, -d))
    else:
        AC, bd = A, b
    m = AC.shape[0]
    A_p = np.vstack((np.concatenate((bd, np.zeros(m))), np.hstack((np.zeros((m, m)), np.eye(m))),
                     np.hstack((AC.T, np.zeros((n_x, m))))))
    B_p = np.vstack((f, AC, np.zeros((n_x, n_x))))
    c_p = np.concatenate((np.zeros(1), bd, -f))
    ys, x, r = pnnls(A_p, B_p, c_p)
    sol = {'min': None, 'argmin': None, 'active_set': None, 'multiplier_inequality': None, 'multiplier_equality': None}
    if r < tol:
        sol['argmin'] = x
        sol['min'] = float(f.dot(x))
        sol['multiplier_inequality'] = ys[:n_ineq]
        sol['active_set'] = sorted(int(i) for i in np.where(sol['multiplier_inequality'] > tol)[0])
        if n_eq > 0:
            sol['multiplier_equality'] = ys[n_ineq:n_ineq + n_eq] - ys[n_ineq + n_eq:m]
    return sol
