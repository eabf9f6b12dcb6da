"""LP side of the `hip` backend: same signature and `sol` dict as pnnls.linear_program, computed by the HIP LP kernel.

Mirrors /root/reference/pympc/optimization/solvers/pnnls.py:41-147:
    linear_program(f, A, b, C=None, d=None, tol=1e-7)
plus the batched form the reference does not have (its callers loop in Python, e.g. polyhedron.py:435-448):
    linear_program_batch(f, A, b, relax_idx=None, relax_val=0., ...)   B LPs of equal shape in one launch

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
    ws = torch.empty(nd, dtype=torch.float64, device=dev)
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
    res = linear_program_batch(f[None, :], AC, bd, tol=tol, n_eq=n_eq)
    return res.sol(0)
