"""Device side of the condensing: host wrapper of pympc_condense_batch and the batch-of-mpQPs object it produces.

Reference behaviour: pympc/control/controllers.py:682-749 (condense_optimal_control_problem) on top of
pympc/dynamics/discrete_time_systems.py:676-751 (condense_pwa_system), one mode sequence per call.  Here B mode
sequences are condensed by one launch of the fp64-MFMA kernel (csrc/condense.hpp) and stay on the device as the
operands of pympc_mpqp_multi_solve.

Stage domains with fewer rows than the largest one are padded with rows 0 x + 0 u <= 1, so that all mpQPs of a batch
share one shape; `rows(k)` maps the reference's (compact) row numbering of sequence k onto the padded one.
"""
import numpy as np
import torch

from .. import _hip
from ..optimization.solvers.hip import BatchResult, TOL


class PwaTables(object):
    """Constant data of a PWA optimal control problem on the device (systems, costs, padded domains, terminal set)."""

    def __init__(self, affine_systems, domains, Q, R, P, X_N):
        self.nm = len(affine_systems)
        self.nx = affine_systems[0].A.shape[0]
        self.nu = affine_systems[0].B.shape[1]
        nx, nu = self.nx, self.nu
        self.rows_per_mode = np.array([D.A.shape[0] for D in domains], dtype=np.int64)
        self.rmax = int(self.rows_per_mode.max()) if self.nm else 0
        self.rN = X_N.A.shape[0]
        Fd = np.zeros((self.nm, self.rmax, nx)); Gd = np.zeros((self.nm, self.rmax, nu)); hd = np.ones((self.nm, self.rmax))
        for i, D in enumerate(domains):
            if D.C.shape[0] != 0 or D.A.shape[1] != nx + nu:
                raise ValueError('stage domains must be inequality-only polyhedra in (x, u).')
            r = D.A.shape[0]
            Fd[i, :r] = D.A[:, :nx]; Gd[i, :r] = D.A[:, nx:]; hd[i, :r] = D.b
        if X_N.C.shape[0] != 0 or X_N.A.shape[1] != nx:
            raise ValueError('the terminal set must be an inequality-only polyhedron in x.')
        t = _hip.to_device
        self.As = t(np.array([S.A for S in affine_systems]).reshape(self.nm, nx, nx))
        self.Bs = t(np.array([S.B for S in affine_systems]).reshape(self.nm, nx, nu))
        self.cs = t(np.array([getattr(S, 'c', np.zeros(nx)) for S in affine_systems]).reshape(self.nm, nx))
        self.Q = t(Q); self.R = t(R); self.P = t(P)
        self.Fd = t(Fd); self.Gd = t(Gd); self.hd = t(hd)
        self.FN = t(X_N.A); self.hN = t(X_N.b)


def condense_batch(tables, mode_sequences, want_prediction=False):
    """Condenses B mode sequences [B x N] in one launch.  Returns a MpqpBatch (device tensors)."""
    lib = _hip.load_library()
    dev = _hip.device()
    modes = _hip.to_device(np.ascontiguousarray(mode_sequences, dtype=np.int32), dtype=torch.int32) \
        if not isinstance(mode_sequences, torch.Tensor) else mode_sequences.to(device=dev, dtype=torch.int32).contiguous()
    if modes.dim() != 2:
        raise ValueError('mode_sequences must be [B x N]')
    B, N = modes.shape
    T = tables
    nx, nu = T.nx, T.nu
    n, p, m, W = N * nu, nx, N * T.rmax + T.rN, nx + N * nu + 1
    f64 = dict(dtype=torch.float64, device=dev)
    o = dict(Huu=torch.empty((B, n, n), **f64), Hux=torch.empty((B, n, p), **f64), Hxx=torch.empty((B, p, p), **f64),
             fu=torch.empty((B, n), **f64), fx=torch.empty((B, p), **f64), g=torch.empty(B, **f64),
             Au=torch.empty((B, m, n), **f64), Ax=torch.empty((B, m, p), **f64), b=torch.empty((B, m), **f64))
    Sbar = torch.empty((B, (N + 1) * nx, W), **f64) if want_prediction else None
    status = torch.empty(B, dtype=torch.int32, device=dev)
    P = _hip.ptr
    _hip.check(lib.pympc_condense_batch(nx, nu, N, T.nm, T.rmax, T.rN, P(modes), B, P(T.As), P(T.Bs), P(T.cs), P(T.Q),
                                        P(T.R), P(T.P), P(T.Fd), P(T.Gd), P(T.hd), P(T.FN), P(T.hN), P(o['Huu']),
                                        P(o['Hux']), P(o['Hxx']), P(o['fu']), P(o['fx']), P(o['g']), P(o['Au']),
                                        P(o['Ax']), P(o['b']), P(Sbar), P(status), _hip.stream_ptr()))
    return MpqpBatch(T, modes, o, status, Sbar)


class MpqpBatch(object):
    """B condensed mpQPs of equal (padded) shape on the device; solve(x) runs them through the wave-per-QP kernel."""

    def __init__(self, tables, modes, mats, status, Sbar=None):
        self.tables = tables
        self.modes = modes
        self.mats = mats
        self.status = status
        self.Sbar = Sbar
        self.B, self.N = modes.shape
        self.n = mats['fu'].shape[1]
        self.p = mats['fx'].shape[1]
        self.m = mats['b'].shape[1]
        self._checked = False

    def check(self):
        """Raises ValueError if a mode index was out of range (one synchronisation, cached)."""
        if not self._checked:
            if bool((self.status != 0).any().item()):
                raise ValueError('mode sequence contains a mode outside [0, %d).' % self.tables.nm)
            self._checked = True

    def rows(self, k):
        """Padded row index of every row of the reference's condensed constraint set for sequence k."""
        T = self.tables
        z = self.modes[k].cpu().numpy()
        idx = [t * T.rmax + np.arange(T.rows_per_mode[z[t]]) for t in range(self.N)]
        idx.append(self.N * T.rmax + np.arange(T.rN))
        return np.concatenate(idx).astype(np.int64)

    def row_mask(self):
        """[B x m] bool: True where the padded row is a row of the reference's problem."""
        T = self.tables
        z = self.modes.cpu().numpy()
        r = T.rows_per_mode[z]                                      # [B x N]
        stage = np.arange(T.rmax)[None, None, :] < r[:, :, None]    # [B x N x rmax]
        return np.concatenate((stage.reshape(self.B, -1), np.ones((self.B, T.rN), bool)), axis=1)

    def get(self, k):
        """The reference's MultiParametricQuadraticProgram of sequence k (host arrays, compact rows)."""
        from ..optimization.parametric_programs import MultiParametricQuadraticProgram
        self.check()
        c = {key: v[k].cpu().numpy() for key, v in self.mats.items()}
        rows = self.rows(k)
        H = {'uu': c['Huu'], 'ux': c['Hux'], 'xx': c['Hxx']}
        f = {'u': c['fu'], 'x': c['fx']}
        A = {'u': c['Au'][rows], 'x': c['Ax'][rows]}
        return MultiParametricQuadraticProgram(H, f, float(c['g']), A, c['b'][rows])

    def prediction(self, k):
        """(A_bar, B_bar, c_bar) of sequence k (discrete_time_systems.py:729); needs want_prediction=True."""
        S = self.Sbar[k].cpu().numpy()
        nx = self.tables.nx
        return S[:, :nx].copy(), S[:, nx:-1].copy(), S[:, -1].copy()

    def solve(self, x, tol=TOL, want_mult=True, want_active=True, want_iters=False, want_argmin=True):
        """get_mpqp(ms).solve(x) for every sequence: x [p] (shared) or [B x p].  Returns a BatchResult over the
        padded rows (pad rows never carry a multiplier).  want_argmin=False (costs only, e.g. the arg-min over mode
        sequences): the per-sequence factorisation then skips the 40 % of its workspace that only the minimiser needs."""
        lib = _hip.load_library()
        self.check()
        dev = self.modes.device
        X = _hip.to_device(x)
        if X.dim() == 1:
            x_stride = 0
            if X.shape[0] != self.p:
                raise ValueError('x must have %d entries' % self.p)
        else:
            x_stride = self.p
            if tuple(X.shape) != (self.B, self.p):
                raise ValueError('x must be [%d x %d]' % (self.B, self.p))
        nd = lib.pympc_mpqp_multi_workspace_doubles(self.n, self.m, self.p, self.B)
        if nd == 0:
            raise NotImplementedError('mpQP with n=%d, m=%d is outside the wave kernel limits (n<=%d, m<=%d)'
                                      % ((self.n, self.m) + _hip.limits()))
        ws = torch.empty(nd, dtype=torch.float64, device=dev)
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        B, n, m = self.B, self.n, self.m
        out = dict(status=torch.empty(B, **i32), argmin=torch.empty((B, n), **f64) if want_argmin else None,
                   min=torch.empty(B, **f64))
        out['multiplier_inequality'] = torch.empty((B, m), **f64) if want_mult else None
        out['active_bits'] = torch.empty((B, (m + 31) // 32), **i32) if want_active else None
        out['iters'] = torch.empty((B, 2), **i32) if want_iters else None
        P = _hip.ptr
        M = self.mats
        _hip.check(lib.pympc_mpqp_multi_solve(n, m, self.p, B, P(M['Huu']), P(M['Hux']), P(M['Hxx']), P(M['fu']),
                                              P(M['fx']), P(M['g']), P(M['Au']), P(M['Ax']), P(M['b']), P(X), x_stride,
                                              float(tol), P(ws), P(out['argmin']), P(out['min']),
                                              P(out['multiplier_inequality']), P(out['active_bits']), P(out['status']),
                                              P(out['iters']), _hip.stream_ptr()))
        res = BatchResult(m, **out)
        res._keepalive = (ws, X)
        return res
