"""`hip` backend: same signatures and `sol` dict as pympc's pnnls backend, computed by the HIP kernels.

Mirrors /root/reference/pympc/optimization/solvers/pnnls.py:
    quadratic_program(H, f, A, b, C=None, d=None, tol=1e-7)   pnnls.py:149-241
    linear_program(f, A, b, C=None, d=None, tol=1e-7)          pnnls.py:41-147   (pympc_amd.optimization.solvers.hip_lp)
plus the batched forms the reference does not have (its callers loop in Python):
    quadratic_program_batch(H, f, A, b, ...)        B problems of equal shape in one launch
    PreparedMpqp(...).solve_batch(X)                one mpQP at B parameter values (shared matrices)

Result contract (pnnls.py:128-134): dict with keys min, argmin, active_set, multiplier_inequality,
multiplier_equality; every value None when infeasible.  Errors: ValueError('missing C or d.'),
numpy.linalg.LinAlgError (H not positive definite), RuntimeError (NNLS iteration limit).
"""
import numpy as np
import torch

from ... import _hip

TOL = 1.e-7


def _empty_sol():
    return {'min': None, 'argmin': None, 'active_set': None, 'multiplier_inequality': None,
            'multiplier_equality': None}


def _raise_for_status(status):
    if status == _hip.MAXITER:
        raise RuntimeError('Maximum number of iterations reached.')
    if status == _hip.NOT_PD:
        raise np.linalg.LinAlgError('Matrix is not positive definite')


def unpack_active(bits, m):
    """[B x ceil(m/32)] uint32 words -> [B x m] bool."""
    b = np.ascontiguousarray(bits).view(np.uint8)
    return np.unpackbits(b, axis=1, bitorder='little')[:, :m].astype(bool)


class BatchResult(object):
    """Outputs of a batched solve, kept on the device until asked for.

    Tensors (device): status [B] int32, min [B], argmin [B x n_out], multiplier_inequality [B x m] (optional),
    active_bits [B x ceil(m/32)] int32 (optional), iters [B x 2] int32 (optional).  Infeasible rows hold NaN.
    """

    def __init__(self, m, n_eq=0, n_ineq=None, **tensors):
        self.m = m
        self.n_eq = n_eq
        self.n_ineq = m if n_ineq is None else n_ineq
        self.t = tensors
        self._np = {}

    def numpy(self, key):
        if key not in self._np:
            v = self.t.get(key)
            self._np[key] = None if v is None else v.cpu().numpy()
        return self._np[key]

    def __len__(self):
        return int(self.t['status'].shape[0])

    @property
    def feasible(self):
        return self.numpy('status') == _hip.OPTIMAL

    def active_mask(self):
        bits = self.numpy('active_bits')
        return None if bits is None else unpack_active(bits, self.m)

    def sol(self, i):
        """Reference-style sol dict of problem i (raises like the scalar API on iteration limit / non-PD)."""
        st = int(self.numpy('status')[i])
        _raise_for_status(st)
        sol = _empty_sol()
        if st != _hip.OPTIMAL:
            return sol
        sol['min'] = float(self.numpy('min')[i])
        sol['argmin'] = self.numpy('argmin')[i].copy()
        lam = self.numpy('multiplier_inequality')
        if lam is not None:
            lam = lam[i]
            sol['multiplier_inequality'] = lam[:self.n_ineq].copy()
            if self.n_eq > 0:
                sol['multiplier_equality'] = lam[self.n_ineq:self.n_ineq + self.n_eq] - lam[self.n_ineq + self.n_eq:]
        act = self.numpy('active_bits')
        if act is not None:
            row = unpack_active(act[i:i + 1], self.m)[0][:self.n_ineq]
            sol['active_set'] = [int(j) for j in np.where(row)[0]]
        return sol


def _alloc_outputs(B, n_out, m, dev, want_mult, want_active, want_iters, want_min=True):
    f64 = dict(dtype=torch.float64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    out = dict(status=torch.empty(B, **i32))
    out['argmin'] = torch.empty((B, n_out), **f64) if n_out > 0 else None
    out['min'] = torch.empty(B, **f64) if want_min else None
    out['multiplier_inequality'] = torch.empty((B, m), **f64) if want_mult else None
    out['active_bits'] = torch.empty((B, (m + 31) // 32), **i32) if want_active else None
    out['iters'] = torch.empty((B, 2), **i32) if want_iters else None
    return out


class PreparedMpqp(object):
    """One mpQP (parametric_programs.py:18-40) factorised once on the device; solve_batch(X) runs the wave kernel.

    Hoists pnnls.py:210-214 (Cholesky, inverses, M) out of the per-state loop.
    """

    def __init__(self, Huu, Hux, Hxx, fu, fx, g, Au, Ax, b):
        lib = _hip.load_library()
        self.dev = _hip.device()
        self.m, self.n = Au.shape
        self.p = Ax.shape[1]
        t = _hip.to_device
        self._inputs = [t(Huu), t(Hux), t(Hxx), t(fu), t(fx), t(Au), t(Ax), t(b)]
        nd = lib.pympc_mpqp_workspace_doubles(self.n, self.m, self.p)
        self.large = False
        if nd == 0:
            nmax, mmax = _hip.limits()
            if lib.pympc_qp_workspace_doubles(self.n, self.m, 1) == 0:
                raise NotImplementedError('mpQP with n=%d, m=%d is outside the kernel limits (n<=%d prepared once, n<=1023 '
                                          'solved per state, m<=%d)' % (self.n, self.m, nmax, mmax))
            # 64 .. 1023 variables, or more than 2048 constraints: no prepared form (the wave kernel's factors and
            # per-constraint bit sets live in LDS and registers); every state is solved as
            # a QP of its own by the workgroup-per-problem kernels of csrc/qp_block.hpp (solve_batch below)
            self.large = True
            self.g = float(g)
            self.ws = None
            self.cuts = None
            self._seeded = True
            st = quadratic_program_batch(self._inputs[0], self._inputs[3].reshape(1, -1), self._inputs[5],
                                         self._inputs[7].reshape(1, -1), want_mult=False, want_active=False).t['status']
            if int(st.item()) == _hip.NOT_PD:
                raise np.linalg.LinAlgError('Matrix is not positive definite')
            return
        self.ws = torch.empty(nd, dtype=torch.float64, device=self.dev)
        self.ws_status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        Huu_, Hux_, Hxx_, fu_, fx_, Au_, Ax_, b_ = self._inputs
        _hip.check(lib.pympc_mpqp_prepare(self.n, self.m, self.p, _hip.ptr(Huu_), _hip.ptr(Hux_), _hip.ptr(Hxx_),
                                          _hip.ptr(fu_), _hip.ptr(fx_), float(g), _hip.ptr(Au_), _hip.ptr(Ax_),
                                          _hip.ptr(b_), _hip.ptr(self.ws), _hip.ptr(self.ws_status), _hip.stream_ptr()))
        if int(self.ws_status.item()) == _hip.NOT_PD:
            raise np.linalg.LinAlgError('Matrix is not positive definite')
        # learned infeasibility certificates of this mpQP (include/pympc_hip.h: pympc_mpqp_solve_batch_cuts)
        nc = lib.pympc_mpqp_cuts_doubles(self.p)
        self.cuts = torch.zeros(nc, dtype=torch.float64, device=self.dev) if nc > 0 else None
        self._seeded = False

    # batches from this size on seed the certificate table first (once per prepared mpQP); smaller ones learn as they go
    SEED_MIN_BATCH = 65536
    SEED_SCALES = (0.25, 0.4, 0.6, 0.85, 1.15, 1.6)
    SEED_STATES = 131072       # per scale and round (>= the batch size from which a launch starts with a learning pre-pass)
    SEED_ROUNDS = 2
    SEED_PROFILE = False       # True: synchronise after every phase of seed_cuts and keep the times in self.seed_timing

    def feasible_box(self):
        """Bounding box (lo, hi) of the feasible parameter set {x : exists u, Au u + Ax x <= b} (the polytope of
        parametric_programs.py:243-260), by 2p LPs in (x, u) in one launch; None when it is empty, unbounded or beyond
        the LP kernel's size."""
        from .hip_lp import linear_program_batch
        Au_, Ax_, b_ = self._inputs[5], self._inputs[6], self._inputs[7]
        p, n = self.p, self.n
        F = torch.zeros((2 * p, p + n), dtype=torch.float64, device=self.dev)
        F[torch.arange(p), torch.arange(p)] = 1.
        F[p + torch.arange(p), torch.arange(p)] = -1.
        # The objective does not see u at all: massively dual degenerate, the active-set LP kernel walks thousands of
        # pivots through ties (3 295 for the cart-pole problem, 57 ms).  A fixed 1e-7 tilt on u breaks the ties (45 pivots,
        # 1.2 ms); the box is read off the x part of the minimiser, which the tilt moves by nothing that matters for a
        # sampling box.
        g = torch.Generator(device='cpu').manual_seed(0)
        F[:, p:] = 1.e-7 * torch.randn((2 * p, n), generator=g, dtype=torch.float64).to(self.dev)
        try:
            res = linear_program_batch(F, torch.cat((Ax_, Au_), dim=1).contiguous(), b_, want_argmin=True,
                                       want_mult=False, want_active=False)
        except NotImplementedError:
            return None
        if not bool((res.t['status'] == _hip.OPTIMAL).all()):
            return None
        xs = res.t['argmin'][:, :p]
        v = xs[torch.arange(2 * p), torch.arange(2 * p) % p]
        return v[:p].clone(), v[p:].clone()

    def seed_cuts(self, seed=0):
        """Certificates at set-up time.  Synthetic states spread over the bounding box of the feasible set (several
        scales, fixed seed: no user data involved) are solved by the batched path itself, outputs discarded: states a
        stored certificate already rejects cost nothing there, so every further batch spends its time on the part of
        the infeasible region that is not covered yet.  The certificates found are marked as part of the prepared
        mpQP (reset_cuts() returns to them).  Returns the number of certificates in the table."""
        if self.cuts is None:
            return 0
        self._seeded = True
        import time
        prof = self.SEED_PROFILE

        def tick(label, t0):
            if prof:
                torch.cuda.synchronize()
                self.seed_timing[label] = self.seed_timing.get(label, 0.) + (time.perf_counter() - t0) * 1e3
            return time.perf_counter()
        self.seed_timing = {}
        t0 = time.perf_counter()
        box = self.feasible_box()
        t0 = tick("feasible_box", t0)
        if box is None:
            return self.n_cuts()
        lib = _hip.load_library()
        lo, hi = box
        c, h = .5 * (lo + hi), .5 * (hi - lo)
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        k = self.SEED_STATES
        keep = []
        for rnd in range(self.SEED_ROUNDS):
            for sc in self.SEED_SCALES:
                U = torch.rand((k, self.p), dtype=torch.float64, device=self.dev, generator=gen) * 2. - 1.
                X = (c[None, :] + sc * h[None, :] * U).contiguous()
                st = torch.empty(k, dtype=torch.int32, device=self.dev)
                keep.append((X, st))
                _hip.check(lib.pympc_mpqp_solve_batch_cuts(self.n, self.m, self.p, _hip.ptr(self.ws), _hip.ptr(self.cuts),
                                                           _hip.ptr(X), k, TOL, 0, None, None, None, None, _hip.ptr(st), None,
                                                           _hip.stream_ptr()))
                t0 = tick("batch scale %.2f" % sc, t0)
        self._select_cuts(keep)
        t0 = tick("selection", t0)
        _hip.check(lib.pympc_mpqp_cuts_mark(self.p, _hip.ptr(self.cuts), _hip.stream_ptr()))
        return self.n_cuts()

    SEED_KEEP = 96            # certificates kept after set-up (every state of a batch is tested against all of them)
    SEED_STRIDE = 16          # every SEED_STRIDE-th seed state takes part in the selection

    def _select_cuts(self, batches):
        """Keep the certificates that matter, most useful first.  The kernels test EVERY state against EVERY stored
        certificate (p + 1 FMAs each), and sampling finds many that overlap.  Greedy cover on a subsample of the
        infeasible seed states: repeatedly take the certificate that rejects most states no earlier pick rejects (the
        kernels' own test: c + d'x below -1e-6 (|c| + |d|'|x|)), until SEED_KEEP are taken or nothing is gained.  What
        the dropped ones covered is solved in full by the kernels, as before; no result depends on the table.  (On the
        host with numpy: a ranking heuristic of a few milliseconds, not part of the solve path.)"""
        n = self.n_cuts()
        if n <= 1:
            return
        KC = 256                                                    # kCutMax (csrc/qp_wave.hpp)
        H = self.cuts.numel() - (self.p + 1) * KC                   # header doubles (kCutHeader)
        body = self.cuts[H:H + (self.p + 1) * KC].view(self.p + 1, KC)
        tab = body[:, :n].cpu().numpy()
        cc, D = tab[0], tab[1:]                                     # c [n], d [p x n]
        k = self.SEED_STRIDE
        Xs = torch.cat([x[::k] for x, _ in batches]).cpu().numpy()
        ss = torch.cat([s_[::k] for _, s_ in batches]).cpu().numpy()
        X = Xs[ss == _hip.INFEASIBLE]
        if X.shape[0] == 0:
            return
        v = cc[None, :] + X.dot(D)
        a = np.abs(cc)[None, :] + np.abs(X).dot(np.abs(D))
        fires = v < -1e-6 * a                                        # [N x n]
        total = fires.shape[0]
        order = []
        for _ in range(min(n, self.SEED_KEEP)):
            gain = fires.sum(axis=0)
            j = int(gain.argmax())
            if gain[j] == 0:
                break
            order.append(j)
            fires = fires[~fires[:, j]]                              # the states no pick rejects yet
        if not order:
            return
        new = np.zeros((self.p + 1, KC))
        new[:, :len(order)] = tab[:, order]
        body.copy_(torch.as_tensor(new).to(self.dev))
        self.cuts[:1].view(torch.int32)[0] = len(order)
        self.seed_coverage = 1. - fires.shape[0] / float(total)      # of the infeasible seed states, by the kept certificates

    def reset_cuts(self, hard=False, table=None):
        """Forget the infeasibility certificates learned from the batches solved so far; the ones stored at set-up time
        (seed_cuts) stay.  hard=True empties the table altogether.  Asynchronous on the current stream.  `table`: a copy
        made by clone_cuts() instead of the prepared mpQP's own table."""
        table = self.cuts if table is None else table
        if table is not None:
            lib = _hip.load_library()
            if hard:
                _hip.check(lib.pympc_mpqp_cuts_reset(self.p, _hip.ptr(table), _hip.stream_ptr()))
            else:
                _hip.check(lib.pympc_mpqp_cuts_restore(self.p, _hip.ptr(table), _hip.stream_ptr()))

    def clone_cuts(self):
        """A copy of the certificate table (set-up mark included) for launches that run CONCURRENTLY with others of this
        mpQP on another stream: launches that share one table must be ordered on one stream (include/pympc_hip.h)."""
        if self.cuts is not None and not self._seeded:
            self.seed_cuts()
        return None if self.cuts is None else self.cuts.clone()

    def n_cuts(self):
        return 0 if self.cuts is None else int(self.cuts[:1].view(torch.int32)[0].item())

    def solve_batch(self, X, tol=TOL, n_out=None, want_mult=True, want_active=True, want_iters=False, want_min=True,
                    out=None, use_cuts=True, cuts_table=None):
        """X: [B x p] (numpy or device tensor).  Returns a BatchResult (device tensors, nothing synchronised).

        use_cuts: reject states that a stored Farkas certificate proves infeasible before iterating, and store the
        certificates of the states found infeasible by the full solve (same statuses either way)."""
        lib = _hip.load_library()
        X = _hip.to_device(X)
        if X.dim() != 2 or X.shape[1] != self.p:
            raise ValueError('X must be [B x %d]' % self.p)
        B = X.shape[0]
        n_out = self.n if n_out is None else n_out
        if self.large:
            if out is not None:
                raise NotImplementedError('preallocated outputs are for the prepared form (n <= 63)')
            Huu, Hux, Hxx, fu, fx, Au, Ax, b = self._inputs
            F = (fu.reshape(1, -1) + X.matmul(Hux.reshape(self.n, self.p).t())).contiguous()     # f(x) = fu + Hux x
            Bv = (b.reshape(1, -1) - X.matmul(Ax.reshape(self.m, self.p).t())).contiguous()      # b(x) = b - Ax x
            res = quadratic_program_batch(Huu, F, Au, Bv, tol=tol, want_mult=want_mult, want_active=want_active,
                                          want_iters=want_iters)
            t_ = res.t
            if want_min:   # + x'Hxx x / 2 + fx'x + g (parametric_programs.py:171-173)
                t_['min'] = t_['min'] + .5 * (X.matmul(Hxx.reshape(self.p, self.p)) * X).sum(dim=1) + X.matmul(fx.reshape(-1)) + self.g
            else:
                t_['min'] = None
            t_['argmin'] = t_['argmin'][:, :n_out].contiguous() if n_out > 0 else None
            return BatchResult(self.m, **t_)
        if out is None:
            out = _alloc_outputs(B, n_out, self.m, self.dev, want_mult, want_active, want_iters, want_min)
        cuts = (self.cuts if cuts_table is None else cuts_table) if use_cuts else None
        if cuts is not None and cuts_table is None and not self._seeded and B >= self.SEED_MIN_BATCH:
            self.seed_cuts()
        _hip.check(lib.pympc_mpqp_solve_batch_cuts(self.n, self.m, self.p, _hip.ptr(self.ws), _hip.ptr(cuts), _hip.ptr(X), B,
                                                   float(tol), n_out, _hip.ptr(out['argmin']), _hip.ptr(out['min']),
                                                   _hip.ptr(out['multiplier_inequality']), _hip.ptr(out['active_bits']),
                                                   _hip.ptr(out['status']), _hip.ptr(out['iters']), _hip.stream_ptr()))
        return BatchResult(self.m, **out)


def quadratic_program_batch(H, f, A, b, tol=TOL, n_eq=0, want_mult=True, want_active=True, want_iters=False):
    """B strictly convex QPs of equal shape in one launch.

    H: [n x n] (shared) or [B x n x n];  f: [B x n];  A: [m x n] (shared) or [B x m x n];  b: [B x m].
    Equalities must already be folded as rows (C, -C) (pnnls.py:202-207); n_eq tells BatchResult.sol how to
    unfold the multipliers.
    """
    lib = _hip.load_library()
    dev = _hip.device()
    H = _hip.to_device(H); f = _hip.to_device(f); A = _hip.to_device(A); b = _hip.to_device(b)
    if f.dim() != 2 or b.dim() != 2:
        raise ValueError('f and b must be [B x n] and [B x m]')
    B, n = f.shape
    m = b.shape[1]
    h_stride = 0 if H.dim() == 2 else n * n
    a_stride = 0 if A.dim() == 2 else m * n
    nd = lib.pympc_qp_workspace_doubles(n, m, B)
    if nd == 0:
        nmax, mmax = _hip.limits()
        raise NotImplementedError('QP with n=%d, m=%d is outside the kernel limits (n<=%d on the wave kernel, n<=1023 on the '
                                  'workgroup kernel of csrc/qp_block.hpp; m<=%d on the wave kernel, m<=32768 beyond)' % (n, m, nmax, mmax))
    ws = _hip.scratch(nd)
    out = _alloc_outputs(B, n, m, dev, want_mult, want_active, want_iters)
    _hip.check(lib.pympc_qp_solve_batch(n, m, B, _hip.ptr(H), h_stride, _hip.ptr(f), _hip.ptr(A), a_stride, _hip.ptr(b),
                                        float(tol), _hip.ptr(ws), _hip.ptr(out['argmin']), _hip.ptr(out['min']),
                                        _hip.ptr(out['multiplier_inequality']), _hip.ptr(out['active_bits']),
                                        _hip.ptr(out['status']), _hip.ptr(out['iters']), _hip.stream_ptr()))
    return BatchResult(m, n_eq=n_eq, n_ineq=m - 2 * n_eq, **out)


def quadratic_program(H, f, A, b, C=None, d=None, tol=TOL):
    """Drop-in for pnnls.quadratic_program (pnnls.py:149-241): one strictly convex QP on the GPU."""
    if (C is None) != (d is None):
        raise ValueError('missing C or d.')
    H = np.asarray(H, dtype=np.float64); f = np.asarray(f, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    n_eq = 0 if C is None else np.asarray(C).shape[0]
    if n_eq > 0:   # state equalities as inequalities (pnnls.py:202-207)
        C = np.asarray(C, dtype=np.float64); d = np.asarray(d, dtype=np.float64)
        AC = np.vstack((A, C, -C))
        bd = np.concatenate((b, d, -d))
    else:
        AC, bd = A, b
    res = quadratic_program_batch(H, f[None, :], AC, bd[None, :], tol=tol, n_eq=n_eq)
    return res.sol(0)
