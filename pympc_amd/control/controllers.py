"""MPC controllers on the batched HIP path, mirroring /root/reference/pympc/control/controllers.py.

    ModelPredictiveController(S, N, Q, R, P, D, X_N)            :11-250
        feedforward(x) / feedback(x)                            :76-124    x 1-D: reference semantics (None when
                                                                           infeasible); x [B x nx]: one launch, arrays
        store_explicit_solution / feedforward_explicit / feedback_explicit   :126-186
    HybridModelPredictiveController(S, N, Q, R, P, X_N)         :252-680   fixed-mode-sequence side and big-M set-up:
        get_mpqp(mode_sequence)                                 :668-680
        _get_bigM_dynamics() / _get_bigM_domains()              :289-405   one batched LP launch per domain
        get_mpqp_batch(mode_sequences) / solve_mode_sequences(x, mode_sequences)      (batched, new)
        feedforward(x) / feedback(x)                            :611-666   exact pruned enumeration of the mode sequences
                                                                           in place of the Gurobi MIQP
    condense_optimal_control_problem(S, Q, R, P, X_N, mode_sequence)                :682-749

The mixed-integer program itself (mpMIQP condensing and its solution through Gurobi, :407-666) is outside the pnnls path.
"""
import numpy as np
import torch

from ..dynamics.discrete_time_systems import AffineSystem, PieceWiseAffineSystem
from .condense import PwaTables, condense_batch


def _is_batch(x):
    return (isinstance(x, torch.Tensor) and x.dim() == 2) or (not isinstance(x, torch.Tensor) and np.ndim(x) == 2)


class ModelPredictiveController(object):
    """
    Linear MPC: V*(x0) = min 1/2 sum_{t<N} (x'Qx + u'Ru) + 1/2 x_N'P x_N  s.t. x+ = A x + B u, (x, u) in D, x_N in X_N.
    """

    def __init__(self, S, N, Q, R, P, D, X_N):
        self.S = S
        self.N = N
        self.Q = Q
        self.R = R
        self.P = P
        self.D = D
        self.X_N = X_N
        self.explicit_solution = None
        self.mpqp = self._condense_program()

    def _condense_program(self):
        """Condensed mpQP of the optimal control problem: the PWA condenser with a single mode (:58-74)."""
        S = PieceWiseAffineSystem([AffineSystem(self.S.A, self.S.B, np.zeros(self.S.nx))], [self.D])
        return condense_optimal_control_problem(S, self.Q, self.R, self.P, self.X_N, [0] * self.N)

    # ------------------------------------------------------------------ implicit (online QP)
    def feedforward(self, x):
        """
        x 1-D: (list of the N optimal inputs, V) or (None, None) when infeasible (:76-102).
        x [B x nx]: (U [B x N x nu], V [B], feasible [B] bool) as host arrays; infeasible rows hold NaN.
        """
        if _is_batch(x):
            res = self.mpqp.solve(x, want_mult=False, want_active=False)
            U = res.numpy('argmin').reshape(len(res), self.N, self.S.nu)
            return U, res.numpy('min'), res.feasible
        sol = self.mpqp.solve(x)
        if sol['min'] is None:
            return None, None
        nu = self.S.nu
        return [sol['argmin'][nu * t:nu * (t + 1)] for t in range(self.N)], sol['min']

    def feedback(self, x):
        """
        x 1-D: first optimal input or None (:104-124).  x [B x nx]: [B x nu] array with NaN rows where infeasible
        (only u(0) leaves the device).
        """
        if _is_batch(x):
            return self.feedback_batch(x).numpy('argmin')
        u = self.feedforward(x)[0]
        return None if u is None else u[0]

    def feedback_batch(self, X, **kwargs):
        """Device-resident batched feedback: BatchResult with argmin = u(0) [B x nu], min = V, status; nothing is
        synchronised or copied.  This is the call the headline benchmark times."""
        kwargs.setdefault('want_mult', False)
        kwargs.setdefault('want_active', False)
        return self.mpqp.solve(X, n_out=self.S.nu, **kwargs)

    # ------------------------------------------------------------------ explicit (table look-up)
    def store_explicit_solution(self, samples=None, **kwargs):
        """
        Stores the explicit solution of the condensed mpQP (:126-136): facet-flipping exploration as in the reference
        (kwargs: step_size, verbose), or, with samples [B x nx], the table of the active sets the batched QP path finds
        at those points (one launch; regions in order of first appearance).
        """
        if samples is None:
            self.explicit_solution = self.mpqp.explicit_solve(**kwargs)
        else:
            self.explicit_solution = self.mpqp.explicit_solve_from_samples(samples, **kwargs)

    def feedforward_explicit(self, x):
        """(u, V) from the stored table (:138-164); batch: (U [B x N x nu], V [B], region index [B])."""
        if self.explicit_solution is None:
            raise ValueError('explicit solution not stored.')
        nu = self.S.nu
        if _is_batch(x):
            idx, U, V = self.explicit_solution.locate(x)
            return U.cpu().numpy().reshape(-1, self.N, nu), V.cpu().numpy(), idx.cpu().numpy()
        u = self.explicit_solution.u(x)
        if u is not None:
            u = [u[t * nu:(t + 1) * nu] for t in range(self.N)]
        return u, self.explicit_solution.V(x)

    def feedback_explicit(self, x):
        """First input from the stored table (:166-186); batch: [B x nu] with NaN rows where uncovered."""
        if self.explicit_solution is None:
            raise ValueError('explicit solution not stored.')
        if _is_batch(x):
            return self.explicit_solution.locate(x, nu=self.S.nu)[1].cpu().numpy()
        u = self.feedforward_explicit(x)[0]
        return None if u is None else u[0]


class HybridModelPredictiveController(object):
    """
    Hybrid MPC for PWA systems, restricted to what runs through the pnnls path in the reference: the convex QP of a
    FIXED mode sequence (get_mpqp, :668-680).  The batched forms condense and solve many mode sequences at once.
    """

    def __init__(self, S, N, Q, R, P, X_N):
        self.S = S
        self.N = N
        self.Q = Q
        self.R = R
        self.P = P
        self.X_N = X_N
        self._tables = None
        self._ptables = None

    def tables(self):
        if self._tables is None:
            self._tables = PwaTables(self.S.affine_systems, self.S.domains, self.Q, self.R, self.P, self.X_N)
        return self._tables

    def get_mpqp(self, mode_sequence):
        return condense_optimal_control_problem(self.S, self.Q, self.R, self.P, self.X_N, mode_sequence, self.tables())

    def _domain_lps(self, j, F):
        """min f'z over the domain D_j for every row f of F [K x (nx + nu)]: ONE batched launch against the shared
        (A, b) of D_j, equalities folded as rows (C, -C) like pnnls.py:101-106.  Returns the K optimal values (a
        ValueError names the domain if one of them is unbounded or the domain is empty)."""
        from ..optimization.solvers import hip_lp
        D = self.S.domains[j]
        A, b = D.A, D.b
        if D.C.shape[0] > 0:
            A = np.vstack((A, D.C, -D.C))
            b = np.concatenate((b, D.d, -D.d))
        res = hip_lp.linear_program_batch(np.ascontiguousarray(F), A, b, want_argmin=False, want_mult=False,
                                          want_active=False)
        st = res.numpy('status')
        if np.any(st == 3):
            raise RuntimeError('Maximum number of iterations reached.')
        if np.any(st != 0):
            raise ValueError('big-M LP over domain %d has no optimum (unbounded or empty domain).' % j)
        return res.numpy('min')

    def _get_bigM_dynamics(self):
        """
        Big-Ms of the PWA dynamics (controllers.py:289-358): alpha[i][j] = min over D_j of A_i x + B_i u + c_i,
        beta[i][j] = max of the same, component by component (column vectors, as the reference returns them).  The
        reference solves 2 nx LPs per pair (i, j) one after the other; here every LP over the same domain D_j, for all
        i and both signs, is one batched launch.
        """
        S = self.S.affine_systems
        s = len(S)
        rows = np.vstack([np.hstack((Si.A, Si.B)) for Si in S])      # [sum nx x (nx + nu)], system i at rows off[i]:off[i+1]
        off = np.cumsum([0] + [Si.nx for Si in S])
        alpha = [[None] * s for _ in range(s)]
        beta = [[None] * s for _ in range(s)]
        for j in range(s):
            v = self._domain_lps(j, np.vstack((rows, -rows)))
            lo, hi = v[:rows.shape[0]], -v[rows.shape[0]:]
            for i, Si in enumerate(S):
                alpha[i][j] = (lo[off[i]:off[i + 1]] + Si.c).reshape(-1, 1)
                beta[i][j] = (hi[off[i]:off[i + 1]] + Si.c).reshape(-1, 1)
        return alpha, beta

    def _get_bigM_domains(self):
        """
        Big-Ms of the PWA domains (controllers.py:360-405): gamma[i][j] = max over D_j of F_i x + G_i u - h_i, row by
        row.  One batched launch per domain D_j for the rows of every D_i.
        """
        Ds = self.S.domains
        s = len(Ds)
        rows = np.vstack([D.A for D in Ds])
        off = np.cumsum([0] + [D.A.shape[0] for D in Ds])
        gamma = [[None] * s for _ in range(s)]
        for j in range(s):
            v = -self._domain_lps(j, -rows)
            for i, D in enumerate(Ds):
                gamma[i][j] = (v[off[i]:off[i + 1]] - D.b).reshape(-1, 1)
        return gamma

    def get_mpqp_batch(self, mode_sequences):
        """Condenses every row of mode_sequences [B x N] in one launch -> MpqpBatch (device)."""
        return condense_batch(self.tables(), mode_sequences)

    def solve_mode_sequences(self, x, mode_sequences, **kwargs):
        """get_mpqp(ms).solve(x) for every row ms of mode_sequences; returns (BatchResult, MpqpBatch)."""
        batch = self.get_mpqp_batch(mode_sequences)
        return batch.solve(x, **kwargs), batch

    # ------------------------------------------------------------------ mode-sequence search (in place of the MIQP)
    def _prefix_tables(self):
        """Tables of the prefix relaxation: horizon t < N with the mode sequence's first t modes, terminal weight Q and no
        terminal set.  Its optimal value is a lower bound for every completion of the prefix (the completion keeps all of
        the prefix's constraints and costs, x_t is weighted with Q there as well, everything added is non-negative)."""
        if getattr(self, '_ptables', None) is None:
            from ..geometry.polyhedron import Polyhedron
            free = Polyhedron(np.zeros((0, self.S.nx)), np.zeros(0))
            self._ptables = PwaTables(self.S.affine_systems, self.S.domains, self.Q, self.R, self.Q, free)
        return self._ptables

    def _evaluate(self, x, seqs):
        """Optimal values (inf where infeasible) and solutions of the fixed-mode-sequence QPs of seqs [B x t]: prefix
        relaxations for t < N, the full problem (terminal cost P, terminal set X_N) for t = N."""
        t = seqs.shape[1]
        batch = condense_batch(self.tables() if t == self.N else self._prefix_tables(), seqs)
        res = batch.solve(x, want_mult=False, want_active=False, want_argmin=(t == self.N))   # (prefixes: costs only)
        V = torch.where(res.t['status'] == 0, res.t['min'], torch.full_like(res.t['min'], float('inf')))
        return V.cpu().numpy(), res

    def feedforward(self, x, verbose=False):
        """
        Optimal input sequence, state trajectory, mode sequence and cost for the state x, or (None, None, None, None)
        when no mode sequence is feasible -- the return values of controllers.py:611-648, which the reference obtains
        from a mixed-integer QP (Gurobi).  Here: exact enumeration of the mode sequences with pruning, every level of
        the tree one batched launch (condensing kernel + QP kernel) over the surviving prefixes x modes:
          * a prefix whose relaxation (see _prefix_tables) is infeasible has no feasible completion;
          * a prefix whose relaxation already costs more than the best complete sequence found so far cannot improve it
            (the incumbent comes from one greedy dive before the sweep).
        """
        nm, N, nu = self.S.nm, self.N, self.S.nu
        x = np.asarray(x, dtype=np.float64)
        best_V, best_seq, best_u = float('inf'), None, None

        def children(front):
            return np.hstack((np.repeat(front, nm, axis=0), np.tile(np.arange(nm, dtype=np.int32), front.shape[0])[:, None]))

        def complete(V, res, seqs):
            nonlocal best_V, best_seq, best_u
            k = int(np.argmin(V))
            if V[k] < best_V:
                best_V, best_seq, best_u = float(V[k]), seqs[k].copy(), res.t['argmin'][k].cpu().numpy()

        # greedy dive: follow the cheapest feasible child
        front = np.zeros((1, 0), dtype=np.int32)
        for t in range(1, N + 1):
            seqs = children(front)
            V, res = self._evaluate(x, seqs)
            if not np.isfinite(V).any():
                break
            if t == N:
                complete(V, res, seqs)
            front = seqs[[int(np.argmin(V))]]
        # level-by-level sweep with pruning
        front = np.zeros((1, 0), dtype=np.int32)
        for t in range(1, N + 1):
            seqs = children(front)
            V, res = self._evaluate(x, seqs)
            if t == N:
                if np.isfinite(V).any():
                    complete(V, res, seqs)
                break
            keep = np.isfinite(V) & (V <= best_V + 1e-9 * max(1., abs(best_V) if np.isfinite(best_V) else 1.))
            if verbose:
                print('level %d: %d prefixes, %d survive (incumbent %g)' % (t, seqs.shape[0], int(keep.sum()), best_V))
            front = seqs[keep]
            if front.shape[0] == 0:
                break
        if best_seq is None:
            return None, None, None, None
        u_list = [best_u[nu * t:nu * (t + 1)] for t in range(N)]
        x_list = [x]
        for t in range(N):
            Sz = self.S.affine_systems[int(best_seq[t])]
            x_list.append(Sz.A.dot(x_list[-1]) + Sz.B.dot(u_list[t]) + getattr(Sz, 'c', np.zeros(self.S.nx)))
        return u_list, x_list, [int(z) for z in best_seq], best_V

    def feedback(self, x):
        """First input of feedforward(x), or None (controllers.py:650-666)."""
        u = self.feedforward(x)[0]
        return None if u is None else u[0]

    def feedforward_over(self, x, mode_sequences):
        """
        Best fixed-mode-sequence solution among the given candidates: (u list, V, mode_sequence) of the feasible
        sequence with the lowest cost, or (None, None, None).  (The reference obtains the same optimum from its MIQP,
        test_controllers.py:270-289.)
        """
        res, batch = self.solve_mode_sequences(x, mode_sequences, want_mult=False, want_active=False)
        V = torch.where(res.t['status'] == 0, res.t['min'], torch.full_like(res.t['min'], float('inf')))
        k = int(torch.argmin(V).item())
        if not np.isfinite(float(V[k].item())):
            return None, None, None
        u = res.t['argmin'][k].cpu().numpy()
        nu = self.S.nu
        return [u[t * nu:(t + 1) * nu] for t in range(self.N)], float(V[k].item()), [int(z) for z in batch.modes[k].cpu().numpy()]


def graph_representation(S):
    """Graphs {(x, u, x+) | F_i x + G_i u <= h_i, x+ = A_i x + B_i u + c_i} of the modes of the PWA system S as a list of
    Polyhedron, the equality written as two inequalities (examples/pwa_from_urdf/utils.py:349-369)."""
    from ..geometry.polyhedron import Polyhedron
    P = []
    for Si, Di in zip(S.affine_systems, S.domains):
        nx = S.nx
        Ai = np.vstack((np.hstack((Di.A, np.zeros((Di.A.shape[0], nx)))), np.hstack((Si.A, Si.B, -np.eye(nx))),
                        np.hstack((-Si.A, -Si.B, np.eye(nx)))))
        P.append(Polyhedron(Ai, np.concatenate((Di.b, -Si.c, Si.c))))
    return P


def big_m(P_list, tol=1.e-6):
    """m[i][j][k] = max over P_j of A_i[k] x - b_i[k] for a list of polyhedra {x | A_i x <= b_i}, values below tol set to
    zero (examples/pwa_from_urdf/utils.py:371-389: one LP per (i, j, k), solved one after the other there).  The LPs
    over the same P_j share their constraint set: one batched launch per j with the rows of every P_i as objectives."""
    from ..optimization.solvers import hip_lp
    rows = np.vstack([P.A for P in P_list])
    rhs = np.concatenate([P.b for P in P_list])
    off = np.cumsum([0] + [P.A.shape[0] for P in P_list])
    m = [[None] * len(P_list) for _ in P_list]
    for j, Pj in enumerate(P_list):
        res = hip_lp.linear_program_batch(-rows, Pj.A, Pj.b, want_argmin=False, want_mult=False, want_active=False)
        st = res.numpy('status')
        if np.any(st == 3):
            raise RuntimeError('Maximum number of iterations reached.')
        if np.any(st != 0):
            raise TypeError("bad operand type for unary -: 'NoneType' (an LP of big_m over polyhedron %d has no optimum)" % j)
        v = -res.numpy('min') - rhs
        v[np.abs(v) < tol] = 0.
        for i in range(len(P_list)):
            m[i][j] = v[off[i]:off[i + 1]].copy()
    return m


def condense_optimal_control_problem(S, Q, R, P, X_N, mode_sequence, tables=None):
    """Condensed mpQP of the PWA optimal control problem for one mode sequence (:682-749), by the MFMA kernel."""
    if tables is None:
        tables = PwaTables(S.affine_systems, S.domains, Q, R, P, X_N)
    batch = condense_batch(tables, np.asarray(mode_sequence, dtype=np.int32).reshape(1, -1))
    return batch.get(0)
