"""Discrete-time linear / affine / piecewise-affine systems: the part of
/root/reference/pympc/dynamics/discrete_time_systems.py that the MPC hot path needs.

    LinearSystem(A, B)                      :12-283   simulate, simulate_closed_loop, solve_dare, mcais, condense,
                                                      controllable, from_continuous
    AffineSystem(A, B, c)                   :285-432  simulate, condense, from_continuous
    PieceWiseAffineSystem(systems, domains) :434-582  condense, simulate, get_mode, is_well_posed
    mcais(A, X)                             :584-674  one LP per facet per iteration -> one batched launch per iteration
    condense_pwa_system(systems, modes)     :676-729  prediction matrices -> fp64 MFMA kernel (csrc/condense.hpp)

The symbolic constructors (sympy) are set-up conveniences outside the path and are not provided.
"""
from copy import copy

import numpy as np
from scipy.linalg import expm, solve_discrete_are

from ..geometry.polyhedron import Polyhedron


def _check_affine_system(A, B, c=None, h=None):
    """Shape checks of dynamics/utils.py:4-37 (same messages)."""
    if A.shape[0] != A.shape[1]:
        raise ValueError('A must be a square matrix.')
    if A.shape[0] != B.shape[0]:
        raise ValueError('A and B must have the same number of rows.')
    if c is not None:
        if c.ndim > 1:
            raise ValueError('c must be a 1-dimensional array.')
        if A.shape[0] != c.size:
            raise ValueError('A and c must have the same number of rows.')
    if h is not None and h < 0:
        raise ValueError('the time step h must be positive.')


def explicit_euler(A, B, c, h):
    """x(t+1) ~ x(t) + h dx/dt (discretization_methods.py:8-41)."""
    _check_affine_system(A, B, c, h)
    return np.eye(A.shape[0]) + h * A, h * B, h * c


def zero_order_hold(A, B, c, h):
    """Exact discretisation for piecewise constant inputs: top rows of expm([[A B c]; 0] h) (discretization_methods.py:43-104)."""
    _check_affine_system(A, B, c, h)
    nx, nu = B.shape
    M = np.zeros((nx + nu + 1, nx + nu + 1))
    M[:nx, :nx] = A; M[:nx, nx:nx + nu] = B; M[:nx, -1] = c
    Md = expm(M * h)
    return Md[:nx, :nx], Md[:nx, nx:nx + nu], Md[:nx, -1]


def _discretize(A, B, c, h, method):
    if method == 'zero_order_hold':
        return zero_order_hold(A, B, c, h)
    if method == 'explicit_euler':
        return explicit_euler(A, B, c, h)
    raise ValueError('unknown discretization method.')


class LinearSystem(object):
    """x(t+1) = A x(t) + B u(t)."""

    def __init__(self, A, B):
        _check_affine_system(A, B)
        self.A = A
        self.B = B
        self.nx, self.nu = B.shape
        self._controllable = None

    def simulate(self, x0, u):
        x = [x0]
        for v in u:
            x.append(self.A.dot(x[-1]) + self.B.dot(v))
        return x

    def simulate_closed_loop(self, x0, N, K):
        Acl = self.A + self.B.dot(K)
        x = [x0]
        for _ in range(N):
            x.append(Acl.dot(x[-1]))
        return x

    def solve_dare(self, Q, R):
        """Cost-to-go Hessian P and LQR gain K (u = K x) of the infinite-horizon problem (:94-126)."""
        if not self.controllable:
            raise ValueError('uncontrollable system, cannot solve Riccati equation.')
        P = solve_discrete_are(self.A, self.B, Q, R)
        K = -np.linalg.solve(self.B.T.dot(P).dot(self.B) + R, self.B.T.dot(P).dot(self.A))
        return P, K

    def mcais(self, K, D, **kwargs):
        """Maximal constraint-admissible invariant set of x(t+1) = (A + B K) x(t) with (x, K x) in D (:128-158)."""
        X_cl = Polyhedron(D.A[:, :self.nx] + D.A[:, self.nx:].dot(K), D.b)
        return mcais(self.A + self.B.dot(K), X_cl, **kwargs)

    def condense(self, N):
        """A_bar, B_bar with x_bar = A_bar x(0) + B_bar u_bar (:160-184)."""
        S = AffineSystem(self.A, self.B, np.zeros(self.nx))
        A_bar, B_bar, _ = condense_pwa_system([S], [0] * N)
        return A_bar, B_bar

    @property
    def controllable(self):
        if self._controllable is None:
            blocks = [self.B]
            for _ in range(self.nx - 1):
                blocks.append(self.A.dot(blocks[-1]))
            self._controllable = bool(np.linalg.matrix_rank(np.hstack(blocks)) == self.nx)
        return self._controllable

    @staticmethod
    def from_continuous(A, B, h, method='zero_order_hold'):
        _check_affine_system(A, B, None, h)
        A_d, B_d, _ = _discretize(A, B, np.zeros(A.shape[0]), h, method)
        return LinearSystem(A_d, B_d)


class AffineSystem(object):
    """x(t+1) = A x(t) + B u(t) + c."""

    def __init__(self, A, B, c):
        _check_affine_system(A, B, c)
        self.A = A
        self.B = B
        self.c = c
        self.nx, self.nu = B.shape

    def simulate(self, x0, u):
        x = [x0]
        for v in u:
            x.append(self.A.dot(x[-1]) + self.B.dot(v) + self.c)
        return x

    def condense(self, N):
        return condense_pwa_system([self], [0] * N)

    @staticmethod
    def from_continuous(A, B, c, h, method='zero_order_hold'):
        return AffineSystem(*_discretize(A, B, c, h, method))


class PieceWiseAffineSystem(object):
    """x(t+1) = A_i x(t) + B_i u(t) + c_i  if  (x(t), u(t)) in D_i."""

    def __init__(self, affine_systems, domains):
        if len(affine_systems) != len(domains):
            raise ValueError('the number of affine systems has to be equal to the number of domains.')
        if len(set(S.nx for S in affine_systems)) != 1:
            raise ValueError('all the affine systems must have the same number of states.')
        if len(set(S.nu for S in affine_systems)) != 1:
            raise ValueError('all the affine systems must have the same number of inputs.')
        self.nx, self.nu = affine_systems[0].nx, affine_systems[0].nu
        dims = set(D.A.shape[1] for D in domains)
        if len(dims) != 1:
            raise ValueError('all the domains must have equal dimnesionality.')
        if dims.pop() != self.nx + self.nu:
            raise ValueError('the domains and the affine systems must have coherent dimensions.')
        for i, S in enumerate(affine_systems):   # linear systems become affine ones, in place like the reference
            if isinstance(S, LinearSystem):
                affine_systems[i] = AffineSystem(S.A, S.B, np.zeros(self.nx))
        self.affine_systems = affine_systems
        self.domains = domains
        self.nm = len(affine_systems)

    def condense(self, mode_sequence):
        return condense_pwa_system(self.affine_systems, mode_sequence)

    def get_mode(self, x, u):
        xu = np.concatenate((x, u))
        for i, D in enumerate(self.domains):
            if D.contains(xu):
                return i
        return None

    def simulate(self, x0, u):
        x = [x0]
        mode_sequence = []
        for t in range(len(u)):
            mode = self.get_mode(x[t], u[t])
            if mode is None:
                raise ValueError('simulation reached an unfeasible point x = ' + str(x[t]) + ', u = ' + str(u[t]) + '.')
            S = self.affine_systems[mode]
            x.append(S.A.dot(x[t]) + S.B.dot(u[t]) + S.c)
            mode_sequence.append(mode)
        return x, mode_sequence

    def is_well_posed(self, tol=1.e-7):
        """True iff the interiors of the domains are pairwise disjoint (Chebyshev radius of each intersection <= tol)."""
        for i in range(self.nm):
            for j in range(i + 1, self.nm):
                if self.domains[i].intersection(self.domains[j]).radius > tol:
                    return False
        return True


def mcais(A, X, verbose=False):
    """
    Maximal constraint-admissible invariant set of x(t+1) = A x(t), x in X (Gilbert-Tan Algorithm 3.2,
    discrete_time_systems.py:584-674).  The reference solves one LP per facet of X at every iteration
    (:646-648); all of them share the constraint set O_inf, so each iteration is ONE batched launch.

    Several iterations per launch: the rows X.A A^t, ..., X.A A^(t+k-1) are all tested against the set as it stands before
    iteration t.  That set has fewer rows than the ones the later iterations would see, so a row found redundant is
    redundant for the reference too, and a row found necessary is at worst a redundant extra that the final
    remove_redundant_inequalities takes out again; the first iteration of the block without a new row ends the
    algorithm as it does in the reference (whose own stop can only come earlier, after which every further row is
    redundant by invariance).  Same minimal set, a launch and a host round trip per k iterations instead of per iteration.
    """
    from ..optimization.solvers import hip_lp
    if np.max(np.absolute(np.linalg.eigvals(A))) > 1.:
        raise ValueError('unstable system, cannot derive maximal constraint-admissible set.')
    nx = X.A.shape[1]
    if not X.contains(np.zeros(nx)):
        raise ValueError('the origin is not contained in the constraint set, cannot derive maximal constraint-admissible set.')
    if not X.bounded:
        raise ValueError('unbounded constraint set, cannot derive maximal constraint-admissible set.')
    O_inf = copy(X)
    At = np.eye(nx)
    t = 1
    mX = X.A.shape[0]
    block = 2
    done = False
    while not done:
        Js = []
        for _ in range(block):
            At = At.dot(A)
            Js.append(X.A.dot(At))
        res = hip_lp.linear_program_batch(-np.vstack(Js), O_inf.A, O_inf.b, want_argmin=False, want_mult=False,
                                          want_active=False)
        hip_lp.retry_stalled(res, -np.vstack(Js), O_inf.A, O_inf.b)
        st = res.numpy('status')
        if np.any(st == 3):
            raise RuntimeError('Maximum number of iterations reached.')
        if np.any(st != 0):
            raise TypeError('an LP of the invariant-set iteration has no optimum (the reference fails on -None here).')
        vals = -res.numpy('min')
        for s, J in enumerate(Js):
            residuals = vals[s * mX:(s + 1) * mX] - X.b
            if verbose:
                print('Time horizon: %d. Convergence index: %g. Number of facets: %d.' % (t, residuals.max(), O_inf.A.shape[0]))
            new_facets = np.where(residuals > 0.)[0]
            if new_facets.size == 0:
                done = True
                break
            O_inf.add_inequality(J[new_facets], X.b[new_facets])
            t += 1
        block = min(2 * block, 16)
    O_inf.remove_redundant_inequalities()
    if verbose:
        print('minimal facets are ' + str(O_inf.A.shape[0]) + '.')
    return O_inf


def condense_pwa_system(affine_systems, mode_sequence):
    """A_bar, B_bar, c_bar with x_bar = A_bar x(0) + B_bar u_bar + c_bar (discrete_time_systems.py:676-729), computed by
    the condensing kernel (its prediction-matrix output)."""
    from ..control.condense import PwaTables, condense_batch
    nx, nu = affine_systems[0].nx, affine_systems[0].nu
    nothing = Polyhedron(np.zeros((0, nx + nu)), np.zeros(0))
    tables = PwaTables(affine_systems, [nothing] * len(affine_systems), np.zeros((nx, nx)), np.zeros((nu, nu)),
                       np.zeros((nx, nx)), Polyhedron(np.zeros((0, nx)), np.zeros(0)))
    batch = condense_batch(tables, np.asarray(mode_sequence, dtype=np.int32).reshape(1, -1), want_prediction=True)
    batch.check()
    return batch.prediction(0)
