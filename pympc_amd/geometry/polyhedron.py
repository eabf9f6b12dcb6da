"""H-representation polyhedra {x | A x <= b, C x = d}: the part of /root/reference/pympc/geometry/polyhedron.py that sits
on the pnnls hot path, with the per-facet LP loops issued as batches to the HIP LP kernel.

Same names, arguments and error behaviour as the reference for
    Polyhedron(A, b, C=None, d=None)                         polyhedron.py:20-61
    add_inequality / add_equality / add_*_bound(s)           :63-199      (host-side builders)
    from_lower_bound / from_upper_bound / from_bounds        :239-336
    normalize(tol)                                           :379-401
    minimal_facets(tol) / remove_redundant_inequalities()    :403-466     one LP per facet -> batched
    _remove_equalities()                                     :468-503
    empty                                                    :505-526     one QP
    bounded                                                  :528-576     one LP
    contains(x, tol)                                         :578-604     x 1-D (bool) or [B x n] (bool array, HIP)
    is_included_in(P2, tol)                                  :606-641     one LP per facet of P2 -> batched
    intersection / cartesian_product                         :643-685
    radius / center / _chebyshev()                           :687-768     one LP
    vertices                                                 :770-824     qhull on the host (HalfspaceIntersection)
    from_convex_hull(points)                                 :856-877     qhull on the host
    project_to(residual_dimensions)                          :826-854     convex-hull method, its LPs batched
                                                                          (geometry/orthogonal_projection.py)
Plotting (matplotlib glue) is outside the hot path.
"""
from copy import copy

import numpy as np

from ..optimization.programs import linear_program, quadratic_program
from ..optimization.solvers import hip_lp


def nullspace_basis(A):
    """Orthonormal basis of ker(A) from the SVD (geometry/utils.py:4-26)."""
    V = np.linalg.svd(A)[2].T
    return V[:, np.linalg.matrix_rank(A):]


def block_diag(*blocks):
    rows = sum(B.shape[0] for B in blocks)
    cols = sum(B.shape[1] for B in blocks)
    out = np.zeros((rows, cols))
    r = c = 0
    for B in blocks:
        out[r:r + B.shape[0], c:c + B.shape[1]] = B
        r += B.shape[0]
        c += B.shape[1]
    return out


class Polyhedron(object):
    """Polyhedron in the form {x in R^n | A x <= b, C x = d}."""

    def __init__(self, A, b, C=None, d=None):
        if len(b.shape) > 1:
            raise ValueError('b must be a one dimensional array.')
        self._same_number_rows(A, b)
        self.A = A
        self.b = b
        if (C is None) != (d is None):
            raise ValueError('missing C or d.')
        if C is None:
            self.C = np.zeros((0, A.shape[1]))
            self.d = np.zeros(0)
        else:
            if len(d.shape) > 1:
                raise ValueError('b must be a one dimensional array.')
            self._same_number_rows(C, d)
            self.C = C
            self.d = d
        self._delete_attributes()

    # ------------------------------------------------------------------ builders (host side)
    def _delete_attributes(self):
        self._empty = None
        self._bounded = None
        self._radius = None
        self._center = None
        self._vertices = None

    @staticmethod
    def _same_number_rows(E, f):
        if E.shape[0] != f.size:
            raise ValueError("incoherent size of the inputs.")

    def _selection_matrix(self, indices=None):
        """S such that S x = x[indices] (rows of the identity, in increasing index order)."""
        n = self.A.shape[1]
        if indices is None:
            return np.eye(n)
        keep = [i for i in range(n) if i in indices]
        return np.eye(n)[keep]

    def add_inequality(self, A, b, indices=None):
        self._same_number_rows(A, b)
        self._delete_attributes()
        S = self._selection_matrix(indices)
        self.A = np.vstack((self.A, A.dot(S)))
        self.b = np.concatenate((self.b, b))

    def add_equality(self, C, d, indices=None):
        self._same_number_rows(C, d)
        self._delete_attributes()
        S = self._selection_matrix(indices)
        self.C = np.vstack((self.C, C.dot(S)))
        self.d = np.concatenate((self.d, d))

    def add_lower_bound(self, x_min, indices=None):
        if isinstance(x_min, float):
            x_min = np.array([x_min])
        self.add_inequality(-self._selection_matrix(indices), -x_min)

    def add_upper_bound(self, x_max, indices=None):
        if isinstance(x_max, float):
            x_max = np.array([x_max])
        self.add_inequality(self._selection_matrix(indices), x_max)

    def add_bounds(self, x_min, x_max, indices=None):
        self.add_lower_bound(x_min, indices)
        self.add_upper_bound(x_max, indices)

    @staticmethod
    def _blank(size, indices, n):
        if indices is not None and n is None:
            raise ValueError("specify the length of x to instantiate the polyhedron.")
        n = size if n is None else n
        return Polyhedron(np.zeros((0, n)), np.zeros(0))

    @staticmethod
    def from_lower_bound(x_min, indices=None, n=None):
        p = Polyhedron._blank(x_min.size, indices, n)
        p.add_lower_bound(x_min, indices)
        return p

    @staticmethod
    def from_upper_bound(x_max, indices=None, n=None):
        p = Polyhedron._blank(x_max.size, indices, n)
        p.add_upper_bound(x_max, indices)
        return p

    @staticmethod
    def from_bounds(x_min, x_max, indices=None, n=None):
        if indices is not None and n is None:
            raise ValueError("specify the length of x to instantiate the polyhedron.")
        if x_min.size != x_max.size:
            raise ValueError("bounds must have the same size.")
        p = Polyhedron._blank(x_min.size, indices, n)
        p.add_bounds(x_min, x_max, indices)
        return p

    def normalize(self, tol=1.e-7):
        """Divides every row of [A b] and [C d] by the norm of its left-hand side (rows with norm <= tol untouched)."""
        for M, v in ((self.A, self.b), (self.C, self.d)):
            for i in range(M.shape[0]):
                r = np.linalg.norm(M[i])
                if r > tol:
                    M[i] = M[i] / r
                    v[i] = v[i] / r

    def intersection(self, P2):
        P3 = copy(P2)
        P3.add_inequality(self.A, self.b)
        P3.add_equality(self.C, self.d)
        return P3

    def cartesian_product(self, P2):
        return Polyhedron(block_diag(self.A, P2.A), np.concatenate((self.b, P2.b)),
                          block_diag(self.C, P2.C), np.concatenate((self.d, P2.d)))

    # ------------------------------------------------------------------ equality elimination
    def _remove_equalities(self):
        """x = N n + R r with N = null(C), R = null(N'), r = (C R)^-1 d  ->  {n | E n <= f} (polyhedron.py:468-503)."""
        N = nullspace_basis(self.C)
        if N.shape[1] == 0:
            raise ValueError('equality constraints C x = d do not have a nullspace.')
        if N.shape[1] != self.C.shape[1] - self.C.shape[0]:
            raise ValueError('equality constraints C x = d are linearly dependent.')
        R = nullspace_basis(N.T)
        E = self.A.dot(N)
        r = np.linalg.inv(self.C.dot(R)).dot(self.d)
        f = self.b - self.A.dot(R.dot(r))
        return E, f, N, R

    # ------------------------------------------------------------------ redundancy removal (batched LPs)
    def minimal_facets(self, tol=1.e-7):
        """
        Indices of a minimal set of inequalities (None if the polyhedron is empty), polyhedron.py:403-450.

        The reference tests facet i against the set from which the facets found redundant before i have already been
        removed (one LP at a time, :435-448).  The same list is obtained from two batched launches plus a short
        sequential tail:
          1. every facet against the FULL set (b_i relaxed by 1): a facet that is not redundant here is not redundant
             against any subset, so it is kept for sure; the others are candidates;
          2. every candidate against the surely-kept facets only: redundant there means redundant against any
             superset, so it is removed for sure;
          3. the candidates that survive 2 (duplicated or touching facets) are re-tested one by one, in index order,
             against exactly the set the reference would hold at that point.
        """
        if self.empty:
            return None
        if self.C.shape[0] != 0:
            E, f, _, _ = self._remove_equalities()
        else:
            E, f = self.A, self.b
        m = E.shape[0]
        if m == 0:
            return []
        E = np.ascontiguousarray(E, dtype=np.float64)
        f = np.ascontiguousarray(f, dtype=np.float64)

        def values(res, idx):
            st = res.numpy('status')
            mn = res.numpy('min')
            if np.any(st == 3):
                raise RuntimeError('Maximum number of iterations reached.')
            return np.where(st == 0, -mn - f[idx], np.inf)

        removed = np.zeros(m, bool)
        staged = self._minimal_facets_staged(E, f, tol, values) if m >= self.STAGED_MIN_FACETS else None
        if staged is not None:
            cand, sure, undecided = staged
        else:
            # 1. all facets against the full relaxed set
            res = hip_lp.linear_program_batch(-E, E, f, relax_idx=np.arange(m, dtype=np.int32), relax_val=1.,
                                              want_argmin=False, want_mult=False, want_active=False)
            hip_lp.retry_stalled(res, -E, E, f, relax_idx=np.arange(m, dtype=np.int32), relax_val=1.)
            v1 = values(res, np.arange(m))
            cand = np.where(v1 < tol)[0]
            sure = np.where(~(v1 < tol))[0]
            undecided = cand
            # 2. candidates against the surely kept facets
            if cand.size > 0 and sure.size > 0:
                res = hip_lp.linear_program_batch(-E[cand], E[sure], f[sure], want_argmin=False, want_mult=False,
                                                  want_active=False)
                hip_lp.retry_stalled(res, -E[cand], E[sure], f[sure])
                v2 = values(res, cand)
                gone = v2 < tol
                undecided = cand[~gone]
        # 3. the rest in index order, against the reference's current set
        pending = np.zeros(m, dtype=bool)                             # removed when their turn comes
        pending[cand] = True
        pending[undecided] = False
        for i in undecided:
            removed[:i] |= pending[:i]
            cur = np.where(~removed)[0]
            rel = f.copy()
            rel[i] += 1.
            sol = linear_program(-E[i], E[cur], rel[cur])
            if - sol['min'] - f[i] < tol:
                removed[i] = True
        removed |= pending
        return np.where(~removed)[0].tolist()

    STAGED_MIN_FACETS = 2000

    @staticmethod
    def _minimal_facets_staged(E, f, tol, values):
        """
        Steps 1 and 2 of minimal_facets for many facets, with the same outcome (cand, sure, undecided) at a fraction of the
        row sweeps.  Step 1 solves one LP per facet against ALL m rows although most facets are redundant against a
        small subset already:
          a. Z = the facets hit first by rays from an interior point along every facet normal (one matrix product; a
             guess only, nothing below relies on it);
          b. every facet outside Z against the rows of Z alone (no relaxation: its own row is not among them): redundant
             there (value < tol) means redundant against any superset, hence value < tol against the full relaxed set
             too: a candidate of step 1 -- and, if Z turns out to lie in `sure`, "gone" in step 2;
          c. only Z and the facets that b. did not settle get step 1's LP against the full relaxed set.
        If some member of Z is not surely kept (duplicated or touching facets: its own full test says so), b. proves
        nothing about step 2 and None is returned: the caller runs the plain steps.
        """
        import torch
        from .. import _hip
        m, n = E.shape
        if np.min(f) <= 1.e-9:            # (the origin must be strictly inside; callers with other data take the plain path)
            return None
        dev = _hip.device()
        E64 = _hip.to_device(E); f64 = _hip.to_device(f)               # (the LP batches below index these on the device)
        # (single precision: the rays only propose a subset)
        Ed = E64.to(torch.float32); fd = f64.to(torch.float32)
        Dn = (Ed / torch.linalg.norm(Ed, dim=1, keepdim=True)).T.contiguous()        # [n x m] ray directions
        hit = torch.empty(m, dtype=torch.int64, device=dev)
        for c0 in range(0, m, 4096):                                   # rays x = t a_i / |a_i|: first facet k with a_k x = f_k
            S = Ed.mm(Dn[:, c0:c0 + 4096])                             # a_k . d_i
            S.clamp_(min=1.e-12).reciprocal_().mul_(fd[:, None])       # t = f_k / (a_k . d_i); rays that never meet row k: huge
            hit[c0:c0 + 4096] = S.argmin(dim=0)
        inZ = torch.zeros(m, dtype=torch.bool, device=dev); inZ[hit] = True
        Zt = torch.nonzero(inZ).reshape(-1)
        rest_t = torch.nonzero(~inZ).reshape(-1)
        Z = Zt.cpu().numpy()
        if Z.size < n + 1 or Z.size > m // 2:
            return None
        rest = rest_t.cpu().numpy()
        res = hip_lp.linear_program_batch(-E64[rest_t], E64[Zt], f64[Zt], want_argmin=False, want_mult=False, want_active=False)
        settled = values(res, rest) < tol                              # redundant against Z alone
        full = np.concatenate((Z, rest[~settled]))
        full.sort()
        full_t = torch.as_tensor(full, device=dev)
        res = hip_lp.linear_program_batch(-E64[full_t], E64, f64, relax_idx=full_t.to(torch.int32), relax_val=1.,
                                          want_argmin=False, want_mult=False, want_active=False)
        v1 = values(res, full)
        sure = full[~(v1 < tol)]
        if not np.all(np.isin(Z, sure)):
            return None
        cand_full = full[v1 < tol]
        cand = np.sort(np.concatenate((rest[settled], cand_full)))
        undecided = cand_full
        if cand_full.size > 0 and sure.size > 0:                       # step 2 for the candidates b. did not settle
            res = hip_lp.linear_program_batch(-E[cand_full], E[sure], f[sure], want_argmin=False, want_mult=False,
                                              want_active=False)
            undecided = cand_full[~(values(res, cand_full) < tol)]
        return cand, sure, undecided

    def remove_redundant_inequalities(self):
        """Removes the redundant facets of the polyhedron, it modifies the attributes A and b (polyhedron.py:452-466)."""
        minimal_facets = self.minimal_facets()
        if minimal_facets is None:
            raise ValueError('empty polyhedron, cannot remove redundant inequalities.')
        self.A = self.A[minimal_facets]
        self.b = self.b[minimal_facets]

    # ------------------------------------------------------------------ emptiness / boundedness / membership
    @property
    def empty(self):
        """True iff the minimum-norm QP over the polyhedron is infeasible (polyhedron.py:505-526)."""
        if self._empty is not None:
            return self._empty
        n = self.A.shape[1]
        m_qp = self.A.shape[0] + 2 * self.C.shape[0]
        from .. import _hip
        n_max, m_max = _hip.limits()
        if m_qp == 0:
            self._empty = False
        elif _hip.load_library().pympc_qp_workspace_doubles(n, m_qp, 1) > 0:
            # (up to 63 variables and 2048 rows on the wave kernel, up to 1023 and 32768 on the workgroup kernels of
            # csrc/qp_block.hpp)
            sol = quadratic_program(np.eye(n), np.zeros(n), self.A, self.b, self.C, self.d)
            self._empty = sol['min'] is None
        elif self.C.shape[0] > 0 and n > min(n_max, _hip.lp_limits()[0]):
            # more variables than a wavefront holds: same decision in the null space of the equalities
            try:
                E, f, _, _ = self._remove_equalities()
            except ValueError:
                # no null space, or dependent equalities: C x = d has at most the least-squares point
                x = np.linalg.lstsq(self.C, self.d, rcond=None)[0]
                self._empty = not (np.allclose(self.C.dot(x), self.d, atol=1e-9) and np.all(self.A.dot(x) <= self.b + 1e-9))
            else:
                self._empty = Polyhedron(E, f).empty
        else:
            # more rows than the wave-per-QP kernel holds: same decision from the feasibility LP min 0'x
            sol = linear_program(np.zeros(n), self.A, self.b, self.C, self.d)
            self._empty = sol['min'] is None
        return self._empty

    @property
    def bounded(self):
        """
        Bounded iff ker(A) = 0 and the polyhedron has no recession direction (polyhedron.py:528-576).  The reference
        decides the second part with Stiemke's alternative, an LP with one variable per ROW of A (some y >= 1 with
        A'y = 0); that LP is used here while it fits the LP paths (up to 64 rows on the active-set kernel, up to 256 rows with
        1 + 2 rows + 2 dim <= 1024 in the reference's KKT form on the workgroup kernels).  Beyond that the same decision
        is taken in the primal space: a non-empty polyhedron is bounded iff the 2 dim LPs max +-x_i over it all have an
        optimum (one batched launch against the shared constraint matrix).
        """
        if self._bounded is not None:
            return self._bounded
        if self.empty:
            return True
        A = np.vstack((self.A, self.C, -self.C))
        if nullspace_basis(A).shape[1] > 0:
            return False
        n, m = A.shape
        from .. import _hip
        n_lp_max = _hip.lp_limits()[0]
        if n <= n_lp_max or (n <= 256 and 1 + 2 * n + 2 * m <= 1024):
            # (beyond the LP kernel's 64 variables linear_program solves the reference's KKT form on the workgroup kernels)
            sol = linear_program(np.ones(n), -np.eye(n), -np.ones(n), A.T, np.zeros(m))
            self._bounded = sol['min'] is not None
        elif m <= n_lp_max:
            b = np.concatenate((self.b, self.d, -self.d))
            F = np.vstack((np.eye(m), -np.eye(m)))
            res = hip_lp.linear_program_batch(F, A, b, want_argmin=False, want_mult=False, want_active=False)
            st = res.numpy('status')
            if np.any(st == 3):
                raise RuntimeError('Maximum number of iterations reached.')
            self._bounded = bool(np.all(st == 0))   # (non-empty: a failed LP is an unbounded direction)
        else:
            raise NotImplementedError('bounded: %d rows and %d variables are both beyond the LP kernel (%d variables)'
                                      % (n, m, n_lp_max))
        return self._bounded

    def contains(self, x, tol=1.e-7):
        """
        max(A x - b) <= tol and |max(C x - d)| <= tol (polyhedron.py:578-604).  x 1-D: bool.  x [B x n] (array or device
        tensor): bool array, evaluated by the HIP point-location kernel with this polyhedron as the only region.
        """
        import torch
        if isinstance(x, torch.Tensor) or np.ndim(x) == 2:
            return self._contains_batch(x, tol)
        in_ineq = np.max(self.A.dot(x) - self.b) <= tol
        in_eq = True
        if self.C.shape[0] > 0:
            in_eq = np.abs(np.max(self.C.dot(x) - self.d)) <= tol
        return bool(in_ineq and in_eq)

    def _contains_batch(self, X, tol):
        import torch
        from .. import _hip
        lib = _hip.load_library()
        X = _hip.to_device(X)
        B, p = X.shape

        def inside(A, b, t):
            """max(A x - b) <= t for every row x of X: the point-location kernel with one region."""
            m = A.shape[0]
            Ad = _hip.to_device(A); bd = _hip.to_device(b)
            row_ptr = _hip.to_device(np.array([0, m], np.int32), dtype=torch.int32)
            z = torch.zeros(max(p * p, 1), dtype=torch.float64, device=X.device)
            idx = torch.empty(B, dtype=torch.int32, device=X.device)
            V = torch.empty(B, dtype=torch.float64, device=X.device)
            _hip.check(lib.pympc_point_location(1, m, p, 0, _hip.ptr(Ad), _hip.ptr(bd), _hip.ptr(row_ptr), None, None,
                                                _hip.ptr(z), _hip.ptr(z), _hip.ptr(z), None, None, 0, None, None, None, 1, _hip.ptr(X), B,
                                                float(t), _hip.ptr(idx), None, _hip.ptr(V), _hip.stream_ptr()))
            return idx == 0

        if self.C.shape[0] == 0:
            return inside(self.A, self.b, tol).cpu().numpy()
        # the reference's equality test is |max(C x - d)| <= tol (polyhedron.py:598-602): max <= tol, and not max < -tol
        upper = inside(np.vstack((self.A, self.C)), np.concatenate((self.b, self.d)), tol)
        below = inside(self.C, self.d, np.nextafter(-tol, -np.inf))
        return (upper & ~below).cpu().numpy()

    def is_included_in(self, P2, tol=1.e-7):
        """P subset of P2?  One LP per facet of P2 (polyhedron.py:606-641), issued as one batch."""
        A1 = np.vstack((self.A, self.C, -self.C))
        b1 = np.concatenate((self.b, self.d, -self.d))
        A2 = np.vstack((P2.A, P2.C, -P2.C))
        b2 = np.concatenate((P2.b, P2.d, -P2.d))
        if A2.shape[0] == 0:
            return True
        res = hip_lp.linear_program_batch(-A2, A1, b1, want_argmin=False, want_mult=False, want_active=False)
        st = res.numpy('status')
        if np.any(st == 3):
            raise RuntimeError('Maximum number of iterations reached.')
        if np.any(st != 0):
            # an LP without optimum (P unbounded along a facet normal of P2, or P empty): the reference evaluates
            # -sol['min'] with sol['min'] None at polyhedron.py:636 and raises TypeError; same error here
            raise TypeError("bad operand type for unary -: 'NoneType' (a facet LP of is_included_in has no optimum)")
        penetration = - res.numpy('min') - b2
        return bool(not np.any(penetration > tol))

    # ------------------------------------------------------------------ Chebyshev centre
    @property
    def radius(self):
        if self._radius is not None:
            return self._radius
        self._radius, self._center = self._chebyshev()
        return self._radius

    @property
    def center(self):
        if self._center is not None:
            return self._center
        self._radius, self._center = self._chebyshev()
        return self._center

    def _chebyshev(self):
        """LP  min e  s.t.  F z - ||F_i|| e <= g  (polyhedron.py:727-768); radius = -min, centre = z."""
        if self.C.shape[0] > 0:
            A, b, N, R = self._remove_equalities()
        else:
            A, b = self.A, self.b
        f_lp = np.concatenate((np.zeros(A.shape[1]), np.ones(1)))
        A_lp = np.hstack((A, -np.linalg.norm(A, axis=1).reshape(-1, 1)))
        sol = linear_program(f_lp, A_lp, b)
        radius = sol['min']
        center = sol['argmin']
        if radius is not None:
            radius = -radius
            center = center[:-1]
            if self.C.shape[0] > 0:
                r = np.linalg.inv(self.C.dot(R)).dot(self.d)
                center = np.hstack((N, R)).dot(np.concatenate((center, r)))
        return radius, center

    # ------------------------------------------------------------------ vertices / convex hull / projection
    @property
    def vertices(self):
        """
        Vertices of a bounded, full-dimensional polytope as a list of arrays, None if unbounded or (nearly) flat
        (polyhedron.py:770-824): boundedness and the Chebyshev centre come from the LP kernel, the enumeration itself is
        qhull's half-space intersection on the host, as in the reference.
        """
        if self._vertices is not None:
            return self._vertices
        if not self.bounded:
            return None
        if self.radius < 1.e-7:
            return None
        from scipy.spatial import HalfspaceIntersection
        if self.C.shape[0] > 0:
            A, b, N, R = self._remove_equalities()
            T = np.hstack((N, R))
            center = np.linalg.inv(T).dot(self.center)[:N.shape[1]]
        else:
            A, b, center = self.A, self.b, self.center
        if A.shape[1] == 1:
            p = Polyhedron(A.copy(), b.copy())
            p.remove_redundant_inequalities()
            verts = [np.array([p.b[i] / p.A[i, 0]]) for i in (0, 1)]
        else:
            V = HalfspaceIntersection(np.column_stack((A, -b)), center).intersections
            verts = [V[i] for i in range(V.shape[0])]
        if self.C.shape[0] > 0:
            r = np.linalg.inv(self.C.dot(R)).dot(self.d)
            verts = [T.dot(np.concatenate((v, r))) for v in verts]
        self._vertices = verts
        return self._vertices

    @staticmethod
    def from_convex_hull(points):
        """Polytope {x | A x <= b} spanned by the given points (polyhedron.py:856-877, qhull on the host)."""
        from scipy.spatial import ConvexHull
        hull = ConvexHull(np.vstack(points))
        p = Polyhedron(hull.equations[:, :-1].copy(), -hull.equations[:, -1].copy())
        p._vertices = [hull.points[i] for i in hull.vertices]
        return p

    def project_to(self, residual_dimensions):
        """
        Orthogonal projection onto the given coordinates (polyhedron.py:826-854): ValueError for empty, unbounded or
        lower-dimensional polyhedra; the convex-hull method with one batched LP launch per round.
        """
        if self.empty:
            raise ValueError('cannot project empty polyhedra.')
        if not self.bounded:
            raise ValueError('cannot project unbounded polyhedra.')
        if self.C.shape[0] > 0:
            raise ValueError('cannot project lower-dimensional polyhedra.')
        from .orthogonal_projection import convex_hull_method
        A, b, vertices = convex_hull_method(self.A, self.b, residual_dimensions)
        proj = Polyhedron(A, b)
        proj._vertices = [v for v in vertices]
        return proj
