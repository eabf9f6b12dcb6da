"""mpQP object and explicit solution, mirroring /root/reference/pympc/optimization/parametric_programs.py.

    MultiParametricQuadraticProgram(H, f, g, A, b)          :10-40
        .solve(x)                                           :155-179   x 1-D -> sol dict; x 2-D [B x p] -> BatchResult
        .explicit_solve_given_active_set(active_set)        :42-116
    CriticalRegion(active_set, u, p, V, polyhedron)         :262-424
    ExplicitSolution(critical_regions)                      :426-532   get_critical_region / u / p / V, scalar or batched

The numerical work of solve() and of the batched look-ups runs in the HIP kernels (one launch per batch).
"""
from copy import copy

import numpy as np
import torch

from .. import _hip
from .solvers import hip as hip_backend


class MultiParametricQuadraticProgram(object):
    """
    mpQP in the form
                      |u|' |Huu  Hux| |u|   |fu|' |u|
    V(x) := min_u 1/2 |x|  |Hux' Hxx| |x| + |fx|  |x| + g
             s.t. Au u + Ax x <= b
    """

    def __init__(self, H, f, g, A, b):
        self.H = H
        self.Huu_inv = np.linalg.inv(self.H['uu'])
        self.f = f
        self.g = g
        self.A = A
        self.b = b
        self._prepared = None

    # ------------------------------------------------------------------ device side
    def prepared(self):
        """Factorise the constant part on the device once (Cholesky of Huu, M, affine maps)."""
        if self._prepared is None:
            self._prepared = hip_backend.PreparedMpqp(self.H['uu'], self.H['ux'], self.H['xx'], self.f['u'],
                                                      self.f['x'], float(self.g), self.A['u'], self.A['x'], self.b)
        return self._prepared

    def solve(self, x, **kwargs):
        """
        Solves the QP at the point x (1-D array: returns the reference's sol dict, parametric_programs.py:155-179)
        or at every row of x (2-D array / device tensor [B x p]: returns a hip.BatchResult).
        """
        if isinstance(x, torch.Tensor):
            batched = x.dim() == 2
        else:
            x = np.asarray(x, dtype=np.float64)
            batched = x.ndim == 2
        if batched:
            return self.prepared().solve_batch(x, **kwargs)
        x = np.asarray(x, dtype=np.float64).reshape(1, -1)
        return self.prepared().solve_batch(x).sol(0)

    # ------------------------------------------------------------------ explicit solution (host formulas)
    def explicit_solve_given_active_set(self, active_set):
        """
        KKT solution for a fixed active set (parametric_programs.py:42-116): u(x) = ux x + u0, p(x) = px x + p0,
        V(x) = .5 x'Vxx x + Vx'x + V0 and the polyhedron where the active set is optimal.  Returns None when the
        active rows of Au are linearly dependent (LICQ).
        """
        from ..geometry.polyhedron import Polyhedron
        active_set = [int(i) for i in active_set]
        Au, Ax, b = self.A['u'], self.A['x'], self.b
        Huu, Hux, Hxx = self.H['uu'], self.H['ux'], self.H['xx']
        fu, fx = self.f['u'], self.f['x']
        Hi = self.Huu_inv
        m = Ax.shape[0]
        Aua = Au[active_set]
        if len(active_set) > 0 and np.linalg.matrix_rank(Aua) < Aua.shape[0]:
            return None
        inactive = np.setdiff1d(np.arange(m), active_set)
        # multipliers of the active rows
        AH = Aua.dot(Hi)
        S = np.linalg.inv(AH.dot(Aua.T))
        pax = S.dot(Ax[active_set] - AH.dot(Hux))
        pa0 = -S.dot(b[active_set] + AH.dot(fu))
        px = np.zeros(Ax.shape); p0 = np.zeros(m)
        px[active_set] = pax; p0[active_set] = pa0
        # primal law
        ux = -Hi.dot(Hux + Aua.T.dot(pax))
        u0 = -Hi.dot(fu + Aua.T.dot(pa0))
        # region: multipliers of active rows >= 0, inactive rows feasible
        Acr = np.vstack((-pax, Au[inactive].dot(ux) + Ax[inactive]))
        bcr = np.concatenate((pa0, b[inactive] - Au[inactive].dot(u0)))
        cr = Polyhedron(Acr, bcr)
        cr.normalize()
        # value function
        Vxx = ux.T.dot(Huu).dot(ux) + 2. * Hux.T.dot(ux) + Hxx
        Vx = (ux.T.dot(Huu.T) + Hux.T).dot(u0) + ux.T.dot(fu) + fx
        V0 = .5 * u0.dot(Huu).dot(u0) + fu.dot(u0) + self.g
        return CriticalRegion(active_set, {'x': ux, '0': u0}, {'x': px, '0': p0}, {'xx': Vxx, 'x': Vx, '0': V0}, cr)

    def explicit_solve_given_point(self, x, active_set_guess=None, verbose=False):
        """
        Critical region that covers x (None if x is infeasible), parametric_programs.py:118-153: try the guessed
        active set first, otherwise solve the QP at x for the active set.
        """
        if active_set_guess is not None:
            cr = self.explicit_solve_given_active_set(active_set_guess)
            if cr is not None and cr.contains(x):
                return cr
            if verbose:
                print('Wrong active-set guess:')
        sol = self.solve(x)
        if sol['active_set'] is None:
            if verbose:
                print('unfeasible sample.')
            return None
        if verbose:
            print('feasible sample with active set ' + str(sol['active_set']) + '.')
        return self.explicit_solve_given_active_set(sol['active_set'])

    def get_feasible_set(self):
        """Feasible set {x | exists u: Au u + Ax x <= b} of the mpQP as a Polyhedron (parametric_programs.py:243-260):
        projection of the constraint set onto the parameters (convex-hull method, batched LPs)."""
        from ..geometry.polyhedron import Polyhedron
        C = Polyhedron(np.hstack((self.A['x'], self.A['u'])), self.b)
        return C.project_to(range(self.A['x'].shape[1]))

    def explicit_solve(self, step_size=1.e-5, verbose=False):
        """
        Explicit solution by facet flipping (parametric_programs.py:181-241): start at the origin, and for every
        critical region found step across each of its minimal facets (from the Chebyshev centre of the facet) with the
        active set guessed by toggling that facet's constraint.  Regions are listed in the order of discovery, which
        is the order the look-up scans.  Every LP / QP inside (redundancy removal, facet centres, active sets) runs
        on the HIP solvers; the exploration itself is a sequential host loop, as in the reference.
        """
        x_buffer = [(np.zeros(self.f['x'].size), [])]
        crs_found = []
        # The reference drops, before every step, the buffered points that a region found so far contains
        # (parametric_programs.py:213-216: one `contains` per point and region, every time).  The set of regions only
        # grows, so testing a point once, when its turn comes, against all regions found until then processes the same
        # points in the same order; the rows of all regions are stacked so that the test is one matrix-vector product
        # and a segmented max (same expression as Polyhedron.contains: max(A x - b) <= 1e-7).
        p = self.f['x'].size
        A_all, b_all, seg = np.zeros((0, p)), np.zeros(0), []

        others = []   # regions that do not stack (equalities, no rows): tested one by one

        def covered(x):
            if any(cr.contains(x) for cr in others):
                return True
            if not seg:
                return False
            return bool((np.maximum.reduceat(A_all.dot(x) - b_all, seg) <= 1.e-7).any())

        while len(x_buffer) > 0:
            x, guess = x_buffer.pop(0)
            if covered(x):
                continue
            cr = self.explicit_solve_given_point(x, guess, verbose)
            if cr is None:
                continue
            facets = cr.minimal_facets()
            for i, c in zip(facets, cr.facet_centers(facets)):   # one batched launch for all facet centres of the region
                step = c + step_size * cr.A[i]
                flipped = set(cr.active_set).symmetric_difference({i})
                x_buffer.append((step, sorted(flipped)))
            crs_found.append(cr)
            if cr.polyhedron.C.shape[0] == 0 and cr.polyhedron.A.shape[0] > 0:
                seg.append(A_all.shape[0])
                A_all = np.vstack((A_all, cr.polyhedron.A))
                b_all = np.concatenate((b_all, cr.polyhedron.b))
            else:
                others.append(cr)
            if verbose:
                print('CR found, active set: ' + str(cr.active_set) + '.')
        if verbose:
            print('Explicit solution found, CRs are: ' + str(len(crs_found)) + '.')
        return ExplicitSolution(crs_found)

    def explicit_solve_from_samples(self, X, max_regions=None):
        """
        Builds an ExplicitSolution from the active sets the batched QP path finds at the sample points X [B x p]
        (one launch), one CriticalRegion per distinct active set, in order of first appearance.  This is the
        table-construction route used for the explicit-MPC benchmark: the reference's facet-flipping
        exploration (parametric_programs.py:181-241) needs thousands of sequential LP/QP calls and is a
        later row of the scope table.
        """
        res = self.solve(np.asarray(X, dtype=np.float64))
        mask = res.active_mask()
        feas = res.feasible
        seen = {}
        for i in np.where(feas)[0]:
            key = mask[i].tobytes()
            if key not in seen:
                seen[key] = np.where(mask[i])[0]
                if max_regions is not None and len(seen) >= max_regions:
                    break
        crs = []
        for act in seen.values():
            cr = self.explicit_solve_given_active_set(list(act))
            if cr is not None:
                crs.append(cr)
        return ExplicitSolution(crs)


class CriticalRegion(object):
    """Region of the parameter space where a fixed active set is optimal (parametric_programs.py:262-424)."""

    def __init__(self, active_set, u, p, V, polyhedron):
        self.active_set = active_set
        self._u = u
        self._p = p
        self._V = V
        self.polyhedron = polyhedron

    def contains(self, x):
        return self.polyhedron.contains(x)

    def minimal_facets(self):
        return self.polyhedron.minimal_facets()

    def facet_center(self, i):
        """Chebyshev centre of the i-th facet: the region with row i added as an equality (parametric_programs.py:320-347)."""
        P = self.polyhedron
        if P.A.shape[1] == 1:
            return P.b.flatten()[i] / P.A[i][0]
        facet = copy(P)
        facet.add_equality(facet.A[i:i + 1, :], facet.b[i:i + 1])
        return facet.center

    def facet_centers(self, indices):
        """
        Chebyshev centres of several facets at once: the LPs of facet_center(i), i in indices, have one shape
        (rows x p) and go out as ONE batched launch.  Returns a list of points (None where the LP has no optimum).
        """
        from ..geometry.polyhedron import nullspace_basis
        from .solvers import hip_lp
        P = self.polyhedron
        indices = [int(i) for i in indices]
        p = P.A.shape[1]
        if p == 1 or P.C.shape[0] > 0 or len(indices) == 0:
            return [self.facet_center(i) for i in indices]
        A, b = P.A, P.b
        As, bs, maps = [], [], []
        for i in indices:                       # x = N n + R r on the hyperplane a_i x = b_i (polyhedron.py:468-503)
            C = A[i:i + 1]
            N = nullspace_basis(C)
            R = nullspace_basis(N.T)
            r = np.linalg.inv(C.dot(R)).dot(b[i:i + 1])
            E = A.dot(N)
            As.append(np.hstack((E, -np.linalg.norm(E, axis=1).reshape(-1, 1))))
            bs.append(b - A.dot(R.dot(r)))
            maps.append((N, R, r))
        f = np.zeros((len(indices), p)); f[:, -1] = 1.
        res = hip_lp.linear_program_batch(f, np.array(As), np.array(bs), want_mult=False, want_active=False)
        st = res.numpy('status')
        if np.any(st == 3):
            raise RuntimeError('Maximum number of iterations reached.')
        z = res.numpy('argmin')
        out = []
        for k, (N, R, r) in enumerate(maps):
            out.append(None if st[k] != 0 else np.hstack((N, R)).dot(np.concatenate((z[k, :-1], r))))
        return out

    def u(self, x):
        return self._u['x'].dot(x) + self._u['0']

    def p(self, x):
        return self._p['x'].dot(x) + self._p['0']

    def V(self, x):
        return .5 * x.dot(self._V['xx']).dot(x) + self._V['x'].dot(x) + self._V['0']

    @property
    def A(self):
        return self.polyhedron.A

    @property
    def b(self):
        return self.polyhedron.b


class ExplicitSolution(object):
    """
    Explicit solution of an mpQP: list of critical regions scanned in order, first hit wins
    (parametric_programs.py:426-532).  Every look-up accepts one point (1-D, reference semantics: None when
    uncovered) or a batch [B x p] (arrays; index -1 / NaN rows when uncovered); batches run in the HIP
    point-location kernel with the table resident in LDS.
    """

    def __init__(self, critical_regions):
        self.critical_regions = critical_regions
        self._table = None

    # ------------------------------------------------------------------ device table
    def table(self, nu=None):
        """Flattens the regions into the arrays the kernel takes (cached per nu)."""
        crs = self.critical_regions
        nu_all = crs[0]._u['x'].shape[0] if crs else 0
        nu = nu_all if nu is None else nu
        if self._table is not None and self._table['nu'] == nu:
            return self._table
        if self._table is not None:
            # another number of inputs (feedback vs feedforward): the regions, their boxes and the cell grid do not
            # depend on it -- only the laws' arrays are rebuilt
            t = dict(self._table)
            t['nu'] = nu
            p, n_cr = t['p'], len(crs)
            t['ux'] = np.array([cr._u['x'][:nu] for cr in crs]).reshape(n_cr, nu, p)
            t['u0'] = np.array([cr._u['0'][:nu] for cr in crs]).reshape(n_cr, nu)
            t['dev'] = dict(self._table['dev'])
            t['dev']['ux'] = _hip.to_device(t['ux'])
            t['dev']['u0'] = _hip.to_device(t['u0'])
            self._table = t
            return t
        p = crs[0].polyhedron.A.shape[1]
        for cr in crs:
            if cr.polyhedron.C.shape[0] != 0:
                raise NotImplementedError('critical regions with equalities')
        row_ptr = np.zeros(len(crs) + 1, np.int32)
        row_ptr[1:] = np.cumsum([cr.polyhedron.A.shape[0] for cr in crs])
        t = dict(nu=nu, p=p, nreg=len(crs), total_rows=int(row_ptr[-1]),
                 A=np.vstack([cr.polyhedron.A for cr in crs]), b=np.concatenate([cr.polyhedron.b for cr in crs]),
                 row_ptr=row_ptr,
                 ux=np.array([cr._u['x'][:nu] for cr in crs]).reshape(len(crs), nu, p),
                 u0=np.array([cr._u['0'][:nu] for cr in crs]).reshape(len(crs), nu),
                 Vxx=np.array([cr._V['xx'] for cr in crs]), Vx=np.array([cr._V['x'] for cr in crs]),
                 V0=np.array([cr._V['0'] for cr in crs], dtype=np.float64))
        t['box'] = self._bounding_boxes(t, tol=1.e-7)
        t['dev'] = {k: _hip.to_device(t[k]) for k in ('A', 'b', 'ux', 'u0', 'Vxx', 'Vx', 'V0', 'box')}
        t['dev']['row_ptr'] = _hip.to_device(row_ptr, dtype=torch.int32)
        t['grid'] = self._cell_grid(t, tol=1.e-7)
        if t['grid'] is not None:
            t['dev']['grid_geom'] = _hip.to_device(t['grid']['geom'])
            t['dev']['grid_masks'] = t['grid']['masks']
            t['dev']['grid_sure'] = t['grid']['sure']
            t['dev']['grid_lists'] = t['grid']['lists']
        self._table = t
        return t

    @staticmethod
    def _bounding_boxes(t, tol):
        """
        [nreg x 2p] axis-aligned boxes (lower bounds, then upper bounds) that contain every point the kernel's row test
        max(A x - b) <= tol can accept in a region: 2 p LPs per region over {A x <= b + tol}, one batched launch for
        the whole table, inflated by 1e-9 (LP accuracy and the rounding of the row test are orders below that).
        Unbounded or empty directions get an infinite bound, i.e. no pruning.  The boxes only let the kernel skip
        regions; the first-hit semantics of parametric_programs.py:442-463 is untouched.
        """
        from .solvers import hip_lp
        nreg, p, rp = t['nreg'], t['p'], t['row_ptr']
        box = np.empty((nreg, 2 * p))
        box[:, :p] = -np.inf
        box[:, p:] = np.inf
        if nreg == 0:
            return box
        rows = np.diff(rp)
        mmax = int(rows.max())
        n_lp, m_lp = hip_lp._hip.lp_limits()
        if mmax == 0 or p > n_lp or mmax > m_lp:
            return box
        A = np.zeros((nreg, mmax, p)); b = np.ones((nreg, mmax))          # pad rows 0 x <= 1 never bind
        for r in range(nreg):
            A[r, :rows[r]] = t['A'][rp[r]:rp[r + 1]]
            b[r, :rows[r]] = t['b'][rp[r]:rp[r + 1]] + tol + 1.e-9
        f = np.vstack((np.eye(p), -np.eye(p)))                             # min x_i, then min -x_i
        # (one launch takes at most 65535 distinct matrices: larger tables go in chunks of whole regions)
        per = max(1, 65535 // (2 * p))
        st = np.empty((nreg, 2 * p), np.int32)
        v = np.empty((nreg, 2 * p))
        for r0 in range(0, nreg, per):
            r1 = min(nreg, r0 + per)
            res = hip_lp.linear_program_batch(np.tile(f, (r1 - r0, 1)), np.repeat(A[r0:r1], 2 * p, axis=0),
                                              np.repeat(b[r0:r1], 2 * p, axis=0), want_argmin=False, want_mult=False,
                                              want_active=False)
            st[r0:r1] = res.numpy('status').reshape(r1 - r0, 2 * p)
            v[r0:r1] = res.numpy('min').reshape(r1 - r0, 2 * p)
        lo = np.where(st[:, :p] == 0, v[:, :p], -np.inf)
        hi = np.where(st[:, p:] == 0, -v[:, p:], np.inf)
        pad = 1.e-9
        box[:, :p] = lo - pad * (1. + np.abs(lo))
        box[:, p:] = hi + pad * (1. + np.abs(hi))
        return box

    GRID_CELLS_PER_AXIS = {1: 4096, 2: 512, 3: 64, 4: 24}
    SURE_MIN_FRACTION = 0.75   # cells decided or listed, from which the per-cell pass runs ahead of the scan

    @classmethod
    def _cell_grid(cls, t, tol):
        """
        Uniform grid over the union of the region boxes with, per cell,
        masks: the bit set of regions that can contain a point of the cell: region r is dropped from a cell only when one
          of its rows a'x <= b + tol is violated on the whole (slightly inflated) cell, i.e. min over the cell of
          a'x = sum_q min(a_q lo_q, a_q hi_q) > b + tol.  Critical regions are mostly thin slabs: their boxes overlap
          everywhere, this test does not;
        sure: the region of EVERY point of the cell, where the cell alone decides it: r when the whole inflated cell
          satisfies every row of r with a margin (max over the cell of a'x <= b - margin, so for any tol >= 0 of the
          kernel's row test) and r is the lowest bit of the cell's set (first hit, parametric_programs.py:442-463: no
          earlier region can contain a point of the cell); -1 when the set is empty; -2 otherwise.
        Built on the device with torch (rows x cells comparisons: 0.2 G of them for the pendulum table).  None when the
        parameter space has more than 4 dimensions or a region is unbounded.
        """
        nreg, p, rp = t['nreg'], t['p'], t['row_ptr']
        G = cls.GRID_CELLS_PER_AXIS.get(p)
        box = t['box']
        if G is None or nreg == 0:
            return None
        # regions whose box is not finite (an unbounded region, or a sliver the box LPs call infeasible): the grid spans
        # the others; such a region keeps its place in the bit sets of the BORDER cells whatever its rows say (points
        # outside the grid are looked up there), no border cell is decided, and no point counts as outside
        bounded = np.all(np.isfinite(box), axis=1)
        loose = np.where(~bounded)[0]
        if not bounded.any() or loose.size > 8:
            return None
        lo = box[bounded, :p].min(axis=0)
        hi = box[bounded, p:].max(axis=0)
        span = hi - lo
        if not np.all(span > 0):
            return None
        h = span / G
        pad = 1.e-9 * (1. + np.abs(lo) + np.abs(hi))          # rounding of the kernel's cell index, LP accuracy of the boxes
        dev = _hip.device()
        c = np.arange(G)
        # per axis q: lower and upper end of the (inflated) cells c_q = 0 .. G-1
        clo = torch.as_tensor(lo[None, :] + c[:, None] * h[None, :] - pad[None, :], device=dev)       # [G x p]
        chi = torch.as_tensor(lo[None, :] + (c[:, None] + 1) * h[None, :] + pad[None, :], device=dev)
        cabs = torch.maximum(clo.abs(), chi.abs())
        # the kernel clamps points outside the grid into the border cells: those cells extend to the region boxes' hull,
        # and no region accepts a point outside its own box, so the clamped look-up of the bit sets stays exact (the
        # decided cells are only consulted for points inside the grid)
        ncell = G ** p
        words = (nreg + 63) // 64
        masks = torch.zeros((ncell, words), dtype=torch.int64, device=dev)
        first = torch.full((ncell,), -1, dtype=torch.int32, device=dev)
        sure = torch.full((ncell,), -2, dtype=torch.int32, device=dev)
        Ad, bd = t['dev']['A'], t['dev']['b']
        shape = lambda q: [-1] + [G if k == q else 1 for k in range(p)]
        for r in range(nreg):
            A = Ad[rp[r]:rp[r + 1]]                                  # [rows x p]
            b = bd[rp[r]:rp[r + 1]]
            amin = amax = mag = None
            for q in range(p):
                a = A[:, q:q + 1]                                   # [rows x 1] against [1 x G]
                lo_q, hi_q = a * clo[None, :, q], a * chi[None, :, q]
                tmin = torch.minimum(lo_q, hi_q).reshape(shape(q))
                tmax = torch.maximum(lo_q, hi_q).reshape(shape(q))
                tmag = (a.abs() * cabs[None, :, q]).reshape(shape(q))
                amin = tmin if amin is None else amin + tmin
                amax = tmax if amax is None else amax + tmax
                mag = tmag if mag is None else mag + tmag
            bb = b.reshape([-1] + [1] * p)
            ok = (amin <= bb + (tol + 1.e-9)).all(dim=0).reshape(-1)                 # region r can touch the cell
            full = (amax <= bb - 1.e-10 * (mag + bb.abs())).all(dim=0).reshape(-1)   # the cell lies inside region r
            masks[:, r // 64] |= ok.to(torch.int64) << (r % 64)
            lowest = ok & (first < 0)
            first = torch.where(lowest, torch.full_like(first, r), first)
            sure = torch.where(lowest & full, torch.full_like(sure, r), sure)
        sure = torch.where(first < 0, torch.full_like(sure, -1), sure)
        frozen = None
        inside_lo, inside_hi = lo, hi
        if loose.size:
            ax = torch.arange(G, device=dev)
            edge1 = (ax == 0) | (ax == G - 1)
            frozen = torch.zeros([G] * p, dtype=torch.bool, device=dev)
            for q in range(p):
                frozen |= edge1.reshape([G if k == q else 1 for k in range(p)])
            frozen = frozen.reshape(-1)
            for r in loose:
                masks[:, r // 64] |= frozen.to(torch.int64) << int(r % 64)
            sure = torch.where(frozen, torch.full_like(sure, -2), sure)
            inside_lo, inside_hi = np.full(p, -np.inf), np.full(p, np.inf)
        decided = float((sure > -2).double().mean().item())
        lists, sure, listed = cls._cell_lists(t, G, clo, chi, cabs, sure, tol, frozen)
        return dict(n=G, geom=np.concatenate((lo, 1. / h, inside_hi, inside_lo)), masks=masks, sure=sure, lists=lists,
                    decided_fraction=decided, listed_fraction=listed, use_sure=(decided + listed) >= cls.SURE_MIN_FRACTION,
                    needs_scan=bool((sure == -2).any().item()))

    @staticmethod
    def _assemble_lists(cand_u, cand_r, cand_k, row_i, ncell, cap):
        """
        The int32 pool of the cells' decision lists from the candidates sorted by (cell, region): cand_u / cand_r / cand_k
        = cell, region and number of crossing rows of every candidate, row_i = the rows of all candidates in that order.
        A cell's list ends at its first candidate without a crossing row; lists longer than `cap` ints are dropped (code -2).
        Returns (pool with a dummy entry 0, codes [ncell]: -3 - offset of the cell's list).  numpy only (tests/test_abi.py).
        """
        ncand = cand_u.size
        cstart = np.searchsorted(cand_u, np.arange(ncell + 1))
        rstart = np.concatenate(([0], np.cumsum(cand_k)))            # rows of candidate j: row_i[rstart[j]:rstart[j + 1]]
        zero = (cand_k == 0).astype(np.int64)
        cz = np.cumsum(zero) - zero                                   # candidates without rows before j (all cells)
        first = np.minimum(cstart[:-1], max(ncand - 1, 0))
        keep = (cz - cz[first][cand_u]) == 0
        ints = np.where(keep, 2 + cand_k, 0)
        ci = np.cumsum(ints) - ints
        nonempty = cstart[1:] > cstart[:-1]
        cell_ints = 1 + np.add.reduceat(np.concatenate((ints, [0])), cstart[:-1]) * nonempty
        cell_ok = (cell_ints <= cap) & nonempty
        used = np.where(cell_ok, cell_ints, 0)
        cell_off = 1 + np.cumsum(used) - used                         # pool[0] is a dummy
        pool = np.zeros(1 + int(used.sum()), np.int32)
        kc = keep & cell_ok[cand_u]
        pos = cell_off[cand_u] + 1 + (ci - ci[first][cand_u])
        pool[cell_off[cell_ok]] = np.add.reduceat(np.concatenate((keep.astype(np.int64), [0])), cstart[:-1])[cell_ok]
        pool[pos[kc]] = cand_r[kc]
        pool[pos[kc] + 1] = cand_k[kc]
        rcand = np.repeat(np.arange(ncand), cand_k)                   # candidate of every row entry (same order as row_i)
        rk = kc[rcand]
        pool[(pos[rcand] + 2 + (np.arange(rcand.size) - rstart[rcand]))[rk]] = row_i[rk]
        return pool, np.where(cell_ok, -3 - cell_off, -2).astype(np.int64)

    LIST_MAX_INTS = 48        # longest decision list of a cell (beyond it the cell is left to the scan)

    @classmethod
    def _cell_lists(cls, t, G, clo, chi, cabs, sure, tol, frozen=None):
        """
        Decision lists for the cells no single region decides (those a region boundary crosses).  Over one small cell
        nearly every row of a candidate region holds on the whole cell (max over the cell of a'x <= b - margin): only the
        few rows that CROSS the cell can tell its points apart.  Per undecided cell, in region order (first hit,
        parametric_programs.py:442-463), the candidate regions (those no row excludes from the cell, as in the bit sets)
        with their crossing rows; the list ends at the first candidate without a crossing row (it holds the whole cell).
        A point of the cell lies in a candidate iff it satisfies the crossing rows -- evaluated by the kernel exactly as
        the scan evaluates them -- so the first candidate that passes is the scan's region, and none passing means no
        region.  int32 pool: [ncand, (region, nrows, row ...) ...]; sure[cell] = -3 - offset of the cell's list.
        Returns (pool tensor, sure, fraction of cells with a list).
        """
        nreg, p, rp = t['nreg'], t['p'], t['row_ptr']
        dev = sure.device
        und = torch.nonzero((sure == -2) if frozen is None else ((sure == -2) & ~frozen)).reshape(-1)   # (frozen: border cells that stay with the scan)
        if und.numel() == 0:
            return torch.zeros(1, dtype=torch.int32, device=dev), sure, 0.
        # cell coordinates per axis (cell index = ((c_0 G + c_1) G + ...))
        coords = []
        rem = und.clone()
        for q in range(p - 1, -1, -1):
            coords.insert(0, rem % G)
            rem = rem // G
        Ad, bd = t['dev']['A'], t['dev']['b']
        cand_u, cand_r, cand_k, row_u, row_r, row_i = [], [], [], [], [], []
        for r in range(nreg):
            A = Ad[rp[r]:rp[r + 1]]
            b = bd[rp[r]:rp[r + 1]][:, None]
            amin = amax = mag = 0.
            for q in range(p):
                a = A[:, q:q + 1]
                lo_q, hi_q = a * clo[coords[q], q][None, :], a * chi[coords[q], q][None, :]
                amin = amin + torch.minimum(lo_q, hi_q)
                amax = amax + torch.maximum(lo_q, hi_q)
                mag = mag + a.abs() * cabs[coords[q], q][None, :]
            touch = (amin <= b + (tol + 1.e-9)).all(dim=0)
            cross = touch[None, :] & ~(amax <= b - 1.e-10 * (mag + b.abs()))
            tu = torch.nonzero(touch).reshape(-1)
            if tu.numel() == 0:
                continue
            cand_u.append(tu)
            cand_r.append(torch.full_like(tu, r))
            cand_k.append(cross[:, tu].sum(dim=0))
            ri, ru = torch.nonzero(cross, as_tuple=True)
            row_u.append(ru)
            row_r.append(torch.full_like(ru, r))
            row_i.append(ri + int(rp[r]))
        cand_u = torch.cat(cand_u).cpu().numpy(); cand_r = torch.cat(cand_r).cpu().numpy(); cand_k = torch.cat(cand_k).cpu().numpy()
        row_u = torch.cat(row_u).cpu().numpy(); row_r = torch.cat(row_r).cpu().numpy(); row_i = torch.cat(row_i).cpu().numpy()
        o = np.lexsort((cand_r, cand_u)); cand_u, cand_r, cand_k = cand_u[o], cand_r[o], cand_k[o]
        o = np.lexsort((row_i, row_r, row_u)); row_u, row_i = row_u[o], row_i[o]
        pool, codes = cls._assemble_lists(cand_u, cand_r, cand_k, row_i, und.numel(), cls.LIST_MAX_INTS)
        sure = sure.clone()
        sure[und] = torch.as_tensor(codes, device=dev).to(torch.int32)
        return torch.as_tensor(pool, device=dev), sure, float((codes <= -3).sum()) / sure.numel()

    def locate(self, X, nu=None, tol=1.e-7, use_boxes=True, use_grid=True, use_sure=True):
        """Batched look-up: X [B x p] -> device tensors (idx int32 [B], u [B x nu], V [B]); no synchronisation.
        use_boxes / use_grid / use_sure switch the pruning data off (tests: the result must not depend on them)."""
        lib = _hip.load_library()
        t = self.table(nu)
        X = _hip.to_device(X)
        if X.dim() != 2 or X.shape[1] != t['p']:
            raise ValueError('X must be [B x %d]' % t['p'])
        B = X.shape[0]
        dev = X.device
        idx = torch.empty(B, dtype=torch.int32, device=dev)
        u = torch.empty((B, t['nu']), dtype=torch.float64, device=dev)
        V = torch.empty(B, dtype=torch.float64, device=dev)
        d = t['dev']
        grid = use_grid and 0. <= tol <= 1.e-7 and t['grid'] is not None
        # (few decided cells: the per-cell pass would mark nearly every point for the scan and only cost its own time --
        # measured on a 268-region table with p = 4: 2.0 against 1.5 ms per 10 M points)
        use_sure = use_sure and grid and t['grid']['use_sure']
        _hip.check(lib.pympc_point_location(t['nreg'], t['total_rows'], t['p'], t['nu'], _hip.ptr(d['A']), _hip.ptr(d['b']),
                                            _hip.ptr(d['row_ptr']), _hip.ptr(d['ux']), _hip.ptr(d['u0']), _hip.ptr(d['Vxx']),
                                            _hip.ptr(d['Vx']), _hip.ptr(d['V0']),
                                            _hip.ptr(d['box']) if (use_boxes and tol <= 1.e-7) else None,
                                            _hip.ptr(d['grid_geom']) if grid else None, t['grid']['n'] if grid else 0,
                                            _hip.ptr(d['grid_masks']) if grid else None,
                                            _hip.ptr(d['grid_sure']) if (grid and use_sure) else None,
                                            _hip.ptr(d['grid_lists']) if (grid and use_sure) else None,
                                            int(t['grid']['needs_scan']) if (grid and use_sure) else 1, _hip.ptr(X), B, float(tol),
                                            _hip.ptr(idx), _hip.ptr(u), _hip.ptr(V), _hip.stream_ptr()))
        return idx, u, V

    @staticmethod
    def _is_batch(x):
        return (isinstance(x, torch.Tensor) and x.dim() == 2) or (not isinstance(x, torch.Tensor) and np.ndim(x) == 2)

    def get_critical_region(self, x):
        """One point: the CriticalRegion or None (parametric_programs.py:442-463).  Batch: int32 indices, -1 = none."""
        if self._is_batch(x):
            return self.locate(x)[0].cpu().numpy()
        i = int(self.locate(np.asarray(x, dtype=np.float64).reshape(1, -1))[0].item())
        return None if i < 0 else self.critical_regions[i]

    def u(self, x):
        if self._is_batch(x):
            return self.locate(x)[1].cpu().numpy()
        idx, u, _ = self.locate(np.asarray(x, dtype=np.float64).reshape(1, -1))
        return None if int(idx.item()) < 0 else u[0].cpu().numpy()

    def p(self, x):
        """Dual optimiser (parametric_programs.py:488-509): region index from the kernel, affine law on the host."""
        if self._is_batch(x):
            idx = self.locate(x)[0].cpu().numpy()
            X = np.asarray(x.cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float64)
            m = self.critical_regions[0]._p['x'].shape[0]
            out = np.full((len(idx), m), np.nan)
            for r in np.unique(idx[idx >= 0]):
                sel = idx == r
                cr = self.critical_regions[r]
                out[sel] = X[sel].dot(cr._p['x'].T) + cr._p['0']
            return out
        cr = self.get_critical_region(x)
        return None if cr is None else cr.p(np.asarray(x, dtype=np.float64))

    def V(self, x):
        if self._is_batch(x):
            return self.locate(x)[2].cpu().numpy()
        idx, _, V = self.locate(np.asarray(x, dtype=np.float64).reshape(1, -1))
        return None if int(idx.item()) < 0 else float(V.item())
