"""Orthogonal projection of polytopes by the convex-hull method, with the LPs of every round in ONE batched launch.

Reference: /root/reference/pympc/geometry/polyhedron.py:941-1132 (convex_hull_method, _get_two_vertices,
_get_inner_simplex, _expand_simplex), called by Polyhedron.project_to (:826-854).  There every facet of the inner
approximation costs one LP, solved one after the other, and the hull grows by ONE point before all facets are looked
at again.  The LPs of a round share the constraint set {A x <= b} and differ only in the objective, so here a round
is one launch of the LP kernel (pympc_lp_solve_batch with a shared matrix) over all facets that have not been
explored yet, and every point that expands its facet joins the hull at once.  The projection is the same polytope
(every LP optimum is a point of it, a facet that does not expand is a facet of it); qhull stays on the host as in the
reference (scipy.spatial.ConvexHull, incremental).
"""
import numpy as np
from scipy.spatial import ConvexHull

from ..optimization.solvers import hip_lp


def _lp_points(F, A, b, n):
    """argmin (first n coordinates) and min of  min f'x s.t. A x <= b  for every row f of F: one batched launch."""
    res = hip_lp.linear_program_batch(np.ascontiguousarray(F), A, b, want_mult=False, want_active=False)
    st = res.numpy('status')
    if np.any(st == 3):
        raise RuntimeError('Maximum number of iterations reached.')
    if np.any(st != 0):
        raise ValueError('an LP of the convex-hull method has no optimum (unbounded or empty polyhedron).')
    return res.numpy('argmin')[:, :n].copy(), res.numpy('min').copy()


def _plane_through(points):
    """Unit normal a and offset d of the hyperplane {x | a'x = d} through i points of R^i (the role of
    geometry/utils.py:54-92, without its random offset)."""
    P = np.vstack(points)
    a = np.linalg.svd(P[1:] - P[0])[2][-1]
    return a, float(a.dot(P[0]))


def convex_hull_method(A, b, residual_dimensions, tol=1.e-7):
    """
    Projection of the bounded, full-dimensional polytope {x | A x <= b} onto the given coordinates.
    Returns (E, f, vertices): half-spaces {y | E y <= f} (unit normals, one per facet of the hull as qhull lists them)
    and the vertices of the projection [V x n].
    """
    residual_dimensions = list(residual_dimensions)
    n = len(residual_dimensions)
    dropped = [i for i in range(A.shape[1]) if i not in residual_dimensions]
    A = np.ascontiguousarray(np.hstack((A[:, residual_dimensions], A[:, dropped])), dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    nz = A.shape[1] - n

    def lift(D):
        return np.hstack((D, np.zeros((D.shape[0], nz))))

    # two vertices along the first residual coordinate (:1003-1035)
    e0 = np.zeros((1, n)); e0[0, 0] = 1.
    pts, _ = _lp_points(lift(np.vstack((e0, -e0))), A, b, n)
    vertices = [pts[0], pts[1]]
    if n == 1:
        hi = max(v[0] for v in vertices); lo = min(v[0] for v in vertices)
        return np.array([[1.], [-1.]]), np.array([hi, -lo]), np.array([[hi], [lo]])
    # inner simplex: one more dimension per step, both directions of the new normal in one launch (:1037-1078)
    for i in range(2, n + 1):
        a, d = _plane_through([v[:i] for v in vertices])
        ai = np.zeros((1, n)); ai[0, :i] = a
        pts, _ = _lp_points(lift(np.vstack((ai, -ai))), A, b, n)
        k = 0 if abs(a.dot(pts[0][:i]) - d) >= abs(a.dot(pts[1][:i]) - d) else 1
        if abs(a.dot(pts[k][:i]) - d) < tol:
            raise ValueError('the projection is not full dimensional.')
        vertices.append(pts[k])
    # expansion (:1080-1132): all unexplored facets of the inner approximation per round
    hull = ConvexHull(np.vstack(vertices), incremental=True)
    explored = np.zeros((0, n))
    while True:
        eq = hull.equations
        normals = eq[:, :-1] / np.linalg.norm(eq[:, :-1], axis=1)[:, None]
        offsets = -eq[:, -1] / np.linalg.norm(eq[:, :-1], axis=1)
        new = np.array([not np.any(np.all(np.abs(explored - a) <= 1e-8 + 1e-5 * np.abs(a), axis=1)) for a in normals], dtype=bool)
        if not new.any():
            break
        explored = np.vstack((explored, normals[new]))
        pts, mins = _lp_points(lift(-normals[new]), A, b, n)
        grow = (-mins - offsets[new]) > tol
        if not grow.any():
            break
        cand = pts[grow]
        # (the same vertex can answer several facets: qhull wants each point once)
        keep = [0]
        for r in range(1, cand.shape[0]):
            if not np.any(np.all(np.abs(cand[keep] - cand[r]) <= 1e-9, axis=1)):
                keep.append(r)
        hull.add_points(cand[keep])
    hull.close()
    E = hull.equations[:, :-1].copy()
    f = -hull.equations[:, -1].copy()
    return E, f, hull.points[hull.vertices].copy()
