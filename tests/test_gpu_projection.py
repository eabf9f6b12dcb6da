"""Orthogonal projection / convex-hull method (polyhedron.py:826-854, 941-1132) with its LPs batched on the GPU.

Known answers of the reference's tests (test_polyhedron.py:808-899: cubes, error cases) and golden projections computed
by the REFERENCE's convex_hull_method in the build container (tests/golden/projections.npz, make_golden.py::gen_proj;
random polytopes with 10 and 16 facets, n = 3..5 -> 2..n-1): same minimal half-spaces and same vertices, as sets.
"""
from itertools import product

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _same_rows(M1, M2, tol=1e-6):
    """Same set of rows after scaling each row to a unit left-hand side."""
    def norm(M):
        M = M / np.linalg.norm(M[:, :-1], axis=1)[:, None]
        return M[np.lexsort(np.round(M, 6).T[::-1])]
    M1, M2 = norm(M1), norm(M2)
    return M1.shape == M2.shape and np.allclose(M1, M2, atol=tol)


def _same_points(P1, P2, tol=1e-6):
    P1, P2 = np.vstack(P1), np.vstack(P2)
    if P1.shape != P2.shape:
        return False
    return all(np.min(np.linalg.norm(P2 - p, axis=1)) < tol for p in P1) and all(np.min(np.linalg.norm(P1 - p, axis=1)) < tol for p in P2)


def test_projection_known_answers():
    from pympc_amd.geometry.polyhedron import Polyhedron
    from pympc_amd.geometry.orthogonal_projection import convex_hull_method
    p = Polyhedron.from_lower_bound(-np.ones(3))                       # unbounded
    with pytest.raises(ValueError):
        p.project_to(range(2))
    p.add_upper_bound(np.ones(3))                                      # cube 3d -> 2d
    proj = p.project_to(range(2))
    assert _same_points(proj.vertices, [np.array(v) for v in product([1., -1.], repeat=2)])
    proj.remove_redundant_inequalities()
    assert _same_rows(np.column_stack((proj.A, proj.b)), np.column_stack((np.vstack((np.eye(2), -np.eye(2))), np.ones(4))))
    p.add_equality(np.array([[1., 1., 0.]]), np.zeros(1))              # lower dimensional
    with pytest.raises(ValueError):
        p.project_to(range(2))
    with pytest.raises(ValueError):                                    # empty
        Polyhedron.from_bounds(-np.ones(3), -2. * np.ones(3)).project_to(range(2))
    for n in range(2, 6):                                              # cubes n -> n - 1 (test_polyhedron.py:849-866)
        c = Polyhedron.from_bounds(-np.ones(n), np.ones(n))
        E, f, V = convex_hull_method(c.A, c.b, range(n - 1))
        q = Polyhedron(E, f)
        q.remove_redundant_inequalities()
        assert _same_rows(np.column_stack((q.A, q.b)), np.column_stack((np.vstack((np.eye(n - 1), -np.eye(n - 1))), np.ones(2 * (n - 1)))))
        assert _same_points(V, [np.array(v) for v in product([1., -1.], repeat=n - 1)])
    # projection onto one coordinate and onto a non-leading subset
    c = Polyhedron.from_bounds(np.array([-1., -2., -3.]), np.array([1., 2., 3.]))
    E, f, V = convex_hull_method(c.A, c.b, [1])
    assert _same_points(V, [np.array([2.]), np.array([-2.])])
    q = c.project_to([0, 2])
    assert _same_points(q.vertices, [np.array(v) for v in product([1., -1.], [3., -3.])])


def test_projection_against_reference_golden():
    from pympc_amd.geometry.polyhedron import Polyhedron
    g = golden("projections")
    n = 0
    for k in range(int(g["count"])):
        if not bool(g["ok%d" % k]):
            continue
        A, b, n_proj = g["A%d" % k], g["b%d" % k], int(g["n_proj%d" % k])
        P = Polyhedron(A.copy(), b.copy())
        proj = P.project_to(range(n_proj))
        assert _same_points(proj.vertices, g["V%d" % k]), k
        proj.remove_redundant_inequalities()
        assert _same_rows(np.column_stack((proj.A, proj.b)), np.column_stack((g["E%d" % k], g["f%d" % k]))), k
        # the vertices of the polytope itself (qhull on the host, centre and boundedness from the LP kernel) project into it
        V = np.vstack(P.vertices)[:, :n_proj]
        assert np.all(V.dot(proj.A.T) <= proj.b[None, :] + 1e-7)
        n += 1
    assert n >= 15


def test_feasible_set_of_an_mpqp_is_where_the_qp_is_feasible():
    """MultiParametricQuadraticProgram.get_feasible_set (parametric_programs.py:243-260) on the C1 controller: the
    batched QP path is feasible exactly on that polytope."""
    from pympc_amd.optimization.parametric_programs import MultiParametricQuadraticProgram
    m = golden("c1_model")
    H = {'uu': m["Huu"], 'ux': m["Hux"], 'xx': m["Hxx"]}
    mp = MultiParametricQuadraticProgram(H, {'u': m["fu"], 'x': m["fx"]}, float(m["g"]), {'u': m["Au"], 'x': m["Ax"]}, m["b"])
    F = mp.get_feasible_set()
    X = np.random.RandomState(0).uniform([-5., -1.], [5., 1.], (4000, 2))
    res = mp.solve(X, want_mult=False, want_active=False)
    margin = (X.dot(F.A.T) - F.b[None, :]).max(axis=1)
    clear = np.abs(margin) > 1e-6
    assert np.array_equal(res.feasible[clear], (margin <= 0)[clear]) and 0.3 < res.feasible.mean() < 0.7
