"""The reference's Polyhedron known-answer tests for the methods on the hot path
(pympc/test/test_geometry/test_polyhedron.py:301-393, 449-470, 472-520, 581-632), run against
pympc_amd.geometry.polyhedron on the GPU.  Inputs and expected values are the data those tests hold."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _P():
    from pympc_amd.geometry.polyhedron import Polyhedron
    return Polyhedron


def same_rows(E, F, tol=1e-7):
    """Every row of E equals (up to a positive factor) a row of F and vice versa."""
    def norm(M):
        return np.array([r / np.linalg.norm(r) for r in M])
    E, F = norm(E), norm(F)
    if E.shape != F.shape:
        return False
    return all(np.min(np.linalg.norm(F - e, axis=1)) < tol for e in E) and \
        all(np.min(np.linalg.norm(E - f, axis=1)) < tol for f in F)


def test_remove_redundant_inequalities():
    Polyhedron = _P()
    # test_polyhedron.py:303-324: duplicated facet (0 and 5) and a zero row
    A = np.array([[1., 1.], [-1., 1.], [0., -1.], [0., 1.], [0., 1.], [2., 2.]])
    b = np.array([1., 1., 1., 1., 2., 2.])
    p = Polyhedron(A, b)
    assert p.minimal_facets() == [1, 2, 5]          # the reference keeps the LATER duplicate
    p.add_inequality(np.zeros((1, 2)), np.ones(1))
    assert p.minimal_facets() == [1, 2, 5]
    p.remove_redundant_inequalities()
    assert same_rows(np.column_stack((np.array([[-1., 1.], [0., -1.], [1., 1.]]), np.ones(3))),
                     np.column_stack((p.A, p.b)))
    # :326-362 with equalities
    p = Polyhedron.from_bounds(-np.ones(2), np.ones(2))
    C = np.ones((1, 2)); d = np.ones(1)
    p.add_equality(C, d)
    assert sorted(p.minimal_facets()) == [2, 3]
    p.remove_redundant_inequalities()
    Amin = np.column_stack((np.array([[1., 0.], [0., 1.]]), np.ones(2)))
    assert same_rows(Amin, np.column_stack((p.A, p.b)))
    p.add_inequality(C, d)
    p.remove_redundant_inequalities()
    assert same_rows(Amin, np.column_stack((p.A, p.b)))
    p.add_inequality(np.array([[-1., 1.]]), np.array([1.1]))
    p.remove_redundant_inequalities()
    assert same_rows(Amin, np.column_stack((p.A, p.b)))
    # :364-374 empty polyhedra raise
    p.add_equality(np.ones((1, 2)), -np.ones(1))
    with pytest.raises(ValueError):
        p.remove_redundant_inequalities()
    with pytest.raises(ValueError):
        Polyhedron.from_bounds(np.ones(2), np.zeros(2)).remove_redundant_inequalities()


def test_empty():
    # test_polyhedron.py:376-393
    p = _P().from_bounds(1. * np.ones(2), 2. * np.ones(2))
    assert not p.empty
    p.add_equality(np.ones((1, 2)), np.array([3.]))
    assert not p.empty
    p.add_upper_bound(np.array([.5]), [0])
    assert p.empty


def test_bounded():
    # test_polyhedron.py:395-447
    Polyhedron = _P()
    assert Polyhedron.from_bounds(np.ones(2), -np.ones(2)).bounded
    assert Polyhedron.from_bounds(np.array([1.]), np.array([-1.]), [0], 2).bounded
    assert Polyhedron.from_bounds(1. * np.ones(2), 2. * np.ones(2)).bounded
    p = Polyhedron.from_lower_bound(np.array([0.]), [0], 2)
    assert not p.bounded
    p.add_lower_bound(np.array([0.]), [1])
    assert not p.bounded
    p.add_upper_bound(np.array([1.]), [0])
    assert not p.bounded
    p.add_lower_bound(np.array([1.]), [0])
    assert not p.bounded
    p.add_upper_bound(np.array([1000.]), [1])
    assert p.bounded
    p = Polyhedron.from_bounds(np.array([-1., -2.]), np.array([1., 2.]), [0, 1], 3)
    p.add_equality(np.array([[1., 0., -1.]]), np.zeros(1))
    assert p.bounded


def test_contains():
    # test_polyhedron.py:449-470 + the batched form
    x1 = 2. * np.ones(2); x2 = 2.5 * np.ones(2); x3 = 3.5 * np.ones(2)
    p = _P().from_bounds(1. * np.ones(2), 3. * np.ones(2))
    assert p.contains(x1) and p.contains(x2) and not p.contains(x3)
    assert list(p.contains(np.array([x1, x2, x3, [3. + 5e-8, 1.], [3. + 2e-7, 1.]]))) == [True, True, False, True, False]
    p.add_equality(np.ones((1, 2)), np.array([4.]))
    assert p.contains(x1) and not p.contains(x2) and not p.contains(x3)


def test_is_included_in():
    # test_polyhedron.py:472-521
    Polyhedron = _P()
    p1 = Polyhedron.from_bounds(1. * np.ones(2), 2. * np.ones(2))
    p2 = Polyhedron.from_bounds(.5 * np.ones(2), 2.5 * np.ones(2))
    assert p1.is_included_in(p1)
    assert p1.is_included_in(p2)
    assert not p2.is_included_in(p1)
    p2 = Polyhedron.from_bounds(.5 * np.ones(2), 1.5 * np.ones(2))       # intersects p1
    assert not p1.is_included_in(p2) and not p2.is_included_in(p1)
    p2 = Polyhedron.from_bounds(.5 * np.ones(2), 2. * np.ones(2))        # includes p1, two equal facets
    assert p1.is_included_in(p2) and not p2.is_included_in(p1)
    p2.add_lower_bound(1.5 * np.ones(2))
    assert not p1.is_included_in(p2) and p2.is_included_in(p1)
    C = np.ones((1, 2)); d = np.array([3.5])                             # with equalities
    p1.add_equality(C, d)
    assert p1.is_included_in(p2) and not p2.is_included_in(p1)
    p2.add_equality(C, d)
    assert p1.is_included_in(p2) and p2.is_included_in(p1)
    p2.add_upper_bound(np.array([1.75]), [1])
    assert not p1.is_included_in(p2) and p2.is_included_in(p1)


def test_chebyshev():
    # test_polyhedron.py:581-632
    Polyhedron = _P()
    p = Polyhedron.from_bounds(np.zeros(2), 2. * np.ones(2))
    assert abs(p.radius - 1.) < 1e-7
    np.testing.assert_array_almost_equal(p.center, np.ones(2))
    p.add_inequality(np.zeros((1, 2)), np.ones(1))                     # nasty zero row
    assert abs(p.radius - 1.) < 1e-7
    np.testing.assert_array_almost_equal(p.center, np.ones(2))
    p.add_equality(np.ones((1, 2)), np.array([3.]))
    assert abs(p.radius - np.sqrt(2.) / 2.) < 1e-7
    np.testing.assert_array_almost_equal(p.center, 1.5 * np.ones(2))
    p = Polyhedron.from_bounds(np.ones(2), -np.ones(2))               # empty: negative radius
    assert abs(p.radius + 1.) < 1e-7
    np.testing.assert_array_almost_equal(p.center, np.zeros(2))
    p = Polyhedron.from_lower_bound(np.ones(2))                        # unbounded
    assert p.radius is None and p.center is None
    p.add_upper_bound(np.array([2.]), [1])                             # "bounded very difficult"
    assert abs(p.radius - .5) < 1e-7
    assert abs(p.center[0] - 1.5) < 1e-7
    p = Polyhedron.from_bounds(np.array([-1., -2.]), np.array([1., 2.]), [0, 1], 3)
    p.add_equality(np.array([[1., 0., -1.]]), np.zeros(1))
    assert abs(p.radius - np.sqrt(2.)) < 1e-7
    assert abs(p.center[0]) < 1e-7 and abs(p.center[2]) < 1e-7
    assert np.abs(p.center[1]) < 2. - p.radius + 1.e-6


def test_bounded_with_more_rows_than_the_stiemke_lp_holds():
    """polyhedron.py:528-576 has no size limit; beyond 32 rows the decision is taken in the primal space."""
    Polyhedron = _P()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_model.npz"))
    XN = Polyhedron(g["XN_A"], g["XN_b"])                               # the 130-facet cart-pole terminal set (mcais)
    assert XN.A.shape[0] == 130 and XN.bounded
    rng = np.random.RandomState(4)
    A = rng.randn(60, 5); A /= np.linalg.norm(A, axis=1)[:, None]
    assert Polyhedron(A, np.ones(60)).bounded                           # 60 random facets around the origin
    A[:, 0] = -np.abs(A[:, 0])                                          # every normal has a non-positive first entry:
    assert not Polyhedron(A, np.ones(60)).bounded                       # +e_0 is a recession direction
    p = Polyhedron.from_bounds(-np.ones(3), np.ones(3))
    for k in range(40):
        p.add_inequality(rng.randn(1, 3), np.array([5.]))
    assert p.bounded


def test_contains_batch_with_equalities():
    """The batched form of polyhedron.py:578-604 with equalities: same decisions as the scalar form, row by row
    (including the reference's |max(C x - d)| <= tol, which only looks at the largest entry)."""
    Polyhedron = _P()
    p = Polyhedron.from_bounds(1. * np.ones(2), 3. * np.ones(2))
    p.add_equality(np.ones((1, 2)), np.array([4.]))
    X = np.array([[2., 2.], [2.5, 2.5], [3.5, 3.5], [1., 3.], [3., 1. + 5e-8], [3., 1. + 2e-7], [2., 2. - 5e-8], [2., 2. - 2e-7],
                  [0.5, 3.5]])
    got = p.contains(X)
    assert list(got) == [bool(p.contains(x)) for x in X]
    assert list(got) == [True, False, False, True, True, False, True, False, False]
    p = Polyhedron.from_bounds(-np.ones(3), np.ones(3))
    p.add_equality(np.array([[1., 1., 0.], [0., 1., -1.]]), np.array([0., .5]))
    rng = np.random.RandomState(0)
    X = rng.uniform(-1.2, 1.2, (2000, 3))
    X[::2, 0] = -X[::2, 1]; X[::2, 2] = X[::2, 1] - .5                  # half of the points on the equalities
    X[::4, 2] -= 1e-3                                                   # second residual negative, first zero: still "inside"
    got = p.contains(X)
    assert list(got) == [bool(p.contains(x)) for x in X] and got.sum() > 100


def test_errors_where_the_reference_fails():
    """is_included_in with an LP without optimum: the reference raises TypeError (-None, polyhedron.py:636)."""
    Polyhedron = _P()
    half = Polyhedron.from_lower_bound(np.zeros(2))                     # unbounded
    box = Polyhedron.from_bounds(-np.ones(2), np.ones(2))
    assert box.is_included_in(Polyhedron.from_bounds(-2. * np.ones(2), 2. * np.ones(2)))
    with pytest.raises(TypeError):
        half.is_included_in(box)


@pytest.mark.parametrize("m,dup", [(3000, 0), (2500, 40)])
def test_staged_redundancy_removal_same_list(m, dup):
    """Polytopes with >= 2000 facets take minimal_facets' staged first steps (ray-shot subset, every other facet against
    that subset, the full-set LP only for what is left): the index list must be the plain path's, also with duplicated and
    scaled-duplicate facets (which the reference's sequential rule, polyhedron.py:435-448, resolves by index order)."""
    from pympc_amd.geometry.polyhedron import Polyhedron
    rng = np.random.RandomState(m)
    A = rng.randn(m, 4); A /= np.linalg.norm(A, axis=1)[:, None]
    b = 1. + rng.rand(m)
    if dup:
        src = rng.choice(m, dup, replace=False)
        dst = rng.choice(m, dup, replace=False)
        scale = 1. + rng.rand(dup)
        A[dst] = A[src] * scale[:, None]; b[dst] = b[src] * scale
    staged = Polyhedron(A.copy(), b.copy()).minimal_facets()
    plain_poly = Polyhedron(A.copy(), b.copy())
    plain_poly.STAGED_MIN_FACETS = 10 ** 9
    plain = plain_poly.minimal_facets()
    assert staged == plain
    assert 30 < len(staged) < m // 2
    # every kept facet is tight somewhere on the polytope, every removed one is implied: vertices of the kept set satisfy all rows
    P = Polyhedron(A[staged], b[staged])
    V = np.array(P.vertices)
    assert (A.dot(V.T) - b[:, None]).max() < 1e-6
