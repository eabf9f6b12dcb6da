"""GPU parity of the LP path (pympc_lp_solve_batch through the drop-in API).

  * the reference's own LP known-answer tests (pympc/test/test_optimization/test_solvers.py:12-256), same data and
    assertions as tests/test_oracle.py applies to the oracle;
  * golden vectors produced by the reference in the build container (tests/golden/lp_polytopes.npz);
  * seeded random LPs (bounded, infeasible, unbounded, degenerate, duplicated / zero rows) against the CPU oracle's
    restatement of pnnls.linear_program.
Tolerances: solvability and active sets exact; min / argmin / multipliers within 1e-6 (relative to max(1, |.|)).
Degenerate LPs (non-unique primal or dual optimum) are compared through the optimal value only, as SURVEY 8(c) says.
"""
import numpy as np
import pytest

from conftest import golden
from test_oracle import _lp_kats

pytestmark = pytest.mark.gpu


def _lp():
    from pympc_amd.optimization.programs import linear_program
    return linear_program


def test_reference_lp_kats():
    _lp_kats(_lp())


def test_lp_errors():
    lp = _lp()
    with pytest.raises(ValueError):
        lp(np.ones(2), -np.eye(2), np.zeros(2), C=np.ones((1, 2)))                 # pnnls.py:90-91
    with pytest.raises(ValueError):
        lp(np.ones(2), -np.eye(2), np.zeros(2), solver='nope')                     # programs.py:55


def test_programs_lp_kats():
    # test_programs.py:14-64 (dispatch layer): same two LPs through solver='pnnls' alias and the default
    f = np.ones(2); A = np.array([[1., 0.], [-1., 0.], [0., 1.], [0., -1.]]); b = np.array([5., 5., 1., 1.])
    for kw in ({}, {'solver': 'pnnls'}, {'solver': 'hip'}):
        sol = _lp()(f, A, b, **kw)
        assert abs(sol['min'] + 6.) < 1e-7
        np.testing.assert_array_almost_equal(sol['argmin'], [-5., -1.])
        assert sol['active_set'] == [1, 3]


def test_generic_lps_vs_reference_golden():
    g = golden("lp_polytopes")
    from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
    res = linear_program_batch(g["lp_f"], g["lp_A"], g["lp_b"])
    valid = g["lp_valid"].astype(bool)
    solv = g["lp_solvable"].astype(bool)
    assert np.array_equal(res.feasible[valid], solv[valid])
    sel = valid & solv
    assert np.allclose(res.numpy('min')[sel], g["lp_min"][sel], rtol=1e-6, atol=1e-6)
    assert np.allclose(res.numpy('argmin')[sel], g["lp_argmin"][sel], rtol=1e-6, atol=1e-6)
    assert np.allclose(res.numpy('multiplier_inequality')[sel], g["lp_mult"][sel], rtol=1e-6, atol=1e-6)
    assert np.array_equal(res.active_mask()[sel], g["lp_mult"][sel] > 1e-7)
    assert np.all(np.isnan(res.numpy('min')[~res.feasible]))


@pytest.mark.parametrize("m", [30, 60, 100])
def test_polytope_vs_reference_golden(m):
    """Polyhedron.minimal_facets / remove_redundant_inequalities / radius / center (config C3 recipe, small m)."""
    from pympc_amd.geometry.polyhedron import Polyhedron
    g = golden("lp_polytopes")
    A = g["m%d_A" % m]; b = g["m%d_b" % m]
    P = Polyhedron(A.copy(), b.copy())
    mf = P.minimal_facets()
    assert mf == list(g["m%d_minimal_facets" % m])                       # bit-exact index list
    assert abs(P.radius - float(g["m%d_radius" % m])) < 1e-6
    assert np.allclose(P.center, g["m%d_center" % m], rtol=1e-6, atol=1e-6)
    P.remove_redundant_inequalities()
    assert np.array_equal(P.A, A[mf]) and np.array_equal(P.b, b[mf])
    # idempotence: a minimal representation has no redundant facet left
    assert Polyhedron(P.A.copy(), P.b.copy()).minimal_facets() == list(range(len(mf)))
    # Chebyshev multipliers and the per-facet LP values of polyhedron.py:447 against the full set (step 1)
    from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
    Alp = np.hstack((A, -np.linalg.norm(A, axis=1)[:, None]))
    res = linear_program_batch(np.concatenate((np.zeros(6), [1.]))[None, :], Alp, b)
    assert np.allclose(res.numpy('multiplier_inequality')[0], g["m%d_cheb_mult" % m], atol=1e-6)


@pytest.mark.parametrize("m", [30, 60, 100])
def test_lockstep_kernel_on_reference_polytopes(m):
    """lp_tile_kernel (eight LPs per workgroup in lockstep: the kernel config C3 is timed on) normally starts at 1024
    rows / 1024 LPs, sizes the reference cannot run.  With the launch thresholds lowered (pympc_hip_tune) it runs the
    reference-run golden polytopes:
      * the reference's own sequence of LPs (polyhedron.py:435-448: facet i against the set with the earlier redundant
        facets removed), one launch each: the value -min - b_i of :447 to 1e-6 against the reference's, for every facet;
      * all m LPs against the full set in one launch: statuses, active sets and pivot counts identical to the
        one-LP-per-workgroup kernel, values to rounding;
      * Polyhedron.minimal_facets with every LP batch on this kernel: the reference's index list, exact."""
    from pympc_amd import _hip
    from pympc_amd.geometry.polyhedron import Polyhedron
    from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
    g = golden("lp_polytopes")
    assert bool(g["m%d_valid" % m])
    A = g["m%d_A" % m]; b = g["m%d_b" % m]; vals = g["m%d_facet_values" % m]
    rx = np.arange(m, dtype=np.int32)
    small = linear_program_batch(-A, A, b, relax_idx=rx, relax_val=1., want_iters=True)
    with _hip.tuned(lp_tile_min_rows=1, lp_tile_min_batch=1):
        tile = linear_program_batch(-A, A, b, relax_idx=rx, relax_val=1., want_iters=True)
        mf = Polyhedron(A.copy(), b.copy()).minimal_facets()
        keep = list(range(m))
        got = np.zeros(m)
        for i in range(m):
            pos = keep.index(i)
            r = linear_program_batch(-A[i][None, :], A[keep], b[keep], relax_idx=np.array([pos], dtype=np.int32), relax_val=1.)
            assert r.feasible.all()
            got[i] = -r.numpy('min')[0] - b[i]
            if vals[i] < 1e-7:            # the reference dropped facet i here (polyhedron.py:447-448)
                keep.remove(i)
    assert _hip.tune("lp_tile_min_rows") == 1024 and _hip.tune("lp_tile_min_batch") == 1024
    assert keep == list(g["m%d_minimal_facets" % m])
    assert np.allclose(got, vals, rtol=1e-6, atol=1e-6)
    assert mf == list(g["m%d_minimal_facets" % m])                            # bit-exact index list (reference run)
    assert tile.feasible.all()
    assert np.array_equal(tile.numpy('status'), small.numpy('status'))
    assert np.array_equal(tile.numpy('iters'), small.numpy('iters'))
    assert np.array_equal(tile.numpy('active_bits'), small.numpy('active_bits'))
    assert np.allclose(tile.numpy('min'), small.numpy('min'), rtol=1e-12, atol=1e-12)
    assert np.allclose(tile.numpy('argmin'), small.numpy('argmin'), rtol=1e-9, atol=1e-9)
    # a facet that is not redundant against the full relaxed set is in the reference's list
    assert set(np.where(-tile.numpy('min') - b >= 1e-7)[0]) <= set(mf)


def test_random_lps_vs_oracle(oracle):
    rng = np.random.RandomState(5)
    lp = _lp()
    n_solv = 0
    for t in range(200):
        n = rng.randint(1, 7); m = rng.randint(1, 25)
        kind = t % 5
        A = rng.randn(m, n)
        if kind == 0:
            b = 1 + rng.rand(m)
        elif kind == 1:
            b = rng.randn(m)
        elif kind == 2:
            A = np.round(A); b = np.round(rng.randn(m) * 2)               # integer data: degenerate vertices
        elif kind == 3:
            e = max(1, m // 3)
            A = np.vstack((A, A[:e])); b = np.concatenate((1 + rng.rand(m), 1 + rng.rand(e)))   # duplicated rows
        else:
            A[rng.randint(m)] = 0; b = rng.rand(m) - .1                    # a zero row, sometimes infeasible
        f = rng.randn(n) if kind != 2 else np.round(rng.randn(n) * 2)
        ref = oracle.linear_program(f, A, b)
        sol = lp(f, A, b)
        assert (sol['min'] is None) == (ref['min'] is None), (t, kind)
        if ref['min'] is None:
            assert all(v is None for v in sol.values())
            continue
        n_solv += 1
        assert abs(sol['min'] - ref['min']) < 1e-6 * max(1., abs(ref['min'])), (t, kind)
        # primal feasibility, dual feasibility and complementarity of what the kernel returned
        x, y = sol['argmin'], sol['multiplier_inequality']
        assert (A.dot(x) - b).max() < 1e-7 * max(1., np.abs(b).max())
        assert y.min() >= 0. and np.abs(A.T.dot(y) + f).max() < 1e-7 * max(1., np.abs(y).max())
        if kind in (0, 1):   # generic data: unique primal-dual pair, active set bit-exact
            assert np.allclose(x, ref['argmin'], rtol=1e-6, atol=1e-6), (t, kind)
            assert np.allclose(y, ref['multiplier_inequality'], rtol=1e-6, atol=1e-6), (t, kind)
            assert sol['active_set'] == ref['active_set'], (t, kind)
    assert n_solv > 50


@pytest.mark.parametrize("n", [33, 48, 64])
def test_many_variables_vs_oracle(oracle, n):
    """LPs with more than 32 variables (up to the kernel's 64: one lane per row of the dual system), against the oracle's
    restatement of pnnls.linear_program: boxes cut by random rows (bounded, generic), the same with a row that makes
    them infeasible, and an unbounded one."""
    rng = np.random.RandomState(100 + n)
    lp = _lp()
    for t in range(6):
        m_extra = 20 + 10 * t
        A = np.vstack((np.eye(n), -np.eye(n), rng.randn(m_extra, n)))
        b = np.concatenate((np.ones(2 * n), 1. + rng.rand(m_extra)))
        f = rng.randn(n)
        if t == 4:                                   # infeasible: x_0 <= -2 against x_0 >= -1
            A = np.vstack((A, np.eye(n)[:1])); b = np.concatenate((b, [-2.]))
        if t == 5:                                   # unbounded: no bound along -f
            A = A[2 * n:]; b = b[2 * n:]
            A = A[A.dot(f) > 0]; b = b[:A.shape[0]]
        ref = oracle.linear_program(f, A, b)
        sol = lp(f, A, b)
        assert (sol['min'] is None) == (ref['min'] is None), (n, t)
        if t >= 4:
            assert ref['min'] is None
            continue
        assert ref['min'] is not None
        assert abs(sol['min'] - ref['min']) < 1e-6 * max(1., abs(ref['min'])), (n, t)
        assert np.allclose(sol['argmin'], ref['argmin'], rtol=1e-6, atol=1e-6), (n, t)
        assert np.allclose(sol['multiplier_inequality'], ref['multiplier_inequality'], rtol=1e-6, atol=1e-6), (n, t)
        assert sol['active_set'] == ref['active_set'], (n, t)


def test_lp_batch_shared_matrix_relaxation(oracle):
    """The minimal_facets batch form: shared A, shared b, per-problem relaxed row (polyhedron.py:439-441)."""
    from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
    rng = np.random.RandomState(11)
    m, n = 40, 4
    A = rng.randn(m, n); A /= np.linalg.norm(A, axis=1)[:, None]
    b = 1. + rng.rand(m)
    res = linear_program_batch(-A, A, b, relax_idx=np.arange(m, dtype=np.int32), relax_val=1., want_iters=True)
    assert res.feasible.all()
    for i in range(m):
        rel = b.copy(); rel[i] += 1.
        ref = oracle.linear_program(-A[i], A, rel)
        assert abs(res.numpy('min')[i] - ref['min']) < 1e-6
        x = res.numpy('argmin')[i]
        assert (A.dot(x) - rel).max() < 1e-9
        if abs(ref['min'] + rel[i]) > 1e-6:
            # the optimum stops before the relaxed facet: unique vertex.  (Otherwise every point of the face
            # {a_i x = b_i + 1} is optimal and only the value is comparable.)
            assert np.allclose(x, ref['argmin'], rtol=1e-6, atol=1e-6)
    # same LPs with per-problem matrices and right-hand sides (a_stride / b_stride != 0)
    bb = np.tile(b, (m, 1)); bb[np.arange(m), np.arange(m)] += 1.
    res2 = linear_program_batch(-A, np.tile(A, (m, 1, 1)), bb)
    assert np.array_equal(res2.numpy('min'), res.numpy('min'))
    assert np.array_equal(res2.numpy('argmin'), res.numpy('argmin'))


@pytest.mark.parametrize("m", [1000, 10000])
def test_full_size_polytope_properties(m):
    """Config C3 at full size (SURVEY 8d recipe: A = randn(m, 6) row-normalised, b = 1 + rand(m), seed 0).  The
    reference needs a dense (1+m+n) x 2m matrix per LP (1.6 GB at m = 10 000), so parity is checked through
    size-independent properties and an independent computation: the non-redundant facets of a bounded full-dimensional
    polytope are exactly the facets that carry >= n vertices, and qhull enumerates the vertices."""
    from scipy.spatial import HalfspaceIntersection
    from pympc_amd.geometry.polyhedron import Polyhedron
    from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
    rng = np.random.RandomState(0)
    A = rng.randn(m, 6); A /= np.linalg.norm(A, axis=1)[:, None]
    b = 1. + rng.rand(m)
    P = Polyhedron(A.copy(), b.copy())
    mf = P.minimal_facets()
    r, c = P.radius, P.center
    # Chebyshev ball: inside, touches >= n + 1 = 7 facets (generic data), multipliers certify optimality
    slack = b - A.dot(c)
    assert slack.min() >= r - 1e-9 and (np.abs(slack - r) < 1e-8).sum() >= 7
    V = HalfspaceIntersection(np.hstack((A, -b[:, None])), c).intersections
    on = np.abs(V.dot(A.T) - b) < 1e-9
    assert mf == [int(i) for i in np.where(on.sum(axis=0) >= 6)[0]]          # bit-exact index list
    # idempotence and in-place semantics of remove_redundant_inequalities (polyhedron.py:452-466)
    P.remove_redundant_inequalities()
    assert np.array_equal(P.A, A[mf]) and np.array_equal(P.b, b[mf])
    assert Polyhedron(P.A.copy(), P.b.copy()).minimal_facets() == list(range(len(mf)))
    # every vertex of the polytope satisfies the reduced description, every dropped facet is implied by it:
    # max a_i x over the reduced set <= b_i
    dropped = np.setdiff1d(np.arange(m), mf)[:512]
    res = linear_program_batch(-A[dropped], P.A, P.b, want_argmin=False, want_mult=False, want_active=False)
    assert res.feasible.all() and np.all(-res.numpy('min') <= b[dropped] + 1e-7)


@pytest.mark.parametrize("n,m", [(6, 4000), (3, 2500), (8, 2048)])
def test_shared_matrix_lockstep_kernel_vs_workgroup_kernel(n, m):
    """Large batches against one shared (A, b) run eight LPs per workgroup in lockstep (csrc/lp_tile.hpp); the same LPs
    in slices below that kernel's batch threshold run one LP per workgroup (csrc/lp_block.hpp).  Same method, same
    pivots: statuses, active sets and pivot counts must be identical, values equal to rounding."""
    from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
    rng = np.random.RandomState(100 + n)
    A = rng.randn(m, n); A /= np.linalg.norm(A, axis=1)[:, None]
    b = 1. + rng.rand(m)
    f = -A.copy()
    f[5] = 0.                      # zero objective: phase 1 has nothing to do
    f[7:40] = rng.randn(33, n)     # generic objectives (no relaxed row of their own)
    rx = np.arange(m, dtype=np.int32)
    big = linear_program_batch(f, A, b, relax_idx=rx, relax_val=1., want_iters=True)
    assert big.feasible.all()
    S = 512
    for lo in range(0, m, S):
        sl = slice(lo, min(lo + S, m))
        ref = linear_program_batch(f[sl], A, b, relax_idx=rx[sl], relax_val=1., want_iters=True)
        assert np.array_equal(big.numpy('status')[sl], ref.numpy('status'))
        assert np.array_equal(big.numpy('active_bits')[sl], ref.numpy('active_bits'))
        assert np.array_equal(big.numpy('iters')[sl], ref.numpy('iters'))
        assert np.allclose(big.numpy('min')[sl], ref.numpy('min'), rtol=1e-12, atol=1e-12)
        assert np.allclose(big.numpy('argmin')[sl], ref.numpy('argmin'), rtol=1e-9, atol=1e-9)
        assert np.allclose(big.numpy('multiplier_inequality')[sl], ref.numpy('multiplier_inequality'), rtol=1e-9, atol=1e-9)
    # no relaxed row at all: generic objectives over the polytope
    F = rng.randn(1500, n)
    big = linear_program_batch(F, A, b, want_iters=True)
    assert big.feasible.all()
    for lo in range(0, 1500, 500):
        ref = linear_program_batch(F[lo:lo + 500], A, b, want_iters=True)
        assert np.array_equal(big.numpy('iters')[lo:lo + 500], ref.numpy('iters'))
        assert np.array_equal(big.numpy('active_bits')[lo:lo + 500], ref.numpy('active_bits'))
        assert np.allclose(big.numpy('argmin')[lo:lo + 500], ref.numpy('argmin'), rtol=1e-9, atol=1e-9)
    # an empty polyhedron: every LP ends with a Farkas certificate in phase 2 (status 1, NaN outputs)
    A2 = np.vstack((A, np.eye(n)[0], -np.eye(n)[0]))
    b2 = np.concatenate((b, [-5., -5.]))
    f2 = np.vstack((f, np.zeros((2, n))))
    bad = linear_program_batch(f2, A2, b2, want_iters=True)
    assert (bad.numpy('status') == 1).all() and np.isnan(bad.numpy('min')).all()
    ref = linear_program_batch(f2[:300], A2, b2, want_iters=True)
    assert np.array_equal(bad.numpy('status')[:300], ref.numpy('status'))
    assert np.array_equal(bad.numpy('iters')[:300], ref.numpy('iters'))


def test_dual_degenerate_lp_is_retried_with_a_tilted_objective(oracle):
    """The Chebyshev LP of the cart-pole constraint polytope in (x, u) (370 rows, 24 + 1 variables; the objective sees
    the radius only) stalls the active-set kernel at its iteration limit; linear_program() then solves it once more
    with the objective tilted by 1e-9 and reports the value for the objective asked for: the oracle's value, a feasible
    point, no RuntimeError."""
    model = golden("c2_model")
    A = np.hstack((model["Ax"], model["Au"])); b = model["b"]
    nrm = np.linalg.norm(A, axis=1)
    Ac = np.hstack((A, nrm[:, None]))
    f = np.zeros(A.shape[1] + 1); f[-1] = -1.
    sol = _lp()(f, Ac, b)
    ref = oracle.linear_program(f, Ac, b)
    assert sol['min'] is not None and ref['min'] is not None
    assert abs(sol['min'] - ref['min']) < 1e-6 * max(1., abs(ref['min']))
    assert (Ac.dot(sol['argmin']) - b).max() < 1e-7
    from pympc_amd.geometry.polyhedron import Polyhedron
    assert abs(Polyhedron(A.copy(), b.copy()).radius + ref['min']) < 1e-6
    # the same for callers of the batched form that read the statuses anyway (minimal_facets, mcais): retry_stalled patches
    # the host copies of status and min
    from pympc_amd.optimization.solvers import hip_lp
    F = np.vstack((f, np.random.RandomState(1).randn(Ac.shape[1])))          # the stalled LP and a generic one
    res = hip_lp.linear_program_batch(F, Ac, b, want_mult=False, want_active=False)
    st0 = res.numpy('status').copy(); mn0 = res.numpy('min').copy()
    if st0[0] == 3:
        assert hip_lp.retry_stalled(res, F, Ac, b) == 1
    assert res.numpy('status')[0] == 0 and abs(res.numpy('min')[0] - ref['min']) < 1e-6
    assert res.numpy('status')[1] == st0[1] and (st0[1] != 0 or res.numpy('min')[1] == mn0[1])
