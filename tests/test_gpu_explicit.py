"""GPU parity of the explicit-MPC side: critical-region construction, point location and affine-law evaluation
(pympc_point_location; config C4).

Pins:
  * the reference's 1-D mpQP known-answer test (pympc/test/test_optimization/test_parametric_programs.py:15-71);
  * examples/linear_mpc.ipynb outputs held by the reference: 71 critical regions (cell 21), V*([.39,.39]) and u* of the
    explicit solution (cell 25) -- tests/golden/nb_solves.npz;
  * implicit == explicit on feasible points, None / NaN on infeasible ones (test_controllers.py:103-163);
  * the CPU oracle's restatement of the linear scan (oracle_point_location): region indices bit-exact, u and V to
    1e-9, on random points and on points placed on region boundaries.
explicit_solve itself cannot be run on the reference in the build container (scipy 1.15.3's nnls makes it
raise), so there is no golden region list: "parity unpinned" beyond the anchors above for the ORDER of the regions.
"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _mpqp_1d():
    from pympc_amd.optimization.parametric_programs import MultiParametricQuadraticProgram
    H = {'uu': np.array([[1.]]), 'xx': np.zeros((1, 1)), 'ux': np.zeros((1, 1))}
    f = {'u': np.zeros(1), 'x': np.zeros(1)}
    A = {'u': np.array([[1.], [-1.], [0.], [0.]]), 'x': np.array([[-1.], [1.], [1.], [-1.]])}
    return MultiParametricQuadraticProgram(H, f, 0., A, np.array([1., 1., 2., 2.]))


def test_reference_1d_mpqp_kats():
    mpqp = _mpqp_1d()
    cr = mpqp.explicit_solve_given_active_set([])
    for v in (cr._V['xx'], cr._V['x'], cr._V['0'], cr._u['x'], cr._u['0']):
        assert abs(float(np.squeeze(v))) < 1e-7
    cr = mpqp.explicit_solve_given_active_set([0])
    got = [float(np.squeeze(v)) for v in (cr._V['xx'], cr._V['x'], cr._V['0'], cr._u['x'], cr._u['0'])]
    assert np.allclose(got, [1., 1., .5, 1., 1.], atol=1e-7)
    cr = mpqp.explicit_solve_given_point(np.array([1.5]))
    got = [float(np.squeeze(v)) for v in (cr._V['xx'], cr._V['x'], cr._V['0'], cr._u['x'], cr._u['0'])]
    assert np.allclose(got, [1., -1., .5, 1., -1.], atol=1e-7)
    sol = mpqp.solve(np.array([1.5]))
    assert abs(sol['min'] - .125) < 1e-7 and np.allclose(sol['argmin'], .5) and sol['active_set'] == [1]
    exp_sol = mpqp.explicit_solve()
    for x in (.5, 1.5, -1.5, 2.5, -2.5):
        x = np.array([x])
        sol = mpqp.solve(x)
        if sol['min'] is not None:
            assert abs(sol['min'] - exp_sol.V(x)) < 1e-7
            np.testing.assert_array_almost_equal(sol['argmin'], exp_sol.u(x))
            np.testing.assert_array_almost_equal(sol['multiplier_inequality'], exp_sol.p(x))
        else:
            assert exp_sol.V(x) is None and exp_sol.u(x) is None and exp_sol.p(x) is None
    # batched forms of the same look-ups
    X = np.array([[.5], [1.5], [-1.5], [2.5], [-2.5]])
    idx = exp_sol.get_critical_region(X)
    assert list(idx >= 0) == [True, True, True, False, False]
    assert np.allclose(exp_sol.V(X)[:3], [0., .125, .125], atol=1e-7) and np.all(np.isnan(exp_sol.V(X)[3:]))
    assert np.all(np.isnan(exp_sol.p(X)[3:]))


@pytest.fixture(scope="module")
def nb():
    from test_gpu_condense import _linear_controller
    g = golden("nb_model")
    ctrl = _linear_controller(g)
    ctrl.store_explicit_solution()
    return ctrl


def test_notebook_explicit_solution_anchors(nb):
    rec = golden("nb_solves")
    es = nb.explicit_solution
    assert len(es.critical_regions) == 71                                  # linear_mpc.ipynb cell 21
    x0 = rec['nb_x0']
    u, V = nb.feedforward_explicit(x0)
    assert abs(V - float(rec['nb_V_explicit'])) < 1e-6 * float(rec['nb_V_explicit'])      # cell 25
    assert np.allclose(np.concatenate(u), rec['nb_u_implicit'], rtol=1e-6, atol=1e-6)
    assert np.allclose(nb.feedback_explicit(x0), rec['nb_u_implicit'][:1], rtol=1e-6, atol=1e-6)
    # distinct active sets, every region non-empty
    assert len(set(tuple(cr.active_set) for cr in es.critical_regions)) == 71


def test_explicit_equals_implicit(nb):
    """test_controllers.py:146-163: same optimum from the table and from the online QP; both None when infeasible."""
    rec = golden("nb_solves")
    X = rec['X']; valid = rec['oracle_valid'].astype(bool); feas = rec['feasible'].astype(bool)
    U, V, idx = nb.feedforward_explicit(X)
    inside = idx >= 0
    # points well inside / outside the feasible set agree exactly; the golden grid has points ON its boundary
    # (|x_i| = .5), where covered-by-a-region (tol 1e-7) and QP-feasible (residual test) may legitimately differ
    edge = np.abs(np.abs(X).max(axis=1) - .5) < 1e-9
    assert np.array_equal(inside[valid & ~edge], feas[valid & ~edge])
    sel = valid & feas & inside
    assert sel.sum() > 100
    assert np.allclose(V[sel], rec['min'][sel], rtol=1e-6, atol=1e-6)
    assert np.allclose(U.reshape(len(X), -1)[sel], rec['argmin'][sel], rtol=1e-6, atol=1e-6)
    assert np.all(np.isnan(V[~inside])) and np.all(np.isnan(U[~inside]))
    # scalar API on an infeasible point
    bad = np.where(valid & ~feas & ~inside)[0]
    if bad.size:
        assert nb.feedforward_explicit(X[bad[0]]) == (None, None) and nb.feedback_explicit(X[bad[0]]) is None
    # multipliers from the table
    Pm = nb.explicit_solution.p(X)
    assert np.allclose(Pm[sel], rec['multiplier_inequality'][sel], rtol=1e-6, atol=1e-6)


def _table(es, nu=None):
    t = es.table(nu)
    return {k: t[k] for k in ('A', 'b', 'row_ptr', 'ux', 'u0', 'Vxx', 'Vx', 'V0')}


def test_point_location_vs_oracle(nb, oracle):
    es = nb.explicit_solution
    rng = np.random.RandomState(0)
    X = rng.uniform(-.6, .6, (200000, 2))
    # points on facets of regions: ties of the first-hit rule and of the 1e-7 tolerance
    crs = es.critical_regions
    onb = []
    for cr in crs[:20]:
        c = cr.polyhedron.center
        for i in range(0, cr.A.shape[0], 7):
            a = cr.A[i]
            t = (cr.b[i] - a.dot(c)) / a.dot(a)
            for eps in (0., 5e-8, 2e-7, -5e-8):
                onb.append(c + (t + eps) * a)
    X = np.vstack((X, np.array(onb)))
    for nu in (None, 1):
        idx, u, V = es.locate(X, nu=nu)
        idx = idx.cpu().numpy(); u = u.cpu().numpy(); V = V.cpu().numpy()
        ridx, ru, rV = oracle.point_location(_table(es, nu), X)
        # a point whose max(Ax - b) sits within rounding of the tolerance in some region may be classified either way
        A, b, rp = es.table(nu)['A'], es.table(nu)['b'], es.table(nu)['row_ptr']
        s = X.dot(A.T) - b
        tie = np.zeros(len(X), bool)
        for r in range(len(rp) - 1):
            tie |= np.abs(s[:, rp[r]:rp[r + 1]].max(axis=1) - 1e-7) < 1e-12
        assert tie.sum() < 50
        ok = ~tie
        assert np.array_equal(idx[ok], ridx[ok])
        cov = ok & (ridx >= 0)
        assert np.allclose(u[cov], ru[cov], rtol=1e-9, atol=1e-9) and np.allclose(V[cov], rV[cov], rtol=1e-9, atol=1e-9)
        unc = ok & (ridx < 0)
        assert np.all(np.isnan(V[unc])) and np.all(np.isnan(u[unc]))
    assert (ridx >= 0).mean() > .3 and (ridx < 0).sum() > 1000
    # the bounding-box pruning only skips work: identical outputs with and without it, ties included
    c = es.locate(X, nu=1, use_boxes=False, use_grid=False)      # plain list scan
    for kw in (dict(use_boxes=True, use_grid=False), dict(use_boxes=False, use_grid=True), dict(use_sure=False), dict()):
        a = es.locate(X, nu=1, **kw)
        assert np.array_equal(a[0].cpu().numpy(), c[0].cpu().numpy())
        assert np.array_equal(a[1].cpu().numpy(), c[1].cpu().numpy(), equal_nan=True)
        assert np.array_equal(a[2].cpu().numpy(), c[2].cpu().numpy(), equal_nan=True)
    grid = es.table(1)['grid']
    G = es.GRID_CELLS_PER_AXIS[2]
    assert grid is not None and tuple(grid['masks'].shape) == (G * G, 2) and tuple(grid['sure'].shape) == (G * G,)
    # decided cells: most of the grid, and each one is right at its centre and corners (first hit of the plain scan)
    sure = grid['sure'].cpu().numpy()
    assert (sure > -2).mean() > .5 and (sure >= 0).any() and (sure == -1).any()
    lo, inv, hi = grid['geom'][:2], grid['geom'][2:4], grid['geom'][4:6]
    cc = np.stack(np.meshgrid(np.arange(G), np.arange(G), indexing='ij'), axis=-1).reshape(-1, 2)
    for off in ((.5, .5), (1e-6, 1e-6), (1 - 1e-6, 1e-6), (1e-6, 1 - 1e-6), (1 - 1e-6, 1 - 1e-6)):
        Xc = lo + (cc + np.array(off)) / inv
        plain = es.locate(Xc, nu=1, use_boxes=False, use_grid=False)[0].cpu().numpy()
        dec = sure > -2
        assert np.array_equal(plain[dec], sure[dec])
    # points outside the grid (clamped into the border cells) and far outside it
    far = np.vstack((X[:2000] * 3.0, X[:2000] * 1e6, X[:100] * 0.0))
    a = es.locate(far, nu=1); c = es.locate(far, nu=1, use_boxes=False, use_grid=False)
    assert np.array_equal(a[0].cpu().numpy(), c[0].cpu().numpy())
    box = es.table(1)['box']
    assert np.isfinite(box).all() and np.all(box[:, :2] < box[:, 2:])
    # scalar look-ups
    i = int(np.where(ridx >= 0)[0][0])
    assert es.get_critical_region(X[i]) is crs[ridx[i]]
    j = int(np.where(ridx < 0)[0][0])
    assert es.get_critical_region(X[j]) is None and es.u(X[j]) is None and es.V(X[j]) is None


def test_table_from_samples_matches_facet_flipping(nb):
    """The sample-based table builder finds the same critical regions (as active sets) wherever it has a sample."""
    g = np.linspace(-.5, .5, 201)
    X = np.array([[a, b] for a in g for b in g])
    es2 = nb.mpqp.explicit_solve_from_samples(X)
    s1 = set(tuple(cr.active_set) for cr in nb.explicit_solution.critical_regions)
    s2 = set(tuple(cr.active_set) for cr in es2.critical_regions)
    assert s2 <= s1 and len(s2) >= 60
    idx1, u1, V1 = nb.explicit_solution.locate(X)
    idx2, u2, V2 = es2.locate(X)
    both = ((idx1 >= 0) & (idx2 >= 0)).cpu().numpy()
    assert both.mean() > .3
    assert np.allclose(V1.cpu().numpy()[both], V2.cpu().numpy()[both], rtol=1e-7, atol=1e-7)


def test_full_size_grid_properties(nb):
    """Config C4 at full size: 10 M grid points (3163 x 3163 over X); size-independent properties."""
    import torch
    es = nb.explicit_solution
    n = 3163
    g = torch.linspace(-.5, .5, n, dtype=torch.float64, device='cuda')
    X = torch.stack(torch.meshgrid(g, g, indexing='ij'), dim=-1).reshape(-1, 2).contiguous()
    idx, u, V = es.locate(X, nu=1)
    assert idx.shape[0] == n * n
    nreg = len(es.critical_regions)
    assert int(idx.min()) >= -1 and int(idx.max()) < nreg
    cov = idx >= 0
    assert bool(torch.isnan(V[~cov]).all()) and bool(torch.isfinite(V[cov]).all())
    # idempotence / determinism
    idx2, u2, V2 = es.locate(X, nu=1)
    assert torch.equal(idx, idx2) and torch.equal(V[cov], V2[cov])
    # central symmetry of the problem: region partition and value function are symmetric under x -> -x
    Vf = V.reshape(n, n)
    Vr = torch.flip(Vf, dims=(0, 1))
    both = torch.isfinite(Vf) & torch.isfinite(Vr)
    assert float(both.double().mean()) > .3
    assert float((Vf[both] - Vr[both]).abs().max()) < 1e-6 * float(Vf[both].abs().max())
    # the table agrees with the online QP on a sample of the grid
    sel = torch.randperm(n * n, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))[:20000]
    res = nb.mpqp.solve(X[sel], want_mult=False, want_active=False)
    feas = res.feasible
    c = cov[sel].cpu().numpy()
    Vs = V[sel].cpu().numpy(); us = u[sel].cpu().numpy()
    agree = feas & c
    assert (feas == c).mean() > .999 and agree.sum() > 5000
    assert np.allclose(Vs[agree], res.numpy('min')[agree], rtol=1e-6, atol=1e-6)
    assert np.allclose(us[agree], res.numpy('argmin')[agree][:, :1], rtol=1e-6, atol=1e-6)


def _oracle_locate_threaded(oracle, table, X, threads=16):
    from concurrent.futures import ThreadPoolExecutor
    chunks = np.array_split(np.arange(len(X)), threads)
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda c: oracle.point_location(table, X[c]), chunks))
    return tuple(np.concatenate([q[i] for q in parts]) for i in range(3))


def _tie_mask(table, X, chunk=20000):
    """Points whose max(Ax - b) sits within rounding (1e-12) of the 1e-7 tolerance in some region: either side is right."""
    A, b, rp = table['A'], table['b'], table['row_ptr']
    tie = np.zeros(len(X), bool)
    for lo in range(0, len(X), chunk):
        s = X[lo:lo + chunk].dot(A.T) - b
        seg = np.maximum.reduceat(s, rp[:-1], axis=1)
        tie[lo:lo + chunk] = (np.abs(seg - 1e-7) < 1e-12).any(axis=1)
    return tie


def test_full_size_grid_vs_oracle(nb, oracle):
    """Config C4 at full size against the CPU oracle's linear scan (parametric_programs.py:442-463), index by index: on
    the 3163 x 3163 grid nearly every wavefront's 64 points are neighbours, so this is where the hull-corner decisions of
    point_location_kernel (all points of the wave inside / one row violated at every corner) decide almost everything;
    random points (test_point_location_vs_oracle) never reach them."""
    import torch
    es = nb.explicit_solution
    n = 3163
    g = torch.linspace(-.5, .5, n, dtype=torch.float64, device='cuda')
    Xd = torch.stack(torch.meshgrid(g, g, indexing='ij'), dim=-1).reshape(-1, 2).contiguous()
    idx, u, V = es.locate(Xd, nu=1)
    # (the decided cells change which lanes scan, never the result: bitwise the same without them)
    idx0, u0, V0 = es.locate(Xd, nu=1, use_sure=False)
    assert torch.equal(idx, idx0) and torch.equal(torch.nan_to_num(u, nan=-7.), torch.nan_to_num(u0, nan=-7.))
    assert torch.equal(torch.nan_to_num(V, nan=-7.), torch.nan_to_num(V0, nan=-7.))
    del idx0, u0, V0
    idx = idx.cpu().numpy(); u = u.cpu().numpy(); V = V.cpu().numpy()
    X = Xd.cpu().numpy()
    table = _table(es, 1)
    ridx, ru, rV = _oracle_locate_threaded(oracle, table, X)
    # a point whose max(Ax - b) sits within rounding (1e-12) of the 1e-7 tolerance in some region may fall either way:
    # every disagreement must be such a point, and there must be few
    mis = np.where(idx != ridx)[0]
    print("10 M grid: %d points differ from the oracle (all must be 1e-12 ties of the tolerance), %.1f %% covered"
          % (mis.size, 100. * (ridx >= 0).mean()))
    assert mis.size <= 100
    if mis.size:
        assert _tie_mask(es.table(1), X[mis]).all()
    ok = idx == ridx
    cov = ok & (ridx >= 0)
    assert cov.mean() > .3
    assert np.allclose(u[cov], ru[cov], rtol=1e-9, atol=1e-9) and np.allclose(V[cov], rV[cov], rtol=1e-9, atol=1e-9)
    unc = ok & (ridx < 0)
    assert np.all(np.isnan(V[unc])) and np.all(np.isnan(u[unc]))


@pytest.mark.parametrize("p", [3, 4])
def test_point_location_higher_dimensional_tables_vs_oracle(oracle, p):
    """Tables with p = 3 and p = 4 parameters (the cart-pole controller of config C2 has p = 4): the hull-corner decisions
    of point_location_kernel test 2^p corners per row.  Clusters of 64 neighbouring points (one wavefront each: the hull
    is tiny, so nearly every region is decided by its corners) plus uniformly random points, against the oracle's scan."""
    from pympc_amd.optimization.parametric_programs import MultiParametricQuadraticProgram
    rng = np.random.RandomState(40 + p)
    n, m = 4, 18
    L = rng.randn(n, n)
    H = {'uu': L.dot(L.T) + n * np.eye(n), 'ux': rng.randn(n, p), 'xx': np.eye(p)}
    f = {'u': rng.randn(n), 'x': rng.randn(p)}
    A = {'u': rng.randn(m, n), 'x': rng.randn(m, p)}
    b = np.abs(rng.randn(m)) + 1.
    # a box on the parameters keeps every critical region bounded (the table's bounding boxes need that)
    A = {'u': np.vstack((A['u'], np.zeros((2 * p, n)))), 'x': np.vstack((A['x'], np.eye(p), -np.eye(p)))}
    b = np.concatenate((b, 2. * np.ones(2 * p)))
    mpqp = MultiParametricQuadraticProgram(H, f, .3, A, b)
    es = mpqp.explicit_solve_from_samples(rng.uniform(-2., 2., (30000, p)))
    assert len(es.critical_regions) >= 8
    centres = rng.uniform(-1.9, 1.9, (4000, p))
    X = (centres[:, None, :] + 2e-3 * rng.uniform(-1., 1., (4000, 64, p))).reshape(-1, p)
    X = np.vstack((X, rng.uniform(-2.2, 2.2, (50000, p))))
    idx, u, V = es.locate(X, nu=2)
    idx = idx.cpu().numpy(); u = u.cpu().numpy(); V = V.cpu().numpy()
    ridx, ru, rV = oracle.point_location(_table(es, 2), X)
    mis = np.where(idx != ridx)[0]
    assert mis.size <= 20
    if mis.size:
        assert _tie_mask(es.table(2), X[mis]).all()
    ok = idx == ridx
    cov = ok & (ridx >= 0)
    assert cov.mean() > .2 and (ridx < 0).sum() > 100
    assert np.allclose(u[cov], ru[cov], rtol=1e-9, atol=1e-9) and np.allclose(V[cov], rV[cov], rtol=1e-9, atol=1e-9)
    # identical with every pruning device switched off (plain list scan)
    c = es.locate(X, nu=2, use_boxes=False, use_grid=False)
    assert np.array_equal(c[0].cpu().numpy(), idx)
    # ... and with the per-cell pass forced on and off (launch policy: it runs when most cells are decided or listed);
    # the p = 4 table has regions without a finite box: border cells stay with the scan, no point counts as outside
    grid = es.table(2)['grid']
    assert grid is not None
    for force in (True, False):
        grid['use_sure'] = force
        d = es.locate(X, nu=2)
        assert np.array_equal(d[0].cpu().numpy(), idx)
        assert np.array_equal(d[1].cpu().numpy(), u, equal_nan=True) and np.array_equal(d[2].cpu().numpy(), V, equal_nan=True)


def test_nan_points_and_empty_table(nb, oracle):
    """A point with a NaN coordinate lies in no region (contains(): max(A x - b) <= tol is False, polyhedron.py:596),
    also when the other 63 points of its wavefront all lie in one region (the hull-corner shortcut); a table without
    regions leaves every point uncovered and reads nothing."""
    import ctypes
    import torch
    from pympc_amd import _hip
    es = nb.explicit_solution
    # a tight cluster inside one region: the whole wavefront shares it
    c = es.critical_regions[0].polyhedron.center
    X = np.tile(c, (4096, 1)) + 1e-6 * np.random.RandomState(0).randn(4096, 2)
    bad = np.arange(5, 4096, 37)
    X[bad, bad % 2] = np.nan
    idx, u, V = es.locate(X, nu=1)
    idx = idx.cpu().numpy(); V = V.cpu().numpy(); u = u.cpu().numpy()
    good = np.setdiff1d(np.arange(4096), bad)
    ridx, ru, rV = oracle.point_location(_table(es, 1), X[good])
    assert np.all(idx[bad] == -1) and np.all(np.isnan(V[bad])) and np.all(np.isnan(u[bad]))
    assert np.array_equal(idx[good], ridx) and (ridx >= 0).all()
    assert np.allclose(V[good], rV, rtol=1e-9, atol=1e-9)
    # empty table through the C ABI
    lib = _hip.load_library()
    Xd = torch.as_tensor(np.random.RandomState(1).randn(1000, 2)).cuda()
    oi = torch.zeros(1000, dtype=torch.int32, device='cuda'); ou = torch.zeros((1000, 1), dtype=torch.float64, device='cuda')
    oV = torch.zeros(1000, dtype=torch.float64, device='cuda')
    rp = torch.zeros(1, dtype=torch.int32, device='cuda')
    _hip.check(lib.pympc_point_location(0, 0, 2, 1, None, None, _hip.ptr(rp), None, None, None, None, None, None, None, 0, None,
                                        None, None, 1, _hip.ptr(Xd), 1000, 1e-7, _hip.ptr(oi), _hip.ptr(ou), _hip.ptr(oV), _hip.stream_ptr()))
    torch.cuda.synchronize()
    assert bool((oi == -1).all()) and bool(torch.isnan(oV).all()) and bool(torch.isnan(ou).all())


def test_table_is_reused_across_input_counts_and_boxes_come_in_chunks(nb, monkeypatch):
    """The look-up table's regions, boxes and cell grid do not depend on how many inputs are evaluated: asking for
    another nu rebuilds the laws only.  Bounding boxes of a table with more regions than one LP launch takes distinct
    matrices (65535 / 2p) are computed in chunks of whole regions: same boxes as in one launch."""
    from pympc_amd.optimization import parametric_programs as pp
    es = nb.explicit_solution
    es._table = None
    t1 = es.table(nu=1)
    calls = []
    orig = pp.ExplicitSolution._bounding_boxes
    monkeypatch.setattr(pp.ExplicitSolution, "_bounding_boxes", staticmethod(lambda t, tol: calls.append(1) or orig(t, tol)))
    t_all = es.table()                                   # all inputs of the horizon
    assert not calls and t_all['nu'] > 1 and t_all['box'] is t1['box'] and t_all['dev']['box'] is t1['dev']['box']
    X = np.random.RandomState(3).uniform(-.4, .4, (5000, 2))
    i_all, u_all, V_all = es.locate(X)
    es._table = None
    i_one, u_one, V_one = es.locate(X, nu=1)
    assert np.array_equal(i_all.cpu().numpy(), i_one.cpu().numpy())
    ok = i_one.cpu().numpy() >= 0
    assert np.array_equal(u_all.cpu().numpy()[ok][:, :1], u_one.cpu().numpy()[ok])
    # chunking
    monkeypatch.undo()
    ref_box = orig(es.table(), 1e-7)
    t = es.table()
    # (a region's box only depends on its own rows: the table split by hand into chunks of 10 regions gives the same boxes)
    parts = []
    for r0 in range(0, t['nreg'], 10):
        r1 = min(t['nreg'], r0 + 10)
        rp = t['row_ptr'][r0:r1 + 1] - t['row_ptr'][r0]
        sub = dict(nreg=r1 - r0, p=t['p'], row_ptr=rp, A=t['A'][t['row_ptr'][r0]:t['row_ptr'][r1]], b=t['b'][t['row_ptr'][r0]:t['row_ptr'][r1]])
        parts.append(orig(sub, 1e-7))
    assert np.allclose(np.vstack(parts), ref_box, rtol=1e-12, atol=1e-12)
