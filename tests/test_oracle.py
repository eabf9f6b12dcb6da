"""CPU tests (no GPU): pin the oracle (oracle/pnnls_oracle.c) to
  (a) the reference's own known-answer tests (pympc/test/test_optimization/test_solvers.py:12-345,
      test_programs.py:14-122) -- inputs and expected values transcribed as data, assertions as the reference makes them;
  (b) the golden vectors written by tests/golden/make_golden.py from the reference run in the build container
      (KKT-certified; uncertified / degenerate records are excluded from the bit-exact claims);
and check the numpy model of the HIP kernel's update formulas (tests/kernel_model.py) against the oracle.
"""
import numpy as np
import pytest

from conftest import golden, unpack_golden_active


# ------------------------------------------------------------------ (a) reference KATs
def _lp_kats(linear_program):
    f = np.ones(2)
    A = np.array([[1., 0.], [-1., 0.], [0., 1.], [0., -1.]]); b = np.array([5., 5., 1., 1.])
    sol = linear_program(f, A, b)                                              # test_solvers.py:15-37
    assert abs(sol['min'] + 6.) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], [-5., -1.])
    assert sol['active_set'] == [1, 3]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [0., 1., 0., 1.])
    assert sol['multiplier_equality'] is None
    A = -np.eye(2); b = np.zeros(2)
    sol = linear_program(f, A, b)                                              # :39-61
    assert abs(sol['min']) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], np.zeros(2))
    assert sol['active_set'] == [0, 1]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], np.ones(2))
    C = np.array([[2., 1.]]); d = np.array([2.])
    sol = linear_program(f, A, b, C, d)                                        # :63-86
    assert abs(sol['min'] - 1.) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], [1., 0.])
    assert sol['active_set'] == [1]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [0., .5])
    np.testing.assert_array_almost_equal(sol['multiplier_equality'], [-.5])
    A = np.array([[0., 1.], [0., -1.]]); b = np.array([0., -1.])
    assert all(v is None for v in linear_program(f, A, b).values())            # :88-93 infeasible
    A = np.array([[1., 0.], [0., 1.], [0., -1.]]); b = np.array([0., 0., 1.])
    assert all(v is None for v in linear_program(f, A, b).values())            # :95-100 unbounded
    C = np.array([[0., 1.]]); d = np.array([-.1])
    assert all(v is None for v in linear_program(f, A, b, C, d).values())      # :102-107
    C = np.array([[1., -1.]]); d = np.array([0.])
    sol = linear_program(f, A, b, C, d)                                        # :109-132
    assert abs(sol['min'] + 2.) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], -np.ones(2))
    assert sol['active_set'] == [2]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [0., 0., 2.])
    np.testing.assert_array_almost_equal(sol['multiplier_equality'], [-1.])
    f = np.array([0., 1.]); A = np.array([[0., -1.]]); b = np.zeros(1)
    sol = linear_program(f, A, b)                                              # :134-157
    assert abs(sol['min']) < 1e-7 and abs(sol['argmin'][1]) < 1e-7
    assert sol['active_set'] == [0]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [1.])
    A = np.array([[1., 1., 1.], [-1., 1., 1.], [1., -1., 1.], [1., 1., -1.], [-1., -1., 1.], [1., -1., -1.],
                  [-1., 1., -1.], [-1., -1., -1.], [1., 0., 0.]])
    b = np.concatenate((np.ones(8), [.5]))
    sol = linear_program(np.array([-1., 0., 0.]), A, b)                        # :159-189
    assert abs(sol['min'] + .5) < 1e-7 and abs(sol['argmin'][0] - .5) < 1e-7
    assert 8 in sol['active_set']
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], np.concatenate((np.zeros(8), [1.])))
    sol = linear_program(np.array([-1., -.1, 0.]), A, b)                       # :191-212
    assert abs(sol['min'] + .55) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], [.5, .5, 0.])
    assert sol['active_set'] == [0, 3, 8]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [.05, 0., 0., .05, 0., 0., 0., 0., .9])
    sol = linear_program(np.array([1., 1., 0.]), A, b)                         # :214-231
    assert abs(sol['min'] + 1.) < 1e-7 and abs(sol['argmin'][0] + sol['argmin'][1] + 1.) < 1e-7
    assert 4 in sol['active_set'] and 7 in sol['active_set']
    assert abs(sol['multiplier_inequality'][4] - .5) < 1e-7 and abs(sol['multiplier_inequality'][7] - .5) < 1e-7
    sol = linear_program(np.array([1., 0.]), np.array([[1., 0.], [-1., 0.]]), np.zeros(2))   # :233-256
    assert abs(sol['min']) < 1e-7 and abs(sol['argmin'][0]) < 1e-7
    assert sol['active_set'] == [1]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [0., 1.])


def _qp_kats(quadratic_program):
    H = np.eye(2); f = np.ones(2); A = -np.eye(2); b = -np.ones(2)
    sol = quadratic_program(H, f, A, b)                                        # :265-289
    assert abs(sol['min'] - 3.) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], [1., 1.])
    assert sol['active_set'] == [0, 1]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [2., 2.])
    assert sol['multiplier_equality'] is None
    C = np.array([[0., 1.]]); d = np.array([2.])
    sol = quadratic_program(H, f, A, b, C, d)                                  # :291-314
    assert abs(sol['min'] - 5.5) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], [1., 2.])
    assert sol['active_set'] == [0]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [2., 0.])
    np.testing.assert_array_almost_equal(sol['multiplier_equality'], [-3.])
    assert all(v is None for v in quadratic_program(H, f, A, b, C, -d).values())   # :316-319
    sol = quadratic_program(np.eye(2), np.zeros(2), np.array([[1., 0.], [-1., 0.]]), np.array([1., -1.]))  # :321-345
    assert abs(sol['min'] - .5) < 1e-7 and abs(sol['argmin'][0] - 1.) < 1e-7
    assert sol['active_set'] == [1]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], [0., 1.])


def test_oracle_reference_lp_kats(oracle):
    _lp_kats(oracle.linear_program)


def test_oracle_reference_qp_kats(oracle):
    _qp_kats(oracle.quadratic_program)


def test_oracle_errors(oracle):
    with pytest.raises(ValueError):
        oracle.quadratic_program(np.eye(2), np.ones(2), -np.eye(2), -np.ones(2), C=np.ones((1, 2)))
    with pytest.raises(np.linalg.LinAlgError):
        oracle.quadratic_program(np.diag([1., -1.]), np.ones(2), -np.eye(2), -np.ones(2))


# ------------------------------------------------------------------ (b) golden vectors from the reference
def test_oracle_nnls_vs_scipy_golden(oracle):
    g = golden("nnls_cases")
    for k in range(int(g["count"])):
        if not bool(g["ok%d" % k]):
            continue
        x, rn, _ = oracle.nnls(g["A%d" % k], g["b%d" % k])
        assert np.allclose(x, g["x%d" % k], rtol=0, atol=1e-10)
        assert abs(rn - float(g["r%d" % k])) < 1e-10


def test_oracle_nnls_vs_scipy_regression_problems(oracle):
    """scipy's own NNLS regression problems (scipy/optimize/tests/test_nnls.py, recorded by make_golden.py)."""
    g = golden("scipy_nnls_tests")
    n = 0
    for k in range(int(g["count"])):
        if not bool(g["ok%d" % k]):
            continue
        A, b = g["A%d" % k], g["b%d" % k]
        x, rn, _ = oracle.nnls(A, b)
        scale = max(1.0, np.abs(g["x%d" % k]).max())
        assert np.allclose(x, g["x%d" % k], rtol=0, atol=1e-8 * scale), str(g["names"][k])
        assert abs(rn - float(g["r%d" % k])) < 1e-8 * max(1.0, np.linalg.norm(b))
        n += 1
    assert n >= 6


def test_oracle_pnnls_against_numpy_restatement(oracle):
    """pnnls(A, B, c) (pnnls.py:5-39): the projector formulation written out with numpy + the golden-certified nnls."""
    rng = np.random.RandomState(5)
    for rows, nv, nu in ((6, 4, 2), (12, 16, 3), (9, 20, 4), (8, 3, 5)):
        A = rng.randn(rows, nv); B = rng.randn(rows, nu); c = rng.randn(rows)
        v, u, r = oracle.pnnls(A, B, c)
        Bp = np.linalg.pinv(B)
        Bbar = np.eye(rows) - B.dot(Bp)
        v2, r2, _ = oracle.nnls(Bbar.dot(A), Bbar.dot(c))
        assert np.allclose(v, v2, atol=1e-10) and abs(r - r2) < 1e-10
        assert np.allclose(u, -Bp.dot(A.dot(v) - c), atol=1e-10)
        assert abs(np.linalg.norm(A.dot(v) + B.dot(u) - c) - r) < 1e-9


@pytest.mark.parametrize("tag", ["s03", "s10"])
def test_oracle_mpqp_vs_10k_reference_states(oracle, tag):
    """10 000 reference-run C2 states per distribution (compact records, make_golden.py::gen_c2big)."""
    model = golden("c2_model"); rec = golden("c2_big_" + tag)
    m = model["Au"].shape[0]
    ob = oracle.mpqp_solve_batch(model, rec["X"])
    trusted = rec["oracle_valid"] & ~rec["degenerate"]
    assert (~trusted).sum() <= 10
    feas = ob["status"] == 0
    assert np.array_equal(feas[trusted], rec["feasible"][trusted])
    sel = trusted & feas
    assert np.array_equal(ob["active"][sel], unpack_golden_active(rec, m)[sel])
    assert np.max(np.abs(ob["argmin"][sel][:, :3] - rec["argmin"][sel]) / np.maximum(1, np.abs(rec["argmin"][sel]))) < 1e-6
    assert np.max(np.abs(ob["min"][sel] - rec["min"][sel]) / np.maximum(1, np.abs(rec["min"][sel]))) < 1e-6
    lam = ob["multiplier_inequality"][sel]
    assert np.max(np.abs(lam.max(axis=1) - rec["multiplier_max"][sel]) / np.maximum(1, rec["multiplier_max"][sel])) < 1e-6


@pytest.mark.parametrize("model_name,solve_name", [("c1_model", "c1_solves"), ("nb_model", "nb_solves"),
                                                   ("c2_model", "c2_solves_s03"), ("c2_model", "c2_solves_s10")])
def test_oracle_mpqp_vs_reference_golden(oracle, model_name, solve_name):
    model = golden(model_name); rec = golden(solve_name)
    m = model["Au"].shape[0]
    ob = oracle.mpqp_solve_batch(model, rec["X"])
    trusted = rec["oracle_valid"] & ~rec["degenerate"]
    feas = ob["status"] == 0
    assert np.array_equal(feas[trusted], rec["feasible"][trusted])
    sel = trusted & feas
    assert np.array_equal(ob["active"][sel], unpack_golden_active(rec, m)[sel])
    assert np.max(np.abs(ob["argmin"][sel] - rec["argmin"][sel]) / np.maximum(1, np.abs(rec["argmin"][sel]))) < 1e-6
    assert np.max(np.abs(ob["min"][sel] - rec["min"][sel]) / np.maximum(1, np.abs(rec["min"][sel]))) < 1e-6
    # the records scipy 1.15.3 got wrong: the oracle's answer must satisfy the QP's KKT conditions
    for i in np.where(~rec["oracle_valid"])[0]:
        if not feas[i]:
            continue
        x = rec["X"][i]
        f = model["Hux"].dot(x) + model["fu"]; b = model["b"] - model["Ax"].dot(x)
        u, lam = ob["argmin"][i], ob["multiplier_inequality"][i]
        assert (model["Au"].dot(u) - b).max() < 1e-9 * max(1., np.abs(b).max())
        assert lam.min() >= 0
        assert np.abs(model["Huu"].dot(u) + f + model["Au"].T.dot(lam)).max() < 1e-7 * max(1., np.abs(lam).max())


def test_oracle_lp_vs_reference_golden(oracle):
    g = golden("lp_polytopes")
    for k in range(len(g["lp_f"])):
        if not bool(g["lp_valid"][k]):
            continue
        sol = oracle.linear_program(g["lp_f"][k], g["lp_A"][k], g["lp_b"][k])
        assert (sol['min'] is not None) == bool(g["lp_solvable"][k])
        if sol['min'] is not None:
            assert abs(sol['min'] - g["lp_min"][k]) < 1e-6 * max(1., abs(g["lp_min"][k]))
            assert np.allclose(sol['argmin'], g["lp_argmin"][k], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("m", [30, 60])
def test_oracle_polytope_lps_vs_reference_golden(oracle, m):
    """minimal_facets (polyhedron.py:403-450) and Chebyshev LP (:727-768) restated with the oracle's LP."""
    g = golden("lp_polytopes")
    A = g["m%d_A" % m]; b = g["m%d_b" % m]
    minimal = list(range(m)); vals = []
    for i in range(m):
        rel = b.copy(); rel[i] += 1.
        sol = oracle.linear_program(-A[i], A[minimal], rel[minimal])
        vals.append(-sol['min'] - b[i])
        if vals[-1] < 1e-7:
            minimal.remove(i)
    assert minimal == list(g["m%d_minimal_facets" % m])
    assert np.allclose(vals, g["m%d_facet_values" % m], rtol=1e-6, atol=1e-7)
    f = np.concatenate((np.zeros(6), [1.]))
    Alp = np.hstack((A, -np.linalg.norm(A, axis=1)[:, None]))
    sol = oracle.linear_program(f, Alp, b)
    assert abs(-sol['min'] - float(g["m%d_radius" % m])) < 1e-6
    assert np.allclose(sol['argmin'][:-1], g["m%d_center" % m], rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ kernel math model vs oracle
def test_kernel_model_matches_oracle(oracle):
    from kernel_model import nnls_invqr, qp_via_nnls
    g = golden("nnls_cases")
    for k in range(int(g["count"])):
        x, rn, _, st = nnls_invqr(g["A%d" % k], g["b%d" % k])
        xo, rno, _ = oracle.nnls(g["A%d" % k], g["b%d" % k])
        assert st == 0 and np.allclose(x, xo, rtol=0, atol=1e-10) and abs(rn - rno) < 1e-10
    model = golden("c1_model"); rec = golden("c1_solves")
    X = rec["X"][:120]
    ob = oracle.mpqp_solve_batch(model, X)
    for i, x in enumerate(X):
        r = qp_via_nnls(model["Huu"], model["Hux"].dot(x) + model["fu"], model["Au"], model["b"] - model["Ax"].dot(x))
        assert (r["status"] == 0) == (ob["status"][i] == 0)
        if r["status"] == 0:
            assert np.array_equal(r["active"], ob["active"][i])
            assert np.allclose(r["argmin"], ob["argmin"][i], rtol=1e-8, atol=1e-9)


def test_point_location_oracle_semantics(oracle):
    """First region in list order wins; boundary within tol belongs to the earlier region (polyhedron.py:596)."""
    A = np.array([[1.], [-1.], [1.], [-1.]]); b = np.array([0., 1., 1., 0.])   # regions [-1,0] and [0,1] on the line
    table = dict(A=A, b=b, row_ptr=np.array([0, 2, 4]), ux=np.array([[[2.]], [[3.]]]), u0=np.array([[1.], [0.]]),
                 Vxx=np.array([[[2.]], [[4.]]]), Vx=np.array([[0.], [1.]]), V0=np.array([0., 1.]))
    X = np.array([[-.5], [0.], [5e-8], [2e-7], [.5], [1.5]])
    idx, u, V = oracle.point_location(table, X)
    assert list(idx) == [0, 0, 0, 1, 1, -1]
    assert np.allclose(u[:5, 0], [0., 1., 1. + 1e-7, 6e-7, 1.5])
    assert np.isnan(u[5, 0]) and np.isnan(V[5])
    assert abs(V[4] - (.5 * 4 * .25 + .5 + 1.)) < 1e-12
