"""Raw NNLS and PNNLS on the GPU (pympc_nnls_batch / pympc_pnnls_batch through the C ABI).

NNLS: scipy.optimize.nnls as the reference calls it (pnnls.py:36,218) -- golden instances with scipy's certified
answers (tests/golden/nnls_cases.npz) and scipy's own regression problems (tests/golden/scipy_nnls_tests.npz, gh-20302,
gh-21021 ...; problems with more than 64 rows are reduced by a QR factorisation first, which leaves the minimiser
unchanged; square problems beyond 64 rows are outside the wave kernel and only run against the oracle on the CPU).
PNNLS: pnnls(A, B, c) of pnnls.py:5-39 against the oracle's restatement, on random instances, on rank-deficient B and
on the LP-shaped systems the reference builds at pnnls.py:109-124 from the golden LPs.
"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _reduce_rows(A, b):
    """min |A x - b| has the minimisers of min |R x - Q'b| (A = Q R); residual^2 differs by |b|^2 - |Q'b|^2."""
    if A.shape[0] <= 64:
        return A, b, 0.0
    Q, R = np.linalg.qr(A)
    qb = Q.T.dot(b)
    return R, qb, max(0.0, b.dot(b) - qb.dot(qb))


def test_nnls_golden_instances():
    from pympc_amd.optimization.solvers.hip_lp import nnls
    g = golden("nnls_cases")
    n = 0
    for k in range(int(g["count"])):
        if not bool(g["ok%d" % k]):
            continue
        A, b = g["A%d" % k], g["b%d" % k]
        x, rn = nnls(A, b)
        assert abs(rn - float(g["r%d" % k])) < 1e-9
        assert x.min() >= 0.0 and abs(np.linalg.norm(A.dot(x) - b) - rn) < 1e-9
        if float(g["r%d" % k]) > 1e-9:      # (a zero residual of a wide matrix has many minimisers: scipy's is one of them)
            assert np.allclose(x, g["x%d" % k], rtol=0, atol=1e-9)
        n += 1
    assert n >= 40


def test_nnls_scipy_regression_problems():
    from pympc_amd.optimization.solvers.hip_lp import nnls
    g = golden("scipy_nnls_tests")
    ran = []
    for k in range(int(g["count"])):
        A, b = g["A%d" % k], g["b%d" % k]
        if not bool(g["ok%d" % k]) or (A.shape[0] > 64 and A.shape[1] > 64):
            continue
        Ar, br, extra = _reduce_rows(A, b)
        x, rn = nnls(Ar, br)
        xs = g["x%d" % k]
        rs = float(g["r%d" % k])
        assert abs(np.sqrt(rn * rn + extra) - rs) < 1e-7 * max(1.0, np.linalg.norm(b))
        assert x.min() >= 0.0 and abs(np.linalg.norm(A.dot(x) - b) - rs) < 1e-7 * max(1.0, np.linalg.norm(b))
        if rs > 1e-7 * max(1.0, np.linalg.norm(b)) or np.linalg.matrix_rank(A) == A.shape[1]:   # unique minimiser
            assert np.allclose(x, xs, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(xs).max())), str(g["names"][k])
        ran.append(str(g["names"][k]))
    print("scipy regression problems run on the GPU:", ran)
    assert len(ran) >= 4 and "test_nnls_gh20302" in ran and "test_nnls_gh21021_ex3" in ran


def test_nnls_batch_random_vs_oracle(oracle):
    from pympc_amd.optimization.solvers.hip_lp import nnls_batch
    rng = np.random.RandomState(9)
    for rows, cols in ((1, 4), (2, 2), (7, 30), (21, 370), (40, 41), (64, 200), (30, 700)):
        A = rng.randn(12, rows, cols); b = rng.randn(12, rows)
        x, rn, st = nnls_batch(A, b)
        x, rn, st = x.cpu().numpy(), rn.cpu().numpy(), st.cpu().numpy()
        assert (st == 0).all()
        for k in range(12):
            xo, ro, _ = oracle.nnls(A[k], b[k])
            assert abs(rn[k] - ro) < 1e-9 * max(1.0, ro)
            assert x[k].min() >= 0.0 and abs(np.linalg.norm(A[k].dot(x[k]) - b[k]) - rn[k]) < 1e-9 * max(1.0, ro)
            if ro > 1e-9:
                assert np.allclose(x[k], xo, rtol=0, atol=1e-8 * max(1.0, np.abs(xo).max())), (rows, cols, k)
    # shared matrix, many right-hand sides
    A = rng.randn(11, 78); b = rng.randn(300, 11)
    x, rn, st = nnls_batch(A, b)
    x = x.cpu().numpy()
    for k in (0, 17, 299):
        xo, ro, _ = oracle.nnls(A, b[k])
        assert abs(float(rn[k]) - ro) < 1e-9 and (ro < 1e-9 or np.allclose(x[k], xo, atol=1e-8))


def test_nnls_all_zero_matrix_never_iterates():
    from pympc_amd.optimization.solvers.hip_lp import nnls
    x, rn = nnls(np.zeros((3, 4)), np.ones(3))
    assert np.all(x == 0.0) and abs(rn - np.sqrt(3.0)) < 1e-12


def test_iteration_limit_maps_to_runtime_error():
    """scipy raises RuntimeError('Maximum number of iterations reached.') at maxiter = 3 cols (_nnls.py:91-95) and the
    reference lets it propagate (pnnls.py:36, :218).  No well-posed instance here reaches 3 cols iterations, so the
    limit is lowered through the launch-policy knob (factor 0: the first column that enters is one too many): batch
    APIs report status 3, scalar APIs raise -- nnls, pnnls and quadratic_program."""
    from pympc_amd import _hip
    from pympc_amd.optimization.solvers.hip_lp import nnls, nnls_batch, pnnls
    from pympc_amd.optimization.solvers.hip import quadratic_program
    rng = np.random.RandomState(4)
    A = rng.randn(5, 9); b = rng.randn(5)
    ok = nnls(A, b)
    with _hip.tuned(nnls_itmax_factor=0):
        x, rn, st = nnls_batch(A, np.vstack((b, np.zeros(5))))
        assert st.cpu().numpy().tolist() == [_hip.MAXITER, _hip.OPTIMAL]      # (b = 0: no column ever enters)
        with pytest.raises(RuntimeError, match="Maximum number of iterations"):
            nnls(A, b)
        with pytest.raises(RuntimeError, match="Maximum number of iterations"):
            pnnls(A, rng.randn(5, 2), b)
        with pytest.raises(RuntimeError, match="Maximum number of iterations"):
            quadratic_program(np.eye(2), np.array([-3., 0.]), np.array([[1., 0.]]), np.array([1.]))   # one active constraint
    again = nnls(A, b)                                                        # knob restored: scipy's limit again
    assert np.array_equal(ok[0], again[0]) and _hip.tune("nnls_itmax_factor") == 3


def test_pnnls_random_vs_oracle(oracle):
    from pympc_amd.optimization.solvers.hip_lp import pnnls, pnnls_batch
    rng = np.random.RandomState(5)
    for rows, nv, nu in ((6, 4, 2), (12, 16, 3), (30, 20, 4), (8, 3, 5), (33, 64, 6), (64, 30, 20), (64, 100, 20), (3, 5, 1)):
        A = rng.randn(rows, nv); B = rng.randn(rows, nu); c = rng.randn(rows)
        v, u, r = pnnls(A, B, c)
        vo, uo, ro = oracle.pnnls(A, B, c)
        assert abs(r - ro) < 1e-9 * max(1.0, ro)
        assert v.min() >= 0.0 and abs(np.linalg.norm(A.dot(v) + B.dot(u) - c) - r) < 1e-8 * max(1.0, r)
        if ro > 1e-9:    # (a zero residual with more columns than rows has many minimisers)
            assert np.allclose(v, vo, atol=1e-8 * max(1.0, np.abs(vo).max())), (rows, nv, nu)
            assert np.allclose(u, uo, atol=1e-7 * max(1.0, np.abs(uo).max())), (rows, nv, nu)
    # rank-deficient B: pinv's minimum-norm u
    A = rng.randn(10, 5); B0 = rng.randn(10, 2); B = np.column_stack((B0, B0[:, 0] - 2 * B0[:, 1], np.zeros(10))); c = rng.randn(10)
    v, u, r = pnnls(A, B, c)
    vo, uo, ro = oracle.pnnls(A, B, c)
    assert np.allclose(v, vo, atol=1e-8) and np.allclose(u, uo, atol=1e-8) and abs(r - ro) < 1e-9
    # batch
    A = rng.randn(20, 18, 9); B = rng.randn(20, 18, 3); c = rng.randn(20, 18)
    v, u, rn, st = pnnls_batch(A, B, c)
    assert (st.cpu().numpy() == 0).all()
    for k in (0, 7, 19):
        vo, uo, ro = oracle.pnnls(A[k], B[k], c[k])
        assert abs(float(rn[k]) - ro) < 1e-9
        if ro > 1e-9:
            assert np.allclose(v[k].cpu().numpy(), vo, atol=1e-8) and np.allclose(u[k].cpu().numpy(), uo, atol=1e-8)


def test_pnnls_on_the_lp_systems_of_the_reference():
    """The reference's LP -> PNNLS system (pnnls.py:109-124) built from the golden LPs (solved by the reference in the
    build container): unknowns v = (y, s) >= 0, u = x; solvable iff the residual is below 1e-7 (pnnls.py:137), then
    x = argmin, y = multipliers."""
    from pympc_amd.optimization.solvers.hip_lp import pnnls
    g = golden("lp_polytopes")
    f, A, b = g["lp_f"], g["lp_A"], g["lp_b"]
    n_solv = 0
    for k in range(len(f)):
        if not bool(g["lp_valid"][k]):
            continue
        m, n = A[k].shape
        Ap = np.vstack((np.concatenate((b[k], np.zeros(m))), np.hstack((np.zeros((m, m)), np.eye(m))),
                        np.hstack((A[k].T, np.zeros((n, m))))))
        Bp = np.vstack((f[k], A[k], np.zeros((n, n))))
        cp = np.concatenate((np.zeros(1), b[k], -f[k]))
        v, u, r = pnnls(Ap, Bp, cp)
        assert (r < 1e-7) == bool(g["lp_solvable"][k]), k
        if g["lp_solvable"][k]:
            n_solv += 1
            assert np.allclose(u, g["lp_argmin"][k], rtol=1e-6, atol=1e-7)
            assert np.allclose(v[:m], g["lp_mult"][k], rtol=1e-6, atol=1e-7)
            assert abs(f[k].dot(u) - g["lp_min"][k]) < 1e-6 * max(1.0, abs(g["lp_min"][k]))
    assert n_solv >= 10
