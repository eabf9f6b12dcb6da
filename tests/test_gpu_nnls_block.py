"""NNLS / PNNLS / the reference's LP form beyond 64 rows: the workgroup-per-problem kernels of csrc/nnls_block.hpp
(65 .. 1024 rows; 256, 512 or 1024 threads, a row per thread) against the oracle's Lawson-Hanson and against scipy's own regression problems, unreduced
(tests/golden/scipy_nnls_tests.npz; test_gpu_nnls.py runs the same files through a QR reduction on the wave kernel).
Tolerances: residual norms to 1e-9 relative, unique minimisers to 1e-8 (fp64; the block kernel sums column-wise dot
products in one thread and row-wise norms through a tree, the oracle serially)."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_nnls_block_random_vs_oracle(oracle):
    from pympc_amd.optimization.solvers.hip_lp import nnls_batch
    rng = np.random.RandomState(19)
    for rows, cols, B in ((65, 30, 6), (100, 300, 6), (128, 128, 4), (200, 64, 4), (256, 500, 3), (255, 2048, 2), (70, 1, 3),
                          (257, 90, 3), (300, 400, 2), (512, 300, 2), (511, 2048, 1), (513, 120, 2), (800, 500, 1), (1024, 300, 1),
                          (30, 5000, 2), (200, 6000, 1), (600, 2500, 1)):
        A = rng.randn(B, rows, cols); b = rng.randn(B, rows)
        x, rn, st = nnls_batch(A, b)
        x, rn, st = x.cpu().numpy(), rn.cpu().numpy(), st.cpu().numpy()
        assert (st == 0).all(), (rows, cols, st)
        for k in range(B):
            xo, ro, _ = oracle.nnls(A[k], b[k])
            assert abs(rn[k] - ro) < 1e-9 * max(1.0, ro), (rows, cols, k, rn[k], ro)
            assert x[k].min() >= 0.0 and abs(np.linalg.norm(A[k].dot(x[k]) - b[k]) - rn[k]) < 1e-9 * max(1.0, ro)
            if ro > 1e-9:
                assert np.allclose(x[k], xo, rtol=0, atol=1e-8 * max(1.0, np.abs(xo).max())), (rows, cols, k)
    # a shared matrix with many right-hand sides; non-negative data (many columns leave again: Givens deletions)
    A = np.abs(rng.randn(90, 140)); b = rng.randn(40, 90) + 0.5
    x, rn, st = nnls_batch(A, b)
    x = x.cpu().numpy()
    for k in (0, 13, 39):
        xo, ro, it = oracle.nnls(A, b[k])
        assert abs(float(rn[k]) - ro) < 1e-9 * max(1.0, ro) and np.allclose(x[k], xo, atol=1e-8), (k, it)


def test_nnls_block_secondary_loop_and_iteration_counts(oracle):
    """Problems whose active set shrinks on the way (the secondary loop with its Givens deletions runs): same minimiser,
    and the same number of outer / secondary passes as the oracle on well-conditioned data."""
    from pympc_amd import _hip
    from pympc_amd.optimization.solvers.hip_lp import nnls_batch
    import torch
    rng = np.random.RandomState(23)
    deletions = 0
    for rows, cols in ((80, 60), (120, 200), (160, 90)):
        X = rng.rand(rows, 5)
        A = np.hstack([X[:, [i % 5]] ** (1 + i // 5) + 0.05 * rng.rand(rows, 1) for i in range(cols)])   # strongly correlated columns
        b = A.dot(np.maximum(rng.randn(cols), 0.0)) + 0.1 * rng.randn(rows)
        xo, ro, it = oracle.nnls(A, b)
        lib = _hip.load_library()
        dev = _hip.device()
        At = torch.as_tensor(A, device=dev); bt = torch.as_tensor(b[None], device=dev)
        ws = torch.empty(lib.pympc_nnls_workspace_doubles(rows, cols, 1), dtype=torch.float64, device=dev)
        x = torch.empty((1, cols), dtype=torch.float64, device=dev); rn = torch.empty(1, dtype=torch.float64, device=dev)
        st = torch.empty(1, dtype=torch.int32, device=dev); its = torch.zeros(2, dtype=torch.int32, device=dev)
        _hip.check(lib.pympc_nnls_batch(rows, cols, 1, _hip.ptr(At), 0, _hip.ptr(bt), _hip.ptr(ws), _hip.ptr(x), _hip.ptr(rn),
                                        _hip.ptr(st), _hip.ptr(its), _hip.stream_ptr()))
        assert int(st.item()) == 0 and abs(float(rn.item()) - ro) < 1e-9 * max(1.0, ro)
        assert np.allclose(x[0].cpu().numpy(), xo, atol=1e-7 * max(1.0, np.abs(xo).max()))
        deletions += it[1] - it[0]
        print(rows, cols, "oracle (outer, secondary):", it, "kernel:", its.cpu().numpy().tolist())
    assert deletions > 0


def test_nnls_block_scipy_regression_problems_unreduced():
    from pympc_amd.optimization.solvers.hip_lp import nnls
    g = golden("scipy_nnls_tests")
    ran = []
    for k in range(int(g["count"])):
        A, b = g["A%d" % k], g["b%d" % k]
        if not bool(g["ok%d" % k]) or not (64 < A.shape[0] <= 1024) or A.shape[1] > 2048:
            continue
        x, rn = nnls(A, b)
        xs = g["x%d" % k]
        rs = float(g["r%d" % k])
        assert abs(rn - rs) < 1e-7 * max(1.0, np.linalg.norm(b)), str(g["names"][k])
        assert x.min() >= 0.0 and abs(np.linalg.norm(A.dot(x) - b) - rs) < 1e-7 * max(1.0, np.linalg.norm(b))
        if rs > 1e-7 * max(1.0, np.linalg.norm(b)) or np.linalg.matrix_rank(A) == A.shape[1]:
            assert np.allclose(x, xs, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(xs).max())), str(g["names"][k])
        ran.append((str(g["names"][k]), A.shape))
    print("scipy regression problems on the block kernel:", ran)
    assert len(ran) >= 1


def test_nnls_block_iteration_limit():
    from pympc_amd import _hip
    from pympc_amd.optimization.solvers.hip_lp import nnls, nnls_batch
    rng = np.random.RandomState(4)
    A = rng.randn(70, 9); b = rng.randn(70)
    with _hip.tuned(nnls_itmax_factor=0):
        x, rn, st = nnls_batch(A, np.vstack((b, np.zeros(70))))
        assert st.cpu().numpy().tolist() == [_hip.MAXITER, _hip.OPTIMAL]
        with pytest.raises(RuntimeError, match="Maximum number of iterations"):
            nnls(A, b)
    with pytest.raises(NotImplementedError):
        nnls(rng.randn(1025, 4), rng.randn(1025))


def test_pnnls_block_random_vs_oracle(oracle):
    from pympc_amd.optimization.solvers.hip_lp import pnnls, pnnls_batch
    rng = np.random.RandomState(6)
    for rows, nv, nu in ((65, 40, 3), (100, 150, 20), (128, 60, 64), (200, 300, 10), (256, 100, 33), (300, 150, 40), (512, 200, 30), (900, 300, 12),
                         (100, 50, 65), (300, 200, 128), (600, 100, 256), (40, 30, 70), (1024, 50, 200)):
        A = rng.randn(rows, nv); B = rng.randn(rows, nu); c = rng.randn(rows)
        v, u, r = pnnls(A, B, c)
        vo, uo, ro = oracle.pnnls(A, B, c)
        assert abs(r - ro) < 1e-9 * max(1.0, ro), (rows, nv, nu, r, ro)
        assert v.min() >= 0.0 and abs(np.linalg.norm(A.dot(v) + B.dot(u) - c) - r) < 1e-8 * max(1.0, r)
        if ro > 1e-9:
            assert np.allclose(v, vo, atol=1e-8 * max(1.0, np.abs(vo).max())), (rows, nv, nu)
            assert np.allclose(u, uo, atol=1e-7 * max(1.0, np.abs(uo).max())), (rows, nv, nu)
    # rank-deficient B: pinv's minimum-norm u
    A = rng.randn(90, 30); B0 = rng.randn(90, 2); B = np.column_stack((B0, B0[:, 0] - 2 * B0[:, 1], np.zeros(90))); c = rng.randn(90)
    v, u, r = pnnls(A, B, c)
    vo, uo, ro = oracle.pnnls(A, B, c)
    assert np.allclose(v, vo, atol=1e-8) and np.allclose(u, uo, atol=1e-8) and abs(r - ro) < 1e-9
    # rank-deficient B with more than 64 columns (the recover kernel's systems in device scratch)
    A = rng.randn(150, 40); B0 = rng.randn(150, 60); B = np.column_stack((B0, B0[:, :20] - B0[:, 20:40], np.zeros((150, 3)))); c = rng.randn(150)
    v, u, r = pnnls(A, B, c)
    vo, uo, ro = oracle.pnnls(A, B, c)
    assert abs(r - ro) < 1e-9 and np.allclose(v, vo, atol=1e-8) and np.allclose(u, uo, atol=1e-7 * max(1.0, np.abs(uo).max()))
    # batch
    A = rng.randn(5, 77, 50); B = rng.randn(5, 77, 6); c = rng.randn(5, 77)
    v, u, rn, st = pnnls_batch(A, B, c)
    assert (st.cpu().numpy() == 0).all()
    for k in range(5):
        vo, uo, ro = oracle.pnnls(A[k], B[k], c[k])
        assert abs(float(rn[k]) - ro) < 1e-9
        assert np.allclose(v[k].cpu().numpy(), vo, atol=1e-8) and np.allclose(u[k].cpu().numpy(), uo, atol=1e-8)


def test_lp_in_the_reference_form_beyond_64_rows(oracle):
    """linear_program_pnnls (the KKT system of pnnls.py:109-124, decided by the residual as pnnls.py:137 does) on LPs
    with 1 + m + n up to 1024: optimum, multipliers and the solvable / not solvable decision against the oracle's
    linear_program -- including LPs that are infeasible by a margin between the two decision rules' tolerances, where
    the active-set LP kernel and the reference differ."""
    from pympc_amd.optimization.solvers.hip_lp import linear_program_pnnls
    rng = np.random.RandomState(12)
    n_solv = 0
    for n, m in ((3, 70), (6, 120), (10, 200), (20, 60), (40, 100), (64, 191), (12, 400), (30, 480), (10, 900)):
        A = rng.randn(m, n); x0 = rng.randn(n)
        b = A.dot(x0) + rng.rand(m) + 0.1
        A = np.vstack((A, np.eye(n)[: min(n, 4)], -np.eye(n)[: min(n, 4)]))[:m]   # keep m rows
        b = np.concatenate((b, np.full(2 * min(n, 4), 50.0)))[:m]
        f = rng.randn(n)
        # bounded: add the box through the objective's sign pattern only if the oracle says so
        ref = oracle.linear_program(f, A, b)
        sol = linear_program_pnnls(f, A, b)
        assert (sol['min'] is None) == (ref['min'] is None), (n, m)
        if ref['min'] is not None:
            n_solv += 1
            assert abs(sol['min'] - ref['min']) < 1e-6 * max(1.0, abs(ref['min']))
            assert np.allclose(sol['argmin'], ref['argmin'], atol=1e-6 * max(1.0, np.abs(ref['argmin']).max()))
            assert np.allclose(sol['multiplier_inequality'], ref['multiplier_inequality'], atol=1e-6 * max(1.0, np.abs(ref['multiplier_inequality']).max()))
            assert sol['active_set'] == ref['active_set']
    # a bounded polytope: always solvable
    for n, m in ((4, 100), (8, 150)):
        A = np.vstack((rng.randn(m - 2 * n, n), np.eye(n), -np.eye(n))); b = np.concatenate((rng.rand(m - 2 * n) + 0.5, np.ones(2 * n)))
        f = rng.randn(n)
        ref = oracle.linear_program(f, A, b); sol = linear_program_pnnls(f, A, b)
        assert ref['min'] is not None and abs(sol['min'] - ref['min']) < 1e-7 and sol['active_set'] == ref['active_set']
        n_solv += 1
        # infeasible by margins around the reference's residual tolerance: x_0 <= 1 and x_0 >= 1 + gap
        for gap in (1e-9, 3e-8, 3e-7, 1e-5, 1e-2):
            A2 = np.vstack((A, -np.eye(n)[0])); b2 = np.concatenate((b, [-(1.0 + gap)]))
            ref = oracle.linear_program(f, A2, b2); sol = linear_program_pnnls(f, A2, b2)
            assert (sol['min'] is None) == (ref['min'] is None), (n, m, gap)
    assert n_solv >= 4
    # more variables than the active-set LP kernel takes: linear_program itself goes to the KKT form
    from pympc_amd.optimization.solvers.hip_lp import linear_program as lp_scalar
    for n, m in ((80, 200), (150, 320)):
        A = np.vstack((rng.randn(m - 2 * n, n), np.eye(n), -np.eye(n))); b = np.concatenate((rng.rand(m - 2 * n) + 0.5, np.ones(2 * n)))
        f = rng.randn(n)
        ref = oracle.linear_program(f, A, b); sol = lp_scalar(f, A, b)
        assert ref['min'] is not None and abs(sol['min'] - ref['min']) < 1e-7 * max(1.0, abs(ref['min']))
        assert np.allclose(sol['argmin'], ref['argmin'], atol=1e-6) and sol['active_set'] == ref['active_set']
        A2 = np.vstack((A, -np.eye(n)[0])); b2 = np.concatenate((b, [-1.5]))          # x_0 >= 1.5 against x_0 <= 1
        assert lp_scalar(f, A2, b2)['min'] is None and oracle.linear_program(f, A2, b2)['min'] is None
    # the same through the dispatch of optimization/programs.py
    from pympc_amd.optimization.programs import linear_program
    sol = linear_program(f, A, b, solver='hip_kkt'); ref = oracle.linear_program(f, A, b)
    assert abs(sol['min'] - ref['min']) < 1e-7 and sol['active_set'] == ref['active_set']
    with pytest.raises(ValueError, match='unknown solver'):
        linear_program(f, A, b, solver='gurobi')


def test_nnls_block_degenerate_inputs(oracle):
    """Ties, dependent and empty columns, a right-hand side inside the cone: the residual norm is unique and must agree;
    the minimiser is compared where it is unique, and the entering order (first maximum among equal duals) where the
    data are exact in binary."""
    from pympc_amd.optimization.solvers.hip_lp import nnls
    rng = np.random.RandomState(31)
    rows = 96
    base = rng.randint(-4, 5, size=(rows, 20)).astype(float)          # small integers: products and sums are exact
    cases = {
        "duplicate columns": np.hstack((base, base[:, :7])),
        "zero columns": np.hstack((base[:, :10], np.zeros((rows, 3)), base[:, 10:])),
        "dependent columns": np.hstack((base, base[:, :5] + base[:, 5:10], 2.0 * base[:, 3:4])),
        "zero rows": np.vstack((base[:40], np.zeros((16, 20)), base[56:])),
        "one column": base[:, :1],
    }
    for name, A in cases.items():
        for b in (rng.randint(-6, 7, size=rows).astype(float), A.dot(np.maximum(rng.randint(-2, 4, size=A.shape[1]), 0).astype(float)),
                  np.zeros(rows)):
            x, rn = nnls(A, b)
            xo, ro, it = oracle.nnls(A, b)
            assert abs(rn - ro) < 1e-9 * max(1.0, np.linalg.norm(b)), (name, rn, ro)
            assert x.min() >= 0.0 and abs(np.linalg.norm(A.dot(x) - b) - rn) < 1e-9 * max(1.0, np.linalg.norm(b)), name
            # the same support as the oracle on the exact data with duplicates: the first of equal columns enters
            if name in ("duplicate columns", "zero columns") and ro > 1e-9:
                assert np.array_equal(x > 0, xo > 0), name
                assert np.allclose(x, xo, atol=1e-9 * max(1.0, np.abs(xo).max())), name


def test_few_rows_many_columns(oracle):
    """More columns than the wave kernel takes with very few rows: the workgroup kernel with a handful of busy
    row-threads; and the QP of one variable with 3000 constraints."""
    from pympc_amd.optimization.solvers.hip_lp import nnls_batch
    from pympc_amd.optimization.solvers.hip import quadratic_program
    rng = np.random.RandomState(41)
    for rows in (1, 2, 5):
        A = rng.randn(3, rows, 3000); b = rng.randn(3, rows)
        x, rn, st = nnls_batch(A, b)
        x, rn = x.cpu().numpy(), rn.cpu().numpy()
        assert (st.cpu().numpy() == 0).all()
        for k in range(3):
            xo, ro, _ = oracle.nnls(A[k], b[k])
            assert abs(rn[k] - ro) < 1e-9 and x[k].min() >= 0.0 and abs(np.linalg.norm(A[k].dot(x[k]) - b[k]) - ro) < 1e-9
    A = rng.randn(3000, 1); b = np.abs(rng.randn(3000)) + 0.5
    ref = oracle.quadratic_program(np.array([[2.]]), np.array([-40.]), A, b)
    sol = quadratic_program(np.array([[2.]]), np.array([-40.]), A, b)
    assert abs(sol['min'] - ref['min']) < 1e-9 * max(1.0, abs(ref['min'])) and sol['active_set'] == ref['active_set']


def test_nnls_block_shape_sweep(oracle):
    """Forty random shapes around the kernels' internal sizes (16-column chunks of the factor, 8-row load blocks, 256 / 512 /
    1024 threads, columns fewer and more than threads), data that force deletions: residual norm and feasibility of
    every problem, the minimiser where it is unique."""
    from pympc_amd.optimization.solvers.hip_lp import nnls_batch
    rng = np.random.RandomState(77)
    special = [(65, 17), (80, 16), (81, 15), (96, 257), (127, 33), (255, 256), (256, 255), (257, 31), (272, 511), (511, 513), (513, 64), (640, 90)]
    shapes = special + [(int(rng.randint(65, 420)), int(rng.randint(1, 640))) for _ in range(28)]
    for rows, cols in shapes:
        A = np.abs(rng.randn(2, rows, cols)) + (0.2 * rng.randn(2, rows, cols) if rows % 2 else 0.0)
        b = rng.randn(2, rows) + 0.3
        x, rn, st = nnls_batch(A, b)
        x, rn, st = x.cpu().numpy(), rn.cpu().numpy(), st.cpu().numpy()
        assert (st == 0).all(), (rows, cols)
        for k in range(2):
            xo, ro, _ = oracle.nnls(A[k], b[k])
            assert abs(rn[k] - ro) < 1e-9 * max(1.0, ro), (rows, cols, k, rn[k], ro)
            assert x[k].min() >= 0.0 and abs(np.linalg.norm(A[k].dot(x[k]) - b[k]) - rn[k]) < 1e-9 * max(1.0, ro), (rows, cols, k)
            if rows >= cols:
                assert np.allclose(x[k], xo, rtol=0, atol=1e-7 * max(1.0, np.abs(xo).max())), (rows, cols, k)
