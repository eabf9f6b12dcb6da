"""The reference's own QP known-answer tests (pympc/test/test_optimization/test_solvers.py:263-345 and
test_programs.py:71-122), run through the drop-in API of the `hip` backend on the GPU.  Inputs and expected
values are the data those tests hold; assertions mirror theirs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _qp():
    from pympc_amd.optimization.programs import quadratic_program
    return quadratic_program


def test_qp_box():
    # test_solvers.py:265-289
    sol = _qp()(np.eye(2), np.ones(2), -np.eye(2), -np.ones(2))
    assert abs(sol['min'] - 3.) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], np.array([1., 1.]))
    assert sol['active_set'] == [0, 1]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], np.array([2., 2.]))
    assert sol['multiplier_equality'] is None


def test_qp_equality_and_infeasible():
    # test_solvers.py:291-319
    H = np.eye(2); f = np.ones(2); A = -np.eye(2); b = -np.ones(2)
    C = np.array([[0., 1.]]); d = np.array([2.])
    sol = _qp()(H, f, A, b, C, d)
    assert abs(sol['min'] - 5.5) < 1e-7
    np.testing.assert_array_almost_equal(sol['argmin'], np.array([1., 2.]))
    assert sol['active_set'] == [0]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], np.array([2., 0.]))
    np.testing.assert_array_almost_equal(sol['multiplier_equality'], np.array([-3.]))
    sol = _qp()(H, f, A, b, C, -d)
    assert all(v is None for v in sol.values())


def test_qp_lower_dimensional():
    # test_solvers.py:321-345
    sol = _qp()(np.eye(2), np.zeros(2), np.array([[1., 0.], [-1., 0.]]), np.array([1., -1.]))
    assert abs(sol['min'] - .5) < 1e-7
    assert abs(sol['argmin'][0] - 1.) < 1e-7
    assert sol['active_set'] == [1]
    np.testing.assert_array_almost_equal(sol['multiplier_inequality'], np.array([0., 1.]))
    assert sol['multiplier_equality'] is None


def test_qp_errors():
    from pympc_amd.optimization.programs import quadratic_program
    with pytest.raises(ValueError):
        quadratic_program(np.eye(2), np.ones(2), -np.eye(2), -np.ones(2), C=np.ones((1, 2)))       # pnnls.py:191-192
    with pytest.raises(ValueError):
        quadratic_program(np.eye(2), np.ones(2), -np.eye(2), -np.ones(2), solver='nope')            # programs.py:105
    with pytest.raises(np.linalg.LinAlgError):
        quadratic_program(np.array([[1., 0.], [0., -1.]]), np.ones(2), -np.eye(2), -np.ones(2))     # pnnls.py:210


def test_qp_batch_vs_oracle(oracle):
    """Per-problem matrices (pympc_qp_solve_batch): random strictly convex QPs, each checked against the oracle."""
    from pympc_amd.optimization.solvers.hip import quadratic_program_batch
    rng = np.random.RandomState(5)
    for (n, m, B) in ((3, 7, 64), (6, 40, 50), (12, 130, 20)):
        G = rng.randn(B, n, n)
        H = np.einsum('bij,bkj->bik', G, G) + np.eye(n)[None]
        f = rng.randn(B, n)
        A = rng.randn(B, m, n)
        b = rng.randn(B, m) + 0.5
        res = quadratic_program_batch(H, f, A, b)
        for k in range(B):
            so = oracle.quadratic_program(H[k], f[k], A[k], b[k])
            sg = res.sol(k)
            assert (so['min'] is None) == (sg['min'] is None)
            if so['min'] is None:
                continue
            lam = so['multiplier_inequality']
            nz = lam[lam != 0]
            if np.any(np.abs(nz - 1e-7) < 1e-9 * max(1., np.linalg.norm(lam))):
                continue
            assert sg['active_set'] == so['active_set']
            assert np.allclose(sg['argmin'], so['argmin'], rtol=1e-6, atol=1e-6 * max(1, np.abs(so['argmin']).max()))
            assert abs(sg['min'] - so['min']) <= 1e-6 * max(1., abs(so['min']))


@pytest.mark.parametrize("n,m,B", [(2, 9, 400), (5, 64, 400), (10, 78, 300), (20, 370, 200), (40, 500, 60), (63, 190, 40)])
def test_qp_badly_scaled_rows_vs_oracle(oracle, n, m, B):
    """Rows scaled over six decades (what the kernel's scaled entering rule reacts to; the oracle keeps
    Lawson-Hanson's max-w rule): same feasibility, active sets and solutions; half of the problems feasible by construction."""
    from pympc_amd.optimization.solvers.hip import quadratic_program_batch
    rng = np.random.RandomState(100 + n)
    G = rng.randn(B, n, n)
    H = np.einsum('bij,bkj->bik', G, G) + np.eye(n)[None]
    f = rng.randn(B, n)
    scale = 10.0 ** rng.uniform(-3, 3, (B, m, 1))
    A = rng.randn(B, m, n) * scale
    xs = rng.randn(B, n)
    slack = np.abs(rng.randn(B, m)) + 1e-3                                     # xs strictly inside: no degenerate vertex by construction
    b = np.einsum('bmn,bn->bm', A, xs) + slack * scale[..., 0]               # feasible by construction ...
    half = B // 2
    b[half:] = (rng.randn(B - half, m) + 0.3) * scale[half:, :, 0]           # ... or random (mostly infeasible for m >> n)
    res = quadratic_program_batch(H, f, A, b)
    nfeas = nlimit = 0
    for k in range(B):
        try:
            so = oracle.quadratic_program(H[k], f[k], A[k], b[k])
        except RuntimeError:      # Lawson-Hanson's max-w rule runs into scipy's 3 * cols iteration limit on this scaling
            nlimit += 1
            continue
        sg = res.sol(k)
        assert (so['min'] is None) == (sg['min'] is None), k
        if so['min'] is None:
            continue
        nfeas += 1
        lam = so['multiplier_inequality']
        nz = lam[lam != 0]
        if np.any(np.abs(nz - 1e-7) < 1e-9 * max(1., np.linalg.norm(lam))):
            continue
        assert sg['active_set'] == so['active_set'], k
        assert np.allclose(sg['argmin'], so['argmin'], rtol=1e-6, atol=1e-6 * max(1, np.abs(so['argmin']).max())), k
        assert abs(sg['min'] - so['min']) <= 1e-6 * max(1., abs(so['min'])), k
    assert 0 < nfeas < B and nlimit <= B // 4


@pytest.mark.parametrize("n,m", [(3, 40), (10, 250), (16, 70)])
def test_wave_prepare_kernel_same_bits_as_tile_kernel(n, m, oracle):
    """Batches of >= 1024 problems with up to 16 variables are factorised by qp_prepare_wave_kernel (one wavefront per
    problem, constraint rows in registers) instead of qp_prepare_kernel (row tiles in LDS): launch policy
    (pympc_hip_tune "prepare_wave").  Same sums in the same order: every output bitwise the same; and a sample of the
    batch against the oracle's quadratic_program (pnnls.py:149-241)."""
    from pympc_amd import _hip
    from pympc_amd.optimization.solvers.hip import quadratic_program_batch
    rng = np.random.RandomState(10 * n + m)
    B = 1500
    G = rng.randn(B, n, n)
    H = np.einsum('bij,bkj->bik', G, G) + 0.5 * np.eye(n)[None]
    f = rng.randn(B, n)
    A = rng.randn(B, m, n)
    b = 0.3 + rng.rand(B, m)
    b[::7] -= 2.0                                   # some infeasible ones
    outs = []
    for wave in (1, 0):
        with _hip.tuned(prepare_wave=wave):
            r = quadratic_program_batch(H, f, A, b, want_iters=True)
            outs.append({k: v.clone() for k, v in r.t.items() if v is not None})
    for k in outs[0]:
        a, c = outs[0][k], outs[1][k]
        if a.dtype.is_floating_point:
            assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(c, nan=-7.0)), k
        else:
            assert torch.equal(a, c), k
    st = outs[0]["status"].cpu().numpy()
    assert (st == 0).sum() > 100 and (st == 1).sum() > 50
    for i in range(0, B, 97):
        ref = oracle.quadratic_program(H[i], f[i], A[i], b[i])
        assert (ref['min'] is None) == (st[i] != 0), i
        if ref['min'] is not None:
            assert abs(outs[0]["min"][i].item() - ref['min']) < 1e-6 * max(1.0, abs(ref['min']))
            assert np.allclose(outs[0]["argmin"][i].cpu().numpy(), ref['argmin'], rtol=1e-6, atol=1e-6)
