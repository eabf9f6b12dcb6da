"""Edge cases of the batched entry points through the C ABI: empty batches, one-row / one-variable problems, sizes at
and beyond the kernels' limits (clear errors, never a silent fallback), non-positive-definite Hessians and iteration
limits reported per problem."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_batches():
    from pympc_amd.optimization.solvers.hip import PreparedMpqp, quadratic_program_batch
    from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
    from pympc_amd.geometry.polyhedron import Polyhedron
    res = quadratic_program_batch(np.eye(2), np.zeros((0, 2)), np.ones((3, 2)), np.zeros((0, 3)))
    assert len(res) == 0 and res.numpy('argmin').shape == (0, 2)
    res = linear_program_batch(np.zeros((0, 2)), np.ones((3, 2)), np.ones(3))
    assert len(res) == 0
    prep = PreparedMpqp(np.eye(2), np.zeros((2, 1)), np.zeros((1, 1)), np.zeros(2), np.zeros(1), 0., np.ones((3, 2)),
                        np.zeros((3, 1)), np.ones(3))
    assert len(prep.solve_batch(np.zeros((0, 1)))) == 0
    P = Polyhedron.from_bounds(-np.ones(2), np.ones(2))
    assert P.contains(np.zeros((0, 2))).shape == (0,)
    assert Polyhedron(np.zeros((0, 2)), np.zeros(0)).minimal_facets() == []


def test_smallest_problems(oracle):
    from pympc_amd.optimization.programs import linear_program, quadratic_program
    # one variable, one constraint
    sol = quadratic_program(np.array([[2.]]), np.array([-4.]), np.array([[1.]]), np.array([1.]))
    ref = oracle.quadratic_program(np.array([[2.]]), np.array([-4.]), np.array([[1.]]), np.array([1.]))
    assert abs(sol['min'] - ref['min']) < 1e-9 and sol['active_set'] == ref['active_set'] == [0]
    sol = quadratic_program(np.array([[2.]]), np.array([-4.]), np.array([[1.]]), np.array([10.]))   # inactive
    assert abs(sol['argmin'][0] - 2.) < 1e-9 and sol['active_set'] == []
    sol = linear_program(np.array([-1.]), np.array([[1.]]), np.array([3.]))
    assert abs(sol['min'] + 3.) < 1e-9 and sol['active_set'] == [0]
    sol = linear_program(np.array([1.]), np.array([[1.]]), np.array([3.]))                          # unbounded
    assert all(v is None for v in sol.values())
    sol = linear_program(np.array([1.]), np.array([[1.], [-1.]]), np.array([-1., -1.]))             # infeasible
    assert all(v is None for v in sol.values())


def test_limits_are_errors_not_fallbacks():
    from pympc_amd import _hip
    from pympc_amd.optimization.solvers.hip import quadratic_program_batch
    from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
    nmax, mmax = _hip.limits()
    nbig = 1023      # per-problem QPs beyond the wave kernel's nmax go to the workgroup kernels (csrc/qp_block.hpp) up to here
    res = quadratic_program_batch(np.eye(nmax + 1), np.zeros((1, nmax + 1)), np.ones((2, nmax + 1)), np.ones((1, 2)))
    assert res.feasible.all() and np.abs(res.numpy('argmin')).max() < 1e-12
    with pytest.raises(NotImplementedError):
        quadratic_program_batch(np.eye(nbig + 1), np.zeros((1, nbig + 1)), np.ones((2, nbig + 1)), np.ones((1, 2)))
    # more rows than the wave kernel takes: the workgroup kernels, with their vectors over columns in device memory
    res = quadratic_program_batch(np.eye(2), np.array([[-3., 0.]]), np.vstack((np.array([[1., 0.]]), np.ones((mmax, 2)))),
                                  np.concatenate(([1.], np.full(mmax, 50.)))[None, :])
    assert res.feasible.all() and np.allclose(res.numpy('argmin')[0], [1., 0.], atol=1e-9) and res.sol(0)['active_set'] == [0]
    with pytest.raises(NotImplementedError):
        quadratic_program_batch(np.eye(2), np.zeros((1, 2)), np.ones((32769, 2)), np.ones((1, 32769)))
    n_lp, m_lp = _hip.lp_limits()
    with pytest.raises(NotImplementedError):
        linear_program_batch(np.zeros((1, n_lp + 1)), np.ones((3, n_lp + 1)), np.ones(3))
    # the largest LP variable count works
    rng = np.random.RandomState(0)
    A = np.vstack((np.eye(n_lp), -np.eye(n_lp))); b = np.ones(2 * n_lp); f = rng.randn(2, n_lp)
    res = linear_program_batch(f, A, b)
    assert res.feasible.all() and np.allclose(res.numpy('min'), -np.abs(f).sum(axis=1), atol=1e-9)
    assert np.allclose(res.numpy('argmin'), -np.sign(f), atol=1e-9)


def test_per_problem_status_in_a_mixed_batch(oracle):
    """One launch with optimal, infeasible and non-PD problems: every problem reports its own status."""
    from pympc_amd.optimization.solvers.hip import quadratic_program_batch
    H = np.array([np.eye(2), np.eye(2), np.array([[1., 2.], [2., 1.]]), np.eye(2)])
    f = np.array([[-1., -1.], [0., 0.], [0., 0.], [3., 0.]])
    A = np.array([[1., 0.], [0., 1.], [-1., -1.]])
    b = np.array([[.5, .5, 5.], [-1., -1., 1.], [1., 1., 1.], [1., 1., 1.]])
    res = quadratic_program_batch(H, f, A, b)
    assert list(res.numpy('status')) == [0, 1, 4, 0]
    assert np.allclose(res.numpy('argmin')[0], [.5, .5]) and np.all(np.isnan(res.numpy('argmin')[1:3]))
    assert res.sol(0)['active_set'] == [0, 1] and all(v is None for v in res.sol(1).values())
    with pytest.raises(np.linalg.LinAlgError):
        res.sol(2)
    ref = oracle.quadratic_program(H[3], f[3], A, b[3])
    assert np.allclose(res.numpy('argmin')[3], ref['argmin'], atol=1e-9)
