"""GPU parity of the dynamics layer that feeds the MPC hot path: mcais (one batched LP launch per iteration of
Gilbert-Tan, pympc/dynamics/discrete_time_systems.py:584-674) against the terminal sets the REFERENCE computed for the
golden models (tests/golden/*_model.npz: XN_A, XN_b are the outputs of S.mcais(K, D) run in the build container), and
the reference's own property test of the invariant set (pympc/test/test_dynamics/test_discrete_time_systems.py:418-453)."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _same_rows(E, F, tol=1e-6):
    En = E / np.linalg.norm(E[:, :-1], axis=1)[:, None]
    Fn = F / np.linalg.norm(F[:, :-1], axis=1)[:, None]
    if En.shape != Fn.shape:
        return False
    return all(np.min(np.linalg.norm(Fn - e, axis=1)) < tol for e in En) and \
        all(np.min(np.linalg.norm(En - f, axis=1)) < tol for f in Fn)


@pytest.mark.parametrize("model", ["c1_model", "nb_model", "c2_model"])
def test_mcais_vs_reference_golden(model):
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    g = golden(model)
    S = LinearSystem(g['S_A'], g['S_B'])
    D = Polyhedron(g['D_A'].copy(), g['D_b'].copy())
    # the gain the reference used (solve_dare): ours must agree, and the invariant set is computed with ours
    P, K = S.solve_dare(g['Q'], g['R'])
    assert np.allclose(P, g['P'], rtol=1e-9, atol=1e-9) and np.allclose(K, g['K'], rtol=1e-9, atol=1e-9)
    O = S.mcais(K, D)
    assert O.A.shape == g['XN_A'].shape                       # 18 / ... / 130 facets (cart-pole 130)
    assert _same_rows(np.column_stack((O.A, O.b)), np.column_stack((g['XN_A'], g['XN_b'])))


def test_mcais_invariance_property():
    """test_discrete_time_systems.py:418-453: vertices of O_inf stay inside O_inf and inside the constraints forever."""
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem, mcais
    from pympc_amd.geometry.polyhedron import Polyhedron
    np.random.seed(1)
    nx, nu = 2, 1
    for _ in range(5):
        A = np.random.rand(nx, nx); B = np.random.rand(nx, nu)
        S = LinearSystem(A, B)
        if not S.controllable:
            continue
        K = S.solve_dare(np.eye(nx), np.eye(nu))[1]
        A_cl = A + B.dot(K)
        x_min = -np.random.rand(nx); x_max = np.random.rand(nx)
        u_min = -np.random.rand(nu); u_max = np.random.rand(nu)
        X = Polyhedron.from_bounds(x_min, x_max); U = Polyhedron.from_bounds(u_min, u_max)
        D = X.cartesian_product(U)
        O_inf = S.mcais(K, D)
        # sample the boundary of O_inf: points on each facet obtained by maximising random directions (vertices)
        from pympc_amd.optimization.solvers.hip_lp import linear_program_batch
        dirs = np.random.randn(64, nx)
        res = linear_program_batch(-dirs, O_inf.A, O_inf.b)
        assert res.feasible.all()
        for v in res.numpy('argmin'):
            x = v.copy()
            for t in range(50):
                u = K.dot(x)
                assert X.contains(x, tol=1e-6) and U.contains(u, tol=1e-6) and O_inf.contains(x, tol=1e-6)
                x = A_cl.dot(x)
        # module-level function with the closed-loop matrices gives the same set
        X_cl = Polyhedron(D.A[:, :nx] + D.A[:, nx:].dot(K), D.b)
        O2 = mcais(A_cl, X_cl)
        assert _same_rows(np.column_stack((O_inf.A, O_inf.b)), np.column_stack((O2.A, O2.b)))
    with pytest.raises(ValueError):
        mcais(2. * np.eye(2), Polyhedron.from_bounds(-np.ones(2), np.ones(2)))          # unstable
    with pytest.raises(ValueError):
        mcais(.5 * np.eye(2), Polyhedron.from_bounds(.5 * np.ones(2), np.ones(2)))       # origin outside
    with pytest.raises(ValueError):
        mcais(.5 * np.eye(2), Polyhedron.from_lower_bound(-np.ones(2)))                  # unbounded
