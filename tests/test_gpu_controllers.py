"""The reference's randomized property tests of ModelPredictiveController
(pympc/test/test_control/test_controllers.py:13-101 implicit, :103-163 explicit), same seed and same recipe, run
end to end on the HIP path: solve_dare -> mcais (batched LPs + redundancy removal) -> condensing (MFMA kernel) ->
feedforward / feedback (QP kernel) -> explicit solution (LP + QP kernels, point location).
Fewer systems than the reference's 100 / 10 to keep the GPU suite short; the recipe consumes the random stream in
the same order, so system i here is system i there."""
from copy import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    from pympc_amd.control.controllers import ModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    return ModelPredictiveController, LinearSystem, Polyhedron


def test_implicit_solution_properties():
    ModelPredictiveController, LinearSystem, Polyhedron = _mods()
    np.random.seed(1)
    n_inside = n_outside = n_infeasible = 0
    for i in range(12):
        n = np.random.randint(2, 5)
        m = np.random.randint(1, n)
        controllable = False
        while not controllable:
            A = np.random.rand(n, n) / 10.
            B = np.random.rand(n, m) / 10.
            S = LinearSystem(A, B)
            controllable = S.controllable
        x_min = -np.random.rand(n); x_max = np.random.rand(n)
        u_min = -np.random.rand(m); u_max = np.random.rand(m)
        X = Polyhedron.from_bounds(x_min, x_max)
        U = Polyhedron.from_bounds(u_min, u_max)
        D = X.cartesian_product(U)
        N = np.random.randint(5, 10)
        Q = np.eye(n); R = np.eye(m)
        P, K = S.solve_dare(Q, R)
        X_N = S.mcais(K, D)
        controller = ModelPredictiveController(S, N, Q, R, P, D, X_N)
        xs = [np.multiply(np.random.rand(n), x_max - x_min) + x_min for j in range(10)]
        # the batched form answers all ten states in one launch and must agree with the scalar calls
        Ub, Vb, okb = controller.feedforward(np.array(xs))
        for j, x in enumerate(xs):
            u_mpc, V_mpc = controller.feedforward(x)
            assert (V_mpc is not None) == bool(okb[j])
            if V_mpc is not None:
                assert abs(V_mpc - Vb[j]) <= 1e-12 * max(1., abs(V_mpc))
                assert np.array_equal(np.concatenate(u_mpc), Ub[j].reshape(-1))
            V_lqr = .5 * x.dot(P).dot(x)
            u_lqr = []
            x_next = x
            for t in range(N):
                u_lqr.append(K.dot(x_next))
                x_next = (A + B.dot(K)).dot(x_next)
            if X_N.contains(x):
                n_inside += 1
                assert abs(V_mpc - V_lqr) < 5e-8
                for t in range(N):
                    np.testing.assert_array_almost_equal(u_mpc[t], u_lqr[t])
            elif V_mpc is not None:
                n_outside += 1
                assert V_mpc > V_lqr
                np.testing.assert_array_almost_equal(u_mpc[0], controller.feedback(x))
                x_mpc = S.simulate(x, u_mpc)
                for t in range(N):
                    assert X.contains(x_mpc[t]) and U.contains(u_mpc[t])
                assert X_N.contains(x_mpc[N])
            else:
                n_infeasible += 1
                assert controller.feedback(x) is None
                # certificate of the reference (:63-73, 101): the un-condensed constraint set is empty
                constraints = copy(D)
                C = np.hstack((np.eye(n), np.zeros((n, m))))
                constraints.add_equality(C, x)
                for t in range(N - 1):
                    constraints = constraints.cartesian_product(D)
                    C = np.zeros((n, constraints.A.shape[1]))
                    C[:, -2 * (n + m):] = np.hstack((A, B, -np.eye(n), np.zeros((n, m))))
                    constraints.add_equality(C, np.zeros(n))
                constraints = constraints.cartesian_product(X_N)
                # x_N is tied to the last stage by the dynamics (the reference's set leaves it free, which can only
                # make its set larger; emptiness of ours with the tie is what infeasibility of the QP means)
                C = np.zeros((n, constraints.A.shape[1]))
                C[:, -(2 * n + m):] = np.hstack((A, B, -np.eye(n)))
                constraints.add_equality(C, np.zeros(n))
                assert constraints.empty
    assert n_inside > 10 and n_outside + n_infeasible >= 1


def test_explicit_solution_properties():
    ModelPredictiveController, LinearSystem, Polyhedron = _mods()
    np.random.seed(1)
    checked = 0
    for i in range(3):
        n = np.random.randint(2, 3)
        m = np.random.randint(1, 2)
        controllable = False
        while not controllable:
            A = np.random.rand(n, n)
            B = np.random.rand(n, m)
            S = LinearSystem(A, B)
            controllable = S.controllable
        x_min = -np.random.rand(n); x_max = np.random.rand(n)
        u_min = -np.random.rand(m); u_max = np.random.rand(m)
        X = Polyhedron.from_bounds(x_min, x_max)
        U = Polyhedron.from_bounds(u_min, u_max)
        D = X.cartesian_product(U)
        N = 10
        Q = np.eye(n); R = np.eye(m)
        P, K = S.solve_dare(Q, R)
        X_N = S.mcais(K, D)
        controller = ModelPredictiveController(S, N, Q, R, P, D, X_N)
        with pytest.raises(ValueError):
            controller.feedforward_explicit(np.zeros(n))
        controller.store_explicit_solution()
        for j in range(10):
            x = np.multiply(np.random.rand(n), x_max - x_min) + x_min
            u_implicit, V_implicit = controller.feedforward(x)
            u_explicit, V_explicit = controller.feedforward_explicit(x)
            if V_implicit is not None:
                checked += 1
                assert abs(V_implicit - V_explicit) < 5e-8
                for t in range(N):
                    np.testing.assert_array_almost_equal(u_implicit[t], u_explicit[t])
                np.testing.assert_array_almost_equal(controller.feedback(x), controller.feedback_explicit(x))
            else:
                assert V_explicit is None and u_explicit is None and controller.feedback_explicit(x) is None
    assert checked >= 5
