"""ModelPredictiveController end to end on the HIP path against values the REFERENCE computed.

tests/golden/controllers.npz (written by tests/golden/make_golden.py::gen_ctrl, which runs the reference in the build
container) holds 30 + 6 randomised linear MPC problems in the manner of the reference's own controller tests
(pympc/test/test_control/test_controllers.py:13-163): system matrices, bounds, horizon, the reference's P, K and
terminal set, ten sampled states per system and, for each state, the reference's feedforward (u, V), whether the state
lies in the terminal set, and the nnls certificate of the call.  Here the same problems run through
solve_dare -> mcais (batched LPs + redundancy removal) -> condensing (MFMA kernel) -> feedforward / feedback (QP
kernels) -> explicit solution (LP + QP kernels, point location) and are compared with that data.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "controllers.npz")


def _controller(g, pre):
    from pympc_amd.control.controllers import ModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    A, B = g[pre + "A"], g[pre + "B"]
    n, m = B.shape
    S = LinearSystem(A, B)
    X = Polyhedron.from_bounds(g[pre + "x_min"], g[pre + "x_max"])
    U = Polyhedron.from_bounds(g[pre + "u_min"], g[pre + "u_max"])
    D = X.cartesian_product(U)
    Q, R = np.eye(n), np.eye(m)
    P, K = S.solve_dare(Q, R)
    np.testing.assert_allclose(P, g[pre + "P"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(K, g[pre + "K"], rtol=1e-8, atol=1e-10)
    X_N = S.mcais(K, D)
    return ModelPredictiveController(S, int(g[pre + "N"]), Q, R, P, D, X_N), X_N


def _same_facets(A1, b1, A2, b2):
    """The two H-representations list the same half-spaces (rows scaled to unit normals, any order)."""
    def rows(A, b):
        s = np.linalg.norm(A, axis=1)
        M = np.column_stack((A, b)) / s[:, None]
        return M[np.lexsort(np.round(M, 7).T[::-1])]
    R1, R2 = rows(A1, b1), rows(A2, b2)
    return R1.shape == R2.shape and np.allclose(R1, R2, rtol=0, atol=1e-6)


def test_implicit_solution_against_reference_values():
    g = np.load(GOLD)
    n_feasible = n_infeasible = n_inside = n_checked = n_skipped = 0
    for i in range(int(g["n_imp"])):
        pre = "imp%d_" % i
        ctrl, X_N = _controller(g, pre)
        assert _same_facets(X_N.A, X_N.b, g[pre + "XN_A"], g[pre + "XN_b"]), "terminal set of system %d" % i
        xs, Uref, Vref = g[pre + "X"], g[pre + "U"], g[pre + "V"]
        valid = g[pre + "oracle_valid"].astype(bool)
        n_skipped += int((~valid).sum())
        Ub, Vb, okb = ctrl.feedforward(xs)          # one launch for the ten states
        fb = ctrl.feedback(xs)
        for j, x in enumerate(xs):
            assert bool(X_N.contains(x)) == bool(g[pre + "inside"][j])
            if not valid[j]:
                continue
            n_checked += 1
            feas = np.isfinite(Vref[j])
            assert bool(okb[j]) == feas, (i, j)
            u, V = ctrl.feedforward(x)               # scalar API: reference semantics
            if not feas:
                n_infeasible += 1
                assert u is None and V is None and ctrl.feedback(x) is None
                assert np.all(np.isnan(fb[j]))
                continue
            n_feasible += 1
            n_inside += int(g[pre + "inside"][j])
            assert abs(V - Vref[j]) <= 1e-6 * max(1., abs(Vref[j]))
            assert abs(Vb[j] - Vref[j]) <= 1e-6 * max(1., abs(Vref[j]))
            np.testing.assert_allclose(np.concatenate(u), Uref[j], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(Ub[j].reshape(-1), Uref[j], rtol=1e-6, atol=1e-7)
            nu = ctrl.S.nu
            np.testing.assert_allclose(ctrl.feedback(x), Uref[j][:nu], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(fb[j], Uref[j][:nu], rtol=1e-6, atol=1e-7)
    print("implicit: %d states checked (%d feasible, %d in the terminal set, %d infeasible), %d uncertified skipped"
          % (n_checked, n_feasible, n_inside, n_infeasible, n_skipped))
    assert n_checked >= 290 and n_inside > 50 and n_feasible > n_inside and n_infeasible >= 1 and n_skipped <= 5


def test_explicit_solution_against_reference_values():
    g = np.load(GOLD)
    checked = with_ref_explicit = 0
    for i in range(int(g["n_exp"])):
        pre = "exp%d_" % i
        ctrl, _ = _controller(g, pre)
        with pytest.raises(ValueError):
            ctrl.feedforward_explicit(np.zeros(ctrl.S.nx))
        ctrl.store_explicit_solution()
        xs, Uref, Vref = g[pre + "X"], g[pre + "U"], g[pre + "V"]
        valid = g[pre + "oracle_valid"].astype(bool)
        for j, x in enumerate(xs):
            if not valid[j]:
                continue
            u, V = ctrl.feedforward_explicit(x)
            if not np.isfinite(Vref[j]):
                assert u is None and V is None and ctrl.feedback_explicit(x) is None
                continue
            checked += 1
            # the reference's implicit solution (its own test's yardstick, :151-157) ...
            assert abs(V - Vref[j]) <= 1e-6 * max(1., abs(Vref[j]))
            np.testing.assert_allclose(np.concatenate(u), Uref[j], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(ctrl.feedback_explicit(x), Uref[j][:ctrl.S.nu], rtol=1e-5, atol=1e-6)
            # ... and its explicit solution where the reference could build one under the container's scipy
            if g[pre + "have_explicit"][j] and np.isfinite(g[pre + "Ve"][j]):
                with_ref_explicit += 1
                assert abs(V - g[pre + "Ve"][j]) <= 1e-6 * max(1., abs(Vref[j]))
                np.testing.assert_allclose(np.concatenate(u), g[pre + "Ue"][j], rtol=1e-5, atol=1e-6)
    print("explicit: %d feasible states checked, %d also against the reference's explicit solution" % (checked, with_ref_explicit))
    assert checked >= 40 and with_ref_explicit >= 30
