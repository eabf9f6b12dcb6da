"""HybridModelPredictiveController.feedforward / feedback: exact pruned enumeration of the mode sequences in place of the
reference's MIQP (controllers.py:611-666, Gurobi: absent, so the yardstick is brute force over ALL mode sequences --
the property test_controllers.py:270-289 states the other way round: get_mpqp(ms).solve(x) equals the MIQP optimum for
the optimal ms).  PWA cart-pole with two walls of examples/hybrid_mpc_2.ipynb (tests/golden/hybrid_seqs.npz)."""
import itertools

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _controller(N):
    from pympc_amd.control.controllers import HybridModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import AffineSystem, PieceWiseAffineSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    h = golden("hybrid_seqs")
    systems = [AffineSystem(h['S_A'][i], h['S_B'][i], h['S_c'][i]) for i in range(3)]
    domains = [Polyhedron(h['D%d_A' % i], h['D%d_b' % i]) for i in range(3)]
    return HybridModelPredictiveController(PieceWiseAffineSystem(systems, domains), N, h['Q'], h['R'], h['P'],
                                           Polyhedron(h['XN_A'], h['XN_b'])), h


def test_search_equals_brute_force_n10():
    ctrl, h = _controller(10)
    x0 = h['x0']
    allseq = np.array(list(itertools.product(range(3), repeat=10)), dtype=np.int32)
    u_bf, V_bf, ms_bf = ctrl.feedforward_over(x0, allseq)
    u, xs, ms, V = ctrl.feedforward(x0)
    assert ms == ms_bf and abs(V - V_bf) <= 1e-9 * abs(V_bf)
    assert np.allclose(np.concatenate(u), np.concatenate(u_bf), rtol=1e-9, atol=1e-9)
    assert np.allclose(ctrl.feedback(x0), u_bf[0])
    # the golden value of this sequence (reference's condensing + pnnls QP)
    k = [list(s) for s in h['seqs']].index(ms)
    assert abs(V - h['min'][k]) <= 1e-6 * abs(V)
    # the trajectory follows the modes it claims
    for t in range(10):
        D = ctrl.S.domains[ms[t]]
        assert D.contains(np.concatenate((xs[t], u[t])), tol=1e-6)


@pytest.mark.parametrize("seed", [0, 1])
def test_search_equals_brute_force_random_states(seed):
    N = 6
    ctrl, h = _controller(N)
    allseq = np.array(list(itertools.product(range(3), repeat=N)), dtype=np.int32)
    rng = np.random.RandomState(seed)
    x_max = np.array([1.5, np.pi / 6., 7., 5.])
    n_feas = n_switch = 0
    for _ in range(12):
        x0 = rng.uniform(-.7, .7, 4) * x_max
        u_bf, V_bf, ms_bf = ctrl.feedforward_over(x0, allseq)
        u, xs, ms, V = ctrl.feedforward(x0)
        if V_bf is None:
            assert u is None and V is None and ctrl.feedback(x0) is None
            continue
        n_feas += 1
        n_switch += int(len(set(ms_bf)) > 1)
        assert abs(V - V_bf) <= 1e-8 * max(1., abs(V_bf))
        assert ms == ms_bf or abs(V - V_bf) <= 1e-12 * max(1., abs(V_bf))     # (equal-cost sequences: either one)
    assert n_feas >= 2


def test_search_reaches_horizon_14():
    """3^14 = 4.8 M sequences: the sweep visits a small fraction of them."""
    ctrl, h = _controller(14)
    u, xs, ms, V = ctrl.feedforward(h['x0'])
    assert u is not None and len(ms) == 14 and np.isfinite(V)
    res, _ = ctrl.solve_mode_sequences(h['x0'], np.array([ms], dtype=np.int32), want_mult=False, want_active=False)
    assert abs(float(res.numpy('min')[0]) - V) <= 1e-9 * abs(V)
    # a neighbour of the optimum is not better
    for t in (0, 7, 13):
        for z in range(3):
            alt = list(ms); alt[t] = z
            r, _ = ctrl.solve_mode_sequences(h['x0'], np.array([alt], dtype=np.int32), want_mult=False, want_active=False)
            if int(r.numpy('status')[0]) == 0:
                assert float(r.numpy('min')[0]) >= V - 1e-9 * abs(V)
