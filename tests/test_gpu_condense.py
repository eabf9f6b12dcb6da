"""GPU parity of the condensing kernel (pympc_condense_batch, fp64 MFMA) and of the controllers built on it.

  * golden: the condensed mpQP matrices the reference produced for C1 / C2 / the notebook model and for four mode
    sequences of the hybrid cart-pole (tests/golden/*_model.npz, hybrid_seqs.npz: inputs S, Q, R, P, D, X_N and the
    reference's Huu ... b);
  * oracle: oracle_condense (restatement of condense_pwa_system + condense_optimal_control_problem) on random PWA
    systems of other sizes, including nx > 4 (more than one MFMA k-step), nu > 1 and ragged stage domains;
  * end to end: ModelPredictiveController.feedback / feedforward on the golden states; get_mpqp(ms).solve(x0) over a
    batch of mode sequences against the reference's per-sequence optimum (config C5).
Tolerance: 1e-9 relative on the matrices (fp64, different summation order), 1e-6 on optimal values / minimisers.
"""
import numpy as np
import pytest

from conftest import golden, unpack_golden_active

pytestmark = pytest.mark.gpu

KEYS = ('Huu', 'Hux', 'Hxx', 'fu', 'fx', 'Au', 'Ax', 'b')


def _close(a, b, tol=1e-9):
    return np.abs(np.asarray(a) - np.asarray(b)).max() <= tol * max(1., np.abs(b).max())


def _linear_controller(g):
    from pympc_amd.control.controllers import ModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    S = LinearSystem(g['S_A'], g['S_B'])
    return ModelPredictiveController(S, int(g['N']), g['Q'], g['R'], g['P'], Polyhedron(g['D_A'], g['D_b']),
                                     Polyhedron(g['XN_A'], g['XN_b']))


@pytest.mark.parametrize("model,solves", [("c1_model", "c1_solves"), ("nb_model", "nb_solves"), ("c2_model", "c2_solves_s03")])
def test_linear_mpc_condense_and_feedback_vs_reference_golden(model, solves):
    g = golden(model); rec = golden(solves)
    ctrl = _linear_controller(g)
    q = ctrl.mpqp
    got = dict(Huu=q.H['uu'], Hux=q.H['ux'], Hxx=q.H['xx'], fu=q.f['u'], fx=q.f['x'], Au=q.A['u'], Ax=q.A['x'], b=q.b)
    for k in KEYS:
        assert got[k].shape == g[k].shape, k
        assert _close(got[k], g[k]), k
    assert abs(q.g - float(g['g'])) < 1e-12
    # feedback over the golden states: first input (or NaN), V, feasibility
    X = rec['X']; valid = rec['oracle_valid'].astype(bool); feas = rec['feasible'].astype(bool)
    nu = g['S_B'].shape[1]
    U, V, ok = ctrl.feedforward(X)
    assert np.array_equal(ok[valid], feas[valid])
    sel = valid & feas
    assert np.allclose(U.reshape(len(X), -1)[sel], rec['argmin'][sel], rtol=1e-6, atol=1e-6)
    assert np.allclose(V[sel], rec['min'][sel], rtol=1e-6, atol=1e-6)
    u0 = ctrl.feedback(X)
    assert u0.shape == (len(X), nu)
    assert np.allclose(u0[sel], rec['argmin'][sel][:, :nu], rtol=1e-6, atol=1e-6)
    assert np.all(np.isnan(u0[~ok]))
    # scalar API: reference semantics
    i = int(np.where(sel)[0][0])
    uf, Vf = ctrl.feedforward(X[i])
    assert len(uf) == int(g['N']) and abs(Vf - rec['min'][i]) < 1e-6 * max(1., abs(rec['min'][i]))
    assert np.allclose(ctrl.feedback(X[i]), rec['argmin'][i][:nu], rtol=1e-6, atol=1e-6)
    bad = np.where(valid & ~feas)[0]
    if bad.size:
        assert ctrl.feedback(X[bad[0]]) is None and ctrl.feedforward(X[bad[0]]) == (None, None)


def _hybrid():
    from pympc_amd.control.controllers import HybridModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import AffineSystem, PieceWiseAffineSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    h = golden("hybrid_seqs")
    systems = [AffineSystem(h['S_A'][i], h['S_B'][i], h['S_c'][i]) for i in range(3)]
    domains = [Polyhedron(h['D%d_A' % i], h['D%d_b' % i]) for i in range(3)]
    S = PieceWiseAffineSystem(systems, domains)
    N = h['seqs'].shape[1]
    return h, HybridModelPredictiveController(S, N, h['Q'], h['R'], h['P'], Polyhedron(h['XN_A'], h['XN_b']))


def test_hybrid_get_mpqp_vs_reference_golden():
    h, ctrl = _hybrid()
    for k in range(4):
        q = ctrl.get_mpqp([int(z) for z in h['seqs'][k]])
        got = dict(Huu=q.H['uu'], Hux=q.H['ux'], Hxx=q.H['xx'], fu=q.f['u'], fx=q.f['x'], Au=q.A['u'], Ax=q.A['x'], b=q.b)
        for key in KEYS:
            ref = h['seq%d_%s' % (k, key)]
            assert got[key].shape == ref.shape, (k, key)
            assert _close(got[key], ref), (k, key)
        assert abs(q.g - float(h['seq%d_g' % k])) < 1e-9 * max(1., abs(float(h['seq%d_g' % k])))
        sol = q.solve(h['x0'])
        if np.isfinite(h['min'][k]):
            assert abs(sol['min'] - h['min'][k]) < 1e-6 * max(1., abs(h['min'][k]))
            assert np.allclose(sol['argmin'], h['argmin'][k], rtol=1e-6, atol=1e-6)
        else:
            assert sol['min'] is None


def test_hybrid_mode_sequence_batch_vs_reference_golden():
    """Config C5: get_mpqp(ms).solve(x0) for a batch of mode sequences in two launches."""
    h, ctrl = _hybrid()
    res, batch = ctrl.solve_mode_sequences(h['x0'], h['seqs'], want_iters=True)
    valid = h['oracle_valid'].astype(bool)
    feas = np.isfinite(h['min'])
    assert np.array_equal(res.feasible[valid], feas[valid])
    sel = valid & feas
    assert sel.sum() >= 1
    assert np.allclose(res.numpy('min')[sel], h['min'][sel], rtol=1e-6, atol=1e-6)
    assert np.allclose(res.numpy('argmin')[sel], h['argmin'][sel], rtol=1e-6, atol=1e-6)
    # pad rows (0 <= 1) never carry a multiplier
    mult = res.numpy('multiplier_inequality')
    pad = ~batch.row_mask()
    assert np.all(mult[res.feasible][pad[res.feasible]] == 0.)
    # best sequence = argmin over the feasible ones
    u, V, ms = ctrl.feedforward_over(h['x0'], h['seqs'])
    kbest = int(np.nanargmin(np.where(feas, h['min'], np.nan)))
    assert abs(V - h['min'][kbest]) < 1e-6 * max(1., abs(V)) and ms == [int(z) for z in h['seqs'][kbest]]
    # per-sequence parameters (x_stride = p): same answers when every sequence gets its own copy of x0
    res2 = batch.solve(np.tile(h['x0'], (len(h['seqs']), 1)))
    assert np.array_equal(res2.numpy('status'), res.numpy('status'))
    assert np.array_equal(res2.numpy('min')[res.feasible], res.numpy('min')[res.feasible])


def _random_pwa(rng, nx, nu, nm, rows):
    systems = [(rng.randn(nx, nx) * .6, rng.randn(nx, nu), rng.randn(nx) * (i > 0)) for i in range(nm)]
    domains = [(rng.randn(r, nx + nu), 1. + rng.rand(r)) for r in rows]
    return systems, domains


@pytest.mark.parametrize("nx,nu,nm,N,rows,rN", [(4, 1, 3, 10, (11, 12, 11), 30), (3, 2, 2, 7, (5, 9), 17), (6, 2, 3, 9, (16, 17, 33), 40),
                                                  (9, 1, 2, 12, (20, 20), 1), (16, 3, 2, 5, (7, 3), 64), (2, 1, 1, 60, (6,), 18)])
def test_condense_batch_vs_oracle(oracle, nx, nu, nm, N, rows, rN):
    from pympc_amd.control.condense import PwaTables, condense_batch
    from pympc_amd.dynamics.discrete_time_systems import AffineSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    rng = np.random.RandomState(nx * 100 + N)
    systems, domains = _random_pwa(rng, nx, nu, nm, rows)
    Q = rng.randn(nx, nx); Q = Q.dot(Q.T) + np.eye(nx)
    R = rng.randn(nu, nu); R = R.dot(R.T) + np.eye(nu)
    P = rng.randn(nx, nx); P = P.dot(P.T) + np.eye(nx)
    XA = rng.randn(rN, nx); Xb = 1. + rng.rand(rN)
    tables = PwaTables([AffineSystem(*s) for s in systems], [Polyhedron(*d) for d in domains], Q, R, P, Polyhedron(XA, Xb))
    seqs = rng.randint(0, nm, (37, N))
    batch = condense_batch(tables, seqs, want_prediction=True)
    batch.check()
    for k in (0, 1, 17, 36):
        ref = oracle.condense(systems, domains, Q, R, P, XA, Xb, seqs[k])
        q = batch.get(k)
        got = dict(Huu=q.H['uu'], Hux=q.H['ux'], Hxx=q.H['xx'], fu=q.f['u'], fx=q.f['x'], Au=q.A['u'], Ax=q.A['x'], b=q.b)
        for key in KEYS:
            assert got[key].shape == ref[key].shape, key
            assert _close(got[key], ref[key]), (k, key)
        assert abs(q.g - ref['g']) < 1e-9 * max(1., abs(ref['g']))
        Ab, Bb, cb = batch.prediction(k)
        assert _close(Ab, ref['Abar']) and _close(Bb, ref['Bbar']) and _close(cb, ref['cbar'])


def test_condense_pwa_system_and_errors(oracle):
    from pympc_amd.control.condense import PwaTables, condense_batch
    from pympc_amd.dynamics.discrete_time_systems import AffineSystem, LinearSystem, PieceWiseAffineSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    rng = np.random.RandomState(3)
    A = rng.randn(3, 3) * .5; B = rng.randn(3, 2)
    S = LinearSystem(A, B)
    Ab, Bb = S.condense(5)                      # test_discrete_time_systems.py:22-41: condense == simulate
    x0 = rng.randn(3); u = [rng.randn(2) for _ in range(5)]
    xs = np.concatenate(S.simulate(x0, u))
    assert np.allclose(Ab.dot(x0) + Bb.dot(np.concatenate(u)), xs, rtol=1e-10, atol=1e-12)
    Sa = AffineSystem(A, B, rng.randn(3))
    Ab, Bb, cb = Sa.condense(4)
    xs = np.concatenate(Sa.simulate(x0, u[:4]))
    assert np.allclose(Ab.dot(x0) + Bb.dot(np.concatenate(u[:4])) + cb, xs, rtol=1e-10, atol=1e-12)
    # a mode index outside [0, nm) is reported, not dereferenced
    D = Polyhedron.from_bounds(-np.ones(5), np.ones(5))
    tables = PwaTables([Sa], [D], np.eye(3), np.eye(2), np.eye(3), Polyhedron.from_bounds(-np.ones(3), np.ones(3)))
    with pytest.raises(ValueError):
        condense_batch(tables, np.array([[0, 0, 1, 0]])).check()
    with pytest.raises(ValueError):
        PieceWiseAffineSystem([Sa, Sa], [D])


def test_hybrid_bigM_vs_reference_golden():
    """Big-Ms of the hybrid controller's MIQP set-up (controllers.py:289-405): the reference solves 2 nx + rows LPs per
    pair of modes one after the other; here one batched launch per domain.  Golden values from the reference's own
    methods (tests/golden/make_golden.py gen_bigm, certified against HiGHS)."""
    from pympc_amd.control.controllers import HybridModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import AffineSystem, PieceWiseAffineSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    h = golden("hybrid_bigm")
    systems = [AffineSystem(h['S_A'][i], h['S_B'][i], h['S_c'][i]) for i in range(3)]
    domains = [Polyhedron(h['D%d_A' % i], h['D%d_b' % i]) for i in range(3)]
    S = PieceWiseAffineSystem(systems, domains)
    ctrl = HybridModelPredictiveController(S, 3, np.eye(4), np.eye(1), np.eye(4), domains[1])
    alpha, beta = ctrl._get_bigM_dynamics()
    gamma = ctrl._get_bigM_domains()
    for i in range(3):
        for j in range(3):
            for name, got in (("alpha", alpha), ("beta", beta), ("gamma", gamma)):
                ref = h["%s_%d_%d" % (name, i, j)]
                assert got[i][j].shape == ref.shape, (name, i, j)
                assert np.allclose(got[i][j], ref, rtol=1e-7, atol=1e-7), (name, i, j)
    # the tightest choice: in its own domain a mode's constraints need no relaxation (gamma_ii <= 0)
    assert all(gamma[i][i].max() <= 1e-9 for i in range(3))
    # big-Ms over the graphs of the dynamics (examples/pwa_from_urdf/utils.py:349-389), values from the reference's LP path
    from pympc_amd.control.controllers import big_m, graph_representation
    m = big_m(graph_representation(S))
    for i in range(3):
        for j in range(3):
            ref = h["graph_m_%d_%d" % (i, j)]
            assert m[i][j].shape == ref.shape and np.allclose(m[i][j], ref, rtol=1e-7, atol=1e-7), (i, j)
            assert np.all((m[i][j] == 0.) == (ref == 0.))
    # an unbounded domain has no big-M
    bad = PieceWiseAffineSystem.__new__(PieceWiseAffineSystem)
    bad.affine_systems = systems
    bad.domains = [Polyhedron(domains[0].A[:2], domains[0].b[:2])] + domains[1:]
    ctrl_bad = HybridModelPredictiveController(bad, 3, np.eye(4), np.eye(1), np.eye(4), domains[1])
    with pytest.raises(ValueError):
        ctrl_bad._get_bigM_domains()


def test_mfma_f64_needs_no_wait_state_pad():
    """csrc/probes/mfma_f64_hazard.hip: chains of v_mfma_f64_16x16x4_f64 whose accumulator is stored immediately, with and
    without explicit wait states.  Both variants must be exact: the condensing kernel relies on the compiler's own
    hazard handling for straight-line MFMA chains (round 3: no pads, no register constraints; the wrong entries of
    rounds 1 / 2 came from a branch between an MFMA and the read of its accumulator, csrc/condense.hpp header and
    profiles/r03/mfma_settle_isa.txt, removed by making the column-tile count a template parameter).  If the unpadded
    variant ever returns a wrong entry, the wait states have to come back."""
    import ctypes
    from pympc_amd.build import PROBE_PATH
    lib = ctypes.CDLL(PROBE_PATH)
    lib.pympc_probe_mfma_f64.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong),
                                         ctypes.POINTER(ctypes.c_int)]
    for ks in (1, 2, 4):
        for pad in (1, 0):
            wrong = ctypes.c_longlong(-1); mask = ctypes.c_int(0)
            assert lib.pympc_probe_mfma_f64(pad, ks, 2048, ctypes.byref(wrong), ctypes.byref(mask)) == 0
            assert wrong.value == 0, "k_steps %d pad %d: %d wrong entries, accumulator registers %#x" % (ks, pad, wrong.value, mask.value)
