"""quadratic_program beyond 63 variables: the workgroup-per-problem path of csrc/qp_block.hpp (64 .. 1023 variables;
Cholesky, inverses and the NNLS matrix in device memory, nnls_block_kernel, recover) against the oracle's restatement
of pnnls.py:149-241 -- optimum, minimiser, multipliers and active set, the infeasible and the not-positive-definite
outcome, equalities folded as the reference folds them, batches with shared and per-problem matrices.
Tolerances: value 1e-8 relative, minimiser / multipliers 1e-6 relative to their largest entry (fp64)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _qp(rng, n, m, feasible=True):
    G = rng.randn(n, n)
    H = G.dot(G.T) / n + np.eye(n)
    f = rng.randn(n)
    A = rng.randn(m, n)
    x0 = rng.randn(n)
    b = A.dot(x0) + (rng.rand(m) if feasible else -rng.rand(m) - 0.1)
    return H, f, A, b


def _same(sol, ref):
    assert (sol['min'] is None) == (ref['min'] is None)
    if ref['min'] is None:
        return
    assert abs(sol['min'] - ref['min']) < 1e-8 * max(1.0, abs(ref['min']))
    assert np.allclose(sol['argmin'], ref['argmin'], rtol=0, atol=1e-6 * max(1.0, np.abs(ref['argmin']).max()))
    lam = np.asarray(ref['multiplier_inequality'])
    assert np.allclose(sol['multiplier_inequality'], lam, rtol=0, atol=1e-6 * max(1.0, np.abs(lam).max()))
    assert sol['active_set'] == ref['active_set']


@pytest.mark.parametrize("n,m", [(64, 50), (64, 300), (100, 200), (150, 600), (255, 400), (200, 1), (256, 300), (400, 500), (511, 700), (512, 300), (1023, 400),
                                 (10, 3000), (100, 5000), (300, 4000)])
def test_quadratic_program_many_variables_vs_oracle(oracle, n, m):
    from pympc_amd.optimization.solvers.hip import quadratic_program
    rng = np.random.RandomState(n + m)
    H, f, A, b = _qp(rng, n, m)
    ref = oracle.quadratic_program(H, f, A, b)
    sol = quadratic_program(H, f, A, b)
    assert ref['min'] is not None
    _same(sol, ref)
    x = sol['argmin']
    assert (A.dot(x) - b).max() < 1e-7 and len(sol['active_set']) > 0 or m == 1


def test_infeasible_equalities_and_not_positive_definite(oracle):
    from pympc_amd.optimization.solvers.hip import quadratic_program
    rng = np.random.RandomState(3)
    n, m = 80, 120
    H, f, A, b = _qp(rng, n, m)
    # x_0 <= -1 and x_0 >= 1
    A2 = np.vstack((A, np.eye(n)[0], -np.eye(n)[0])); b2 = np.concatenate((b, [-1.0, -1.0]))
    ref = oracle.quadratic_program(H, f, A2, b2); sol = quadratic_program(H, f, A2, b2)
    assert ref['min'] is None and sol['min'] is None and sol['argmin'] is None
    # equalities (pnnls.py:202-207)
    C = rng.randn(5, n); d = C.dot(rng.randn(n)) * 0.1
    ref = oracle.quadratic_program(H, f, A, b + 5.0, C, d); sol = quadratic_program(H, f, A, b + 5.0, C, d)
    _same(sol, ref)
    assert np.allclose(sol['multiplier_equality'], ref['multiplier_equality'], atol=1e-6 * max(1.0, np.abs(ref['multiplier_equality']).max()))
    assert np.abs(C.dot(sol['argmin']) - d).max() < 1e-7
    # indefinite Hessian: numpy.linalg.LinAlgError in the reference (pnnls.py:210)
    Hbad = H.copy(); Hbad[n - 1, n - 1] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        quadratic_program(Hbad, f, A, b)


def test_batches_shared_and_per_problem(oracle):
    from pympc_amd.optimization.solvers.hip import quadratic_program_batch
    rng = np.random.RandomState(8)
    n, m, B = 70, 90, 6
    H, f, A, b = _qp(rng, n, m)
    F = rng.randn(B, n); Bv = b[None, :] + rng.rand(B, m)
    Bv[4] = b - 100.0 * (np.arange(m) % 2)            # one of them most likely infeasible
    res = quadratic_program_batch(H, F, A, Bv, want_iters=True)
    for k in range(B):
        _same(res.sol(k), oracle.quadratic_program(H, F[k], A, Bv[k]))
    Hs = np.stack([H + 0.1 * k * np.eye(n) for k in range(B)]); As = A[None] + 0.01 * rng.randn(B, m, n)
    res = quadratic_program_batch(Hs, F, As, Bv)
    for k in range(B):
        _same(res.sol(k), oracle.quadratic_program(Hs[k], F[k], As[k], Bv[k]))
    with pytest.raises(NotImplementedError):
        quadratic_program_batch(np.eye(1024), np.zeros((1, 1024)), np.ones((1, 1024)), np.ones((1, 1)))


def test_iteration_limit_raises_like_scipy():
    from pympc_amd import _hip
    from pympc_amd.optimization.solvers.hip import quadratic_program
    n = 66
    f = np.zeros(n); f[0] = -3.0
    with _hip.tuned(nnls_itmax_factor=0):
        with pytest.raises(RuntimeError, match="Maximum number of iterations"):
            quadratic_program(np.eye(n), f, np.eye(n)[:1], np.array([1.]))


def test_mpqp_with_many_variables_is_solved_state_by_state(oracle):
    """An mpQP of 72 variables (a condensed double-integrator MPC of horizon 72: beyond the prepared form's 63) through
    PreparedMpqp.solve_batch: every state a QP of its own on the workgroup kernels; decisions, active sets and values
    against the oracle's loop of MultiParametricQuadraticProgram.solve (parametric_programs.py:155-179)."""
    from pympc_amd.optimization.solvers.hip import PreparedMpqp
    N, nx = 72, 2
    Ad = np.array([[1., 1.], [0., 1.]]); Bd = np.array([[.5], [1.]])
    # x_t = Ad^t x_0 + sum_k Ad^(t-1-k) Bd u_k
    Phi = np.vstack([np.linalg.matrix_power(Ad, t + 1) for t in range(N)])                       # [N nx x nx]
    Gam = np.zeros((N * nx, N))
    for t in range(N):
        for k in range(t + 1):
            Gam[t * nx:(t + 1) * nx, k] = np.linalg.matrix_power(Ad, t - k).dot(Bd)[:, 0]
    Qb = np.eye(N * nx); Rb = 0.1 * np.eye(N)
    model = dict(Huu=Gam.T.dot(Qb).dot(Gam) + Rb, Hux=Gam.T.dot(Qb).dot(Phi), Hxx=Phi.T.dot(Qb).dot(Phi) + np.eye(nx),
                 fu=np.zeros(N), fx=np.zeros(nx), g=0.25,
                 Au=np.vstack((np.eye(N), -np.eye(N), Gam[1::nx][:40], -Gam[1::nx][:40])),           # |u| <= 1, |v_t| <= 1.5
                 Ax=np.vstack((np.zeros((2 * N, nx)), Phi[1::nx][:40], -Phi[1::nx][:40])),
                 b=np.concatenate((np.ones(2 * N), 1.5 * np.ones(80))))
    rng = np.random.RandomState(2)
    X = rng.uniform(-1., 1., (24, nx)) * np.array([6., 1.6])
    mp = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], model["g"], model["Au"],
                      model["Ax"], model["b"])
    assert mp.large
    res = mp.solve_batch(X, n_out=3)
    ref = oracle.mpqp_solve_batch(model, X)
    st = res.numpy("status")
    assert np.array_equal(st, ref["status"]) and 0 < (st == 0).sum() < len(X) or (st == 0).all()
    ok = st == 0
    assert np.array_equal(res.active_mask()[ok], ref["active"][ok])
    assert np.allclose(res.numpy("min")[ok], ref["min"][ok], rtol=1e-8, atol=1e-8)
    assert np.allclose(res.numpy("argmin")[ok], ref["argmin"][ok, :3], rtol=0, atol=1e-6)
    assert res.numpy("argmin").shape == (24, 3) and np.isnan(res.numpy("min")[~ok]).all()
    lam = ref["multiplier_inequality"][ok]
    assert np.allclose(res.numpy("multiplier_inequality")[ok], lam, rtol=0, atol=1e-6 * max(1.0, np.abs(lam).max()))


def test_linear_mpc_with_horizon_70_end_to_end(oracle):
    """ModelPredictiveController on the double integrator with N = 70 (70 optimisation variables: refused until the
    workgroup-per-problem QP path existed): condensing kernel -> mpQP without prepared form -> feedforward / feedback
    over a batch of states, against the oracle's per-state loop on the same condensed problem."""
    from pympc_amd.control.controllers import ModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    A = np.array([[1., 1.], [0., 1.]]); B = np.array([[.5], [1.]])
    S = LinearSystem(A, B)
    X = Polyhedron.from_bounds(np.array([-8., -2.]), np.array([8., 2.]))
    U = Polyhedron.from_bounds(np.array([-1.]), np.array([1.]))
    D = X.cartesian_product(U)
    Q, R = np.eye(2), np.eye(1)
    P, K = S.solve_dare(Q, R)
    X_N = S.mcais(K, D)
    ctrl = ModelPredictiveController(S, 70, Q, R, P, D, X_N)
    mp = ctrl.mpqp
    assert mp.A['u'].shape[1] == 70 and mp.prepared().large
    rng = np.random.RandomState(5)
    xs = rng.uniform(-1., 1., (12, 2)) * np.array([7., 1.8])
    model = dict(Huu=mp.H['uu'], Hux=mp.H['ux'], Hxx=mp.H['xx'], fu=mp.f['u'], fx=mp.f['x'], g=mp.g, Au=mp.A['u'], Ax=mp.A['x'], b=mp.b)
    ref = oracle.mpqp_solve_batch(model, xs)
    Ub, Vb, okb = ctrl.feedforward(xs)
    ok = ref["status"] == 0
    assert np.array_equal(np.asarray(okb, bool), ok) and 0 < ok.sum()
    assert np.allclose(np.asarray(Vb)[ok], ref["min"][ok], rtol=1e-7, atol=1e-7)
    assert np.allclose(np.asarray(Ub)[ok].reshape(int(ok.sum()), -1), ref["argmin"][ok], rtol=0, atol=1e-6)
    # one state through the scalar API
    k = int(np.where(ok)[0][0])
    u_seq, V = ctrl.feedforward(xs[k])
    assert abs(V - ref["min"][k]) < 1e-7 * max(1.0, abs(ref["min"][k])) and abs(ctrl.feedback(xs[k])[0] - ref["argmin"][k, 0]) < 1e-6


def test_polyhedron_empty_in_many_dimensions(oracle):
    """Polyhedron.empty (the minimum-norm QP of polyhedron.py:505-526) in 80 and 200 dimensions: the reference's own
    formulation now that the QP takes that many variables, with and without equalities."""
    from pympc_amd.geometry.polyhedron import Polyhedron
    rng = np.random.RandomState(17)
    for n in (80, 200):
        P = Polyhedron.from_bounds(-np.ones(n), np.ones(n))
        assert not P.empty
        A = np.vstack((P.A, np.ones((1, n)), -np.ones((1, n)))); b = np.concatenate((P.b, [-1.0, 0.5]))      # sum <= -1 and sum >= -0.5
        assert Polyhedron(A, b).empty
        C = rng.randn(3, n); d = C.dot(rng.uniform(-.5, .5, n))
        assert not Polyhedron(P.A, P.b, C, d).empty
        assert Polyhedron(P.A, P.b, np.ones((1, n)), np.array([n + 1.0])).empty                               # sum = n + 1 outside the box
        ref = oracle.quadratic_program(np.eye(n), np.zeros(n), P.A, P.b, C, d)
        assert ref['min'] is not None


def test_mpqp_with_more_constraints_than_the_prepared_form_takes(oracle):
    """An mpQP of 6 variables and 2600 constraints (beyond the prepared form's 2048): solved state by state."""
    from pympc_amd.optimization.solvers.hip import PreparedMpqp
    rng = np.random.RandomState(21)
    n, p, m = 6, 3, 2600
    G = rng.randn(n, n)
    model = dict(Huu=G.dot(G.T) + np.eye(n), Hux=rng.randn(n, p), Hxx=np.eye(p) * 3.0, fu=rng.randn(n), fx=rng.randn(p), g=1.5,
                 Au=rng.randn(m, n), Ax=0.3 * rng.randn(m, p), b=1.0 + rng.rand(m))
    mp = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], model["g"], model["Au"], model["Ax"], model["b"])
    assert mp.large
    X = rng.uniform(-1., 1., (16, p)) * 2.0
    res = mp.solve_batch(X)
    ref = oracle.mpqp_solve_batch(model, X)
    st = res.numpy("status")
    assert np.array_equal(st, ref["status"]) and (st == 0).any()
    ok = st == 0
    assert np.array_equal(res.active_mask()[ok], ref["active"][ok])
    assert np.allclose(res.numpy("min")[ok], ref["min"][ok], rtol=1e-8, atol=1e-8)
    assert np.allclose(res.numpy("argmin")[ok], ref["argmin"][ok], rtol=0, atol=1e-7)


def test_bounded_with_more_than_64_rows_and_dimensions():
    """Polyhedron.bounded by Stiemke's LP (polyhedron.py:528-576) for a polytope with 150 rows in 70 dimensions: refused
    while the LP paths ended at 64 variables."""
    from pympc_amd.geometry.polyhedron import Polyhedron
    rng = np.random.RandomState(3)
    n = 70
    box = Polyhedron.from_bounds(-np.ones(n), np.ones(n))
    extra = rng.randn(10, n)
    P = Polyhedron(np.vstack((box.A, extra)), np.concatenate((box.b, np.full(10, 20.0))))
    assert P.bounded
    keep = np.ones(box.A.shape[0], bool)
    keep[int(np.where((box.A[:, 0] > 0.5))[0][0])] = False           # drop x_0 <= 1 ...
    more = np.zeros((10, n)); more[:, 0] = -1.0                      # ... and add ten rows -x_0 <= 20: the ray along +x_0 stays
    Q = Polyhedron(np.vstack((box.A[keep], more)), np.concatenate((box.b[keep], np.full(10, 20.0))))
    assert not Q.bounded
