"""Problems placed ON the reference's feasibility thresholds.

QP (pnnls.py:230): feasible iff the NNLS residual r > 1e-7.  The residual of a feasible QP is 1 / sqrt(1 + 2 (V - V_unc)):
it falls below the threshold when the unconstrained minimiser lies ~1e7 away from the feasible set, and the reference
then reports a FEASIBLE problem as infeasible.  The linear term is scaled over nine decades and the HIP decision must
equal the oracle's (the pinned restatement of the reference, which reports r) wherever r is not within 0.1 % of the
threshold; the records inside that band are counted and must be few.  Both QP kernels: per-problem launch and the
shared-matrix mpQP launch (lane path + wave kernel), where the scale is the parameter.

LP (pnnls.py:137): solvable iff the PNNLS residual r < 1e-7.  POLICY of this build: linear_program() uses a dual
active-set method and calls a row violated above 1e-9 max(1, |b_j|), so an LP that is infeasible by a margin between the
two tolerances is "solvable" for the reference and infeasible here (the stricter answer); outside that band the decisions
agree, inside it linear_program_pnnls() -- the reference's formulation on the device -- reproduces the reference's.
Degenerate LPs (non-unique optimal vertex or multipliers): min and KKT validity are compared, argmin / active_set may be
any optimal pair (tests/test_gpu_lp.py).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _box_qp(rng, n):
    """Strictly convex QP over the unit box; with the linear term scaled by s the unconstrained minimiser lies ~s away
    from the box and the NNLS residual of Bemporad's transformation, r = 1 / sqrt(1 + m'(M M')^-1 m), falls like 1 / s."""
    L = rng.randn(n, n); H = L.dot(L.T) + n * np.eye(n)
    f0 = rng.randn(n)
    A = np.vstack((np.eye(n), -np.eye(n)))
    b = np.ones(2 * n)
    return H, f0, A, b


def test_qp_feasibility_edge_per_problem_kernel(oracle):
    from pympc_amd.optimization.solvers.hip import quadratic_program_batch
    rng = np.random.RandomState(21)
    n = 4
    scales = 10. ** np.linspace(2, 11, 181)
    Hs, fs, As, bs, rn, feas = [], [], [], [], [], []
    for k, sc in enumerate(scales):
        H, f0, A, b = _box_qp(rng, n)
        if k % 6 == 5:                       # truly infeasible ones in between: x_0 <= 1 and x_0 >= 1 + 1e-3
            A = np.vstack((A, -np.eye(n)[:1])); b = np.concatenate((b, [-1.001]))
        else:
            A = np.vstack((A, np.zeros((1, n)))); b = np.concatenate((b, [1.]))
        info = {}
        sol = oracle.quadratic_program(H, sc * f0, A, b, info=info)
        Hs.append(H); fs.append(sc * f0); As.append(A); bs.append(b)
        rn.append(info["rnorm"]); feas.append(sol["min"] is not None)
    rn = np.array(rn); feas = np.array(feas)
    res = quadratic_program_batch(np.array(Hs), np.array(fs), np.array(As), np.array(bs))
    got = res.numpy("status") == 0
    band = np.abs(rn / 1e-7 - 1.0) < 1e-3
    near = (rn > 1e-9) & (rn < 1e-5)
    print("QP edge: %d feasible / %d infeasible by the reference's rule, %d residuals in [1e-9, 1e-5], %d within 0.1 %% of "
          "the threshold (masked)" % (feas.sum(), (~feas).sum(), near.sum(), band.sum()))
    assert near.sum() >= 20 and feas.sum() >= 20 and (~feas).sum() >= 20 and band.sum() <= 2
    assert np.array_equal(got[~band], feas[~band])
    # values against the oracle only where the reference's own recovery x = -H^-1(f + A'y / r^2) still has its digits
    # (1 + m'y = r^2 is a difference of numbers near 1: its error grows like 1e-16 / r^2, in the kernels as in the reference)
    ok = feas & (rn > 1e-3)
    ref_min = np.array([oracle.quadratic_program(Hs[k], fs[k], As[k], bs[k])["min"] for k in np.where(ok)[0]])
    assert np.allclose(res.numpy("min")[ok], ref_min, rtol=1e-6, atol=1e-9)


def test_qp_feasibility_edge_shared_launch(oracle):
    """mpQP whose parameter scales the linear term: f(x) = Hux x with Hux = f0 (one column)."""
    from pympc_amd.optimization.solvers.hip import PreparedMpqp
    rng = np.random.RandomState(22)
    n = 5
    H, f0, A, b = _box_qp(rng, n)
    m = A.shape[0]
    X = np.concatenate((10. ** np.linspace(2, 11, 241), -10. ** np.linspace(2, 11, 61), [0.0]))[:, None]
    prep = PreparedMpqp(H, f0[:, None], np.zeros((1, 1)), np.zeros(n), np.zeros(1), 0., A, np.zeros((m, 1)), b)
    res = prep.solve_batch(X)
    again = prep.solve_batch(X)                                   # with whatever certificates the first pass learned
    rn = np.zeros(len(X)); feas = np.zeros(len(X), bool); ref_min = np.full(len(X), np.nan)
    for k, x in enumerate(X):
        info = {}
        sol = oracle.quadratic_program(H, f0 * x[0], A, b, info=info)
        rn[k] = info["rnorm"]; feas[k] = sol["min"] is not None
        if feas[k]:
            ref_min[k] = sol["min"]
    band = np.abs(rn / 1e-7 - 1.0) < 1e-3
    near = (rn > 1e-9) & (rn < 1e-5)
    print("mpQP edge: %d feasible / %d infeasible by the reference's rule, %d residuals in [1e-9, 1e-5], %d masked"
          % (feas.sum(), (~feas).sum(), near.sum(), band.sum()))
    assert near.sum() >= 20 and feas.sum() >= 20 and (~feas).sum() >= 20 and band.sum() <= 2
    for r in (res, again):
        assert np.array_equal((r.numpy("status") == 0)[~band], feas[~band])
    ok = feas & (rn > 1e-3)   # (closer to the edge the reference's own x = -H^-1(f + A'y / r^2) has lost its digits)
    assert np.allclose(res.numpy("min")[ok], ref_min[ok], rtol=1e-6, atol=1e-9)


def _margin_lp(rng, n, delta):
    """LP over a box whose first coordinate must also satisfy x_0 >= 1 + delta while the box says x_0 <= 1:
    infeasible by the margin delta (delta <= 0: feasible, the extra row is redundant or touching)."""
    A = np.vstack((np.eye(n), -np.eye(n), -np.eye(n)[:1]))
    b = np.concatenate((np.ones(n), np.ones(n), [-(1. + delta)]))
    f = rng.randn(n)
    return f, A, b


def test_lp_solvability_edge_policy(oracle):
    from pympc_amd.optimization.solvers.hip_lp import linear_program, linear_program_pnnls
    rng = np.random.RandomState(23)
    n = 3
    deltas = np.concatenate((10. ** np.linspace(-13, -3, 41), [-1e-3, -1e-6, -1e-9, 0.0]))
    agree = band = 0
    for dl in deltas:
        f, A, b = _margin_lp(rng, n, dl)
        info = {}
        ref = oracle.linear_program(f, A, b, info=info)          # the reference's rule: solvable iff residual < 1e-7
        ref_ok = ref["min"] is not None
        assert ref_ok == (info["rnorm"] < 1e-7)
        if abs(info["rnorm"] / 1e-7 - 1.0) < 1e-3:
            continue
        # the reference's formulation on the device decides as the reference does, everywhere
        via = linear_program_pnnls(f, A, b)
        assert (via["min"] is not None) == ref_ok, (dl, info["rnorm"])
        if ref_ok:
            assert abs(via["min"] - ref["min"]) < 1e-6 * max(1.0, abs(ref["min"]))
        # the active-set kernel: stricter inside the band, identical outside it
        sol = linear_program(f, A, b)
        got_ok = sol["min"] is not None
        if dl > 1e-9 * 1.0 and ref_ok:
            band += 1
            assert not got_ok, dl                                 # infeasible by more than the row tolerance: refused here
        else:
            agree += 1
            assert got_ok == ref_ok, (dl, info["rnorm"])
            if got_ok:
                assert abs(sol["min"] - ref["min"]) < 1e-6 * max(1.0, abs(ref["min"]))
    print("LP edge: %d margins decided identically, %d inside the tolerance band (reference: solvable, active-set kernel: "
          "infeasible)" % (agree, band))
    assert agree >= 20 and band >= 3


def test_lp_via_pnnls_on_the_reference_kats():
    """The reference's LP known-answer tests (test_solvers.py:12-256 subset: box, equality, infeasible, unbounded)
    through the reference's formulation on the device."""
    from pympc_amd.optimization.solvers.hip_lp import linear_program_pnnls
    A = np.array([[1., 0.], [0., 1.], [-1., 0.], [0., -1.]]); b = np.array([1., 1., 0., 0.])   # unit box
    sol = linear_program_pnnls(-np.ones(2), A, b)
    assert abs(sol["min"] + 2.) < 1e-7 and np.allclose(sol["argmin"], [1., 1.], atol=1e-7)
    assert sol["active_set"] == [0, 1] and np.allclose(sol["multiplier_inequality"], [1., 1., 0., 0.], atol=1e-7)
    assert sol["multiplier_equality"] is None
    sol = linear_program_pnnls(-np.ones(2), A, b, np.array([[1., -1.]]), np.array([.5]))       # with x_0 - x_1 = .5
    assert abs(sol["min"] + 1.5) < 1e-7 and np.allclose(sol["argmin"], [1., .5], atol=1e-7)
    sol = linear_program_pnnls(np.ones(2), np.array([[1., 0.], [-1., 0.]]), np.array([-1., -1.]))   # infeasible
    assert all(v is None for v in sol.values())
    sol = linear_program_pnnls(-np.ones(2), np.array([[1., 0.]]), np.array([1.]))                   # unbounded
    assert all(v is None for v in sol.values())
