"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the GPU box and the
tests only ever read the .npz files this script writes.  Nothing from the
reference is copied: the files hold inputs, the reference's outputs and a
certificate per record.

Protocol: the container's scipy (1.15.3) is newer than the
reference's pin (scipy<=1.11.4, /root/reference/setup.py:18) and its nnls is
wrong on about 0.1 % of inputs, so every scipy.optimize.nnls call the reference
makes is intercepted and certified with plain numpy:
  (i)   x >= 0
  (ii)  | ||Ax-b|| - rnorm | <= 1e-8 max(1, ||Ax-b||)
  (iii) w = A'(b-Ax):  w_j <= eps where x_j == 0, |w_j| <= eps where x_j > 0
Records whose nnls call fails the certificate are kept and flagged
``oracle_invalid``; records whose thresholded active set sits on the 1e-7 edge
are flagged ``degenerate``.

Usage:  python tests/golden/make_golden.py [--only c1,c2,c2big,ctrl,lp,nnls,nb,hybrid,bigm]
"""
import os
import sys
import types
import time
import argparse

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("gurobipy", types.ModuleType("gurobipy"))  # .travis.yml:11 trick

import numpy as np
import scipy.optimize

import pympc.optimization.solvers.pnnls as ref_pnnls
from pympc.geometry.polyhedron import Polyhedron
from pympc.dynamics.discrete_time_systems import LinearSystem, AffineSystem, PieceWiseAffineSystem
from pympc.control.controllers import ModelPredictiveController, HybridModelPredictiveController

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-7

# ---------------------------------------------------------------------------
# nnls interception + certificate
# ---------------------------------------------------------------------------
_calls = []


def nnls_certificate(A, b, x, rnorm):
    """numpy-only KKT certificate of an NNLS answer; returns (ok, detail)."""
    r = b - A.dot(x)
    rn = np.linalg.norm(r)
    w = A.T.dot(r)
    eps = 1e-8 * max(1.0, np.linalg.norm(A) * max(np.linalg.norm(b), 1.0))
    ok_sign = bool(np.all(x >= 0.0))
    ok_norm = abs(rn - rnorm) <= 1e-8 * max(1.0, rn)
    zero = x == 0.0
    ok_w = bool(np.all(w[zero] <= eps)) and bool(np.all(np.abs(w[~zero]) <= eps))
    return ok_sign and ok_norm and ok_w, dict(true_rnorm=rn, wmax_zero=(w[zero].max() if zero.any() else 0.0),
                                              wabs_pos=(np.abs(w[~zero]).max() if (~zero).any() else 0.0))


def _recording_nnls(A, b, *args, **kwargs):
    try:
        x, rnorm = scipy.optimize.nnls(A, b, *args, **kwargs)
    except RuntimeError:
        _calls.append(dict(ok=False, raised=True))
        raise
    ok, det = nnls_certificate(A, b, x, rnorm)
    _calls.append(dict(ok=ok, raised=False, **det))
    return x, rnorm


ref_pnnls.nnls = _recording_nnls


def certified_call(fun, *args):
    """Run a reference function; return (result, all nnls calls inside it certified)."""
    del _calls[:]
    try:
        out = fun(*args)
        raised = False
    except RuntimeError:
        out = None
        raised = True
    ok = (not raised) and all(c["ok"] for c in _calls)
    return out, ok


# ---------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------
def build_c1():
    """Double integrator, ZOH h=.1, N=10 (config C1)."""
    A = np.array([[0., 1.], [0., 0.]])
    B = np.array([[0.], [1.]])
    S = LinearSystem.from_continuous(A, B, .1, 'zero_order_hold')
    N = 10
    Q = np.eye(2)
    R = np.eye(1)
    P, K = S.solve_dare(Q, R)
    X = Polyhedron.from_bounds(np.array([-5., -1.]), np.array([5., 1.]))
    U = Polyhedron.from_bounds(np.array([-1.]), np.array([1.]))
    D = X.cartesian_product(U)
    X_N = S.mcais(K, D)
    ctrl = ModelPredictiveController(S, N, Q, R, P, D, X_N)
    return ctrl, dict(K=K)


def cart_pole_pieces():
    """Cart-pole with two elastic walls, examples/hybrid_mpc_2.ipynb cells 5-9 (parameters only)."""
    mc = 1.; mp = 1.; l = 1.; d = 1.; k = 100.; g = 10.; h = .05
    method = 'explicit_euler'
    A1 = np.array([[0., 0., 1., 0.], [0., 0., 0., 1.], [0., (g*mp)/mc, 0., 0.],
                   [k/(l*mp), (g*mp**2 + g*mc*mp - k*l*mc)/(l*mc*mp), 0., 0.]])
    B1 = np.array([[0.], [0.], [1./mc], [1./(l*mc)]])
    c1 = np.array([0., 0., 0., (d*k)/(l*mp)])
    S1 = AffineSystem.from_continuous(A1, B1, c1, h, method)
    A2 = np.array([[0., 0., 1., 0.], [0., 0., 0., 1.], [0., (g*mp)/mc, 0., 0.], [0., g*(mc+mp)/(l*mc), 0., 0.]])
    B2 = np.array([[0.], [0.], [1./mc], [1./(l*mc)]])
    S2 = LinearSystem.from_continuous(A2, B2, h, method)
    c3 = np.array([0., 0., 0., -(d*k)/(l*mp)])
    S3 = AffineSystem.from_continuous(A1.copy(), B1.copy(), c3, h, method)
    x_max = np.array([1.5*d, np.pi/6., 7., 5.])
    x_min = -x_max
    X1 = Polyhedron.from_bounds(x_min, x_max); X1.add_inequality(np.array([[1., -l, 0., 0.]]), np.array([-d]))
    X2 = Polyhedron.from_bounds(x_min, x_max); X2.add_inequality(np.array([[-1., l, 0., 0.], [1., -l, 0., 0.]]), np.array([d, d]))
    X3 = Polyhedron.from_bounds(x_min, x_max); X3.add_inequality(np.array([[-1., l, 0., 0.]]), np.array([-d]))
    U = Polyhedron.from_bounds(np.array([-100.]), np.array([100.]))
    D1 = X1.cartesian_product(U); D2 = X2.cartesian_product(U); D3 = X3.cartesian_product(U)
    return [S1, S2, S3], [D1, D2, D3], x_max


def build_c2():
    """Cart-pole mode 2, N=20 (config C2)."""
    S_list, D_list, x_max = cart_pole_pieces()
    S2, D2 = S_list[1], D_list[1]
    N = 20
    Q = np.eye(4); R = np.eye(1)
    P, K = S2.solve_dare(Q, R)
    t0 = time.time()
    X_N = S2.mcais(K, D2)
    print("  cart-pole mcais: %d facets in %.1f s" % (X_N.A.shape[0], time.time() - t0))
    ctrl = ModelPredictiveController(S2, N, Q, R, P, D2, X_N)
    return ctrl, dict(K=K, x_max=x_max)


def model_dict(ctrl, extra):
    q = ctrl.mpqp
    d = dict(Huu=q.H['uu'], Hux=q.H['ux'], Hxx=q.H['xx'], fu=q.f['u'], fx=q.f['x'], g=np.asarray(q.g),
             Au=q.A['u'], Ax=q.A['x'], b=q.b,
             S_A=ctrl.S.A, S_B=ctrl.S.B, N=np.asarray(ctrl.N), Q=ctrl.Q, R=ctrl.R, P=ctrl.P,
             D_A=ctrl.D.A, D_b=ctrl.D.b, XN_A=ctrl.X_N.A, XN_b=ctrl.X_N.b)
    d.update(extra)
    return d


def qp_kkt_ok(H, f, A, b, sol):
    """Direct KKT check of a reference QP answer (independent of nnls)."""
    x = sol['argmin']; lam = sol['multiplier_inequality']
    scale = max(1.0, np.abs(b).max(), np.abs(x).max())
    prim = (A.dot(x) - b).max() <= 1e-7 * scale
    dual = lam.min() >= -1e-9
    stat = np.abs(H.dot(x) + f + A.T.dot(lam)).max() <= 1e-6 * max(1.0, np.abs(f).max(), np.abs(lam).max())
    comp = np.abs(lam * (A.dot(x) - b)).max() <= 1e-6 * max(1.0, np.abs(lam).max()) * scale
    return bool(prim and dual and stat and comp)


def solve_batch(ctrl, X):
    """Reference mpqp.solve over a batch of states, with certificates."""
    q = ctrl.mpqp
    B = X.shape[0]
    n = q.H['uu'].shape[0]; m = q.A['u'].shape[0]
    feas = np.zeros(B, bool); vmin = np.full(B, np.nan); arg = np.full((B, n), np.nan)
    mult = np.zeros((B, m)); act = np.zeros((B, m), bool)
    valid = np.zeros(B, bool); degen = np.zeros(B, bool)
    t0 = time.time()
    for i in range(B):
        x = X[i]
        sol, ok = certified_call(q.solve, x)
        if sol is None:          # nnls raised (maxiter)
            continue
        valid[i] = ok
        if sol['min'] is not None:
            feas[i] = True
            vmin[i] = sol['min']; arg[i] = sol['argmin']; mult[i] = sol['multiplier_inequality']
            act[i, sol['active_set']] = True
            f = q.H['ux'].dot(x) + q.f['u']; bb = q.b - q.A['x'].dot(x)
            valid[i] = ok and qp_kkt_ok(q.H['uu'], f, q.A['u'], bb, sol)
            lam = sol['multiplier_inequality']
            nz = lam[lam != 0.0]   # exact zeros never entered / were dropped from the passive set: not on the edge
            degen[i] = bool(np.any(np.abs(nz - TOL) < 1e-9 * max(1.0, np.linalg.norm(lam))))
            a = sol['active_set']
            if len(a) > 0 and np.linalg.matrix_rank(q.A['u'][a]) < len(a):
                degen[i] = True
    dt = time.time() - t0
    print("  %d solves in %.2f s (%.0f/s, 1 core): feasible %d, oracle_invalid %d, degenerate %d"
          % (B, dt, B / dt, feas.sum(), (~valid).sum(), degen.sum()))
    return dict(X=X, feasible=feas, min=vmin, argmin=arg, multiplier_inequality=mult, active=np.packbits(act, axis=1),
                oracle_valid=valid, degenerate=degen, ref_seconds=np.asarray(dt))


# ---------------------------------------------------------------------------
def gen_c1():
    print("C1 double integrator N=10")
    ctrl, extra = build_c1()
    np.savez_compressed(os.path.join(HERE, "c1_model.npz"), **model_dict(ctrl, extra))
    X = np.random.RandomState(0).uniform([-5., -1.], [5., 1.], (1000, 2))
    rec = solve_batch(ctrl, X)
    # feedback() contract: first input or None
    np.savez_compressed(os.path.join(HERE, "c1_solves.npz"), **rec)


def gen_c2():
    print("C2 cart-pole N=20")
    ctrl, extra = build_c2()
    np.savez_compressed(os.path.join(HERE, "c2_model.npz"), **model_dict(ctrl, extra))
    x_max = extra['x_max']
    for tag, s, count in (("s03", 0.3, 400), ("s10", 1.0, 400)):
        X = np.random.RandomState(0).uniform(-s * x_max, s * x_max, (1000000, 4))[:count]
        rec = solve_batch(ctrl, X)
        np.savez_compressed(os.path.join(HERE, "c2_solves_%s.npz" % tag), **rec)


def gen_c2big():
    """10 000 reference-run C2 states per distribution (the first 10 000 of the benchmark's own stream).
    Kept compact: u0 .. u2, min, status, active-set bits, certificate flags (no multiplier rows)."""
    print("C2 cart-pole N=20, 10 000 states per distribution")
    ctrl, extra = build_c2()
    x_max = extra['x_max']
    for tag, s in (("s03", 0.3), ("s10", 1.0)):
        X = np.random.RandomState(0).uniform(-s * x_max, s * x_max, (1000000, 4))[:10000]
        rec = solve_batch(ctrl, X)
        lam = rec.pop('multiplier_inequality')
        rec['argmin'] = rec['argmin'][:, :3].copy()
        rec['multiplier_sum'] = lam.sum(axis=1)
        rec['multiplier_max'] = lam.max(axis=1)
        np.savez_compressed(os.path.join(HERE, "c2_big_%s.npz" % tag), **rec)


def gen_ctrl():
    """Randomised linear MPC problems in the manner of the reference's controller tests (random stable-ish systems with
    box constraints, LQR terminal cost, mcais terminal set): every quantity the reference computes is stored as data --
    systems, horizons, the reference's P, K and terminal set, the sampled states, and for each state the reference's
    feedforward (u, V), whether the state lies in the terminal set and the nnls certificate.  Explicit solutions: the
    reference's store_explicit_solution is attempted per system (it fails under the container's scipy for some systems,
    SURVEY 8c); where it works the explicit (u, V) are stored next to the implicit ones."""
    print("randomised MPC controllers (implicit + explicit)")
    rng = np.random.RandomState(1)
    out = {}
    count = 0
    for kind, nsys in (("imp", 30), ("exp", 6)):
        for i in range(nsys):
            if kind == "imp":
                n = rng.randint(2, 5); m = rng.randint(1, n); scale = 10.; N = rng.randint(5, 10)
            else:
                n = 2; m = 1; scale = 1.; N = 10
            while True:
                A = rng.rand(n, n) / scale; B = rng.rand(n, m) / scale
                S = LinearSystem(A, B)
                if S.controllable:
                    break
            x_min = -rng.rand(n); x_max = rng.rand(n); u_min = -rng.rand(m); u_max = rng.rand(m)
            D = Polyhedron.from_bounds(x_min, x_max).cartesian_product(Polyhedron.from_bounds(u_min, u_max))
            Q = np.eye(n); R = np.eye(m)
            P, K = S.solve_dare(Q, R)
            X_N = S.mcais(K, D)
            ctrl = ModelPredictiveController(S, N, Q, R, P, D, X_N)
            xs = rng.rand(10, n) * (x_max - x_min) + x_min
            U = np.full((10, N * m), np.nan); V = np.full(10, np.nan); ok = np.zeros(10, bool); inside = np.zeros(10, bool)
            Ue = np.full((10, N * m), np.nan); Ve = np.full(10, np.nan); have_e = np.zeros(10, bool)
            explicit_ok = False
            if kind == "exp":
                try:
                    ctrl.store_explicit_solution()
                    explicit_ok = True
                except Exception as e:   # scipy-1.15.3 nnls failure inside explicit_solve
                    print("    system %d: store_explicit_solution failed (%s)" % (i, type(e).__name__))
            for j, x in enumerate(xs):
                (u, v), cert = certified_call(ctrl.feedforward, x)
                ok[j] = cert
                inside[j] = X_N.contains(x)
                if v is not None:
                    U[j] = np.concatenate(u); V[j] = v
                if explicit_ok:
                    try:
                        ue, ve = ctrl.feedforward_explicit(x)
                        have_e[j] = True
                        if ve is not None:
                            Ue[j] = np.concatenate(ue); Ve[j] = ve
                    except Exception:
                        pass
            pre = "%s%d_" % (kind, i)
            out.update({pre + "A": A, pre + "B": B, pre + "x_min": x_min, pre + "x_max": x_max, pre + "u_min": u_min,
                        pre + "u_max": u_max, pre + "N": np.asarray(N), pre + "P": P, pre + "K": K,
                        pre + "XN_A": X_N.A, pre + "XN_b": X_N.b, pre + "X": xs, pre + "U": U, pre + "V": V,
                        pre + "oracle_valid": ok, pre + "inside": inside})
            if kind == "exp":
                out.update({pre + "Ue": Ue, pre + "Ve": Ve, pre + "have_explicit": have_e})
            count += 1
        out["n_" + kind] = np.asarray(nsys)
    print("  %d systems" % count)
    np.savez_compressed(os.path.join(HERE, "controllers.npz"), **out)


def gen_nb():
    """examples/linear_mpc.ipynb: inverted pendulum N=6; anchors V*(x0) of cell 25."""
    print("linear_mpc.ipynb inverted pendulum N=6")
    A = np.array([[0., 1.], [10., 0.]]); B = np.array([[0.], [1.]])
    S = LinearSystem.from_continuous(A, B, .1, 'zero_order_hold')
    N = 6; Q = np.eye(2); R = np.eye(1)
    P, K = S.solve_dare(Q, R)
    U = Polyhedron.from_bounds(np.array([-6.]), np.array([6.]))
    X = Polyhedron.from_bounds(np.array([-.5, -.5]), np.array([.5, .5]))
    D = X.cartesian_product(U)
    X_N = S.mcais(K, D)
    ctrl = ModelPredictiveController(S, N, Q, R, P, D, X_N)
    np.savez_compressed(os.path.join(HERE, "nb_model.npz"), **model_dict(ctrl, dict(K=K)))
    g = np.linspace(-.5, .5, 21)
    Xs = np.array([[a, b] for a in g for b in g] + [[.39, .39]])
    rec = solve_batch(ctrl, Xs)
    # notebook cell 25 printed outputs (data the reference holds)
    rec['nb_x0'] = np.array([.39, .39])
    rec['nb_V_implicit'] = np.asarray(100.82362500750787)
    rec['nb_V_explicit'] = np.asarray(100.82362500750855)
    rec['nb_u_implicit'] = np.array([-6.000000000000085, -6.000000000000028, -5.999999999999972,
                                     -5.999999999999957, -5.051626711312593, -3.6811765674366725])
    np.savez_compressed(os.path.join(HERE, "nb_solves.npz"), **rec)


def gen_lp():
    """Random 6-D polytopes (SURVEY 8d C3 recipe, small m) + the LPs inside minimal_facets / chebyshev."""
    print("LP / polyhedron fixtures")
    import pympc.geometry.polyhedron as ref_poly
    out = {}
    lp_records = []
    real_lp = ref_poly.linear_program

    def recording_lp(f, A, b, C=None, d=None, **kw):
        sol = real_lp(f, A, b, C, d, **kw)
        lp_records.append((f, A, b, C, d, sol, all(c["ok"] for c in _calls[-1:])))
        return sol

    ref_poly.linear_program = recording_lp
    rng = np.random.RandomState(0)
    for m in (30, 60, 100):
        A = rng.randn(m, 6); A /= np.linalg.norm(A, axis=1)[:, None]
        b = 1. + rng.rand(m)
        P = Polyhedron(A.copy(), b.copy())
        del lp_records[:]
        mf, ok1 = certified_call(P.minimal_facets)
        ok1 = all(r[6] for r in lp_records)
        # per-facet value  -min - f_i  of polyhedron.py:447 (order dependent)
        vals = np.array([(-r[5]['min'] - b[i]) if r[5]['min'] is not None else np.nan for i, r in enumerate(lp_records)])
        del lp_records[:]
        (rad, cen), ok2 = certified_call(P._chebyshev)
        cheb = lp_records[-1]
        out["m%d_A" % m] = A; out["m%d_b" % m] = b
        out["m%d_minimal_facets" % m] = np.array(mf)
        out["m%d_facet_values" % m] = vals
        out["m%d_radius" % m] = np.asarray(rad); out["m%d_center" % m] = cen
        out["m%d_cheb_mult" % m] = cheb[5]['multiplier_inequality']
        out["m%d_valid" % m] = np.asarray(ok1 and ok2)
        print("  m=%d: %d minimal facets, radius %.6f, certified %s" % (m, len(mf), rad, ok1 and ok2))
    ref_poly.linear_program = real_lp

    # a batch of small generic LPs straight through pnnls.linear_program (bounded, infeasible, unbounded mixes)
    fs, As, bs, mins, args, mults, solv, valid = [], [], [], [], [], [], [], []
    rng = np.random.RandomState(1)
    for t in range(60):
        n = 3; m = 8
        A = rng.randn(m, n)
        if t % 3 == 0:      # bounded polytope around a random point
            A /= np.linalg.norm(A, axis=1)[:, None]
            A = np.vstack((A, -A[:2]))[:m]
            b = 1. + rng.rand(m)
            A = np.vstack((np.eye(n), -np.eye(n), A[:m - 2 * n]));
        else:
            b = rng.randn(m) + (1.0 if t % 3 == 1 else 0.0)
        f = rng.randn(n)
        sol, ok = certified_call(ref_pnnls.linear_program, f, A, b)
        fs.append(f); As.append(A); bs.append(b)
        s = sol['min'] is not None
        solv.append(s); valid.append(ok)
        mins.append(sol['min'] if s else np.nan)
        args.append(sol['argmin'] if s else np.full(n, np.nan))
        mults.append(sol['multiplier_inequality'] if s else np.full(m, np.nan))
    out.update(lp_f=np.array(fs), lp_A=np.array(As), lp_b=np.array(bs), lp_min=np.array(mins), lp_argmin=np.array(args),
               lp_mult=np.array(mults), lp_solvable=np.array(solv), lp_valid=np.array(valid))
    print("  generic LPs: %d solvable of %d, certified %d" % (sum(solv), len(solv), sum(valid)))
    np.savez_compressed(os.path.join(HERE, "lp_polytopes.npz"), **out)


def gen_nnls():
    """Raw NNLS instances: scipy answer + certificate (kernel-level fixtures)."""
    print("raw NNLS fixtures")
    rng = np.random.RandomState(2)
    out = {}
    k = 0
    for (r, c) in ((5, 3), (3, 5), (11, 78), (21, 60), (8, 8), (20, 10)):
        for rep in range(8):
            A = rng.randn(r, c)
            b = rng.randn(r)
            x, rn = scipy.optimize.nnls(A, b)
            ok, _ = nnls_certificate(A, b, x, rn)
            out["A%d" % k] = A; out["b%d" % k] = b; out["x%d" % k] = x; out["r%d" % k] = np.asarray(rn)
            out["ok%d" % k] = np.asarray(ok)
            k += 1
    out["count"] = np.asarray(k)
    print("  %d instances, certified %d" % (k, sum(bool(out["ok%d" % i]) for i in range(k))))
    np.savez_compressed(os.path.join(HERE, "nnls_cases.npz"), **out)


def gen_scipynnls():
    """The NNLS regression problems of the container's scipy (scipy/optimize/tests/test_nnls.py, SURVEY 8c: gh-20168,
    gh-20302, gh-21021 ...): the test methods are RUN with a recording nnls, so the file holds the matrices they build,
    scipy's answers and the KKT certificate of each answer.  (Third-party fixtures, not reference code.)"""
    print("scipy's own NNLS regression problems")
    import scipy.optimize.tests.test_nnls as T
    rec = []
    real = T.nnls

    def recorder(A, b, *args, **kw):
        A = np.asarray(A, dtype=float); b = np.asarray(b, dtype=float)
        x, rn = real(A, b, *args, **kw)
        ok, _ = nnls_certificate(A, b, x, rn)
        rec.append((A.copy(), b.copy(), x, rn, ok, bool(args or kw)))
        return x, rn

    T.nnls = recorder
    t = T.TestNNLS()
    names = []
    for name in sorted(dir(t)):
        if not name.startswith("test_") or name == "test_maxiter":
            continue
        t.setup_method()
        n0 = len(rec)
        try:
            getattr(t, name)()
        except AssertionError:
            pass
        names += [name] * (len(rec) - n0)
    T.nnls = real
    out = dict(count=np.asarray(len(rec)), names=np.array(names))
    for k, (A, b, x, rn, ok, custom) in enumerate(rec):
        out["A%d" % k] = A; out["b%d" % k] = b; out["x%d" % k] = x; out["r%d" % k] = np.asarray(rn)
        out["ok%d" % k] = np.asarray(ok)
        print("  %-32s %3d x %3d certified %s" % (names[k], A.shape[0], A.shape[1], ok))
    np.savez_compressed(os.path.join(HERE, "scipy_nnls_tests.npz"), **out)


def gen_proj():
    """Orthogonal projections (polyhedron.py:826-854, 941-1132): the reference's convex-hull method on random polytopes
    in the manner of test_polyhedron.py:845-899 (m = 10 facets, n = 3..5 -> n_proj = 2..n-1) and on larger ones.  Stored:
    A, b, the residual dimensions, and the reference's projection as minimal half-spaces + vertices (qhull order is
    not part of the contract: the test compares sets)."""
    print("orthogonal projections (convex-hull method)")
    from pympc.geometry.polyhedron import convex_hull_method
    rng = np.random.RandomState(3)
    out = {}
    k = 0
    for m, dims in ((10, (3, 4, 5)), (16, (4,))):
        for n in dims:
            for n_proj in range(2, n):
                for rep in range(3):
                    A = rng.randn(m, n); b = rng.rand(m)
                    P = Polyhedron(A.copy(), b.copy())
                    print("    m=%d n=%d n_proj=%d rep=%d" % (m, n, n_proj, rep), flush=True)
                    try:
                        (verts, ok1) = certified_call(lambda: P.vertices)
                    except Exception as e:
                        print("      vertices failed: %s" % type(e).__name__, flush=True)
                        continue
                    if verts is None:
                        continue
                    import signal

                    def _alarm(*_):
                        raise RuntimeError("timeout")
                    signal.signal(signal.SIGALRM, _alarm)
                    signal.alarm(20)          # (the reference's loop does not end on some instances under this scipy)
                    try:
                        (res, ok2) = certified_call(convex_hull_method, A.copy(), b.copy(), list(range(n_proj)))
                    except Exception as e:
                        print("      convex_hull_method failed: %s" % type(e).__name__, flush=True)
                        res = None
                    finally:
                        signal.alarm(0)
                    if res is None:
                        print("    m=%d n=%d n_proj=%d rep=%d: reference did not finish" % (m, n, n_proj, rep))
                        continue
                    E, f, V = res
                    Q = Polyhedron(E, f)
                    (_, ok3) = certified_call(Q.remove_redundant_inequalities)
                    out["A%d" % k] = A; out["b%d" % k] = b; out["n_proj%d" % k] = np.asarray(n_proj)
                    out["E%d" % k] = Q.A; out["f%d" % k] = Q.b; out["V%d" % k] = np.asarray(V)
                    out["ok%d" % k] = np.asarray(ok1 and ok2 and ok3)
                    k += 1
    out["count"] = np.asarray(k)
    print("  %d projections, certified %d" % (k, sum(bool(out["ok%d" % i]) for i in range(k))))
    np.savez_compressed(os.path.join(HERE, "projections.npz"), **out)


def gen_hybrid():
    """Config C5 anchors: get_mpqp(mode_sequence).solve(x0) for a handful of mode sequences (N=10)."""
    print("hybrid cart-pole fixed-mode-sequence QPs (N=10)")
    S_list, D_list, x_max = cart_pole_pieces()
    S2 = S_list[1]                       # PieceWiseAffineSystem replaces list entries in place
    S = PieceWiseAffineSystem(S_list, D_list)
    N = 10; Q = np.eye(4); R = np.eye(1)
    P, K = S2.solve_dare(Q, R)
    X_N = S2.mcais(K, D_list[1])
    # do not call HybridModelPredictiveController.__init__ (needs gurobi-free big-M LPs but builds MIQP); use the module function
    from pympc.control.controllers import condense_optimal_control_problem
    x0 = np.array([0., 0., 2.5, 0.])
    rng = np.random.RandomState(3)
    seqs = [[1] * N, [1] * 3 + [2] * 4 + [1] * 3, [1] * 2 + [2] * 3 + [1] * 5, [1, 1, 1, 2, 2, 2, 2, 1, 1, 1]]
    for _ in range(28):
        seqs.append(list(rng.randint(0, 3, N)))
    out = dict(S_A=np.array([s.A for s in S_list]), S_B=np.array([s.B for s in S_list]), S_c=np.array([s.c if hasattr(s, 'c') else np.zeros(4) for s in S_list]),
               Q=Q, R=R, P=P, XN_A=X_N.A, XN_b=X_N.b, x0=x0, seqs=np.array(seqs))
    for i, D in enumerate(D_list):
        out["D%d_A" % i] = D.A; out["D%d_b" % i] = D.b
    mins = []; args = []; valid = []
    for k, ms in enumerate(seqs):
        q = condense_optimal_control_problem(S, Q, R, P, X_N, ms)
        if k < 4:
            out["seq%d_Huu" % k] = q.H['uu']; out["seq%d_Hux" % k] = q.H['ux']; out["seq%d_Hxx" % k] = q.H['xx']
            out["seq%d_fu" % k] = q.f['u']; out["seq%d_fx" % k] = q.f['x']; out["seq%d_g" % k] = np.asarray(q.g)
            out["seq%d_Au" % k] = q.A['u']; out["seq%d_Ax" % k] = q.A['x']; out["seq%d_b" % k] = q.b
        sol, ok = certified_call(q.solve, x0)
        s = sol is not None and sol['min'] is not None
        mins.append(sol['min'] if s else np.nan)
        args.append(sol['argmin'] if s else np.full(N, np.nan))
        valid.append(ok)
    out.update(min=np.array(mins), argmin=np.array(args), oracle_valid=np.array(valid))
    print("  %d sequences, feasible %d, certified %d" % (len(seqs), np.isfinite(mins).sum(), sum(valid)))
    np.savez_compressed(os.path.join(HERE, "hybrid_seqs.npz"), **out)


def gen_bigm():
    """Big-Ms of the hybrid controller's MIQP set-up (controllers.py:289-405) for the cart-pole PWA system: the
    reference's own methods run on a bare object that only carries S (its __init__ goes on to build a Gurobi model).
    Every value is certified against scipy.optimize.linprog (HiGHS), an independent LP solver."""
    print("hybrid cart-pole big-Ms")
    S_list, D_list, x_max = cart_pole_pieces()
    S = PieceWiseAffineSystem(S_list, D_list)
    holder = types.SimpleNamespace(S=S)
    alpha, beta = HybridModelPredictiveController._get_bigM_dynamics(holder)
    gamma = HybridModelPredictiveController._get_bigM_domains(holder)
    s = len(S_list)

    def lp_value(f, D):   # min f'z over D
        r = scipy.optimize.linprog(f, A_ub=D.A, b_ub=D.b, bounds=(None, None), method="highs")
        assert r.status == 0
        return r.fun

    worst = 0.0
    for i in range(s):
        Ai = np.hstack((S.affine_systems[i].A, S.affine_systems[i].B))
        for j in range(s):
            for k in range(Ai.shape[0]):
                lo = lp_value(Ai[k], D_list[j]) + S.affine_systems[i].c[k]
                hi = -lp_value(-Ai[k], D_list[j]) + S.affine_systems[i].c[k]
                worst = max(worst, abs(lo - alpha[i][j][k, 0]), abs(hi - beta[i][j][k, 0]))
            for k in range(D_list[i].A.shape[0]):
                g = -lp_value(-D_list[i].A[k], D_list[j]) - D_list[i].b[k]
                worst = max(worst, abs(g - gamma[i][j][k, 0]))
    print("  max |reference - HiGHS| = %.2e" % worst)
    assert worst < 1e-7
    out = dict(S_A=np.array([t.A for t in S_list]), S_B=np.array([t.B for t in S_list]), S_c=np.array([t.c for t in S_list]),
               certified_against_highs=np.asarray(worst))
    for i, D in enumerate(D_list):
        out["D%d_A" % i] = D.A; out["D%d_b" % i] = D.b
    for i in range(s):
        for j in range(s):
            out["alpha_%d_%d" % (i, j)] = alpha[i][j]; out["beta_%d_%d" % (i, j)] = beta[i][j]
            out["gamma_%d_%d" % (i, j)] = gamma[i][j]
    # big-Ms over the GRAPHS of the dynamics (examples/pwa_from_urdf/utils.py:349-389: graph_representation + big_m; that
    # module imports pydrake and cannot be loaded, its big_m is a loop of the reference's linear_program over the rows
    # below -- the same calls are issued here through the reference's LP path and certified against HiGHS)
    graphs = []
    for Si, Di in zip(S_list, D_list):
        nx = Si.A.shape[0]
        Ag = np.vstack((np.hstack((Di.A, np.zeros((Di.A.shape[0], nx)))), np.hstack((Si.A, Si.B, -np.eye(nx))),
                        np.hstack((-Si.A, -Si.B, np.eye(nx)))))
        graphs.append((Ag, np.concatenate((Di.b, -Si.c, Si.c))))
    import pympc.optimization.programs as ref_programs
    worst = 0.0
    for i, (Ai, bi) in enumerate(graphs):
        for j, (Aj, bj) in enumerate(graphs):
            mij = []
            for k in range(Ai.shape[0]):
                sol, ok = certified_call(ref_programs.linear_program, -Ai[k], Aj, bj)
                v = -sol['min'] - bi[k]
                r = scipy.optimize.linprog(-Ai[k], A_ub=Aj, b_ub=bj, bounds=(None, None), method="highs")
                assert r.status == 0
                worst = max(worst, abs(v - (-r.fun - bi[k])))
                mij.append(0. if abs(v) < 1e-6 else v)
            out["graph_m_%d_%d" % (i, j)] = np.array(mij)
    print("  graph big-Ms: max |reference - HiGHS| = %.2e" % worst)
    assert worst < 1e-6
    np.savez_compressed(os.path.join(HERE, "hybrid_bigm.npz"), **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c1,c2,c2big,ctrl,nb,lp,nnls,scipynnls,proj,hybrid,bigm")
    only = ap.parse_args().only.split(",")
    for name, fn in (("c1", gen_c1), ("c2", gen_c2), ("c2big", gen_c2big), ("ctrl", gen_ctrl), ("nb", gen_nb), ("lp", gen_lp), ("nnls", gen_nnls), ("scipynnls", gen_scipynnls), ("proj", gen_proj), ("hybrid", gen_hybrid), ("bigm", gen_bigm)):
        if name in only:
            fn()
