"""Learned infeasibility certificates (pympc_mpqp_solve_batch_cuts): same statuses and solutions as the plain solve.

The reference solves every state from scratch (pnnls.py:216-230); a certificate only replaces the iterations of a
state whose infeasibility it proves with a 1e-6 relative margin, so every output must be identical to the plain
kernel's, which the other GPU tests pin to the oracle and the golden vectors.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _prep(name):
    from pympc_amd.optimization.solvers.hip import PreparedMpqp
    m = dict(np.load(os.path.join(GOLD, name)))
    return m, PreparedMpqp(m["Huu"], m["Hux"], m["Hxx"], m["fu"], m["fx"], float(m["g"]), m["Au"], m["Ax"], m["b"])


@pytest.mark.parametrize("scale,B", [(0.3, 300000), (1.0, 200000)])
def test_cuts_same_results_c2(scale, B):
    model, prep = _prep("c2_model.npz")
    X = np.random.RandomState(3).uniform(-scale * model["x_max"], scale * model["x_max"], (B, 4))
    plain = prep.solve_batch(X, use_cuts=False, want_iters=True)
    cold = prep.solve_batch(X, use_cuts=True, want_iters=True)      # learns while it runs
    ncut = prep.n_cuts()
    warm = prep.solve_batch(X, use_cuts=True, want_iters=True)      # table from the first pass
    torch.cuda.synchronize()
    st = plain.numpy("status")
    assert 0 < ncut <= 256
    for res in (cold, warm):
        assert np.array_equal(res.numpy("status"), st)
        for key in ("min", "argmin", "multiplier_inequality", "active_bits"):
            assert np.array_equal(res.numpy(key), plain.numpy(key), equal_nan=True), key
    it = warm.numpy("iters")
    hit = (st == 1) & (it[:, 0] == 0)
    # a certificate never fires on a feasible state, and the warm table covers nearly every infeasible one
    assert not np.any(hit & (st == 0))
    assert hit.sum() >= 0.9 * (st == 1).sum()
    # feasible states report the plain kernel's counters; an infeasible state may stop early when a certificate stored
    # by another wave during its solve rejects it (never later than the plain solve)
    pit = plain.numpy("iters")
    assert np.array_equal(it[st == 0], pit[st == 0])
    assert np.all(it[st == 1, 0] <= pit[st == 1, 0])


def test_cuts_reset_and_golden_c1():
    """C1 golden vectors (reference outputs, KKT-certified) through the certificate path, twice, then after a reset."""
    from pympc_amd.optimization.solvers.hip import unpack_active
    model, prep = _prep("c1_model.npz")
    g = dict(np.load(os.path.join(GOLD, "c1_solves.npz")))
    X = g["X"]
    feas = g["feasible"].astype(bool)
    valid = g["oracle_valid"].astype(bool)
    for rnd in range(3):
        if rnd == 2:
            prep.reset_cuts()
            torch.cuda.synchronize()
            assert prep.n_cuts() == 0
        res = prep.solve_batch(X, use_cuts=True)
        torch.cuda.synchronize()
        ok = res.numpy("status") == 0
        assert np.array_equal(ok[valid], feas[valid])
        sel = valid & feas
        np.testing.assert_allclose(res.numpy("min")[sel], g["min"][sel], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(res.numpy("argmin")[sel], g["argmin"][sel], rtol=1e-6, atol=1e-8)
    assert prep.n_cuts() > 0


def test_certificates_stored_at_set_up():
    """seed_cuts: certificates learned from samples of the feasible set's bounding box (no user data) are part of the
    prepared mpQP: reset_cuts() returns to them, reset_cuts(hard=True) empties the table; they reject most infeasible
    states of both benchmark distributions before any iteration, and never change a result."""
    model, prep = _prep("c2_model.npz")
    box = prep.feasible_box()
    assert box is not None
    lo, hi = (v.cpu().numpy() for v in box)
    assert np.all(lo < 0) and np.all(hi > 0) and np.all(hi <= model["x_max"] + 1e-6) and np.all(lo >= -model["x_max"] - 1e-6)
    n0 = prep.seed_cuts()
    torch.cuda.synchronize()
    assert 16 <= n0 <= 256
    for scale in (0.3, 1.0):
        X = np.random.RandomState(5).uniform(-scale * model["x_max"], scale * model["x_max"], (100000, 4))
        plain = prep.solve_batch(X, use_cuts=False, want_iters=True)
        prep.reset_cuts()
        torch.cuda.synchronize()
        assert prep.n_cuts() == n0
        res = prep.solve_batch(X, use_cuts=True, want_iters=True)
        torch.cuda.synchronize()
        st = plain.numpy("status")
        assert np.array_equal(res.numpy("status"), st)
        for key in ("min", "argmin", "multiplier_inequality", "active_bits"):
            assert np.array_equal(res.numpy(key), plain.numpy(key), equal_nan=True), key
        hit = (st == 1) & (res.numpy("iters")[:, 0] == 0)
        print("s = %.1f: %d set-up certificates reject %.1f %% of the infeasible states" % (scale, n0, 100. * hit.sum() / (st == 1).sum()))
        assert hit.sum() >= 0.8 * (st == 1).sum()
        assert prep.n_cuts() >= n0
    prep.reset_cuts(hard=True)
    torch.cuda.synchronize()
    assert prep.n_cuts() == 0


def test_cuts_disabled_without_parameters():
    """p = 0 or a null table: plain solve (quadratic_program_batch has no parameter space to cut)."""
    from pympc_amd import _hip
    lib = _hip.load_library()
    assert lib.pympc_mpqp_cuts_doubles(0) == 0
    assert lib.pympc_mpqp_cuts_doubles(9) == 0
    assert lib.pympc_mpqp_cuts_doubles(4) == 48 + 5 * 256
