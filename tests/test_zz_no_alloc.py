"""pympc_mpqp_solve_batch_static (the entry point that allocates nothing: certificate table in the library's static
device scratch) against pympc_mpqp_solve_batch_cuts with a caller-owned, freshly emptied table.

Two calls in a row through the C ABI, the second on a smaller batch: the static table keeps the first call's
certificates, so the second call only matches if the entry point empties it.  Outputs must be bit-identical
(status, argmin, min, active bits; north_star: active sets bit-exact).  No reference counterpart.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _outputs(B, nu, m, dev):
    nw = (m + 31) // 32
    return dict(argmin=torch.full((B, nu), 7.0, dtype=torch.float64, device=dev),
                vmin=torch.full((B,), 7.0, dtype=torch.float64, device=dev),
                active=torch.full((B, nw), 0x55, dtype=torch.int32, device=dev),
                status=torch.full((B,), -7, dtype=torch.int32, device=dev))


def test_static_scratch_matches_caller_table_twice():
    from pympc_amd import _hip
    from pympc_amd.optimization.solvers.hip import PreparedMpqp
    lib = _hip.load_library()
    model = dict(np.load(os.path.join(GOLD, "c2_model.npz")))
    prep = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], float(model["g"]),
                        model["Au"], model["Ax"], model["b"])
    n, m, p, dev = prep.n, prep.m, prep.p, prep.dev
    nu = 1
    rng = np.random.RandomState(11)
    # first call: the stress distribution (most states infeasible: the table fills up); second call: fewer states
    # of the headline distribution
    for B, scale in ((60000, 1.0), (17001, 0.3)):
        X = _hip.to_device(rng.uniform(-scale * model["x_max"], scale * model["x_max"], (B, p)))
        a, b = _outputs(B, nu, m, dev), _outputs(B, nu, m, dev)
        _hip.check(lib.pympc_mpqp_solve_batch_static(n, m, p, _hip.ptr(prep.ws), _hip.ptr(X), B, 1e-7, nu,
                                                     _hip.ptr(a["argmin"]), _hip.ptr(a["vmin"]), None, _hip.ptr(a["active"]),
                                                     _hip.ptr(a["status"]), None, _hip.stream_ptr()))
        table = torch.zeros(lib.pympc_mpqp_cuts_doubles(p), dtype=torch.float64, device=dev)
        _hip.check(lib.pympc_mpqp_solve_batch_cuts(n, m, p, _hip.ptr(prep.ws), _hip.ptr(table), _hip.ptr(X), B, 1e-7, nu,
                                                   _hip.ptr(b["argmin"]), _hip.ptr(b["vmin"]), None, _hip.ptr(b["active"]),
                                                   _hip.ptr(b["status"]), None, _hip.stream_ptr()))
        torch.cuda.synchronize()
        st = b["status"].cpu().numpy()
        assert set(np.unique(st)) <= {0, 1}
        assert (st == 0).any() and (st == 1).any()
        assert np.array_equal(a["status"].cpu().numpy(), st)
        for key in ("argmin", "vmin", "active"):
            assert np.array_equal(a[key].cpu().numpy(), b[key].cpu().numpy(), equal_nan=(key != "active")), (key, B)
        assert int(table[:1].view(torch.int32)[0].item()) > 0      # the caller-owned table did learn certificates


def test_static_scratch_rejects_sizes_beyond_its_table():
    from pympc_amd import _hip
    lib = _hip.load_library()
    st = torch.zeros(4, dtype=torch.int32, device=_hip.device())
    for p in (0, 9):
        rc = lib.pympc_mpqp_solve_batch_static(4, 8, p, _hip.ptr(st), _hip.ptr(st), 4, 1e-7, 0, None, None, None, None,
                                               _hip.ptr(st), None, _hip.stream_ptr())
        assert rc == -2
        assert b"static certificate table" in lib.pympc_hip_last_error()
