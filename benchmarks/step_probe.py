#!/usr/bin/env python
"""One C2 step taken apart: certificates stored at set-up and what they cover, step time with the Gram-form kernel on and
off (pympc_hip_tune), and who finishes what (histograms of outer iterations / active-set sizes).
    python benchmarks/step_probe.py [scale]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pympc_amd import _hip  # noqa: E402
from pympc_amd.optimization.solvers.hip import PreparedMpqp, _alloc_outputs  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
B = 1000000
model = dict(np.load(os.path.join(ROOT, "tests", "golden", "c2_model.npz")))
prep = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], float(model["g"]),
                    model["Au"], model["Ax"], model["b"])
if os.environ.get("PYMPC_SEED_KEEP"):
    PreparedMpqp.SEED_KEEP = int(os.environ["PYMPC_SEED_KEEP"])
torch.cuda.synchronize()
t0 = time.perf_counter()
box = prep.feasible_box()
torch.cuda.synchronize()
print("feasible_box: %.1f ms" % ((time.perf_counter() - t0) * 1e3), [v.cpu().numpy().round(3).tolist() for v in box])
t0 = time.perf_counter()
PreparedMpqp.SEED_PROFILE = True
n0 = prep.seed_cuts()
print({k: round(v, 2) for k, v in prep.seed_timing.items()})
torch.cuda.synchronize()
print("seed_cuts: %d certificates in %.1f ms; they reject %.4f of the infeasible seed states" % (n0, (time.perf_counter() - t0) * 1e3, getattr(prep, "seed_coverage", float("nan"))))
X = torch.as_tensor(np.random.RandomState(0).uniform(-scale * model["x_max"], scale * model["x_max"], (B, 4))).cuda()
out = _alloc_outputs(B, 1, model["Au"].shape[0], X.device, want_mult=False, want_active=True, want_iters=True)
BIG = 1 << 40


def timed(label, restore=True, **knobs):
    with _hip.tuned(**knobs):
        for _ in range(2):
            if restore:
                prep.reset_cuts()
            prep.solve_batch(X, n_out=1, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            if restore:
                prep.reset_cuts()
            prep.solve_batch(X, n_out=1, out=out)
        e1.record()
        torch.cuda.synchronize()
    print("%-44s %.3f ms/step" % (label, e0.elapsed_time(e1) / 10), flush=True)


timed("screen + gram + solve, set-up table")
timed("screen + solve (no gram), set-up table", gram_min_batch=BIG)
timed("screen + gram + solve, table kept", restore=False)
prep.reset_cuts(); prep.solve_batch(X, n_out=1, out=out)
torch.cuda.synchronize()
st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
feas = st == 0
print("feasible %.4f; infeasible states that iterated: %d of %d; certificates after the step: %d"
      % (feas.mean(), ((st == 1) & (it[:, 0] > 0)).sum(), (st == 1).sum(), prep.n_cuts()))
print("outer iterations (feasible):", np.bincount(np.minimum(it[feas, 0], 12)).tolist())
