#!/usr/bin/env python
"""Where the time of the C2 batch goes, by class of state: python benchmarks/qp_classes.py [scale] [batch]

One warm pass over the batch with the iteration counters, then every class of state (by status and number of outer
NNLS iterations) is timed by itself: its states tiled to a fixed count and solved with the warm certificate table.
Prints the share of the batch, the time per state of the class (wall time of the launch / states) and the estimated
contribution to a 1M-state launch.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pympc_amd.optimization.solvers.hip import PreparedMpqp, _alloc_outputs  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
NT = 262144
model = dict(np.load(os.path.join(ROOT, "tests", "golden", "c2_model.npz")))
m = model["Au"].shape[0]
prep = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], float(model["g"]),
                    model["Au"], model["Ax"], model["b"])
Xh = np.random.RandomState(0).uniform(-scale * model["x_max"], scale * model["x_max"], (B, 4))
X = torch.as_tensor(Xh).cuda()


def timed(Xd, reps=5, cuts=True, cold=False):
    out = _alloc_outputs(Xd.shape[0], 1, m, Xd.device, want_mult=False, want_active=True, want_iters=False)
    prep.solve_batch(Xd, n_out=1, out=out, use_cuts=cuts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if cold:
            prep.reset_cuts()
        prep.solve_batch(Xd, n_out=1, out=out, use_cuts=cuts)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


prep.solve_batch(X, n_out=1, want_mult=False)          # learn
res = prep.solve_batch(X, n_out=1, want_mult=False, want_iters=True)
torch.cuda.synchronize()
st = res.numpy("status")
it = res.numpy("iters")
print("whole batch: warm %.3f ms, cold %.3f ms, cuts %d" % (timed(X), timed(X, cold=True), prep.n_cuts()))
prep.solve_batch(X, n_out=1, want_mult=False)
classes = [("feasible k=0", (st == 0) & (it[:, 0] == 0)), ("feasible k=1", (st == 0) & (it[:, 0] == 1))]
for k in (2, 3, 4):
    classes.append(("feasible k=%d" % k, (st == 0) & (it[:, 0] == k)))
classes += [("feasible k=5..8", (st == 0) & (it[:, 0] >= 5) & (it[:, 0] <= 8)),
            ("feasible k>=9", (st == 0) & (it[:, 0] >= 9)),
            ("infeasible by cut", (st == 1) & (it[:, 0] == 0)),
            ("infeasible solved k<12", (st == 1) & (it[:, 0] > 0) & (it[:, 0] < 12)),
            ("infeasible solved k>=12", (st == 1) & (it[:, 0] >= 12))]
tot = 0.0
for name, sel in classes:
    cnt = int(sel.sum())
    if cnt == 0:
        print("%-26s none" % name)
        continue
    idx = np.where(sel)[0]
    Xc = torch.as_tensor(Xh[np.resize(idx, NT)]).cuda()
    ms = timed(Xc)
    contrib = ms * cnt / NT
    tot += contrib
    print("%-26s share %7.4f  %8.3f ms per %d  -> %6.3f ms of the launch  (mean outer %.2f, inner %.2f)"
          % (name, cnt / B, ms, NT, contrib, it[sel, 0].mean(), it[sel, 1].mean()))
print("sum of the classes %.3f ms" % tot)
