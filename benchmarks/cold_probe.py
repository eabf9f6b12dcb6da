#!/usr/bin/env python
"""Cold-table launch of the C2 batch with iteration counters: how the infeasible states were decided."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pympc_amd.optimization.solvers.hip import PreparedMpqp
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
model = dict(np.load(os.path.join(ROOT, "tests", "golden", "c2_model.npz")))
prep = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], float(model["g"]),
                    model["Au"], model["Ax"], model["b"])
X = torch.as_tensor(np.random.RandomState(0).uniform(-scale * model["x_max"], scale * model["x_max"], (1000000, 4))).cuda()
for rep in range(3):
    prep.reset_cuts()
    res = prep.solve_batch(X, n_out=1, want_mult=False, want_iters=True)
    torch.cuda.synchronize()
    st, it = res.numpy("status"), res.numpy("iters")
    inf = st == 1
    print("cuts %d | infeasible %d: by cut at start %d, stopped early (outer<12) %d, full %d | outer sum over infeasible %d, over feasible pending (k>=3) %d"
          % (prep.n_cuts(), inf.sum(), (inf & (it[:, 0] == 0)).sum(), (inf & (it[:, 0] > 0) & (it[:, 0] < 12)).sum(),
             (inf & (it[:, 0] >= 12)).sum(), it[inf, 0].sum(), it[(st == 0) & (it[:, 0] >= 3), 0].sum()))
