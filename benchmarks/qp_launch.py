#!/usr/bin/env python
"""Bare launches of the headline kernel for profilers: python benchmarks/qp_launch.py [scale] [launches] [batch] [cuts]
(C2 model, states ~ U(+-scale x_max), outputs u0, V, status, active bits -- the same call bench.py times; cuts =
cold (default, certificate table emptied before every launch like bench.py), warm or off)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pympc_amd.optimization.solvers.hip import PreparedMpqp, _alloc_outputs  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
model = dict(np.load(os.path.join(ROOT, "tests", "golden", "c2_model.npz")))
prep = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], float(model["g"]),
                    model["Au"], model["Ax"], model["b"])
X = torch.as_tensor(np.random.RandomState(0).uniform(-scale * model["x_max"], scale * model["x_max"], (B, 4))).cuda()
out = _alloc_outputs(B, 1, model["Au"].shape[0], X.device, want_mult=False, want_active=os.environ.get("PYMPC_NO_ACTIVE") is None,
                     want_iters=False)
cuts = sys.argv[4] if len(sys.argv) > 4 else "cold"
for _ in range(launches):
    if cuts == "cold":
        prep.reset_cuts()
    prep.solve_batch(X, n_out=1, out=out, use_cuts=cuts != "off")
torch.cuda.synchronize()
print("done", launches, "launches of", B, "QPs, feasible", float((out["status"] == 0).double().mean()))
