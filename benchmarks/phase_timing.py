"""Debug: per-phase cycle counts of qp_solve_kernel (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPYMPC_PHASE_TIMING -o benchmarks/_dbg/libpympc_phase.so pympc_amd/csrc/pympc_hip.hip)."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pympc_amd.build as b
import pympc_amd._hip as h
h.LIB_PATH = os.path.join(ROOT, "benchmarks", "_dbg", "libpympc_phase.so")
h.load_library(h.LIB_PATH)
import torch
from pympc_amd.optimization.solvers.hip import PreparedMpqp
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
model = dict(np.load(os.path.join(ROOT, "tests", "golden", "c2_model.npz")))
prep = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], float(model["g"]),
                    model["Au"], model["Ax"], model["b"])
X = torch.as_tensor(np.random.RandomState(0).uniform(-scale * model["x_max"], scale * model["x_max"], (B, 4))).cuda()
prep.solve_batch(X, n_out=1, want_mult=True, want_iters=True)
res = prep.solve_batch(X, n_out=1, want_mult=True, want_iters=True)
print('cuts', prep.n_cuts())
torch.cuda.synchronize()
ph = res.numpy('multiplier_inequality')[:, :8]
it = res.numpy('iters')
st = res.numpy('status')
names = ["setup m(x)", "sweep", "pick", "trial", "add", "inner", "(loop exit)", "outputs"]
w = (st == 0) & (it[:, 0] >= 3)            # solved by the wave kernel (the lane path finishes up to two columns)
print("states solved by the wave kernel: %d of %d; mean outer %.2f inner %.2f" % (w.sum(), B, it[w, 0].mean(), it[w, 1].mean()))
tot = ph[w].sum()
nit = it[w, 0].sum()
for i, nme in enumerate(names):
    print("%-12s %8.0f cycles/QP  %5.1f %%   per outer iteration: %7.0f" % (nme, ph[w, i].mean(), 100 * ph[w, i].sum() / tot, ph[w, i].sum() / nit))
print("total per outer iteration: %.0f cycles" % (tot / nit))
g = w & (it[:, 0] <= 8)   # (debug builds with the Gram kernel: its phases are setup, pick, trial, border+solve, secondary, sweep, exit, outputs)
gn = ["setup m(x)", "pick", "trial", "border+solve", "secondary", "sweep", "(exit)", "outputs"]
if g.sum():
    print("Gram kernel states: %d, mean outer %.2f" % (g.sum(), it[g, 0].mean()))
    for i, nme in enumerate(gn):
        print("  %-12s %8.0f cycles/QP   per outer iteration: %7.0f" % (nme, ph[g, i].mean(), ph[g, i].sum() / it[g, 0].sum()))
for k in (3, 4, 6, 8, 12, 16):
    sel = w & (it[:, 0] == k)
    if sel.sum():
        print("k=%2d n=%6d cycles/QP %8.0f  per iteration %6.0f" % (k, sel.sum(), ph[sel].sum(1).mean(), ph[sel].sum(1).mean() / k))
