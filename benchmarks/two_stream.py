#!/usr/bin/env python
"""Steps of the resident benchmark issued alternately on S streams (each with its own outputs and its own copy of the
certificate table), so that the tail of one step's kernels overlaps the head of the next step's.
    python benchmarks/two_stream.py [scale] [streams]"""
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
prep.seed_cuts()
X = torch.as_tensor(np.random.RandomState(0).uniform(-scale * model["x_max"], scale * model["x_max"], (B, 4))).cuda()
m = model["Au"].shape[0]
lib = _hip.load_library()


def run(S, steps=40):
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(S - 1)]
    outs = [_alloc_outputs(B, 1, m, X.device, want_mult=False, want_active=True, want_iters=False) for _ in range(S)]
    tables = [prep.cuts.clone() for _ in range(S)]
    torch.cuda.synchronize()

    def step(t):
        k = t % S
        with torch.cuda.stream(streams[k]):
            sp = ctypes_stream(streams[k])
            _hip.check(lib.pympc_mpqp_cuts_restore(prep.p, _hip.ptr(tables[k]), sp))
            o = outs[k]
            _hip.check(lib.pympc_mpqp_solve_batch_cuts(prep.n, prep.m, prep.p, _hip.ptr(prep.ws), _hip.ptr(tables[k]), _hip.ptr(X), B,
                                                       1e-7, 1, _hip.ptr(o['argmin']), _hip.ptr(o['min']), None,
                                                       _hip.ptr(o['active_bits']), _hip.ptr(o['status']), None, sp))
    for t in range(6):
        step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        step(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = all(torch.equal(outs[0]['status'], o['status']) and torch.equal(outs[0]['active_bits'], o['active_bits']) for o in outs[1:])
    print("%d stream(s): %.3f ms/step, %.1f M QP/s, outputs identical across streams: %s" % (S, dt * 1e3, B / dt / 1e6, ok), flush=True)


def ctypes_stream(s):
    import ctypes
    return ctypes.c_void_p(s.cuda_stream)


for rep in range(2):
    for S in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 2, 3]):
        run(S)
