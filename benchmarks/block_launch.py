#!/usr/bin/env python
"""The workgroup-per-problem kernels (csrc/nnls_block.hpp, qp_block.hpp) alone, for rocprofv3 and for the host clock:

    python3 benchmarks/block_launch.py [nnls|qp] [reps]

nnls: random dense problems of 128 x 300, 256 x 500 and 512 x 700 (one, and a batch); qp: strictly convex QPs of 100,
255 and 500 variables (one, and a batch).  Every line also times the oracle (one scalar core) on the first problem of the batch.
Prints one JSON line per shape."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "nnls"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from oracle import oracle
    rng = np.random.RandomState(0)
    if what == "nnls":
        from pympc_amd.optimization.solvers.hip_lp import nnls_batch
        for rows, cols, B in ((128, 300, 1), (256, 500, 1), (512, 700, 1), (128, 300, 512), (256, 500, 512), (512, 700, 256)):
            A = torch.as_tensor(rng.randn(B, rows, cols), device='cuda'); b = torch.as_tensor(rng.randn(B, rows), device='cuda')
            nnls_batch(A, b); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                x, rn, st = nnls_batch(A, b)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            t0 = time.perf_counter(); _, _, it = oracle.nnls(A[0].cpu().numpy(), b[0].cpu().numpy()); cpu = (time.perf_counter() - t0) * 1e3
            print(json.dumps(dict(kernel="nnls_block_kernel", rows=rows, cols=cols, batch=B, ms=round(ms, 3), ms_per_problem=round(ms / B, 4),
                                  oracle_ms_one_problem=round(cpu, 3), oracle_iters=[int(it[0]), int(it[1])], ok=bool((st == 0).all()))), flush=True)
    else:
        from pympc_amd.optimization.solvers.hip import quadratic_program_batch
        for n, m, B in ((100, 300, 1), (255, 1000, 1), (500, 1000, 1), (100, 300, 512), (255, 1000, 256), (500, 1000, 256)):
            G = rng.randn(n, n); H = G.dot(G.T) / n + np.eye(n); A = rng.randn(m, n)
            F = rng.randn(B, n); Bv = A.dot(rng.randn(n))[None, :] + rng.rand(B, m)
            Ht, At, Ft, Bt = [torch.as_tensor(a, device='cuda') for a in (H, A, F, Bv)]
            quadratic_program_batch(Ht, Ft, At, Bt); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = quadratic_program_batch(Ht, Ft, At, Bt, want_iters=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            t0 = time.perf_counter(); ref = oracle.quadratic_program(H, F[0], A, Bv[0]); cpu = (time.perf_counter() - t0) * 1e3
            print(json.dumps(dict(kernels="qp_block_prepare + nnls_block + qp_block_recover", n=n, m=m, batch=B, ms=round(ms, 3),
                                  ms_per_qp=round(ms / B, 4), oracle_ms_one_qp=round(cpu, 3), active=len(ref['active_set']),
                                  ok=bool(r.feasible.all()))), flush=True)


if __name__ == "__main__":
    main()
