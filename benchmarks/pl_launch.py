#!/usr/bin/env python
"""Config C4's look-up alone (for rocprofv3): python3 benchmarks/pl_launch.py [launches] [sure|scan]."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))


def main():
    import configs
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    sure = (sys.argv[2] if len(sys.argv) > 2 else "sure") == "sure"
    ctrl = configs.controller(configs.golden("nb_model"))
    ctrl.store_explicit_solution()
    es = ctrl.explicit_solution
    n = 3163
    g = torch.linspace(-.5, .5, n, dtype=torch.float64, device='cuda')
    X = torch.stack(torch.meshgrid(g, g, indexing='ij'), dim=-1).reshape(-1, 2).contiguous()
    es.locate(X, nu=1, use_sure=sure)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = es.locate(X, nu=1, use_sure=sure)
    torch.cuda.synchronize()
    print("ms per launch (host clock, %d launches): %.4f" % (reps, (time.perf_counter() - t0) / reps * 1e3))


if __name__ == "__main__":
    main()
