#!/usr/bin/env python
"""Sharded forms of configs C4 and C5: one process per GPU, contiguous shards, ONE exchange at the end.

    python benchmarks/sharded.py --config c4,c5                                   (one rank; still goes through RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        benchmarks/sharded.py --config c4,c5

C4: the 3163^2-point grid of the explicit inverted-pendulum controller is sharded by rows of the point list
    (sharding.shard_range); every rank locates its points and the step ends with an all-gather of (region, u0, V).
C5: the 3^10 mode sequences of the hybrid cart-pole are sharded; every rank condenses and solves its sequences, the
    step ends with an all-gather of the costs (8 B per sequence) and a local arg-min (sharding.argmin_over_ranks).
Rank 0 prints one JSON line per configuration; time = max over ranks of the wall time between two barriers.
"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pympc_amd import sharding  # noqa: E402


def golden(name):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))


def timed(step, steps, warmup, dev):
    for _ in range(warmup):
        step()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dist.barrier(); torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()) / steps


def c4(args, dev, rank, world):
    from pympc_amd.control.controllers import ModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    g = golden("nb_model")
    ctrl = ModelPredictiveController(LinearSystem(g["S_A"], g["S_B"]), int(g["N"]), g["Q"], g["R"], g["P"],
                                     Polyhedron(g["D_A"], g["D_b"]), Polyhedron(g["XN_A"], g["XN_b"]))
    ctrl.store_explicit_solution()
    es = ctrl.explicit_solution
    n = args.grid
    B = n * n
    lo, hi = sharding.shard_range(B)
    k = torch.arange(lo, hi, device=dev)
    ax = torch.linspace(-.5, .5, n, dtype=torch.float64, device=dev)
    X = torch.stack((ax[k // n], ax[k % n]), dim=1).contiguous()       # this rank's rows of the grid's point list
    out = {}

    def step():
        idx, u, V = es.locate(X, nu=1)
        out["idx"] = sharding.gather_rows(idx, B)
        out["u"] = sharding.gather_rows(u, B)
        out["V"] = sharding.gather_rows(V, B)

    t = timed(step, args.steps, args.warmup, dev)
    if rank == 0:
        cov = float((out["idx"] >= 0).double().mean())
        print(json.dumps(dict(config="C4 sharded", n_gpus=world, points=B, value=B / t, unit="points/s", ms_per_step=t * 1e3,
                              covered_fraction=cov, exchange="RCCL all-gather of (region, u0, V): 20 B per point",
                              regions=len(es.critical_regions))), flush=True)


def c5(args, dev, rank, world):
    from pympc_amd.control.controllers import HybridModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import AffineSystem, PieceWiseAffineSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    h = golden("hybrid_seqs")
    systems = [AffineSystem(h['S_A'][i], h['S_B'][i], h['S_c'][i]) for i in range(3)]
    domains = [Polyhedron(h['D%d_A' % i], h['D%d_b' % i]) for i in range(3)]
    N = args.horizon
    ctrl = HybridModelPredictiveController(PieceWiseAffineSystem(systems, domains), N, h['Q'], h['R'], h['P'],
                                           Polyhedron(h['XN_A'], h['XN_b']))
    seqs = np.array(list(itertools.product(range(3), repeat=N)), dtype=np.int32)
    B = seqs.shape[0]
    lo, hi = sharding.shard_range(B)
    mine = torch.as_tensor(seqs[lo:hi]).to(dev)
    x0 = torch.as_tensor(h['x0']).to(dev)
    out = {}

    def step():
        res, _ = ctrl.solve_mode_sequences(x0, mine, want_mult=False, want_active=False)
        cost = torch.where(res.t['status'] == 0, res.t['min'], torch.full_like(res.t['min'], float('inf')))
        out["best"] = sharding.argmin_over_ranks(cost, B)

    t = timed(step, args.steps, args.warmup, dev)
    if rank == 0:
        k, v = out["best"]
        print(json.dumps(dict(config="C5 sharded", n_gpus=world, sequences=B, value=B / t,
                              unit="mode-sequence QPs/s (condense + prepare + solve + exchange)", ms_per_step=t * 1e3,
                              best_sequence=[int(z) for z in seqs[k]] if k >= 0 else None, best_cost=v,
                              exchange="RCCL all-gather of the costs (8 B per sequence) + local arg-min")), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4,c5")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=3163)
    ap.add_argument("--horizon", type=int, default=10)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    for name in args.config.split(","):
        {"c4": c4, "c5": c5}[name](args, dev, rank, world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
