"""CPU tests of the multi-GPU host logic (pympc_amd/sharding.py): world_size-2 (and 3) gloo process groups.
The local solver is the CPU oracle here (test infrastructure); on the GPU box the same functions wrap the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden


def test_shard_range_partitions():
    from pympc_amd.sharding import shard_range
    for B in (0, 1, 7, 8, 1000, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_range(B, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from pympc_amd.sharding import argmin_over_ranks, gather_rows, shard_range, solve_sharded
    model = dict(np.load(os.path.join(ROOT, "tests", "golden", "c1_model.npz")))
    X = torch.as_tensor(np.random.RandomState(0).uniform([-5., -1.], [5., 1.], (B, 2)))

    def local_solve(Xl):
        r = O.mpqp_solve_batch(model, Xl.numpy(), want_mult=False)
        return dict(u0=torch.as_tensor(r["argmin"][:, :1].copy()), status=torch.as_tensor(r["status"]),
                    V=torch.as_tensor(np.where(r["status"] == 0, r["min"], np.nan)))

    got, local = solve_sharded(local_solve, X, ("u0", "status", "V"))
    lo, hi = shard_range(B)
    assert local["status"].shape[0] == hi - lo
    k, v = argmin_over_ranks(local["V"], B)
    # a ragged gather of an int tensor with trailing dims
    tag = torch.full((hi - lo, 3), rank, dtype=torch.int32)
    tags = gather_rows(tag, B)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), u0=got["u0"].numpy(), status=got["status"].numpy(),
             V=got["V"].numpy(), k=k, v=v, tags=tags.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 101), (2, 64), (3, 10)])
def test_sharded_feedback_gloo(tmp_path, oracle, world, B):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    model = dict(golden("c1_model"))
    X = np.random.RandomState(0).uniform([-5., -1.], [5., 1.], (B, 2))
    ref = oracle.mpqp_solve_batch(model, X, want_mult=False)
    V = np.where(ref["status"] == 0, ref["min"], np.inf)
    from pympc_amd.sharding import shard_range
    owner = np.concatenate([np.full(b - a, r) for r in range(world) for a, b in [shard_range(B, r, world)]])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(got["status"], ref["status"])                      # every rank holds the whole result
        feas = ref["status"] == 0
        assert np.array_equal(got["u0"][feas], ref["argmin"][feas][:, :1])       # same code on the same rows: bit-exact
        assert int(got["k"]) == int(np.argmin(V)) and float(got["v"]) == float(V.min())
        assert np.array_equal(got["tags"], np.repeat(owner[:, None], 3, axis=1))
