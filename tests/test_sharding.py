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


@pytest.mark.gpu
def test_sharded_feedback_hip_solver_through_rccl_one_rank():
    """solve_sharded with the HIP local solver under torch.distributed "nccl" (= RCCL) with ONE rank on the GPU box: the
    code path of N ranks (init_process_group, shard_range, all_gather_into_tensor on device tensors), checked against
    the plain batched call.  N > 1 launch lines: README.md."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from pympc_amd import sharding
from pympc_amd.optimization.solvers.hip import PreparedMpqp
m = dict(np.load(os.path.join(%r, "tests", "golden", "c1_model.npz")))
prep = PreparedMpqp(m["Huu"], m["Hux"], m["Hxx"], m["fu"], m["fx"], float(m["g"]), m["Au"], m["Ax"], m["b"])
X = torch.as_tensor(np.random.RandomState(0).uniform([-5., -1.], [5., 1.], (1001, 2))).cuda()
def local(Xs):
    r = prep.solve_batch(Xs, n_out=1, want_mult=False, want_active=False)
    return dict(u0=r.t["argmin"], status=r.t["status"], V=r.t["min"])
got, _ = sharding.solve_sharded(local, X, ("u0", "status", "V"))
ref = local(X)
torch.cuda.synchronize()
assert torch.equal(got["status"], ref["status"])
assert torch.equal(torch.nan_to_num(got["u0"]), torch.nan_to_num(ref["u0"]))
cost = torch.where(ref["status"] == 0, ref["V"], torch.full_like(ref["V"], float("inf")))
k, v = sharding.argmin_over_ranks(cost, X.shape[0])
assert k == int(torch.argmin(cost)) and v == float(cost.min())
dist.barrier(); dist.destroy_process_group()
print("sharded ok", int((ref["status"] == 0).sum()))
''' % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=170)
    assert out.returncode == 0 and "sharded ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
