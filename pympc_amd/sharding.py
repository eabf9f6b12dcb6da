"""Multi-GPU form of the batched paths: one process per GPU, contiguous shards of the batch dimension, and ONE
exchange at the end (an all-gather of the per-problem outputs) -- the problems are independent, so nothing is
exchanged while they are being solved.

The reference is a single-threaded Python loop (controllers.py:94, polyhedron.py:435-448, parametric_programs.py:
442-463); this module is new.  It only uses torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests), never a collective of its own.

    lo, hi = shard_range(B)                      rows of the global batch this rank owns
    full   = gather_rows(local, B)               [B x ...] on every rank, rows in global order
    out    = solve_sharded(fn, X, keys)          fn(X[lo:hi]) -> dict of tensors; gathers the listed keys
    k, v   = argmin_over_ranks(costs_local, B)   global index and value of the smallest finite cost (config C5)
"""
import torch
import torch.distributed as dist


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(B, rank=None, world=None):
    """Contiguous, balanced shard [lo, hi) of B problems: the first B % world ranks own one problem more."""
    r, w = _world()
    rank = r if rank is None else rank
    world = w if world is None else world
    if not 0 <= rank < world:
        raise ValueError('rank %d outside [0, %d)' % (rank, world))
    q, rem = divmod(int(B), world)
    lo = rank * q + min(rank, rem)
    return lo, lo + q + (1 if rank < rem else 0)


def gather_rows(local, B):
    """All-gather of per-problem rows: every rank passes its shard [hi - lo x ...] and receives [B x ...] in global
    order.  Shards are padded to the largest one so that a single fixed-size all_gather_into_tensor does the job."""
    rank, world = _world()
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != B:
            raise ValueError('local shard has %d rows, expected %d' % (local.shape[0], B))
        return local
    # (an initialised group of ONE rank still goes through the collective: the same code path as N ranks)
    lo, hi = shard_range(B, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError('rank %d: local shard has %d rows, expected %d' % (rank, local.shape[0], hi - lo))
    cap = (B + world - 1) // world
    tail = tuple(local.shape[1:])
    send = local
    if hi - lo < cap:
        send = torch.empty((cap,) + tail, dtype=local.dtype, device=local.device)
        send[:hi - lo] = local
        send[hi - lo:] = 0
    recv = torch.empty((world * cap,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous())
    if B == world * cap:
        return recv
    parts = []
    for r in range(world):
        a, b = shard_range(B, r, world)
        parts.append(recv[r * cap:r * cap + (b - a)])
    return torch.cat(parts, dim=0)


def solve_sharded(fn, X, keys):
    """Solves this rank's shard of X [B x p] with fn (-> mapping of per-problem tensors) and all-gathers `keys`.

    X may be the global batch (every rank holds it, e.g. generated from a common seed) -- only rows [lo, hi) are read.
    Returns (gathered dict, local result of fn)."""
    B = X.shape[0]
    lo, hi = shard_range(B)
    local = fn(X[lo:hi])
    return {k: gather_rows(local[k], B) for k in keys}, local


def argmin_over_ranks(costs_local, B):
    """Index (into the global batch) and value of the smallest finite cost; (-1, inf) if there is none.
    costs_local: [hi - lo] with NaN / inf for infeasible problems.  One all-gather of 8 B per problem."""
    costs = gather_rows(torch.nan_to_num(costs_local, nan=float('inf')), B)
    if B == 0:
        return -1, float('inf')
    k = int(torch.argmin(costs).item())
    v = float(costs[k].item())
    return (k, v) if v != float('inf') else (-1, v)
