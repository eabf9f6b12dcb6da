"""Host-to-host feedback over a stream of batches: ModelPredictiveController.feedback for states that arrive in host
memory and whose inputs are wanted back in host memory (the metric counts the upload of the states and
the download of the results).  The reference has no counterpart (controllers.py:104-124 is one state per call); this is
the serving loop around pympc_mpqp_solve_batch_cuts.

Per batch t:  upload X_t (pinned host -> HBM), solve, download (u0_t, status_t) to pinned host memory.  Three HIP
streams and `depth` sets of device buffers, so that the upload of batch t + 1 and the download of batch t - 1 run while
batch t is being solved.

MI355X / ROCm detail that decides whether anything overlaps: HIP maps every stream of a process onto one of a few
hardware queues (GPU_MAX_HW_QUEUES, 4 here) and a hardware queue runs its packets in order, so two streams that share a
queue serialise (measured: 2.0 ms per 1 M-state step with three distinct queues, 2.6 - 3.9 ms when the copy streams
share a queue with each other or with the compute stream; which it is depends on the order streams were created in).
pick_copy_streams() therefore takes candidate streams from torch's pool and MEASURES which ones run independently of
the compute stream and of each other (a 1 ms spin kernel on one, a trivial kernel on the other).
"""
import time

import torch

from . import _hip


_ticks_per_ms = [None]


def _spin_cycles(ms):
    """torch.cuda._sleep counts ticks of a device clock whose rate differs between parts: measured once."""
    if _ticks_per_ms[0] is None:
        torch.cuda._sleep(1000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200000
        e0.record(); torch.cuda._sleep(n); e1.record()
        torch.cuda.synchronize()
        _ticks_per_ms[0] = n / max(e0.elapsed_time(e1), 1e-3)
    return max(1000, int(ms * _ticks_per_ms[0]))


def _independent(a, b, ms=1.0):
    """True when work queued on stream b does not wait for work queued earlier on stream a."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(_spin_cycles(ms))
    with torch.cuda.stream(b):
        torch.cuda._sleep(1000)
    b.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt < 0.5 * ms * 1e-3


def pick_streams(count, comp=None, candidates=12):
    """`count` streams that run independently of `comp` (default: the current stream) and of each other, found by
    measurement among `candidates` streams of torch's pool.  Returns (streams, report); when fewer independent ones
    exist (HIP maps all streams onto GPU_MAX_HW_QUEUES hardware queues) the list is filled up with whatever is nearest and
    report['independent'] tells how many really are."""
    comp = comp or torch.cuda.current_stream()
    torch.cuda._sleep(1000)   # warm the probe kernel up (first launch loads it)
    torch.cuda.synchronize()
    pool = [torch.cuda.Stream() for _ in range(candidates)]
    free = [s for s in pool if _independent(comp, s) and _independent(s, comp)]
    chosen = []
    for s in free:
        if len(chosen) == count:
            break
        if all(_independent(c, s) and _independent(s, c) for c in chosen):
            chosen.append(s)
    report = dict(candidates=candidates, independent_of_compute=len(free), wanted=count, independent=len(chosen))
    for s in free + pool:
        if len(chosen) == count:
            break
        if all(s is not c for c in chosen):
            chosen.append(s)
    return chosen, report


def pick_copy_streams(comp=None, candidates=8):
    """Two streams for uploads and downloads that run independently of `comp` and of each other: (s_in, s_out, report)."""
    (a, b), report = pick_streams(2, comp, candidates)
    report["independent_pair"] = report["independent"] == 2
    return a, b, report


class FeedbackPipeline(object):
    """Batched feedback with transfers overlapped: submit(X_host) queues one batch and returns the slot whose results
    (pinned host tensors u0 [B x nu], status [B]) are complete after wait(slot).

    depth: sets of device / host buffers (batches in flight).  compute_streams: 1 = every solve on the current stream;
    2 = solves of consecutive batches alternate between two streams, each with its own copy of the certificate table, so
    that the tail of one batch's kernels (a few long solves) overlaps the head of the next batch's (measured on the
    cart-pole batch: 1.55 -> 1.36 ms per 1 M states)."""

    def __init__(self, prep, batch, nu, depth=2, cold=True, streams=None, compute_streams=1):
        from .optimization.solvers.hip import _alloc_outputs
        self.prep, self.B, self.nu, self.depth, self.cold = prep, int(batch), int(nu), int(depth), cold
        dev = prep.dev
        main = torch.cuda.current_stream()
        ncs = max(1, min(int(compute_streams), self.depth))
        if streams is None:
            picked, self.stream_report = pick_streams(2 + (ncs - 1), main)
            self.s_in, self.s_out = picked[0], picked[1]
            self.comp = [main] + picked[2:]
            self.stream_report["independent_pair"] = self.stream_report["independent"] >= 2
        else:
            self.s_in, self.s_out = streams[0], streams[1]
            self.comp = [main] + list(streams[2:2 + ncs - 1])
            self.stream_report = dict(given=True)
        self.tables = [None] + [prep.clone_cuts() for _ in self.comp[1:]]    # None: the prepared mpQP's own table
        self.Xd = [torch.empty((self.B, prep.p), dtype=torch.float64, device=dev) for _ in range(depth)]
        self.outs = [_alloc_outputs(self.B, nu, prep.m, dev, want_mult=False, want_active=True, want_iters=False)
                     for _ in range(depth)]
        self.u_host = [torch.empty((self.B, nu), dtype=torch.float64).pin_memory() for _ in range(depth)]
        self.s_host = [torch.empty(self.B, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.ev_in = [torch.cuda.Event() for _ in range(depth)]
        self.ev_done = [torch.cuda.Event() for _ in range(depth)]
        self.ev_out = [torch.cuda.Event() for _ in range(depth)]
        for k in range(depth):
            self.ev_done[k].record(main)
            self.ev_out[k].record(main)
        for c in self.comp[1:]:
            c.wait_stream(main)
        self.t = 0

    def submit(self, X_host):
        """X_host: pinned [B x p] float64 host tensor.  Returns the slot index of this batch."""
        k = self.t % self.depth
        c = self.t % len(self.comp)
        comp, table = self.comp[c], self.tables[c]
        self.t += 1
        self.s_in.wait_event(self.ev_done[k])            # the solve that last read this slot's states has finished
        with torch.cuda.stream(self.s_in):
            self.Xd[k].copy_(X_host, non_blocking=True)
            self.ev_in[k].record(self.s_in)
        comp.wait_event(self.ev_in[k])
        comp.wait_event(self.ev_out[k])                  # ... and this slot's previous results have left the device
        with torch.cuda.stream(comp):
            if self.cold:
                self.prep.reset_cuts(table=table)
            self.prep.solve_batch(self.Xd[k], n_out=self.nu, out=self.outs[k], use_cuts=True, cuts_table=table)
            self.ev_done[k].record(comp)
        self.s_out.wait_event(self.ev_done[k])
        with torch.cuda.stream(self.s_out):
            self.u_host[k].copy_(self.outs[k]["argmin"], non_blocking=True)
            self.s_host[k].copy_(self.outs[k]["status"], non_blocking=True)
            self.ev_out[k].record(self.s_out)
        return k

    def wait(self, k):
        self.ev_out[k].synchronize()
        return self.u_host[k], self.s_host[k]
