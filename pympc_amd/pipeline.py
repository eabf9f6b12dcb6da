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


def pick_copy_streams(comp=None, candidates=8):
    """Two streams for uploads and downloads that run independently of `comp` (default: the current stream) and of each
    other.  Returns (s_in, s_out, report); falls back to whatever is nearest when no such pair is among the candidates."""
    comp = comp or torch.cuda.current_stream()
    # warm the probe kernel up (first launch compiles / loads it)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    pool = [torch.cuda.Stream() for _ in range(candidates)]
    free = [s for s in pool if _independent(comp, s) and _independent(s, comp)]
    report = dict(candidates=candidates, independent_of_compute=len(free))
    for i, a in enumerate(free):
        for b in free[i + 1:]:
            if _independent(a, b) and _independent(b, a):
                report["independent_pair"] = True
                return a, b, report
    report["independent_pair"] = False
    if len(free) >= 1:
        return free[0], free[-1], report
    return pool[0], pool[1], report


class FeedbackPipeline(object):
    """Batched feedback with transfers overlapped: submit(X_host) queues one batch and returns the slot whose results
    (pinned host tensors u0 [B x nu], status [B]) are complete after wait(slot)."""

    def __init__(self, prep, batch, nu, depth=2, cold=True, streams=None):
        from .optimization.solvers.hip import _alloc_outputs
        self.prep, self.B, self.nu, self.depth, self.cold = prep, int(batch), int(nu), int(depth), cold
        dev = prep.dev
        self.comp = torch.cuda.current_stream()
        if streams is None:
            self.s_in, self.s_out, self.stream_report = pick_copy_streams(self.comp)
        else:
            self.s_in, self.s_out = streams
            self.stream_report = dict(given=True)
        self.Xd = [torch.empty((self.B, prep.p), dtype=torch.float64, device=dev) for _ in range(depth)]
        self.outs = [_alloc_outputs(self.B, nu, prep.m, dev, want_mult=False, want_active=True, want_iters=False)
                     for _ in range(depth)]
        self.u_host = [torch.empty((self.B, nu), dtype=torch.float64).pin_memory() for _ in range(depth)]
        self.s_host = [torch.empty(self.B, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.ev_in = [torch.cuda.Event() for _ in range(depth)]
        self.ev_done = [torch.cuda.Event() for _ in range(depth)]
        self.ev_out = [torch.cuda.Event() for _ in range(depth)]
        for k in range(depth):
            self.ev_done[k].record(self.comp)
            self.ev_out[k].record(self.comp)
        self.t = 0

    def submit(self, X_host):
        """X_host: pinned [B x p] float64 host tensor.  Returns the slot index of this batch."""
        k = self.t % self.depth
        self.t += 1
        self.s_in.wait_event(self.ev_done[k])            # the solve that last read this slot's states has finished
        with torch.cuda.stream(self.s_in):
            self.Xd[k].copy_(X_host, non_blocking=True)
            self.ev_in[k].record(self.s_in)
        self.comp.wait_event(self.ev_in[k])
        self.comp.wait_event(self.ev_out[k])             # ... and this slot's previous results have left the device
        if self.cold:
            self.prep.reset_cuts()
        self.prep.solve_batch(self.Xd[k], n_out=self.nu, out=self.outs[k], use_cuts=True)
        self.ev_done[k].record(self.comp)
        self.s_out.wait_event(self.ev_done[k])
        with torch.cuda.stream(self.s_out):
            self.u_host[k].copy_(self.outs[k]["argmin"], non_blocking=True)
            self.s_host[k].copy_(self.outs[k]["status"], non_blocking=True)
            self.ev_out[k].record(self.s_out)
        return k

    def wait(self, k):
        self.ev_out[k].synchronize()
        return self.u_host[k], self.s_host[k]
