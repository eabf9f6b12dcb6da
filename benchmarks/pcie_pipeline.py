#!/usr/bin/env python
"""The PCIe-inclusive step of bench.py in isolation (wall time of the batched call including the upload of
the states and the download of the results), for timing variants of the pipeline and for tracing it:

    python3 benchmarks/pcie_pipeline.py --steps 20                       # every variant, one line each
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/pcie -- \
        python3 benchmarks/pcie_pipeline.py --variant streams3 --steps 6    # one variant under the tracer

Variants: picked* (pympc_amd.pipeline.FeedbackPipeline: streams chosen by measurement; 2c = two compute streams), serial
(one stream), streams3 (round 2's: upload / solve / download on whatever streams torch hands out), streams3+reserved
(persistent launches leave CUs free for copy kernels), chunked uploads.  High-priority copy streams were tried once and
hung the run (the second variant never finished): not offered.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1000000)
    ap.add_argument("--scale", type=float, default=0.3)
    ap.add_argument("--variant", default="all")
    ap.add_argument("--reserved", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=1)
    args = ap.parse_args()
    from pympc_amd import _hip
    from pympc_amd.optimization.solvers.hip import PreparedMpqp, _alloc_outputs
    model = dict(np.load(os.path.join(ROOT, "tests", "golden", "c2_model.npz")))
    x_max = model["x_max"]
    m, n = model["Au"].shape
    p = 4
    nu = 1
    dev = torch.device("cuda", 0)
    prep = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], float(model["g"]),
                        model["Au"], model["Ax"], model["b"])
    prep.seed_cuts()
    B = args.batch
    Xh = torch.as_tensor(np.random.RandomState(0).uniform(-args.scale * x_max, args.scale * x_max, (B, p))).pin_memory()

    def run(name, nbuf, streams, reserved=0, up_chunks=1, steps=args.steps):
        _hip.tune("reserved_cus", reserved)
        uh = [torch.empty((B, nu), dtype=torch.float64).pin_memory() for _ in range(nbuf)]
        sh = [torch.empty(B, dtype=torch.int32).pin_memory() for _ in range(nbuf)]
        Xd = [torch.empty((B, p), dtype=torch.float64, device=dev) for _ in range(nbuf)]
        outs = [_alloc_outputs(B, nu, m, dev, want_mult=False, want_active=True, want_iters=False) for _ in range(nbuf)]
        comp = torch.cuda.current_stream()
        s_in = torch.cuda.Stream() if streams else comp
        s_out = torch.cuda.Stream() if streams else comp
        ev_in = [torch.cuda.Event() for _ in range(nbuf)]
        ev_done = [torch.cuda.Event() for _ in range(nbuf)]
        ev_out = [torch.cuda.Event() for _ in range(nbuf)]
        cb = (B + up_chunks - 1) // up_chunks

        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]

        def host_step(t):
            k = t % nbuf
            s_in.wait_event(ev_done[k])
            with torch.cuda.stream(s_in):
                for c in range(up_chunks):
                    Xd[k][c * cb:(c + 1) * cb].copy_(Xh[c * cb:(c + 1) * cb], non_blocking=True)
                ev_in[k].record(s_in)
            comp.wait_event(ev_in[k])
            comp.wait_event(ev_out[k])
            prep.reset_cuts()
            prep.solve_batch(Xd[k], n_out=nu, out=outs[k], use_cuts=True)
            ev_done[k].record(comp)
            s_out.wait_event(ev_done[k])
            with torch.cuda.stream(s_out):
                uh[k].copy_(outs[k]["argmin"], non_blocking=True)
                sh[k].copy_(outs[k]["status"], non_blocking=True)
                ev_out[k].record(s_out)

        for k in range(nbuf):
            ev_done[k].record(comp)
            ev_out[k].record(comp)
        for t in range(3):
            host_step(t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks[0].record(comp)
        for t in range(steps):
            host_step(t)
            marks[t + 1].record(comp)       # end of solve t on the compute stream
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / steps
        per = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        # the resident step for comparison
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            prep.reset_cuts()
            prep.solve_batch(Xd[0], n_out=nu, out=outs[0], use_cuts=True)
        e1.record()
        torch.cuda.synchronize()
        _hip.tune("reserved_cus", 0)
        print(json.dumps(dict(variant=name, ms_per_step=dt * 1e3, qps=B / dt, resident_ms=e0.elapsed_time(e1) / 5,
                              median_step_ms=float(np.median(per)), steps_ms=[round(v, 2) for v in per],
                              feasible=int((sh[0] == 0).sum()))), flush=True)

    def run_picked(name, depth, reserved=0, ncs=1):
        from pympc_amd.pipeline import FeedbackPipeline
        _hip.tune("reserved_cus", reserved)
        t0 = time.perf_counter()
        pipe = FeedbackPipeline(prep, B, nu, depth=depth, compute_streams=ncs)
        torch.cuda.synchronize()
        setup = time.perf_counter() - t0
        for t in range(40):
            pipe.submit(Xh)
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t1 = time.perf_counter()
        marks[0].record(pipe.comp[0])
        for t in range(args.steps):
            k = pipe.submit(Xh)
            marks[t + 1].record(pipe.comp[0])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / args.steps
        per = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        u, st = pipe.wait(k)
        _hip.tune("reserved_cus", 0)
        print(json.dumps(dict(variant=name, ms_per_step=dt * 1e3, qps=B / dt, median_step_ms=float(np.median(per)),
                              steps_ms=[round(v, 2) for v in per], feasible=int((st == 0).sum()), setup_ms=setup * 1e3,
                              streams=pipe.stream_report)), flush=True)

    variants = {
        "picked": lambda: run_picked("picked", 2),
        "picked_reserved": lambda: run_picked("picked_reserved%d" % args.reserved, 2, reserved=args.reserved),
        "picked_depth3": lambda: run_picked("picked_depth3", 3),
        "picked2c": lambda: run_picked("picked2c_depth2", 2, ncs=2),
        "picked2c_depth3": lambda: run_picked("picked2c_depth3", 3, ncs=2),
        "picked2c_depth4": lambda: run_picked("picked2c_depth4", 4, ncs=2),
        "serial": lambda: run("serial", 1, False),
        "streams3": lambda: run("streams3", 2, True),
        "streams3_buf3": lambda: run("streams3_buf3", 3, True),
        "streams3_reserved": lambda: run("streams3_reserved%d" % args.reserved, 2, True, reserved=args.reserved),
        "streams3_reserved_buf3": lambda: run("streams3_reserved%d_buf3" % args.reserved, 3, True, reserved=args.reserved),
        "streams3_chunks4": lambda: run("streams3_chunks4", 2, True, up_chunks=4),
        "streams3_reserved_chunks4": lambda: run("streams3_reserved%d_chunks4" % args.reserved, 2, True, reserved=args.reserved, up_chunks=4),
    }
    for rep in range(args.repeat):
        for k, fn in variants.items():
            if args.variant in ("all", k) or k in args.variant.split(","):
                fn()


if __name__ == "__main__":
    main()
