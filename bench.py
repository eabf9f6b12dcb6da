#!/usr/bin/env python
"""Headline benchmark: QP solves/sec, cart-pole linear MPC N=20 (config C2 of SURVEY.md section 8), fp64.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (pympc_mpqp_solve_batch: ModelPredictiveController.feedback over a batch) over
1,000,000 states per GPU that are already resident in HBM; outputs per state are u0, V, status and the active-set
bit mask (32 B in, 68 B out).  With N > 1 every rank solves its own 1M-state shard (weak scaling) and
the step ends with the path's one exchange: an RCCL all-gather of (u0, status).  Rank 0 prints ONE JSON line.

The model matrices come from tests/golden/c2_model.npz (condensed by the reference from the cart-pole of
examples/hybrid_mpc_2.ipynb; the one-time controller set-up is outside the metric).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_VALU_TFLOPS = 78.6     # MI355X fp64 vector peak: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(n, m, p, outer, inner):
    """flops of Lawson-Hanson NNLS on the (n+1) x m QP matrix, from the kernel's counters."""
    kmax = int(outer.max()) if outer.size else 0
    j = np.arange(1, kmax + 1)
    jj = np.minimum(j, n + 1)
    per_outer = 2.0 * (n + 1) * m + 4.0 * np.maximum(n + 2 - jj, 0) * np.maximum(m - jj, 0) + jj.astype(float) ** 2
    cum = np.concatenate(([0.0], np.cumsum(per_outer)))
    k = outer.astype(np.int64)
    jbar = np.minimum(k, n + 1).astype(float)
    extra = np.maximum(inner.astype(np.int64) - k, 0)
    F = 2.0 * m * p + cum[k] + 2.0 * (n + 1) * m + 2.0 * n * k + 2.0 * m + extra * (jbar ** 2 + 6.0 * (n + 1))
    return F


def algorithmic_flops_strict(n, m, p, outer, inner):
    """The same count with the states that never enter the active-set loop (no negative entry in m(x), or rejected by a
    stored certificate: outer == 0) credited their m(x) evaluation only, 2 m p, instead of Lawson-Hanson's first dual
    vector and output assembly on top (2 (n + 1) m + 2 m): what a reader who discounts the certificates would count."""
    F = algorithmic_flops(n, m, p, outer, inner)
    return np.where(outer == 0, 2.0 * m * p, F)


def executed_flops(n, m, p, outer, inner):
    """Flops this kernel's formulation needs: it sweeps the UNtransformed matrix (2(n+1)m per outer iteration) and
    updates a small Q / R^-1 pair (~6 R^2) instead of applying each reflector to every column."""
    R = n + 1
    k = outer.astype(np.float64)
    return 2.0 * m * p + k * (2.0 * R * m + 8.0 * R * R) + inner.astype(np.float64) * R * R + 2.0 * n * n + 2.0 * m


def cpu_baseline(model, x_max, scale, seconds_target=12.0):
    """CPU oracle (scalar C port of the reference loop, refactorising per state like pnnls.py:210-213) on a bounded
    sample of the same workload; 1 core, and a thread fan-out over the box's CPU share for a whole-CPU figure."""
    from oracle import oracle as O
    O.build()
    rng = np.random.RandomState(7)
    probe = rng.uniform(-scale * x_max, scale * x_max, (2000, 4))
    t0 = time.perf_counter(); O.mpqp_solve_batch(model, probe, want_mult=False); dt = time.perf_counter() - t0
    count = int(max(2000, min(200000, seconds_target / (dt / 2000))))
    X = rng.uniform(-scale * x_max, scale * x_max, (count, 4))
    t0 = time.perf_counter(); O.mpqp_solve_batch(model, X, want_mult=False); dt = time.perf_counter() - t0
    one = count / dt
    out = dict(value=one, unit="QP solves/s", cores=1, kind="port",
               sample="%d states of the same C2 distribution (s=%.1f), oracle/pnnls_oracle.c oracle_mpqp_solve_batch, %.1f s"
                      % (count, scale, dt))
    try:
        from concurrent.futures import ThreadPoolExecutor
        nth = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
        nth = max(1, min(nth, 16))
        chunks = np.array_split(X, nth)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nth) as ex:
            list(ex.map(lambda c: O.mpqp_solve_batch(model, c, want_mult=False), chunks))
        dtm = time.perf_counter() - t0
        out["all_cores"] = dict(value=count / dtm, cores=nth)
    except Exception as e:  # pragma: no cover
        out["all_cores"] = dict(error=str(e))
    return out


def reference_python_rate(scale):
    """SURVEY 8(d) 'reference-as-is' figure: the reference's own ModelPredictiveController path (CPython + scipy 1.15.3
    nnls, 1 core) timed in the build container while tests/golden/make_golden.py::gen_c2big solved the first 10 000
    states of this benchmark's stream; the reference's files do not travel to the GPU box, so the recorded time is
    read back from the fixture."""
    tag = {0.3: "s03", 1.0: "s10"}.get(round(scale, 1))
    path = os.path.join(ROOT, "tests", "golden", "c2_big_%s.npz" % tag) if tag else None
    if not path or not os.path.exists(path):
        return None
    rec = np.load(path)
    n = int(rec["X"].shape[0])
    return dict(value=n / float(rec["ref_seconds"]), unit="QP solves/s", cores=1, kind="reference",
                sample="%d states (the first of this benchmark's stream, s=%.1f) through the reference's mpqp.solve" % (n, scale),
                where="build container (8 vCPU, no GPU), CPython 3.10 + numpy 2.2 + scipy 1.15.3; recorded in " + os.path.basename(path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1000000, help="states per GPU per step")
    ap.add_argument("--scale", type=float, default=0.3, help="states ~ U(-s x_max, s x_max); 0.3 headline, 1.0 stress")
    ap.add_argument("--cuts", choices=("cold", "warm", "off"), default="cold",
                    help="learned infeasibility certificates: cold = table emptied before EVERY step (each step learns from "
                         "scratch, nothing carried over), warm = kept across steps, off = plain solve")
    ap.add_argument("--compare-plain", action="store_true",
                    help="also time the same launch without certificates (extra launches: keep it off under a profiler)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--force-gather", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the all-gather of (u0, status) even with one rank")
    ap.add_argument("--pcie-chunks", type=int, default=1,
                    help="0: skip the PCIe-inclusive (host-to-host) measurement")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.stderr.write("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)\n" % args.gpus)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pympc_amd.optimization.solvers.hip import PreparedMpqp, _alloc_outputs
    model = dict(np.load(os.path.join(ROOT, "tests", "golden", "c2_model.npz")))
    x_max = model["x_max"]
    m, n = model["Au"].shape
    p = model["Ax"].shape[1]
    nu = model["S_B"].shape[1]
    prep = PreparedMpqp(model["Huu"], model["Hux"], model["Hxx"], model["fu"], model["fx"], float(model["g"]),
                        model["Au"], model["Ax"], model["b"])
    n_seed = prep.seed_cuts()   # set-up: certificates from samples of the feasible box (no benchmark state involved)
    B = args.batch
    X = torch.as_tensor(np.random.RandomState(rank).uniform(-args.scale * x_max, args.scale * x_max, (B, p))).to(dev)
    out = _alloc_outputs(B, nu, m, dev, want_mult=False, want_active=True, want_iters=False)
    gather = use_dist and not args.no_gather
    if gather:
        g_u = torch.empty((world * B, nu), dtype=torch.float64, device=dev)
        g_s = torch.empty(world * B, dtype=torch.int32, device=dev)

    use_cuts = args.cuts != "off"

    def solve():
        if args.cuts == "cold":
            prep.reset_cuts()
        prep.solve_batch(X, n_out=nu, out=out, use_cuts=use_cuts)

    def step():
        solve()
        if gather:
            dist.all_gather_into_tensor(g_u, out["argmin"])
            dist.all_gather_into_tensor(g_s, out["status"])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()                      # HIP events on the stream the kernel is launched on (torch's current stream)
        solve()
        b.record()
        if gather:
            dist.all_gather_into_tensor(g_u, out["argmin"])
            dist.all_gather_into_tensor(g_s, out["status"])
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- PCIe-inclusive figure (wall time incl. H2D of the states and D2H of the results; reported next
    # to `value`, which by the bench contract counts states resident in HBM).  Every step uploads its 1M states from
    # pinned host memory and downloads (u0, status) to pinned host memory through pympc_amd.pipeline.FeedbackPipeline:
    # upload / solve / download on three streams that are MEASURED to map to three different hardware queues, two sets
    # of device buffers.  The first ~50 ms after new streams and pinned buffers come into use show multi-millisecond
    # stalls (profiles/r03/pcie_pipeline.md), so the pipeline runs 40 untimed steps first; a serving loop lives in the
    # steady state that follows.
    pcie = None
    if args.pcie_chunks > 0 and rank == 0:
        from pympc_amd.pipeline import FeedbackPipeline
        Xh = X.cpu().pin_memory()
        pipe = FeedbackPipeline(prep, B, nu, depth=3, cold=(args.cuts == "cold"), compute_streams=2)
        for t in range(max(40, args.warmup)):
            pipe.submit(Xh)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in range(args.steps):
            k = pipe.submit(Xh)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        uh, sh = pipe.wait(k)
        assert int((sh == 0).sum()) == int((out["status"] == 0).sum())
        pcie = {"value": B * args.steps / dt, "unit": "QP solves/s", "ms_per_step": dt / max(args.steps, 1) * 1e3,
                "device_buffer_sets": 3, "compute_streams": 2, "host_bytes_per_state": 8 * p + 8 * nu + 4, "streams": pipe.stream_report,
                "note": "every step: pinned host states in, (u0, status) out to pinned host memory, certificate table "
                        "restored to its set-up state before the step's solve; transfers of neighbouring steps overlap the "
                        "solve and the solves of consecutive steps alternate between two streams, each with its own "
                        "certificate table (pympc_amd/pipeline.py); 40 untimed steps first; one GPU (rank 0)"}

    # roofline accounting: one extra untimed pass that also writes the iteration counters (deterministic kernel)
    if args.cuts == "cold":
        prep.reset_cuts()
    res = prep.solve_batch(X, n_out=nu, want_mult=False, want_active=False, want_iters=True, use_cuts=use_cuts)
    torch.cuda.synchronize()
    iters = res.numpy("iters")
    status = res.numpy("status")
    cut_hits = float(((status == 1) & (iters[:, 0] == 0)).mean())
    n_cuts = prep.n_cuts()
    # the same launch without certificates, for reference (untimed by the metric)
    plain_ms = None
    if use_cuts and rank == 0 and args.compare_plain:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prep.solve_batch(X, n_out=nu, out=out, use_cuts=False)
        e0.record()
        for _ in range(3):
            prep.solve_batch(X, n_out=nu, out=out, use_cuts=False)
        e1.record()
        torch.cuda.synchronize()
        plain_ms = e0.elapsed_time(e1) / 3
    # ... and with the certificate table kept from the previous call (the serving case: the table belongs to the prepared
    # mpQP and lives across calls); reported beside `value`, which pays for an empty table every step
    kept_ms = None
    if use_cuts and rank == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prep.solve_batch(X, n_out=nu, out=out, use_cuts=True)
        e0.record()
        for _ in range(5):
            prep.solve_batch(X, n_out=nu, out=out, use_cuts=True)
        e1.record()
        torch.cuda.synchronize()
        kept_ms = e0.elapsed_time(e1) / 5
    # ... and with two steps in flight: consecutive steps alternate between two streams (own outputs, own copy of the
    # certificate table), so that the tail of one step's kernels overlaps the head of the next step's.  A throughput
    # figure for callers that have more than one batch at hand; `value` stays the one-stream number, whose kernel times
    # the rocprofv3 summaries under profiles/ reproduce.
    two = None
    if use_cuts and rank == 0 and args.cuts == "cold":
        from pympc_amd.pipeline import pick_streams
        (s2,), rep2 = pick_streams(1)
        strs = [torch.cuda.current_stream(), s2]
        outs2 = [out, _alloc_outputs(B, nu, m, dev, want_mult=False, want_active=True, want_iters=False)]
        tabs = [None, prep.clone_cuts()]
        s2.wait_stream(strs[0])

        def step2(t):
            with torch.cuda.stream(strs[t % 2]):
                prep.reset_cuts(table=tabs[t % 2])
                prep.solve_batch(X, n_out=nu, out=outs2[t % 2], use_cuts=True, cuts_table=tabs[t % 2])
        for t in range(6):
            step2(t)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for t in range(args.steps):
            step2(t)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        assert torch.equal(outs2[0]["status"], outs2[1]["status"]) and torch.equal(outs2[0]["active_bits"], outs2[1]["active_bits"])
        two = {"value": B * args.steps / dt2, "unit": "QP solves/s", "ms_per_step": dt2 / max(args.steps, 1) * 1e3,
               "independent_stream_found": rep2["independent"] == 1}
    F = algorithmic_flops(n, m, p, iters[:, 0], iters[:, 1])
    Fs = algorithmic_flops_strict(n, m, p, iters[:, 0], iters[:, 1])
    Fx = executed_flops(n, m, p, iters[:, 0], iters[:, 1])
    alg_bytes = B * (8 * p + 8 * nu + 8 + 4 + 4 * ((m + 31) // 32))

    if rank == 0:
        value = world * B * args.steps / elapsed
        achieved = F.sum() / (kern_ms * 1e-3) / 1e12
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE, separate passes,
        # same command; see profiles/pmc_traffic.json for the raw counters and the gfx950 FETCH_SIZE caveat)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and B == 1000000:
            try:
                traffic = json.load(open(pmc)).get("s%02d" % int(round(args.scale * 10)), {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "QP solves/sec (whole node) for cart-pole MPC N=20",
            "value": value,
            "value_definition": "whole-job throughput with inputs already resident in HBM when the timed region starts (the "
                                "bench contract of this build); the rate including the upload of the states and the download "
                                "of (u0, status) over PCIe, SURVEY 8(d)'s wording of the metric, is pcie_inclusive.value", "unit": "QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: cart-pole linear MPC N=20 (n=%d vars, m=%d rows, p=%d), batched feedback over "
                                   "%d states per GPU, states ~ U(+-%.1f x_max) seed=rank" % (n, m, p, B, args.scale),
                       "states_per_gpu": B, "scale": args.scale, "outputs": "u0, V, status, active-set bits",
                       "exchange": "RCCL all-gather of (u0, status)" if gather else "none",
                       "feasible_fraction": float((status == 0).mean()),
                       "mean_outer_iterations": float(iters[:, 0].mean()), "max_outer_iterations": int(iters[:, 0].max()),
                       "infeasibility_cuts": {"cold": "table restored before EVERY step to the certificates stored at set-up "
                                                      "(PreparedMpqp.seed_cuts: synthetic samples of the feasible box, no benchmark "
                                                      "state); nothing learned in one step reaches the next",
                                              "warm": "table kept across steps", "off": "plain solve"}[args.cuts],
                       "certificates_from_set_up": n_seed, "certificates_after_a_step": n_cuts,
                       "states_rejected_by_a_cut": cut_hits,
                       "plain_kernel_ms_without_cuts": plain_ms, "ms_per_step_with_kept_table": kept_ms,
                       "two_steps_in_flight": two},
            "roofline": {"bound": "fp64_valu", "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json: FETCH_SIZE / WRITE_SIZE passes of benchmarks/qp_launch.py (the "
                                           "same launch) committed with this build, not counters of this run",
                         "frac_strict": Fs.sum() / (kern_ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                         "kernel": "one step = mpqp_screen_kernel<%d> (lane path: up to 2 active constraints) + mpqp_gram_kernel<%d> "
                                   "(Gram form: up to 8) + qp_solve_kernel<%d,true> (Lawson-Hanson: the rest); the learning pre-pass "
                                   "(qp_solve_kernel on 8192 states) returns at once with a seeded table"
                                   % (p, 2 * ((m + 127) // 128), 2 * ((m + 127) // 128)), "kernel_ms": kern_ms,
                         "algorithmic_flops_per_launch": float(F.sum()), "algorithmic_flops_strict_per_launch": float(Fs.sum()),
                         "executed_flops_per_launch": float(Fx.sum()),
                         "executed_tflops": Fx.sum() / (kern_ms * 1e-3) / 1e12,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "hbm_algorithmic_gbs": alg_bytes / (kern_ms * 1e-3) / 1e9, "hbm_peak_gbs": PEAK_HBM_GBS,
                         "note": "SURVEY 8(d): the NNLS kernel is bound by fp64 vector-ALU issue (+LDS), not by HBM or MFMA; "
                                 "achieved = algorithmic Lawson-Hanson flops (SURVEY 8(d) formula on the kernels' own iteration counters; "
                                 "EVERY state is credited 2mp + 2(n+1)m + 2m: m(x), the first dual vector and the output assembly, also "
                                 "the states a certificate rejects; frac_strict credits states that never iterate 2mp only) / HIP-event "
                                 "time of one step (certificate restore + screening, Gram-form and solve kernels); executed_tflops is "
                                 "what the formulation actually issues; "
                                 "peak = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (nominal; independent FMA chains sustain 53 "
                                 "TFLOP/s on this part: benchmarks/fp64_rate.hip, profiles/r01_v5/fp64_rate.txt)",
                         "measured_fp64_fma_tflops": 53.2},
        }
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(model, x_max, args.scale)
            line["cpu_baseline"]["reference_python"] = reference_python_rate(args.scale)
            line["gpu_over_cpu_1core"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
