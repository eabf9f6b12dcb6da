#!/usr/bin/env python
"""Measurements of the non-headline configurations of BASELINE.json (C1, C3, C4, C5) and the PCIe-inclusive rate of
C2, on one MI355X.  `bench.py` stays the headline (C2).  One JSON object per line on stdout; run as

    python benchmarks/configs.py [--only c1,c2pcie,c3,c4,c5] > profiles/<round>/configs.jsonl

Every figure is: inputs resident in HBM unless the line says otherwise, HIP events around the launches on torch's
current stream, the CPU oracle (oracle/, 1 core) on a bounded sample of the same workload next to it."""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

PEAK_HBM = 8000.0
PEAK_F64 = 78.6


def gpu_time(fn, reps=5, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return float(np.median(ts))


def golden(name):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))


def controller(g):
    from pympc_amd.control.controllers import ModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    return ModelPredictiveController(LinearSystem(g['S_A'], g['S_B']), int(g['N']), g['Q'], g['R'], g['P'],
                                     Polyhedron(g['D_A'], g['D_b']), Polyhedron(g['XN_A'], g['XN_b']))


def c1():
    from oracle import oracle as O
    g = golden("c1_model")
    ctrl = controller(g)
    X = np.random.RandomState(0).uniform([-5., -1.], [5., 1.], (1000, 2))
    t0 = time.perf_counter(); u = ctrl.feedback(X); t_wall = time.perf_counter() - t0      # first call: includes prepare
    reps = []
    for _ in range(5):
        t0 = time.perf_counter(); u = ctrl.feedback(X); reps.append(time.perf_counter() - t0)
    Xd = torch.as_tensor(X).cuda()
    t_dev = gpu_time(lambda: ctrl.feedback_batch(Xd))
    t0 = time.perf_counter(); ref = O.mpqp_solve_batch(g, X, want_mult=False); t_cpu = time.perf_counter() - t0
    feas = ref["status"] == 0
    ok = bool(np.array_equal(~np.isnan(u[:, 0]), feas) and np.allclose(u[feas, 0], ref["argmin"][feas, 0], rtol=1e-6, atol=1e-6))
    return dict(config="C1", workload="double integrator N=10, ModelPredictiveController.feedback over 1000 states (host arrays in, host arrays out)",
                value=1000 / float(np.median(reps)), unit="QP solves/s", ms_host_to_host=float(np.median(reps)) * 1e3,
                ms_first_call=t_wall * 1e3, ms_device_resident=t_dev * 1e3, feasible=int(feas.sum()), parity_vs_oracle=ok,
                cpu_baseline=dict(value=1000 / t_cpu, unit="QP solves/s", cores=1, kind="port", sample="all 1000 states"),
                reference_python_1core="648 solves/s (BASELINE.md section 2, measured in the build container)")


def c2pcie():
    """Host-to-host rate of C2 through pympc_amd.pipeline.FeedbackPipeline (the figure bench.py reports as pcie_inclusive)."""
    from pympc_amd.pipeline import FeedbackPipeline
    g = golden("c2_model")
    ctrl = controller(g)
    prep = ctrl.mpqp.prepared()
    prep.seed_cuts()
    B = 1000000
    x_max = g["x_max"]
    Xh = torch.as_tensor(np.random.RandomState(0).uniform(-.3 * x_max, .3 * x_max, (B, 4))).pin_memory()
    out = {}
    for ncs, depth in ((1, 2), (2, 3)):
        pipe = FeedbackPipeline(prep, B, 1, depth=depth, compute_streams=ncs)
        for _ in range(40):
            pipe.submit(Xh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            k = pipe.submit(Xh)
        torch.cuda.synchronize()
        out[ncs] = (time.perf_counter() - t0) / 20
        pipe.wait(k)
    Xd = Xh.cuda()
    t_res = gpu_time(lambda: (prep.reset_cuts(), ctrl.feedback_batch(Xd)), reps=5, warmup=1)
    return dict(config="C2-pcie", workload="C2 (cart-pole N=20, 1M states, s=0.3): pinned host states -> HBM, solve, (u0, status) -> pinned host; "
                                           "transfers overlapped with the solves of neighbouring batches (pympc_amd/pipeline.py)",
                value=B / out[2], unit="QP solves/s", ms_two_compute_streams=out[2] * 1e3, ms_one_compute_stream=out[1] * 1e3,
                ms_resident=t_res * 1e3, resident_value=B / t_res, note="PCIe-inclusive rate; never the headline value")


def c2setup():
    """One-time controller set-up of C2 (outside the headline metric): DARE, mcais, condensing, factorisation."""
    from pympc_amd.control.controllers import ModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import LinearSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    g = golden("c2_model")
    S = LinearSystem(g['S_A'], g['S_B'])
    D = Polyhedron(g['D_A'].copy(), g['D_b'].copy())
    ts = {}
    for rep in range(2):                                   # second pass = warm
        t0 = time.perf_counter(); P, K = S.solve_dare(g['Q'], g['R']); ts['dare'] = time.perf_counter() - t0
        t0 = time.perf_counter(); X_N = S.mcais(K, D); ts['mcais'] = time.perf_counter() - t0
        t0 = time.perf_counter(); ctrl = ModelPredictiveController(S, int(g['N']), g['Q'], g['R'], P, D, X_N); ts['condense'] = time.perf_counter() - t0
        t0 = time.perf_counter(); ctrl.mpqp.prepared(); torch.cuda.synchronize(); ts['factorise'] = time.perf_counter() - t0
    return dict(config="C2-setup", workload="cart-pole N=20 controller set-up: solve_dare, mcais (batched LPs per iteration + redundancy removal), condensing (MFMA kernel), factorisation",
                value=sum(ts.values()) * 1e3, unit="ms (warm, host to host)", ms_dare=ts['dare'] * 1e3, ms_mcais=ts['mcais'] * 1e3,
                ms_condense=ts['condense'] * 1e3, ms_factorise=ts['factorise'] * 1e3, terminal_set_facets=int(X_N.A.shape[0]),
                reference_python_1core="mcais alone: 51 s (BASELINE.md section 2)")


def c3():
    from oracle import oracle as O
    from pympc_amd.geometry.polyhedron import Polyhedron
    from pympc_amd.optimization.solvers import hip_lp
    out = []
    for m in (100, 400, 1000, 10000):
        rng = np.random.RandomState(0)
        A = rng.randn(m, 6); A /= np.linalg.norm(A, axis=1)[:, None]
        b = 1. + rng.rand(m)
        Ad = torch.as_tensor(A).cuda(); bd = torch.as_tensor(b).cuda(); fd = (-Ad).contiguous()
        rx = torch.arange(m, dtype=torch.int32, device='cuda')
        res = hip_lp.linear_program_batch(fd, Ad, bd, relax_idx=rx, relax_val=1., want_argmin=False, want_mult=False,
                                          want_active=False, want_iters=True)
        it = res.numpy('iters')
        t_batch = gpu_time(lambda: hip_lp.linear_program_batch(fd, Ad, bd, relax_idx=rx, relax_val=1., want_argmin=False,
                                                               want_mult=False, want_active=False))
        Polyhedron(A.copy(), b.copy()).minimal_facets(); Polyhedron(A.copy(), b.copy()).radius      # warm-up (first launch of a shape)
        t0 = time.perf_counter(); P = Polyhedron(A.copy(), b.copy()); mf = P.minimal_facets(); t_mf = time.perf_counter() - t0
        t0 = time.perf_counter(); r = P.radius; t_ch = time.perf_counter() - t0
        piv = it[:, 2].astype(float) + it[:, 0]
        flops = float((piv * (4. * m * 6 + 4. * m) + piv ** 3 / 3.).sum())
        sweep_bytes = float((piv * m * 7 * 8).sum())          # every pivot sweeps At (n rows) and b from L2
        line = dict(config="C3", workload="random 6-D polytope, %d facets: Polyhedron.minimal_facets (one LP per facet, batched) + chebyshev" % m,
                    m=m, value=m / t_batch, unit="LP solves/s (step-1 batch, all facets vs the full relaxed set)",
                    ms_step1_batch=t_batch * 1e3, ms_minimal_facets_host_to_host=t_mf * 1e3, ms_chebyshev=t_ch * 1e3,
                    minimal_facets=len(mf), radius=r, mean_pivots=float(piv.mean()), max_pivots=int(piv.max()),
                    roofline=dict(bound="l2-sweep", achieved=sweep_bytes / t_batch / 1e9, unit="GB/s (L2 -> CU sweep traffic of At)",
                                  algorithmic_gflops=flops / t_batch / 1e9,
                                  hbm_algorithmic_bytes=8 * m * 7 + m * (8 + 4), note="the shared matrix is read from HBM once; pivots re-read it from L2"))
        if m <= 400:
            nlp = 20 if m == 100 else 3
            t0 = time.perf_counter()
            for i in range(nlp):
                rel = b.copy(); rel[i] += 1.
                O.linear_program(-A[i], A, rel)
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = dict(value=nlp / dt, unit="LP solves/s", cores=1, kind="port",
                                        sample="%d of the %d step-1 LPs (dense (1+m+n) x 2m PNNLS as pnnls.py:109-124)" % (nlp, m))
        else:
            line["cpu_baseline"] = dict(value=None, note="the reference formulation needs an O(m^2) projector: 1.6 GB and hours per LP at m = 10 000 (SURVEY section 6)")
        out.append(line)
    return out


def c4():
    from oracle import oracle as O
    ctrl = controller(golden("nb_model"))
    t0 = time.perf_counter(); ctrl.store_explicit_solution(); t_build = time.perf_counter() - t0
    es = ctrl.explicit_solution
    n = 3163
    g = torch.linspace(-.5, .5, n, dtype=torch.float64, device='cuda')
    X = torch.stack(torch.meshgrid(g, g, indexing='ij'), dim=-1).reshape(-1, 2).contiguous()
    B = X.shape[0]
    torch.cuda.synchronize(); t0 = time.perf_counter(); tab = es.table(1); torch.cuda.synchronize(); t_table = time.perf_counter() - t0
    idx, u, V = es.locate(X, nu=1)
    t = gpu_time(lambda: es.locate(X, nu=1))
    t_scan = gpu_time(lambda: es.locate(X, nu=1, use_sure=False))
    by = B * (8 * 2 + 4 + 8 + 8)
    # which points / wavefronts the decided cells settle (a wavefront = 64 consecutive points of the list)
    G, geom = tab['grid']['n'], torch.as_tensor(tab['grid']['geom'], device='cuda')
    cell = ((X - geom[:2]) * geom[2:4]).floor().clamp_(0, G - 1).long()
    dec = tab['grid']['sure'][cell[:, 0] * G + cell[:, 1]] != -2
    dec |= ((X < geom[6:8]) | (X > geom[4:6])).any(dim=1)
    wave_dec = torch.nn.functional.pad(dec, (0, (-B) % 64), value=True).reshape(-1, 64).all(dim=1)
    sample = X[::50].cpu().numpy()
    t0 = time.perf_counter(); O.point_location({k: tab[k] for k in ('A', 'b', 'row_ptr', 'ux', 'u0', 'Vxx', 'Vx', 'V0')}, sample)
    t_cpu = time.perf_counter() - t0
    # rows scanned per point (for the flop side): region index r means regions 0..r were visited
    rp = tab['row_ptr']
    return dict(config="C4", workload="explicit MPC look-up, inverted pendulum N=6 (71 critical regions, %d rows), 3163^2 = %d grid points: (region, u0, V)" % (int(rp[-1]), B),
                value=B / t, unit="points/s", ms=t * 1e3, covered_fraction=float((idx >= 0).double().mean()),
                ms_table_construction_facet_flipping=t_build * 1e3, ms_device_table_boxes_and_cell_grid=t_table * 1e3,
                ms_without_decided_cells=t_scan * 1e3, cells_per_axis=G, cells_decided=tab['grid']['decided_fraction'], cells_with_decision_list=tab['grid']['listed_fraction'],
                scan_pass_needed=tab['grid']['needs_scan'],
                points_decided_by_their_cell=float(dec.double().mean()), wavefronts_without_scan=float(wave_dec.double().mean()),
                roofline=dict(bound="hbm", achieved=by / t / 1e9, peak=PEAK_HBM, unit="GB/s", frac=by / t / 1e9 / PEAK_HBM,
                              algorithmic_bytes_per_launch=by, note="36 B per point; the table is read through the scalar cache and L2"),
                cpu_baseline=dict(value=len(sample) / t_cpu, unit="points/s", cores=1, kind="port", sample="every 50th grid point (%d)" % len(sample)))


def _c5_roofline(n, m, p, B, in_bytes, it, t):
    """qp_prepare_kernel + qp_solve_kernel<4, false> of one pympc_mpqp_multi_solve call.  HBM side: every sequence's
    matrices are read once (what the condensing kernel wrote), its factorised workspace is written once and read once by
    the solve (Mt, m0, Gm, cn and the small tables; T is skipped for a costs-only call), outputs are 12 B.  Flop side:
    Cholesky / inverses / M = A L^-T of the prepare (n^3 + 2 m n^2 + 2 m n p) plus the SURVEY 8(d) active-set count on
    the kernel's iteration counters."""
    sys.path.insert(0, ROOT)
    from bench import algorithmic_flops
    mp = 64 * next(b for b in (2, 4, 6, 8, 12, 16, 24, 32) if (m + 63) // 64 <= b)
    ws = 8 * ((n + p + 2) * mp + n + n * p + p * p + p + 1)
    bytes_ = B * (in_bytes + 2 * ws + 12)
    flops = B * (n ** 3 + 2. * m * n * n + 2. * m * n * p) + float(algorithmic_flops(n, m, p, it[:, 0], it[:, 1]).sum())
    return dict(kernels="qp_prepare_kernel + qp_solve_kernel<4, false>", bound="hbm", achieved=bytes_ / t / 1e9, peak=PEAK_HBM, unit="GB/s",
                frac=bytes_ / t / 1e9 / PEAK_HBM, algorithmic_bytes_per_launch=bytes_, algorithmic_tflops=flops / t / 1e12,
                frac_fp64_valu=flops / t / 1e12 / PEAK_F64,
                note="time = the whole call (both kernels; rocprofv3 split in profiles/r03/kernel_stats_configs.csv)")


def c5():
    from oracle import oracle as O
    from pympc_amd.control.controllers import HybridModelPredictiveController
    from pympc_amd.dynamics.discrete_time_systems import AffineSystem, PieceWiseAffineSystem
    from pympc_amd.geometry.polyhedron import Polyhedron
    h = golden("hybrid_seqs")
    systems = [AffineSystem(h['S_A'][i], h['S_B'][i], h['S_c'][i]) for i in range(3)]
    domains = [Polyhedron(h['D%d_A' % i], h['D%d_b' % i]) for i in range(3)]
    N = 10
    ctrl = HybridModelPredictiveController(PieceWiseAffineSystem(systems, domains), N, h['Q'], h['R'], h['P'],
                                           Polyhedron(h['XN_A'], h['XN_b']))
    seqs = torch.as_tensor(np.array(list(itertools.product(range(3), repeat=N)), dtype=np.int32)).cuda()
    B = seqs.shape[0]
    x0 = torch.as_tensor(h['x0']).cuda()
    batch = ctrl.get_mpqp_batch(seqs)
    t_cond = gpu_time(lambda: ctrl.get_mpqp_batch(seqs), reps=3)
    res = batch.solve(x0, want_mult=False, want_active=False, want_iters=True)
    t_solve = gpu_time(lambda: batch.solve(x0, want_mult=False, want_active=False, want_argmin=False), reps=3)   # costs only: arg-min over sequences
    st = res.numpy('status'); V = res.numpy('min')
    feas = st == 0
    kbest = int(np.nanargmin(np.where(feas, V, np.nan)))
    n, p, m = batch.n, batch.p, batch.m
    out_bytes = 8 * (n * n + n * p + p * p + n + p + 1 + m * (n + p + 1))
    # CPU: oracle condense + QP on a sample of sequences
    sysd = [(h['S_A'][i], h['S_B'][i], h['S_c'][i]) for i in range(3)]; domd = [(h['D%d_A' % i], h['D%d_b' % i]) for i in range(3)]
    idx = np.random.RandomState(0).choice(B, 200, replace=False)
    sq = seqs.cpu().numpy()
    t0 = time.perf_counter()
    agree = 0
    for k in idx:
        o = O.condense(sysd, domd, h['Q'], h['R'], h['P'], h['XN_A'], h['XN_b'], sq[k])
        sol = O.quadratic_program(o['Huu'], o['Hux'].dot(h['x0']) + o['fu'], o['Au'], o['b'] - o['Ax'].dot(h['x0']))
        ref_feas = sol['min'] is not None
        agree += int(ref_feas == bool(feas[k]) and (not ref_feas or abs(sol['min'] + .5 * h['x0'].dot(o['Hxx']).dot(h['x0']) + o['fx'].dot(h['x0']) + o['g'] - V[k]) < 1e-6 * max(1., abs(V[k]))))
    t_cpu = time.perf_counter() - t0
    it = res.numpy('iters')
    return dict(config="C5", workload="hybrid cart-pole (3 modes), N=10: all 3^10 = %d mode sequences condensed (fp64 MFMA) and solved at x0=[0,0,2.5,0] (n=%d, m=%d padded rows, p=%d)" % (B, n, m, p),
                value=B / (t_cond + t_solve), unit="mode-sequence QPs/s (condense + prepare + solve)", ms_condense=t_cond * 1e3,
                ms_prepare_and_solve=t_solve * 1e3, feasible=int(feas.sum()), best_sequence=[int(z) for z in sq[kbest]], best_cost=float(V[kbest]),
                mean_outer_iterations=float(it[:, 0].mean()), sample_parity_vs_oracle="%d/200" % agree,
                roofline=dict(kernel="condense_kernel<1, 1>", bound="hbm", achieved=B * out_bytes / t_cond / 1e9, peak=PEAK_HBM, unit="GB/s",
                              frac=B * out_bytes / t_cond / 1e9 / PEAK_HBM, algorithmic_bytes_per_launch=B * out_bytes),
                roofline_prepare_and_solve=_c5_roofline(n, m, p, B, out_bytes, it, t_solve),
                cpu_baseline=dict(value=200 / t_cpu, unit="mode-sequence QPs/s", cores=1, kind="port", sample="200 random sequences: oracle_condense + oracle_qp"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c1,c2pcie,c2setup,c3,c4,c5")
    only = ap.parse_args().only.split(",")
    assert torch.cuda.is_available()
    from oracle import oracle as O
    O.build()
    for name, fn in (("c1", c1), ("c2pcie", c2pcie), ("c2setup", c2setup), ("c3", c3), ("c4", c4), ("c5", c5)):
        if name in only:
            r = fn()
            for line in (r if isinstance(r, list) else [r]):
                print(json.dumps(line), flush=True)
