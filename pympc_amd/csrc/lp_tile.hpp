// LPs that share one constraint matrix and right-hand side (redundancy removal, polyhedron.py:435-448: one LP per facet
// against the same rows): eight LPs per workgroup, advanced in lockstep rounds.
//
// lp_block.hpp gives every LP a whole workgroup and sweeps the transposed matrix from L2 once per pivot: at 10 000 rows
// that is 0.56 MB per pivot and LP, and the kernel runs at the L2 bandwidth (13 TB/s).  Here wavefront g of a workgroup
// owns LP g (same dual active-set method, same update formulas: invqr_wave.hpp, tests/kernel_model.py).  Per round
// every LP needs exactly one sweep  acc_j = a_j . v_g  over all rows (v_g = residual direction in phase 1, current
// vertex in phase 2), so the 512 threads split the rows, load each row ONCE and form the eight dot products from
// registers: L2 traffic and load instructions per pivot drop eightfold.  Per-LP arg-max through LDS, then every owner
// wave does its pivot (its own Q / R^-1 in LDS) in parallel with the other seven.  A wave whose LP is finished takes
// the next one from the workgroup's ticket.
// Working rows are not masked during the sweep: in phase 2 they are tight (violation ~1e-16, below the candidate
// threshold); in phase 1 their dual is zero up to rounding, so a working row can only come out as the maximum when no
// real candidate is left, which the owner wave treats as such.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "invqr_wave.hpp"
#include "lp_block.hpp"
#include "ws_layout.hpp"

namespace pympc {

constexpr int kLpTileWaves = 8;
constexpr int kLpTileMinRows = 1024;    // below this lp_block.hpp (one small workgroup per LP) fills the chip better (8 442 LPs x 1 558 rows: 1.51 against 1.83 ms)
constexpr int kLpTileMinBatch = 1024;
constexpr int kLpTileCtl = 160;   // doubles ahead of the per-wave blocks: 128 control ints, 64 arg-max values, 64 indices

// per-wave LDS doubles: factorisation, three padded vectors, two bitmaps of m bits, rejected list, working-row list
__host__ __device__ inline size_t lp_tile_wave_doubles(int n, int m, int RP)
{
    const size_t nw = (size_t)(m + 31) / 32;
    return (size_t)invqr_lds_doubles(n, RP) + 3 * (size_t)(n + kVecPad) + nw + (kLpRejMax + 1) / 2 + (n + 1) / 2 + 2;
}
__host__ __device__ inline size_t lp_tile_lds_doubles(int n, int m, int RP)
{
    return kLpTileCtl + kLpTileWaves * lp_tile_wave_doubles(n, m, RP);
}

#ifndef PYMPC_LPT_WAVES_PER_SIMD
#define PYMPC_LPT_WAVES_PER_SIMD 2   // one workgroup of eight waves per CU
#endif

enum { kLpIdle = 0, kLpPhase1 = 1, kLpPhase2 = 2 };

// scratch/lpt_profile.hip builds this file with PYMPC_LPT_PROFILE to split a wave's cycles into rounds / sweep / arg-max
// / barrier wait / pivot; the library build has none of it.
#ifdef PYMPC_LPT_PROFILE
__device__ unsigned long long g_lpt_prof[8];
#define LPT_T(var) const long long var = __builtin_readcyclecounter()
#define LPT_ACC(slot, t1, t0) prof[slot] += (t1) - (t0)
#else
#define LPT_T(var)
#define LPT_ACC(slot, t1, t0)
#endif

// One sweep over all rows for the eight LPs of a workgroup: thread t scores rows t, t + T, ... against the eight
// directions (rb of each owner wave, read from LDS as broadcasts) and keeps per LP the best (score, row) it has seen.
// Straight-line code per step of JU rows: no branch on the LP's state (an idle LP's scores are never looked at), the
// rejected-candidate list of phase 1 (rare) only in the REJ instantiation.
template <int NT, bool REJ>
__device__ __forceinline__ void lp_tile_sweep(const double *__restrict__ At, const double *__restrict__ ainv,
                                              const double *__restrict__ bsh, int m, int T, int tid, const double *lds,
                                              int WD, int rb_off, int rej_off, const int (&gmode)[kLpTileWaves],
                                              const int (&grx)[kLpTileWaves], const int (&gnrej)[kLpTileWaves],
                                              double (&best)[kLpTileWaves], int (&bjs)[kLpTileWaves])
{
    constexpr int G = kLpTileWaves;
    constexpr int JU = 4;   // rows per thread whose loads are in flight together
    double vb[G];
    int xr[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bool p2 = gmode[g] == kLpPhase2;
        vb[g] = p2 ? 1.0 : 0.0;      // phase 2 subtracts b_j, phase 1 does not
        xr[g] = p2 ? grx[g] : -1;    // phase 2 leaves the relaxed row to the owner
    }
    for (int jb = tid; jb < m; jb += T * JU) {
        double a[JU][NT], ai[JU], nb[JU];
        int jc[JU];   // rows past the end repeat row m - 1: same score, same index
#pragma unroll
        for (int q = 0; q < JU; ++q) {
            jc[q] = min(jb + q * T, m - 1);
#pragma unroll
            for (int i = 0; i < NT; ++i) a[q][i] = At[(size_t)i * m + jc[q]];
            ai[q] = ainv[jc[q]];
            nb[q] = -lp_row_rhs(bsh[jc[q]]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // direction of LP g.  The offset is made opaque so that the 48 doubles of the eight LPs are re-read here
            // instead of being hoisted out of the row loop into registers.
            int off = kLpTileCtl + g * WD + rb_off;
            asm volatile("" : "+v"(off));
            const double *rg = lds + off;
            double vg[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) vg[i] = rg[i];
            double acc[JU];
#pragma unroll
            for (int q = 0; q < JU; ++q) acc[q] = a[q][0] * vg[0];
#pragma unroll
            for (int i = 1; i < NT; ++i)
#pragma unroll
                for (int q = 0; q < JU; ++q) acc[q] = fma(a[q][i], vg[i], acc[q]);
#pragma unroll
            for (int q = 0; q < JU; ++q) {
                double sc = fma(nb[q], vb[g], acc[q]) * ai[q];
                if (REJ) {   // phase 1: candidates the owner refused in this outer iteration score 0
                    const int *rj = reinterpret_cast<const int *>(lds + kLpTileCtl + (size_t)g * WD + rej_off);
                    for (int r = 0; r < gnrej[g]; ++r)
                        if (rj[r] == jc[q]) sc = 0.0;
                }
                const bool take = sc > best[g] && jc[q] != xr[g];
                best[g] = take ? sc : best[g];
                bjs[g] = take ? jc[q] : bjs[g];
            }
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(64 * kLpTileWaves, PYMPC_LPT_WAVES_PER_SIMD) lp_tile_kernel(LpParams P)
{
    extern __shared__ double lds[];
    constexpr int n = NT;
    constexpr int G = kLpTileWaves;
    const int m = P.m, RP = P.RP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = blockDim.x;
    const int nw = (m + 31) / 32;
    const int WD = (int)lp_tile_wave_doubles(n, m, RP);
    const int rb_off = invqr_lds_doubles(n, RP);
    const int rej_off = rb_off + 3 * (n + kVecPad) + nw;

    int *ctl = reinterpret_cast<int *>(lds);       // [0] ticket, [8 + g] mode, [16 + g] relaxed row, [24 + g] rejected count
    double *slot_v = lds + 64;                     // [G LPs][G waves]
    int *slot_i = reinterpret_cast<int *>(lds + 128);
    double *wbase = lds + kLpTileCtl + (size_t)wave * WD;
    InvQr F;
    invqr_bind(F, wbase, n, RP);
    double *rb = wbase + rb_off;
    double *tb = rb + n + kVecPad;
    double *yb = tb + n + kVecPad;
    uint32_t *inW = reinterpret_cast<uint32_t *>(yb + n + kVecPad);
    uint32_t *abits = inW + nw;
    int *rej = reinterpret_cast<int *>(wbase + rej_off);
    int *wlist = rej + kLpRejMax + (kLpRejMax & 1);

    const double *__restrict__ At = P.At;    // shared matrix (at_stride == 0)
    const double *__restrict__ ainv = At + (size_t)n * m;
    const double *__restrict__ bsh = P.b;    // shared right-hand side (b_stride == 0)
    const double rv = P.relax_val;
    const double nanv = __builtin_nan("");

    for (int e = tid; e < kLpTileCtl + G * WD; e += T) lds[e] = 0.0;
    __syncthreads();

    // LPs of this workgroup: blockIdx.x, blockIdx.x + gridDim.x, ...
    const long long mine = blockIdx.x < P.B ? (P.B - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;

    int mode = kLpIdle;
    long long prob = -1;
    int rx = -1, k = 0, status = kStatusOptimal, outer = 0, inner = 0, piv = 0, nrej = 0;
    bool capped = false;
    int Wl = -1;
    double yl = 0.0, fnorm = 1.0;
    double arx = 0.0, brx = 0.0, airx = 1.0;   // relaxed row: lane i < n holds its i-th entry; right-hand side; 1 / norm
    const int itmax = 3 * m;
    bool exhausted = false;
#ifdef PYMPC_LPT_PROFILE
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (;;) {
        LPT_T(t_top);
        // ------------------------------------------------------------ idle waves take the next LP
        if (mode == kLpIdle && !exhausted) {
            int t = 0;
            if (lane == 0) t = atomicAdd(&ctl[0], 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= mine) {
                exhausted = true;
            } else {
                prob = (long long)blockIdx.x + (long long)t * gridDim.x;
                rx = P.relax_idx ? P.relax_idx[prob] : -1;
                if (rx >= 0) {
                    arx = (lane < n) ? At[(size_t)lane * m + rx] : 0.0;
                    brx = bsh[rx];
                    airx = ainv[rx];
                }
                const double *f = P.f + prob * P.f_stride;
                for (int e = lane; e < rb_off + 3 * (n + kVecPad); e += 64) wbase[e] = 0.0;
                for (int e = lane; e < nw; e += 64) { inW[e] = 0u; abits[e] = 0u; }
                wave_sync_lds();
                if (lane < n) {
                    F.Q[lane * RP + lane] = 1.0;
                    F.cv[lane] = -f[lane];
                }
                wave_sync_lds();
                k = 0; status = kStatusOptimal; outer = 0; inner = 0; piv = 0; nrej = 0; capped = false; Wl = -1; yl = 0.0;
                const double fl = (lane < n) ? F.cv[lane] : 0.0;
                const double fn = sqrt(wsum(fl * fl));
                fnorm = fmax(1.0, fn);
                if (!(fn > 1e-13 * fnorm)) {
                    mode = kLpPhase2;   // f = 0: the dual system has residual zero; phase 2 starts from x = 0
                } else {
                    mode = kLpPhase1;
                    if (lane < n) rb[lane] = F.cv[lane];   // r = Qt'[0; cv[0:]] with Qt = I
                }
                wave_sync_lds();
            }
        }
        if (lane == 0) {
            ctl[8 + wave] = mode;
            ctl[16 + wave] = rx;
            ctl[24 + wave] = nrej;
        }
        LPT_T(t_b1);
        __syncthreads();
        LPT_T(t_sw);
        int gmode[G], grx[G], gnrej[G];   // the same in every thread: kept in scalar registers
        int any = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            gmode[g] = __builtin_amdgcn_readfirstlane(ctl[8 + g]);
            grx[g] = __builtin_amdgcn_readfirstlane(ctl[16 + g]);
            gnrej[g] = __builtin_amdgcn_readfirstlane(ctl[24 + g]);
            any |= gmode[g];
        }
        if (!any) break;

        // ------------------------------------------------------------ one sweep over all rows for the eight LPs
        // phase 1: score (a_j . r) / |a_j|; phase 2: (a_j . x - lp_row_rhs(b_j)) / |a_j|, the distance of x beyond row j;
        // candidates > 0.  The relaxed row of an LP is left out in phase 2 (its b differs: the owner wave scores it).
        double best[G];
        int bjs[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            best[g] = 0.0;
            bjs[g] = 0x7fffffff;
        }
        const bool anyrej = (gnrej[0] | gnrej[1] | gnrej[2] | gnrej[3] | gnrej[4] | gnrej[5] | gnrej[6] | gnrej[7]) != 0;
        if (__builtin_expect(anyrej, 0))
            lp_tile_sweep<NT, true>(At, ainv, bsh, m, T, tid, lds, WD, rb_off, rej_off, gmode, grx, gnrej, best, bjs);
        else
            lp_tile_sweep<NT, false>(At, ainv, bsh, m, T, tid, lds, WD, rb_off, rej_off, gmode, grx, gnrej, best, bjs);
        LPT_T(t_am);
        // per-LP arg-max (largest score, lowest row among equals): wave level, then one slot per (LP, wave)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (gmode[g] == kLpIdle) continue;
            const double gm = wmax(best[g]);
            const int gi = wmin_int(best[g] == gm ? bjs[g] : 0x7fffffff);
            if (lane == 0) { slot_v[g * G + wave] = gm; slot_i[g * G + wave] = gi; }
        }
        LPT_T(t_b2);
        __syncthreads();
        LPT_T(t_pv);
#ifdef PYMPC_LPT_PROFILE
        prof[0] += 1;
        LPT_ACC(1, t_b1, t_top);   // ticket / set-up of a new LP
        LPT_ACC(2, t_sw, t_b1);    // wait at the first barrier
        LPT_ACC(3, t_am, t_sw);    // sweep
        LPT_ACC(4, t_b2, t_am);    // arg-max
        LPT_ACC(5, t_pv, t_b2);    // wait at the second barrier
        if (mode != kLpIdle) prof[7] += 1;
#endif
        if (mode == kLpIdle) continue;
        double best_ = slot_v[wave * G];
        int bj = slot_i[wave * G];
#pragma unroll
        for (int w = 1; w < G; ++w) {
            const double vv = slot_v[wave * G + w];
            const int ii = slot_i[wave * G + w];
            if (vv > best_ || (vv == best_ && ii < bj)) { best_ = vv; bj = ii; }
        }
        // the relaxed row in phase 2, with its own right-hand side (same arithmetic as the sweep)
        if (mode == kLpPhase2 && rx >= 0) {
            double acc = lane_f64(arx, 0) * rb[0];
#pragma unroll
            for (int i = 1; i < NT; ++i) acc = fma(lane_f64(arx, i), rb[i], acc);
            acc -= lp_row_rhs(brx + rv);
            const double sc = acc * airx;
            if (sc > 0.0 && (sc > best_ || (sc == best_ && rx < bj))) { best_ = sc; bj = rx; }
        }
        bool found = best_ > 0.0;
        // a working row as the maximum of phase 1: its dual is rounding noise, no real candidate is left
        if (mode == kLpPhase1 && found && __ballot(lane < k && Wl == bj) != 0ull) found = false;

        bool finish = false;
        if (mode == kLpPhase1) {
            bool p1_end = false;
            if (!found) {
                p1_end = true;
            } else {
                const int j = bj;
                const double gl = (lane < n) ? At[(size_t)lane * m + j] : 0.0;
                const QrTrial tr = qr_trial(F, k, gl);
                const bool ok = tr.independent && (tr.ck / tr.diag) > 0.0;
                if (ok) {
                    qr_add(F, k, tr);
                    if (lane == k) { Wl = j; yl = 0.0; }
                    if (lane == 0) inW[j >> 5] |= 1u << (j & 31);
                    ++k;
                    ++outer;
                    nrej = 0;
                    for (;;) {   // secondary loop of Lawson-Hanson
                        const double zl = qr_coeff(F, k);
                        ++inner;
                        if (inner > itmax) { status = kStatusMaxIter; break; }
                        const bool bad = (lane < k) && (zl <= 0.0);
                        if (__ballot(bad) == 0ull) {
                            if (lane < k) yl = zl;
                            break;
                        }
                        const double tt = bad ? yl / (yl - zl) : 1.0e300;
                        const double alpha = wmin(tt);
                        int jj = wmin_int((bad && tt == alpha) ? lane : 64);
                        if (lane < k) yl = fma(alpha, zl - yl, yl);
                        for (;;) {
                            const int jdel = __shfl(Wl, jj);
                            if (lane == 0) inW[jdel >> 5] &= ~(1u << (jdel & 31));
                            qr_drop(F, k, jj);
                            const int Wn = __shfl_down(Wl, 1);
                            const double yn = __shfl_down(yl, 1);
                            if (lane >= jj && lane < k - 1) { Wl = Wn; yl = yn; }
                            --k;
                            const unsigned long long neg = __ballot(lane < k && yl <= 0.0);
                            if (neg == 0ull) break;
                            jj = __ffsll((long long)neg) - 1;
                        }
                    }
                    wave_sync_lds();
                    if (status != kStatusOptimal) {
                        finish = true;
                    } else {
                        const double cl = (lane >= k && lane < n) ? F.cv[lane] : 0.0;
                        const double rn = sqrt(wsum(cl * cl));
                        if (k >= n || k >= m || !(rn > 1e-13 * fnorm)) p1_end = true;
                        else if (lane < n) rb[lane] = dotu(F.Q + lane * RP + k, F.cv + k, n - k);   // r = Qt'[0; cv[k:]]
                    }
                } else {
                    if (nrej < kLpRejMax) {
                        if (lane == 0) rej[nrej] = j;
                        ++nrej;
                    } else {
                        p1_end = true;
                        capped = true;   // candidates with a positive dual remain untried
                    }
                }
            }
            if (p1_end && !finish) {
                const double cl = (lane >= k && lane < n) ? F.cv[lane] : 0.0;
                const double rn = sqrt(wsum(cl * cl));
                if (!(rn < P.tol)) {
                    // no dual feasible point: LP unbounded or infeasible -- unless phase 1 stopped at its cap of rejected
                    // candidates, then nothing is proven and the host raises (iteration limit) instead of answering None
                    status = capped ? kStatusMaxIter : kStatusInfeasible;
                    finish = true;
                } else {
                    mode = kLpPhase2;
                }
            }
        } else {
            if (!found) {
                finish = true;   // primal feasible: (x, y) optimal
            } else {
                const int e = bj;
                ++piv;
                if (piv > P.maxit) status = kStatusMaxIter;
                const double gl = (lane < n) ? At[(size_t)lane * m + e] : 0.0;
                QrTrial tr = qr_trial(F, k, gl);
                const bool indep = k < n && tr.independent && fabs(tr.diag) > 1e-10 * (tr.unorm + fabs(tr.diag));
                if (status == kStatusOptimal && !indep) {
                    if (lane < n) F.ab[lane] = tr.qi;
                    wave_sync_lds();
                    double dl = 0.0;
                    if (lane < k) dl = dotu(F.Wi + lane * RP, F.ab, k);
                    const double zl = fmax(qr_coeff(F, k), 0.0);
                    const double dmax = wmax(lane < k ? fabs(dl) : 0.0);
                    const bool cand = (lane < k) && (dl > 1e-12 * fmax(1.0, dmax));
                    const double ratio = cand ? zl / dl : 1.0e300;
                    const double rmin = wmin(ratio);
                    const int jj = wmin_int((cand && ratio == rmin) ? lane : 64);
                    wave_sync_lds();
                    if (jj >= 64) {
                        status = kStatusInfeasible;   // Farkas: a_e is a non-positive combination of tight rows
                    } else {
                        const int jdel = __shfl(Wl, jj);
                        if (lane == 0) inW[jdel >> 5] &= ~(1u << (jdel & 31));
                        qr_drop(F, k, jj);
                        const int Wn = __shfl_down(Wl, 1);
                        if (lane >= jj && lane < k - 1) Wl = Wn;
                        --k;
                        tr = qr_trial(F, k, gl);
                        if (!tr.independent) status = kStatusMaxIter;
                    }
                }
                if (status == kStatusOptimal) {
                    qr_add(F, k, tr);
                    if (lane == k) Wl = e;
                    if (lane == 0) inW[e >> 5] |= 1u << (e & 31);
                    ++k;
                } else {
                    finish = true;
                }
            }
        }
        wave_sync_lds();
        if (!finish && mode == kLpPhase2) {
            // x = Qt[:k]' (Wi' b_W): the vertex of the working rows (minimum-norm point for k < n)
            double bw = 0.0;
            if (lane < k) bw = bsh[Wl] + (Wl == rx ? rv : 0.0);
            if (lane < n) F.ab[lane] = bw;
            wave_sync_lds();
            if (lane < n) tb[lane] = (lane < k) ? dots(F.Wi + lane, RP, F.ab, k) : 0.0;
            wave_sync_lds();
            if (lane < n) rb[lane] = dotu(F.Q + lane * RP, tb, k);
            wave_sync_lds();
        }
        if (finish) {
            // -------------------------------------------------------- outputs (pnnls.py:137-145)
            const bool ok = (status == kStatusOptimal);
            const double *f = P.f + prob * P.f_stride;
            const double zl = ok ? fmax(qr_coeff(F, k), 0.0) : 0.0;
            if (lane < k) { yb[lane] = zl; wlist[lane] = Wl; }
            if (P.argmin && lane < n) P.argmin[prob * n + lane] = ok ? rb[lane] : nanv;
            if (P.vmin) {
                const double v = wsum((ok && lane < n) ? f[lane] * rb[lane] : 0.0);
                if (lane == 0) P.vmin[prob] = ok ? v : nanv;
            }
            if (lane == 0) {
                P.status[prob] = status;
                if (P.iters) {
                    P.iters[3 * prob] = outer;
                    P.iters[3 * prob + 1] = inner;
                    P.iters[3 * prob + 2] = piv;
                }
            }
            if (ok && lane < k && zl > P.tol) atomicOr(&abits[Wl >> 5], 1u << (Wl & 31));
            wave_sync_lds();
            if (P.mult) {
                for (int j = lane; j < m; j += 64) {
                    double v = ok ? 0.0 : nanv;
                    if (ok && ((inW[j >> 5] >> (j & 31)) & 1u))
                        for (int l = 0; l < k; ++l)
                            if (wlist[l] == j) v = yb[l];
                    P.mult[prob * m + j] = v;
                }
            }
            if (P.active)
                for (int e = lane; e < nw; e += 64) P.active[prob * nw + e] = abits[e];
            wave_sync_lds();
            mode = kLpIdle;
        }
#ifdef PYMPC_LPT_PROFILE
        {
            LPT_T(t_end);
            LPT_ACC(6, t_end, t_pv);   // pivot + outputs
        }
#endif
    }
#ifdef PYMPC_LPT_PROFILE
    if (lane == 0)
        for (int e = 0; e < 8; ++e) atomicAdd(&g_lpt_prof[e], (unsigned long long)prof[e]);
#endif
}

}  // namespace pympc
