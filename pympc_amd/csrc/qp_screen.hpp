// Screening kernel of the shared-matrix batch (MultiParametricQuadraticProgram.solve over a batch of states,
// /root/reference/pympc/optimization/parametric_programs.py:155-179 -> pympc/optimization/solvers/pnnls.py:149-241):
// one state per LANE, ahead of the wave-per-problem kernel (qp_wave.hpp).  Most states of an MPC batch end with none,
// one or two active constraints; for those the NNLS of pnnls.py:216-230 is a handful of scalar formulas plus sweeps
// over the columns, which a lane does for its own state with every lane of the wavefront busy.
//
//   stage 0  (every state)   m(x) = m0 + Gm'x.  No negative entry: the NNLS stops before its first column (y = 0),
//            the state is finished with the unconstrained minimiser.  A stored infeasibility certificate fires: finished
//            as infeasible.  Otherwise the column the NNLS takes first (largest w^2 / |a|^2 over w = -m(x) > 0, lowest
//            index among equals: the rule of qp_wave_solve) goes into queue 1 with the state's index.
//   stage K  (K = 1 .. kScreenStages, entries of queue K: state + passive columns c_1 .. c_K)
//            NNLS solution on the passive set from the Gram matrix of the constant rows:  with G = Gr[P][P] = M_P M_P',
//            z = G^-1 m_P,  t = m_P'z:   y = -z / (1 + t),   residual^2 = gamma = 1 / (1 + t)   (Sherman-Morrison on
//            G + m_P m_P'; every term of t is positive: no cancellation).  Next dual vector
//            w_j = -gamma m_j - sum_l y_l Gr[c_l][j]:  no positive entry => optimal, the state is finished here
//            (lambda = -z, V = V_unc + t / 2, u = u_unc - sum_l lambda_l T[c_l]);  otherwise the best next column
//            joins the passive set in queue K + 1.  A state leaves the lane path for the wave kernel (status
//            kStatusPending, solved there from scratch) when a coefficient is not positive (Lawson-Hanson would drop a
//            column), a pivot of G is small (nearly dependent rows), a coefficient is large or the residual is small
//            (near-infeasible), or after the last stage.
//
// MI355X mapping.  The constants of a sweep (m0, Gm, cn: the same for every lane) come through the scalar cache and
// enter the FMAs as scalar operands; that path delivers only a few bytes per clock per CU, so stage 0 keeps S states
// per lane (each loaded constant feeds S FMAs).  The Gram entries of a lane's own passive columns are gathered
// (Gr is symmetric: lanes read column c_l of the row the sweep is at, so a wavefront's loads fall into one 3 KB row).
// Candidates are packed: a persistent workgroup pushes them into LDS queues while it walks over its tiles of states
// and runs a stage only on full wavefronts' worth of entries (the remainders are flushed once at the end), so that
// every lane of every stage pass works.  Outputs of a finished state are written by its lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "qp_wave.hpp"

namespace pympc {

#ifndef PYMPC_SCREEN_THREADS
#define PYMPC_SCREEN_THREADS 512
#endif
#ifndef PYMPC_SCREEN_WG_PER_CU
#define PYMPC_SCREEN_WG_PER_CU 2
#endif
constexpr int kScreenThreads = PYMPC_SCREEN_THREADS;
constexpr int kScreenWgPerCu = PYMPC_SCREEN_WG_PER_CU;   // persistent workgroups per CU (x kScreenThreads / 256 = waves per SIMD)
#ifndef PYMPC_SCREEN_U0
#define PYMPC_SCREEN_U0 2
#endif
#ifndef PYMPC_SCREEN_U
#define PYMPC_SCREEN_U 4
#endif
#ifndef PYMPC_SCREEN_KEY0
#define PYMPC_SCREEN_KEY0 0
#endif
constexpr int kScreenU0 = PYMPC_SCREEN_U0;
constexpr int kScreenStages = 2;          // lane-path stages after stage 0: states with up to this many active constraints
constexpr double kScreenPivotMin = 1e-6;  // relative pivot of G below which the rows count as nearly dependent
constexpr double kScreenGammaMin = 1e-8;  // squared residual below which the state counts as near-infeasible
template <int PD>
constexpr int screen_states_per_lane() { return PD <= 4 ? 2 : 1; }

// x'(.5 Sxx x + sx) with the p terms added as the solve kernel's butterfly adds them:
// ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7))
template <int PD>
__device__ __forceinline__ double vunc_terms(cdouble_ptr tab, const WsLayout &L, const double (&x)[PD])
{
    double t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        t[q] = 0.0;
        if (q < PD) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < PD; ++c) s = fma(tab[L.Sxx + (size_t)q * PD + c], x[c], s);
            t[q] = x[q] * fma(0.5, s, tab[L.sx + q]);
        }
    }
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

// Outputs of a state finished on the lane path.  K = 0: no active constraint (okk: optimal, else infeasible by a
// certificate).  K >= 1: passive columns cols[] with multipliers lam[] (all positive), tq = t of the header.
template <int PD, int K>
__device__ __forceinline__ void screen_write(const SolveParams &P, cdouble_ptr tab, long long prob, const double (&x)[PD],
                                             bool okk, const int *cols, const double *lam, double tq)
{
    const WsLayout &L = P.L;
    const int n = P.n, m = P.m, nw = (m + 31) / 32;
    const double nanv = __builtin_nan("");
    P.status[prob] = okk ? kStatusOptimal : kStatusInfeasible;
    if (P.iters) { P.iters[2 * prob] = K; P.iters[2 * prob + 1] = K; }
    if (P.argmin)
        for (int i = 0; i < P.n_out; ++i) {
            double ui = tab[L.u0 + i];
#pragma unroll
            for (int q = 0; q < PD; ++q) ui = fma(tab[L.Ku + (size_t)i * PD + q], x[q], ui);
#pragma unroll
            for (int l = 0; l < K; ++l) ui = fma(-lam[l], P.ws[L.T + (size_t)cols[l] * n + i], ui);
            P.argmin[prob * P.n_out + i] = okk ? ui : nanv;
        }
    if (P.vmin) {
        const double vq = vunc_terms<PD>(tab, L, x);
        P.vmin[prob] = okk ? (tab[L.s0] + vq) + 0.5 * tq : nanv;
    }
    if (P.active)
        for (int e = 0; e < nw; ++e) {
            uint32_t word = 0u;
#pragma unroll
            for (int l = 0; l < K; ++l)
                if (e == (cols[l] >> 5) && lam[l] > P.tol) word |= 1u << (cols[l] & 31);
            P.active[prob * nw + e] = word;
        }
    if (P.mult)
        for (int j = 0; j < m; ++j) {
            double v = okk ? 0.0 : nanv;
#pragma unroll
            for (int l = 0; l < K; ++l)
                if (j == cols[l]) v = lam[l];
            P.mult[prob * m + j] = v;
        }
}

enum { kScreenDone = 0, kScreenNext = 1, kScreenPunt = 2 };

// Stage K for the state of this lane (act: the lane holds an entry).  Returns kScreenDone (outputs written),
// kScreenNext (jnext = column to add) or kScreenPunt (status set to pending).  Full wavefront: the sweep is wave-uniform.
// LAST: no stage follows (only "optimal here or not" is decided); U: columns per step of the sweep (their loads in flight
// together: 8 while the passive set is small, fewer as K gathered entries per column fill the registers).
// The diagonal of the Gram matrix holds kGramDiag instead of |M_j|^2 (qp_gram_kernel), so that the dual of a passive
// column comes out hugely negative by itself: no per-column test for "is this one of mine".
template <int PD, int K, bool LAST, int U>
__device__ __forceinline__ int screen_stage(const SolveParams &P, cdouble_ptr m0, cdouble_ptr Gm, cdouble_ptr cn, cdouble_ptr tab,
                                            bool act, long long prob, const int (&cols)[K], int &jnext)
{
    const WsLayout &L = P.L;
    const int m = P.m, mp = L.mp;
    double x[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) x[q] = act ? P.X[prob * P.x_stride + q] : 0.0;
    // ---- passive rows: m_P(x), G = Gr[P][P] (diagonal = cn), its LDL' factorisation, z = G^-1 m_P
    double mP[K], D[K], Lm[K][K], v[K], z[K], y[K], lam[K];
    bool ok = act;
#pragma unroll
    for (int l = 0; l < K; ++l) {
        const int c = act ? cols[l] : 0;
        double s = P.ws[L.m0 + c];   // (per-lane index: vector loads)
#pragma unroll
        for (int q = 0; q < PD; ++q) s = fma(P.ws[L.Gm + (size_t)q * mp + c], x[q], s);
        mP[l] = s;
    }
    double tq = 0.0, gamma = 1.0;
    if constexpr (K == 1) {
        // the closed form of qp_wave_solve's first column
        const double cj = P.ws[L.cn + (act ? cols[0] : 0)];
        const double dj = fma(mP[0], mP[0], cj);
        y[0] = fast_div(-mP[0], dj);
        gamma = fast_div(cj, dj);
        ok = ok && (y[0] < kGramMaxCoeff) && (fast_sqrt(gamma) > P.tol);
        const double my = mP[0] * y[0];
        lam[0] = y[0] * fast_div(1.0, 1.0 + my);
        tq = -my * fast_div(1.0, 1.0 + my);   // V - V_unc = -.5 m'y / (1 + m'y)
    } else {
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const int ca = act ? cols[a] : 0;
            const double gaa = P.ws[L.cn + ca];
            double d = gaa;
#pragma unroll
            for (int b = 0; b < a; ++b) {
                double g = P.gram[(size_t)ca * mp + (act ? cols[b] : 0)];
#pragma unroll
                for (int c = 0; c < b; ++c) g = fma(-Lm[a][c] * D[c], Lm[b][c], g);
                Lm[a][b] = fast_div(g, D[b]);
                d = fma(-Lm[a][b] * D[b], Lm[a][b], d);
            }
            D[a] = d;
            ok = ok && (d > kScreenPivotMin * gaa);
            if (!(d > 0.0)) D[a] = 1.0;   // (keeps the arithmetic of a punted lane finite)
        }
#pragma unroll
        for (int a = 0; a < K; ++a) {
            double s = mP[a];
#pragma unroll
            for (int b = 0; b < a; ++b) s = fma(-Lm[a][b], v[b], s);
            v[a] = s;
        }
#pragma unroll
        for (int a = 0; a < K; ++a) {
            z[a] = fast_div(v[a], D[a]);
            tq = fma(v[a], z[a], tq);
        }
#pragma unroll
        for (int a = K - 1; a >= 0; --a) {
            double s = z[a];
#pragma unroll
            for (int b = a + 1; b < K; ++b) s = fma(-Lm[b][a], z[b], s);
            z[a] = s;
        }
        gamma = fast_div(1.0, 1.0 + tq);
        ok = ok && (gamma > kScreenGammaMin);
#pragma unroll
        for (int l = 0; l < K; ++l) {
            lam[l] = -z[l];
            y[l] = lam[l] * gamma;
            ok = ok && (lam[l] > 0.0) && (y[l] < kGramMaxCoeff);
        }
    }
    // ---- next dual vector over all columns; the best entering column along the way.
    // Straight-line code per column: the score w^2 / (cn + m^2) is compared through s = w rsq(cn + m^2) (same order for
    // w > 0, negative otherwise) with the column index packed into the low 11 mantissa bits (2047 - j: lowest index among
    // scores equal to 2^-41 relative), so that ONE v_max_f64 per column keeps the arg-max; a last stage only keeps max w.
    bool more = false;
    int jb = 0;
    {
        const double ng = -gamma;
        unsigned boff[K];
#pragma unroll
        for (int l = 0; l < K; ++l) boff[l] = act ? 8u * (unsigned)cols[l] : 0u;   // Gr is symmetric: column c_l of row j
        const char *gbase = reinterpret_cast<const char *>(P.gram);
        double bestkey = 0.0, wtop = -1.0;
        for (int j = 0; j < mp && __ballot(ok) != 0ull; j += U) {
            double mv[U], g[K][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const char *grow = gbase + (size_t)(j + u) * mp * sizeof(double);   // (wave-uniform row + per-lane 32-bit byte offset)
#pragma unroll
                for (int l = 0; l < K; ++l) g[l][u] = *reinterpret_cast<const double *>(grow + boff[l]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) mv[u] = m0[j + u];
#pragma unroll
            for (int q = 0; q < PD; ++q)
#pragma unroll
                for (int u = 0; u < U; ++u) mv[u] = fma(Gm[(size_t)q * mp + j + u], x[q], mv[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double w = ng * mv[u];   // (pad columns: m = 1, Gr = 0: w = -gamma; passive columns: -y kGramDiag)
#pragma unroll
                for (int l = 0; l < K; ++l) w = fma(-y[l], g[l][u], w);
                if (LAST) {
                    wtop = max_raw(wtop, w);
                } else {
                    const double d = fma(mv[u], mv[u], cn[j + u]);
                    const double sc = w * __builtin_amdgcn_rsq(d);
                    const long long key = (__double_as_longlong(sc) & ~0x7ffLL) | (long long)(2047 - (j + u));
                    bestkey = max_raw(bestkey, __longlong_as_double(key));
                }
            }
            if (LAST && __ballot(ok && !(wtop > 0.0)) == 0ull) break;   // nobody left who could still be optimal
        }
        more = LAST ? (wtop > 0.0) : (bestkey > 0.0);
        jb = 2047 - (int)(__double_as_longlong(bestkey) & 0x7ffLL);
    }
    const double best = more ? 1.0 : -1.0;
    if (!act) return kScreenDone;
    if (ok && !more) {
        screen_write<PD, K>(P, tab, prob, x, true, cols, lam, tq);
        return kScreenDone;
    }
    if (ok && !LAST && best > 0.0) {
        jnext = jb;
        return kScreenNext;
    }
    // (a last stage of two columns that is fine but not optimal: the passive set travels with the state)
    P.status[prob] = (LAST && K == 2 && ok && more) ? status_pending_warm(cols[0], cols[K - 1]) : kStatusPending;
    return kScreenPunt;
}

// append (id, cols[0..K]) of the lanes with `push` to a queue in LDS (whole workgroup; no barrier inside)
template <int KC>
__device__ __forceinline__ void screen_push(bool push, int *cnt, int *qid, int (*qcol)[KC], int cap_unused, int id,
                                            const int *cols)
{
    (void)cap_unused;
    const unsigned long long mask = __ballot(push);
    const int lane = threadIdx.x & 63;
    int sb = 0;
    if (lane == 0 && mask != 0ull) sb = atomicAdd(cnt, __popcll(mask));
    sb = __builtin_amdgcn_readfirstlane(sb);
    const int slot = sb + __popcll(mask & ((1ull << lane) - 1ull));
    if (push) {
        qid[slot] = id;
#pragma unroll
        for (int l = 0; l < KC; ++l) qcol[slot][l] = cols[l];
    }
}

template <int PD>
__global__ void __launch_bounds__(kScreenThreads)
#ifdef PYMPC_SCREEN_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(PYMPC_SCREEN_WAVES_PER_EU, PYMPC_SCREEN_WAVES_PER_EU)))
#endif
mpqp_screen_kernel(SolveParams P)
{
    constexpr int S = screen_states_per_lane<PD>();
    constexpr int U0 = kScreenU0;   // columns per step of the stage-0 sweep
    constexpr int NT = kScreenThreads;
    constexpr int TILE = NT * S;
    static_assert(kScreenStages == 2, "queues below are written for two lane-path stages");
    __shared__ int q1_id[TILE + NT], q1_col[TILE + NT][1];
    __shared__ int q2_id[2 * NT], q2_col[2 * NT][2];
    __shared__ int q_cnt[2];
    const WsLayout &L = P.L;
    const int mp = L.mp;
    const cdouble_ptr m0 = as_const(P.ws + L.m0), Gm = as_const(P.ws + L.Gm), tab = as_const(P.ws), cn = as_const(P.ws + L.cn);
    const int ncut = P.cuts ? cut_count(P.cuts) : 0;
    const cdouble_ptr cut = as_const(P.cuts);   // entries [0, ncut) were complete before this launch started
    const int tid = threadIdx.x;
    const bool ws_ok = P.ws_status ? (P.ws_status[0] == 0) : true;
    if (tid < 2) q_cnt[tid] = 0;
    if (P.cuts && blockIdx.x == 0 && tid < 2)   // chunk counters of the kernels that follow in this launch (SolveParams::sched)
        __hip_atomic_store(reinterpret_cast<int *>(P.cuts) + (tid == 0 ? kCutTicketGram : kCutTicketSolve), 0, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();

    // run stage 1 / stage 2 on the top `take` entries of their queues
    auto pass1 = [&](int take) {
        const int cnt = q_cnt[0];
        __syncthreads();
        const bool act = tid < take;
        const int e = act ? cnt - take + tid : 0;
        if ((tid & ~63) < take) {
            const int id = q1_id[e];
            int cols[1] = {q1_col[e][0]};
            int jn = 0;
            const int r = screen_stage<PD, 1, false, PYMPC_SCREEN_U>(P, m0, Gm, cn, tab, act, (long long)id, cols, jn);
            int c2[2] = {cols[0], jn};
            screen_push<2>(act && r == kScreenNext, &q_cnt[1], q2_id, q2_col, 0, id, c2);
        }
        __syncthreads();
        if (tid == 0) q_cnt[0] = cnt - take;
        __syncthreads();
    };
    auto pass2 = [&](int take) {
        const int cnt = q_cnt[1];
        __syncthreads();
        const bool act = tid < take;
        const int e = act ? cnt - take + tid : 0;
        if ((tid & ~63) < take) {
            const int id = q2_id[e];
            int cols[2] = {q2_col[e][0], q2_col[e][1]};
            int jn = 0;
            (void)screen_stage<PD, 2, true, PYMPC_SCREEN_U>(P, m0, Gm, cn, tab, act, (long long)id, cols, jn);
        }
        __syncthreads();
        if (tid == 0) q_cnt[1] = cnt - take;
        __syncthreads();
    };

    const int TS = P.screen_tile;   // states per tile: TILE, or less (a multiple of 64) so that a small batch still spreads over the chip
    const long long ntile = (P.B + TS - 1) / TS;
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long base = tile * TS;
        double x[S][PD];
        bool valid[S], hit[S], neg[S];
        double best[S];
        int jb[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const long long prob = base + (long long)s * NT + tid;
            valid[s] = s * NT + tid < TS && prob < P.B;
#pragma unroll
            for (int q = 0; q < PD; ++q) x[s][q] = valid[s] ? P.X[prob * P.x_stride + q] : 0.0;
            hit[s] = false;
            neg[s] = false;
            best[s] = 0.0;
            jb[s] = 0;
        }
        // ---- stored certificates (same test as cut_fires)
        for (int c = 0; c < ncut; ++c) {
            const double cc = cut[kCutHeader + c];
            double dq[PD];
#pragma unroll
            for (int q = 0; q < PD; ++q) dq[q] = cut[kCutHeader + (size_t)(q + 1) * kCutMax + c];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double v = cc, a = fabs(cc);   // (fused: the margin is ten orders above the rounding either way)
#pragma unroll
                for (int q = 0; q < PD; ++q) {
                    v = fma(dq[q], x[s][q], v);
                    a = fma(fabs(dq[q]), fabs(x[s][q]), a);
                }
                hit[s] = hit[s] || (v < -kCutMargin * a);
            }
        }
        // ---- m(x): is some entry negative?  (a NaN state has none: the solve kernel also finds no column for it)
        // Along the way the column the NNLS would take first.
        for (int j = 0; j < mp; j += U0) {
            double c0[U0], cg[PD][U0];
#pragma unroll
            for (int u = 0; u < U0; ++u) c0[u] = m0[j + u];
#pragma unroll
            for (int q = 0; q < PD; ++q)
#pragma unroll
                for (int u = 0; u < U0; ++u) cg[q][u] = Gm[(size_t)q * mp + j + u];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double mv[U0];
#pragma unroll
                for (int u = 0; u < U0; ++u) mv[u] = c0[u];
#pragma unroll
                for (int q = 0; q < PD; ++q)
#pragma unroll
                    for (int u = 0; u < U0; ++u) mv[u] = fma(cg[q][u], x[s][q], mv[u]);
#if PYMPC_SCREEN_KEY0
                // straight-line arg-max as in screen_stage: s = -m rsq(cn + m^2) with the column packed into its low bits
#pragma unroll
                for (int u = 0; u < U0; ++u) {
                    const double d = fma(mv[u], mv[u], cn[j + u]);
                    const double sc = -mv[u] * __builtin_amdgcn_rsq(d);
                    const long long key = (__double_as_longlong(sc) & ~0x7ffLL) | (long long)(2047 - (j + u));
                    best[s] = max_raw(best[s], __longlong_as_double(key));
                }
#else
                bool neg8 = false;
#pragma unroll
                for (int u = 0; u < U0; ++u) neg8 = neg8 || (mv[u] < 0.0);
                neg[s] = neg[s] || neg8;
                if (__ballot(neg8) != 0ull) {   // (most groups of columns have no negative entry in any lane)
#pragma unroll
                    for (int u = 0; u < U0; ++u) {
                        const double d = fma(mv[u], mv[u], cn[j + u]);
                        const double sc = -mv[u] * __builtin_amdgcn_rsq(d);
                        const long long key = (__double_as_longlong(sc) & ~0x7ffLL) | (long long)(2047 - (j + u));
                        best[s] = max_raw(best[s], __longlong_as_double(key));
                    }
                }
#endif
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            neg[s] = best[s] > 0.0;   // some m_j(x) < 0 (a NaN state has none: the solve kernel also finds no column for it)
            jb[s] = 2047 - (int)(__double_as_longlong(best[s]) & 0x7ffLL);
        }
        // ---- finished here, handed to the wave kernel, or into queue 1
        // (barrier: every wave has left the previous tile's queue loop, whose conditions read the counters pushed to below)
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const long long prob = base + (long long)s * NT + tid;
            const bool cand = valid[s] && ws_ok && neg[s] && !hit[s] && best[s] > 0.0;
            if (valid[s] && !cand) {
                if (!ws_ok || (neg[s] && !hit[s])) P.status[prob] = kStatusPending;
                else screen_write<PD, 0>(P, tab, prob, x[s], !hit[s], nullptr, nullptr, 0.0);
            }
            int c1[1] = {jb[s]};
            screen_push<1>(cand, &q_cnt[0], q1_id, q1_col, 0, (int)prob, c1);
        }
        __syncthreads();
        while (q_cnt[0] >= NT) {   // (workgroup-uniform: the counters only change between barriers; the next tile's pushes wait behind the barrier above)
            pass1(NT);
            if (q_cnt[1] >= NT) pass2(NT);
        }
    }
    // ---- remainders
    while (q_cnt[0] > 0) {
        pass1(min(q_cnt[0], NT));
        if (q_cnt[1] >= NT) pass2(NT);
    }
    while (q_cnt[1] > 0) pass2(min(q_cnt[1], NT));
}

}  // namespace pympc
