// Gram-form kernel of the shared-matrix batch: one state per wavefront, between the lane path (qp_screen.hpp, states
// with up to two active constraints) and the Lawson-Hanson kernel (qp_wave.hpp, everything else).  Same NNLS
// (/root/reference/pympc/optimization/solvers/pnnls.py:214-239 through parametric_programs.py:155-179), same
// entering rule and secondary loop as qp_wave_solve, but no factorisation of the R x k matrix:
//
//   passive set P (k <= kGramK columns), G = Gr[P][P] (Gr = M M', the Gram matrix of the constant rows), H = G^-1,
//   z = H m_P(x),  t = m_P'z:   coefficients y = -z / (1 + t),  squared residual gam = 1 / (1 + t)   (Sherman-Morrison
//   on G + m_P m_P'; every term of t is positive: no cancellation),  multipliers lambda = -z,  V = V_unc + t / 2,
//   next dual vector  w = -gam m - sum_l y_l Gr[P_l][:] :  k rows of Gr through L2.
//
// H is kept explicitly, row a in lane a (symmetric: Hrow[b] = H[a][b], b a compile-time register index).  A column
// enters by bordering (Schur complement s = cn_j - g'H g, g = Gr[j][P]) and leaves by the rank-one downdate; the only
// cross-lane traffic is v_readlane broadcasts and 8-lane butterflies.  A state LEAVES this kernel unsolved (its status
// stays kStatusPending, qp_solve_kernel then solves it from scratch) when the passive set would outgrow kGramK, a
// Schur complement is small (nearly dependent rows), a coefficient is large or the residual small (near-infeasible:
// the decision against the reference's threshold belongs to the factorised solve), or the iteration cap is reached.
//
// MI355X mapping.  Nothing of the 64 KB constraint matrix is touched, so the kernel needs no LDS beyond cn, m0, Gm and
// the small objective tables (19 KB per workgroup for the cart-pole problem).  One persistent workgroup of 12 waves per
// CU (3 per SIMD: at 168 VGPRs almost nothing spills; at 128 VGPRs / 16 waves the kernel moved 440 MB of scratch per
// 1 M-state step through HBM); the waves draw interleaved slices of states through an LDS ticket, as qp_solve_kernel
// does.  An iteration costs ~10 k cycles: one L2 round trip for the candidate's Gram entries, one per two rows of the
// sweep, and ~700 wave instructions of k x k work in which at most 8 lanes are busy (change 9).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qp_wave.hpp"

namespace pympc {

#ifndef PYMPC_GRAM_LEAN
#define PYMPC_GRAM_LEAN 1
#endif
constexpr int kGramThreads = 768;       // one workgroup per CU: 12 waves (3 per SIMD, 168 VGPRs: nothing spills) draw from one pool of slices
constexpr int kGramSlice = 16;           // states per ticket of a wave (<= 64)
constexpr int kGramK = 8;               // passive columns up to which a state stays in this kernel (<= 8: wsum8)
static_assert(kGramK == kStatusSetColumns, "the hand-over format of kStatusPendingSet holds kGramK columns");
constexpr double kGramPivotMin = 1e-6;  // relative Schur complement below which a column counts as nearly dependent
constexpr double kGramGammaMin = 1e-8;  // squared residual below which the state is left to the factorised solve

// LDS image of the constants the kernel reads per state: column norms, m0, Gm (p rows) and the small objective tables
// ws[L.u0 .. L.total) (u0, Ku, Sxx, sx, s0), as doubles
__host__ __device__ inline size_t gram_lds_doubles(int p, const WsLayout &L) { return (size_t)(p + 2) * L.mp + (L.total - L.u0); }

// One state.  cnl / m0l / Gml / tab: LDS copies (tab indexed like the workspace: tab[L.u0 ...]).  Returns with the
// outputs written, or with nothing written (status stays pending).  warm >= 0: c0 | c1 << 11, the two passive columns on
// which the lane path left the state (both coefficients positive, a third column entering): the first two outer iterations
// -- two arg-max sweeps, two borderings, three rows of Gr -- are not repeated; H is built by the same two bordering steps.
template <int MT>
__device__ __forceinline__ void gram_wave_solve(const SolveParams &P, long long prob, const double *cnl, const double *m0l,
                                                const double *Gml, const double *tab, int warm)
{
    constexpr int mp = 64 * MT;
    constexpr int RW = 2;   // rows of Gr in flight in the sweep
    const WsLayout &L = P.L;
    const int lane = threadIdx.x & 63;
    const int n = P.n, m = P.m, p = P.p, R = n + 1;
    const double xq = (lane < p) ? P.X[prob * P.x_stride + lane] : 0.0;
#ifdef PYMPC_PHASE_TIMING
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();   // debug build: cycles per phase, written over mult[0..7]
#endif

    // ---- m(x) = m0 + Gm'x (pnnls.py:214), columns owned as in qp_wave_solve
    double mv[MT], w[MT];
    {
        const double2 *m2 = reinterpret_cast<const double2 *>(m0l) + lane;
#pragma unroll
        for (int h = 0; h < MT / 2; ++h) {
            const double2 v = m2[64 * h];
            mv[2 * h] = v.x;
            mv[2 * h + 1] = v.y;
        }
        for (int q0 = 0; q0 < p; q0 += 2) {
            double2 gv[2][MT / 2];
            double xv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = min(q0 + u, p - 1);
                const double xx = lane_f64(xq, q);
                xv[u] = (q0 + u < p) ? xx : 0.0;
                const double2 *g2 = reinterpret_cast<const double2 *>(Gml + (size_t)q * mp) + lane;
#pragma unroll
                for (int h = 0; h < MT / 2; ++h) gv[u][h] = g2[64 * h];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int h = 0; h < MT / 2; ++h) {
                    mv[2 * h] = fma(gv[u][h].x, xv[u], mv[2 * h]);
                    mv[2 * h + 1] = fma(gv[u][h].y, xv[u], mv[2 * h + 1]);
                }
        }
    }
    PYMPC_TICK(0)
    uint32_t inP = 0;
    int k = 0, iter = 0, outer = 0;
    int Pl = -1;
    double yl = 0.0, mPl = 0.0;
    double Hrow[kGramK];
#pragma unroll
    for (int b = 0; b < kGramK; ++b) Hrow[b] = 0.0;
    double gam = 1.0, gz = 0.0, gtq = 0.0;
    const int itmax = P.itmax_factor * m;
    bool giveup = false, handoff = false;

    // z = H m_P, t, gam; returns the candidate coefficients -z gam (lanes < k)
    auto gram_solve = [&]() __attribute__((always_inline)) -> double {
        double z = 0.0;
#pragma unroll
        for (int b = 0; b < kGramK; ++b)
            if (b < k) z = fma(Hrow[b], lane_f64(mPl, b), z);
        if (lane >= k) z = 0.0;
        gz = z;
        gtq = wsum8(mPl * z);
        gam = fast_div(1.0, 1.0 + gtq);
        return -z * gam;
    };
    // remove position jj (H, Pl, yl, mPl, inP); false: the pivot is not positive (H is spoilt)
    auto gram_drop = [&](int jj) __attribute__((always_inline)) -> bool {
        double hj = 0.0;
#pragma unroll
        for (int b = 0; b < kGramK; ++b) hj = (b == jj) ? Hrow[b] : hj;   // H[a][jj] in lane a
        if (lane >= k) hj = 0.0;
        const double piv = lane_f64(hj, jj);
        const double hp = hj * fast_div(1.0, piv);
#pragma unroll
        for (int b = 0; b < kGramK; ++b)
            if (b < k) Hrow[b] = fma(-hp, lane_f64(hj, b), Hrow[b]);
#pragma unroll
        for (int b = 0; b + 1 < kGramK; ++b) Hrow[b] = (b >= jj) ? Hrow[b + 1] : Hrow[b];   // column jj out
#pragma unroll
        for (int b = 0; b < kGramK; ++b) {                                                    // row jj out
            const double dn = __shfl_down(Hrow[b], 1);
            if (lane >= jj && lane < k - 1) Hrow[b] = dn;
        }
        const int jdel = lane_i32(Pl, jj);
        if (lane == lane_of(jdel)) inP &= ~(1u << slot_of(jdel));
        const int Pn = __shfl_down(Pl, 1);
        const double yn = __shfl_down(yl, 1), mn = __shfl_down(mPl, 1);
        if (lane >= jj && lane < k - 1) { Pl = Pn; yl = yn; mPl = mn; }
        --k;
        return piv > 0.0;
    };

    if (warm >= 0) {
        const int c0 = warm & 2047, c1 = warm >> 11;
        double ms0 = mv[0], ms1 = mv[0];
#pragma unroll
        for (int s = 1; s < MT; ++s) {
            ms0 = (s == slot_of(c0)) ? mv[s] : ms0;
            ms1 = (s == slot_of(c1)) ? mv[s] : ms1;
        }
        const double mc0 = lane_f64(ms0, lane_of(c0)), mc1 = lane_f64(ms1, lane_of(c1));
        const double g00 = cnl[c0], g11 = cnl[c1], g01 = P.gram[(size_t)c1 * mp + c0];
        // (the two bordering steps of the loop below, written out: k = 0 -> 1 -> 2)
        const double i0 = fast_div(1.0, g00);
        const double h = i0 * g01;
        const double sch = g11 - g01 * h;
        if (sch > kGramPivotMin * g11) {
            const double si = fast_div(1.0, sch), hs = h * si;
            if (lane == 0) { Hrow[0] = fma(hs, h, i0); Hrow[1] = -hs; Pl = c0; mPl = mc0; }
            if (lane == 1) { Hrow[0] = -h * si; Hrow[1] = si; Pl = c1; mPl = mc1; }
            if (lane == lane_of(c0)) inP |= 1u << slot_of(c0);
            if (lane == lane_of(c1)) inP |= 1u << slot_of(c1);
            k = 2;
            const double zl = gram_solve();
            const bool fine = !(lane < 2) || (zl > 0.0 && fabs(zl) < kGramMaxCoeff);
            if (__ballot(!fine) == 0ull && gam > kGramGammaMin) {
                if (lane < 2) yl = zl;
                outer = 2;
                iter = 2;
            } else {   // (not what the lane path saw, to rounding: from nothing, then)
                k = 0; inP = 0; Pl = -1; yl = 0.0; mPl = 0.0; gam = 1.0; gz = 0.0; gtq = 0.0;
#pragma unroll
                for (int b = 0; b < kGramK; ++b) Hrow[b] = 0.0;
            }
        }
    }

    for (;;) {
        // ---------------- dual sweep  w = -gam m - sum_l y_l Gr[P_l][:]   (nothing passive yet: w = -m)
        {
            const double ng = -gam;
#pragma unroll
            for (int s = 0; s < MT; ++s) w[s] = ng * mv[s];
            for (int l0 = 0; l0 < k; l0 += RW) {
                double2 a[RW][MT / 2];
                double yv[RW];
#pragma unroll
                for (int u = 0; u < RW; ++u) {
                    const int l = min(l0 + u, k - 1);
                    const double yy = lane_f64(yl, l);
                    yv[u] = (l0 + u < k) ? -yy : 0.0;
                    const double2 *gr = reinterpret_cast<const double2 *>(P.gram + (size_t)lane_i32(Pl, l) * mp) + lane;
#pragma unroll
                    for (int h = 0; h < MT / 2; ++h) a[u][h] = gr[64 * h];
                }
#pragma unroll
                for (int u = 0; u < RW; ++u)
#pragma unroll
                    for (int h = 0; h < MT / 2; ++h) {
                        w[2 * h] = fma(yv[u], a[u][h].x, w[2 * h]);
                        w[2 * h + 1] = fma(yv[u], a[u][h].y, w[2 * h + 1]);
                    }
            }
#if !PYMPC_GRAM_LEAN
#pragma unroll
            for (int s = 0; s < MT; ++s)
                if (col_of(lane, s) >= m || ((inP >> s) & 1u)) w[s] = -1.0;
#endif
            // (lean: no masking needed.  A pad column has m = 1 and a zero Gram row: w = -gam < 0.  A passive column meets
            // its own diagonal entry, stored as kGramDiag = 1e300: w = -y 1e300, hugely negative for its positive y.)
        }
        PYMPC_TICK(5)
        // ---------------- the column of largest guaranteed decrease (the rule of qp_wave_solve), tested in Gram form
        bool accepted = false, stale = false;   // stale: gz / gtq / gam belong to a trial set whose new column was refused
        int j = -1, jl = 0, js = 0;
        double mj = 0.0, zt = 0.0;
        for (;;) {
            double sc[MT];
            {
                const double2 *c2 = reinterpret_cast<const double2 *>(cnl) + lane;
#pragma unroll
                for (int h = 0; h < MT / 2; ++h) {
                    const double2 v = c2[64 * h];
                    sc[2 * h] = v.x;
                    sc[2 * h + 1] = v.y;
                }
            }
#if PYMPC_GRAM_LEAN
            // score and column in one key, straight-line as in the lane path (qp_screen.hpp): w^2 / (cn + m^2) is compared
            // through s = w rsq(cn + m^2) (same order for w > 0, negative otherwise), the low 10 mantissa bits give way to
            // 1023 - column: ONE max per column and one max-reduction return the best column (lowest index among scores
            // equal to 2^-42 relative); a negative key never beats the initial 0
            double best = 0.0;
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                const double d = fma(mv[s], mv[s], sc[s]);
                const double sv = w[s] * __builtin_amdgcn_rsq(d);
                const long long key = (__double_as_longlong(sv) & ~0x3ffLL) | (long long)(1023 - col_of(lane, s));
                best = max_raw(best, __longlong_as_double(key));
            }
#else
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                const double d = fma(mv[s], mv[s], sc[s]);
                sc[s] = (w[s] > 0.0) ? w[s] * w[s] * __builtin_amdgcn_rcp(d) : -1.0;
            }
            // score and column in one key: the low 10 mantissa bits of the (positive) score give way to 1023 - column, so
            // that ONE max-reduction returns the best column (lowest index among scores equal to 2^-42 relative)
            double best = -1.0;
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                const long long key = (__double_as_longlong(sc[s]) & ~0x3ffLL) | (long long)(1023 - col_of(lane, s));
                const double kd = (sc[s] > 0.0) ? __longlong_as_double(key) : -1.0;
                best = max_raw(best, kd);
            }
#endif
            const double gmax = wmax(best);
            // (lean keys: a column with w = 0 exactly packs to a denormal, not to zero: everything below 1e-290 is "no gain")
            if (!(gmax > (PYMPC_GRAM_LEAN ? 1e-290 : 0.0))) {   // no column enters: optimal on the current passive set
                if (stale) (void)gram_solve();   // ... whose solution the refused trials overwrote
                break;
            }
            if (k >= kGramK || k + 1 >= R || k + 1 >= m) {
                giveup = true;
                handoff = k >= kGramK;   // (a consistent passive set: the factorised solve can start from it)
                break;
            }
            j = 1023 - (int)(__double_as_longlong(gmax) & 0x3ffLL);
            jl = lane_of(j);
            js = slot_of(j);
            double mvsel = mv[0];
#pragma unroll
            for (int s = 1; s < MT; ++s) mvsel = (s == js) ? mv[s] : mvsel;
            mj = lane_f64(mvsel, jl);
            PYMPC_TICK(1)
            // Schur complement of column j on the passive rows
            const double cj = cnl[j];
            const double g = (lane < k) ? P.gram[(size_t)j * mp + Pl] : 0.0;
            double h = 0.0;
#pragma unroll
            for (int b = 0; b < kGramK; ++b)
                if (b < k) h = fma(Hrow[b], lane_f64(g, b), h);
            if (lane >= k) h = 0.0;
            const double sch = cj - wsum8(g * h);
            PYMPC_TICK(2)
            if (!(sch > kGramPivotMin * cj)) { giveup = true; break; }
            const double si = fast_div(1.0, sch);
            const double hs = h * si;
#pragma unroll
            for (int b = 0; b < kGramK; ++b) {
                if (b < k) {
                    const double hb = lane_f64(h, b);
                    if (lane < k) Hrow[b] = fma(hs, hb, Hrow[b]);
                    else if (lane == k) Hrow[b] = -hb * si;
                } else if (b == k) {
                    Hrow[b] = (lane < k) ? -hs : si;   // (lane k: the new diagonal entry)
                }
            }
            if (lane == k) { Pl = j; yl = 0.0; mPl = mj; }
            if (lane == jl) inP |= (1u << js);
            ++k;
            // Lawson-Hanson's second acceptance test: the new column's own coefficient must come out positive
            zt = gram_solve();
            PYMPC_TICK(3)
            if (lane_f64(zt, k - 1) > 0.0) { accepted = true; break; }
            if (!gram_drop(k - 1)) { giveup = true; break; }
            stale = true;
            if (lane == jl) {
#pragma unroll
                for (int s = 0; s < MT; ++s)
                    if (s == js) w[s] = -1.0;   // (refused: out of this round's arg-max)
            }
        }
        if (giveup || !accepted) break;
        ++outer;

        // ---------------- secondary loop (column j sits at position k - 1 with coefficient 0)
        for (bool first = true;; first = false) {
            const double zl = first ? zt : gram_solve();   // (the acceptance test already solved on the new set)
            ++iter;
            const bool big = lane < k && !(fabs(zl) < kGramMaxCoeff);
            if (iter > itmax || !(gam > kGramGammaMin) || __ballot(big) != 0ull) { giveup = true; break; }
            const bool bad = (lane < k) && (zl <= 0.0);
            if (__ballot(bad) == 0ull) {
                if (lane < k) yl = zl;
                break;
            }
            const double tt = bad ? yl / (yl - zl) : 1.0e300;
            const double alpha = wmin(tt);
            int jj = __ffsll((long long)__ballot(bad && tt == alpha)) - 1;
            if (lane < k) yl = fma(alpha, zl - yl, yl);
            for (;;) {
                if (!gram_drop(jj)) giveup = true;
                const unsigned long long neg = __ballot(lane < k && yl <= 0.0);
                if (neg == 0ull) break;
                jj = __ffsll((long long)neg) - 1;
            }
            if (giveup) break;
        }
        if (giveup) break;
        PYMPC_TICK(4)
    }
    if (giveup) {
        // the passive set travels with the state (kStatusPendingSet): its kGramK columns in the order they entered, two per
        // word of the state's active-set words
        const int nwords = (m + 31) / 32;
        if (handoff && P.active && k == kGramK && nwords >= kGramK / 2) {
            const int lo = __shfl(Pl, 2 * (lane & 31)), hi = __shfl(Pl, 2 * (lane & 31) + 1);
            if (lane < kGramK / 2) P.active[prob * nwords + lane] = (uint32_t)lo | ((uint32_t)hi << 16);
            if (lane == 0) P.status[prob] = kStatusPendingSet;
        }
        return;
    }
    PYMPC_TICK(6)

    // ---------------- outputs (pnnls.py:230-239, parametric_programs.py:176-177): optimal with multipliers -z
    const int nw = (m + 31) / 32;
    const double lam = (lane < k) ? -gz : 0.0;
    if (P.argmin) {
        double tv[kGramK];   // u -= sum_l lam_l (Huu^-1 a_Pl): rows of T, all gathers in flight together
#pragma unroll
        for (int l = 0; l < kGramK; ++l) {
            const int pj = lane_i32(Pl, min(l, max(k - 1, 0)));
            tv[l] = (l < k && lane < P.n_out) ? P.ws[L.T + (size_t)pj * n + lane] : 0.0;
        }
        double ui = 0.0;
        if (lane < P.n_out) {
            ui = tab[L.u0 + lane];
            for (int q = 0; q < p; ++q) ui = fma(tab[L.Ku + (size_t)lane * p + q], lane_f64(xq, q), ui);
        }
#pragma unroll
        for (int l = 0; l < kGramK; ++l)
            if (l < k) ui = fma(-lane_f64(lam, l), tv[l], ui);
        if (lane < P.n_out) P.argmin[prob * P.n_out + lane] = ui;
    }
    if (P.vmin) {
        double vq = 0.0;
        if (lane < p) {
            double sx = 0.0;
            for (int c = 0; c < p; ++c) sx = fma(tab[L.Sxx + (size_t)lane * p + c], lane_f64(xq, c), sx);
            vq = xq * fma(0.5, sx, tab[L.sx + lane]);
        }
        vq = wsum8(vq);   // p <= 8
        if (lane == 0) P.vmin[prob] = (tab[L.s0] + vq) + 0.5 * gtq;
    }
#ifndef PYMPC_PHASE_TIMING
    if (P.mult) {
#pragma unroll
        for (int s = 0; s < MT; ++s) {
            const int jc = col_of(lane, s);
            if (jc < m && !((inP >> s) & 1u)) P.mult[prob * m + jc] = 0.0;
        }
        if (lane < k) P.mult[prob * m + Pl] = lam;
    }
#endif
    if (P.active) {
        const unsigned long long act = __ballot(lane < k && lam > P.tol);
        uint32_t word = 0u;
        for (int l = 0; l < k; ++l) {
            const int pj = lane_i32(Pl, l);
            if (((act >> l) & 1ull) && (pj >> 5) == lane) word |= 1u << (pj & 31);
        }
        if (lane < nw) P.active[prob * nw + lane] = word;
    }
#ifdef PYMPC_PHASE_TIMING
    PYMPC_TICK(7)
    if (P.mult && lane < 8) {
        double v = 0.0;
        for (int i = 0; i < 8; ++i) v = (lane == i) ? (double)tph[i] : v;
        P.mult[prob * m + lane] = v;
    }
#endif
    if (lane == 0) {
        P.status[prob] = kStatusOptimal;
        if (P.iters) {
            P.iters[2 * prob] = outer;
            P.iters[2 * prob + 1] = iter;
        }
    }
}

// Persistent workgroups over interleaved slices of P.chunk states (the scheme of qp_solve_kernel<MT, true>); only
// states the screening pass left pending are touched.  Requires P.screened, P.gram, p <= 8 and no raw right-hand side.
template <int MT>
__global__ void __launch_bounds__(kGramThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) mpqp_gram_kernel(SolveParams P)
{
    constexpr int mp = 64 * MT;
    extern __shared__ double glds[];   // cn | m0 | Gm (p rows) | tables: gram_lds_doubles(p, L)
    __shared__ int ctl[kSchedCtlInts];   // slice ticket and chunk list (sched_next)
    if (P.ws_status && P.ws_status[0] != 0) return;   // (a workspace that is not positive definite: the solve kernel reports it)
    const WsLayout &L = P.L;
    const int p = P.p;
    double *cnl = glds, *m0l = glds + mp, *Gml = glds + 2 * mp, *tabl = glds + (size_t)(p + 2) * mp;
    for (int e = threadIdx.x; e < mp; e += blockDim.x) { cnl[e] = P.ws[L.cn + e]; m0l[e] = P.ws[L.m0 + e]; }
    for (int e = threadIdx.x; e < p * mp; e += blockDim.x) Gml[e] = P.ws[L.Gm + e];
    for (int e = threadIdx.x; e < (int)(L.total - L.u0); e += blockDim.x) tabl[e] = P.ws[L.u0 + e];
    sched_init(P, ctl);
    __syncthreads();
    if (P.sched) {
        sched_first(P, ctl);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, S = P.chunk;
    const long long nslice = (P.B + S - 1) / S;
    const long long mine = blockIdx.x < nslice ? (nslice - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    ChunkSched W;
    W.solo = -1; W.solo_s = 0;
    for (;;) {
        long long base;
        if (P.sched) {
            base = sched_next(P, ctl, W, lane, S);
            if (base < 0) break;
        } else {
            int t = 0;
            if (lane == 0) t = atomicAdd(ctl, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= mine) break;
            base = ((long long)blockIdx.x + (long long)t * gridDim.x) * S;
        }
        const int st = (lane < S && base + lane < P.B) ? P.status[base + lane] : 0;
        unsigned long long todo = __ballot(status_is_pending(st));
        while (todo != 0ull) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            gram_wave_solve<MT>(P, base + b, cnl, m0l, Gml, tabl - L.u0, status_warm_columns(lane_i32(st, b)));
        }
    }
}

}  // namespace pympc
