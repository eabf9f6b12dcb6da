// One strictly convex QP per wavefront, solved as the NNLS problem of Bemporad's Theorem 1
// (/root/reference/pympc/optimization/solvers/pnnls.py:149-241):
//
//        min || A y - e_R ||   s.t. y >= 0,      A = [ -M' ; -m' ]   (R = n + 1 rows, m columns)
//
// with the Lawson-Hanson active-set method that scipy.optimize.nnls implements (pnnls.py:218).
//
// MI355X mapping (64-lane wavefront, no tracing compiler, no library calls):
//   * columns of A are strided over the 64 lanes, MT columns per lane; the per-problem last row m(x) and the
//     dual vector w live in registers (2*MT doubles per lane);
//   * the first n rows of A are identical for every state of an mpQP, so the workgroup stages them ONCE in LDS
//     (Mt, [n x 64*MT], row i contiguous over columns -> conflict-free ds_read_b64) and every wave sweeps them
//     for w = A' r with r broadcast from LDS;
//   * instead of Lawson-Hanson's transformed copy of A per problem (62 KB for the cart-pole QP) each wave keeps
//     only  Qt (R x R, Qt A_P = [Rk;0]),  Wi = Rk^-1  and  cv = Qt e_R  in LDS (7 KB): adding a column is one
//     Householder reflector applied to Qt plus one new column of Wi; dropping a column is a run of Givens
//     rotations read off the dropped row of Wi.  Small-matrix work maps lane <-> row (or column) so that the
//     mat-vecs need no cross-lane reduction; the only reductions are the arg-max of w and three dot products.
//   * selection rules are Lawson-Hanson's: first maximum of w, 0.01 independence factor, positive trial
//     coefficient, alpha interpolation, itmax = 3 * columns.
//
// tests/kernel_model.py is the scalar numpy model of exactly these update formulas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "invqr_wave.hpp"
#include "ws_layout.hpp"

namespace pympc {

struct SolveParams {
    int n, m, p, RP;            // RP: odd pitch of the R x R per-wave matrices
    WsLayout L;
    const double *ws;           // prepared workspace(s)
    long long ws_stride;        // doubles between problems' workspaces (0: one shared workspace)
    const int32_t *ws_status;   // per-workspace prepare status (may be null); stride 1 if ws_stride != 0
    const double *X;            // [B x p] or null when p == 0
    long long x_stride;         // doubles between problems' parameter vectors (0: one shared x)
    long long B;
    double tol;
    int n_out;
    double *argmin;             // [B x n_out] or null
    double *vmin;               // [B] or null
    double *mult;               // [B x m] or null
    uint32_t *active;           // [B x ceil(m/32)] or null
    int32_t *status;            // [B]
    int32_t *iters;             // [B x 2] or null
    int chunk;                  // problems per workgroup
    int wave_doubles;           // per-wave LDS doubles
};

// per-wave LDS doubles for R rows, n variables, m columns
__host__ __device__ inline int wave_lds_doubles(int n, int p, int RP)
{
    const int R = n + 1;
    return invqr_lds_doubles(R, RP) + (R + kVecPad) + n + p + 1;   // Q, Wi, cv, ab | rb | ub | x0
}

// Column ownership: lane l owns the MT / 2 pairs of adjacent columns 128 h + 2 l, 128 h + 2 l + 1, so that a row of
// the staged matrix is read with 16-byte LDS loads (ds_read_b128 moves 256 B/clk; the paired 8-byte form the
// compiler builds from strided columns, ds_read2st64_b64, half of that).  Slot s = 2 h + e.
__device__ __forceinline__ int col_of(int lane, int s) { return 128 * (s >> 1) + 2 * lane + (s & 1); }
__device__ __forceinline__ int lane_of(int j) { return (j & 127) >> 1; }
__device__ __forceinline__ int slot_of(int j) { return 2 * (j >> 7) + (j & 1); }

// Solve problem `prob`.  Mt (row pitch mtp) / m0 / Gm point to LDS (shared-matrix launch) or to this problem's global
// workspace; tab[L.u0 ...] are the small objective tables (LDS copy or the global workspace), wsg the global workspace
// (rows of T for the primal recovery).
#ifdef PYMPC_PHASE_TIMING
#define PYMPC_TICK(i) { const long long now_ = clock64(); tph[i] += now_ - tlast; tlast = now_; }
#else
#define PYMPC_TICK(i)
#endif

template <int MT>
__device__ void qp_wave_solve(const SolveParams &P, const double *__restrict__ Mt, int mtp, const double *__restrict__ m0,
                              const double *__restrict__ Gm, const double *__restrict__ tab,
                              const double *__restrict__ wsg, long long prob, double *wl, bool prepared_ok)
{
    constexpr int mp = 64 * MT;
    constexpr int SW = MT <= 8 ? 4 : (MT <= 16 ? 2 : 1);   // rows of the dual sweep in flight (MT * SW loads)
    const int lane = threadIdx.x & 63;
    const int n = P.n, m = P.m, p = P.p, R = n + 1, RP = P.RP;
    InvQr F;
    invqr_bind(F, wl, R, RP);
    double *rb = wl + invqr_lds_doubles(R, RP);
    double *ub = rb + R + kVecPad;
    double *x0 = ub + n;   // the parameter vector, staged once (every later use is an LDS broadcast, not an HBM round trip)
    if (lane < p) x0[lane] = P.X[prob * P.x_stride + lane];
    if (lane < R) F.cv[lane] = (lane == n) ? 1.0 : 0.0;   // cv = Qt e_R with Qt = I
    wave_sync_lds();

#ifdef PYMPC_PHASE_TIMING
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();   // debug build: cycles per phase, written over mult[0..7]
#endif
    // ---- m(x) = m0 + Gm' x : the per-problem row of the NNLS matrix (pnnls.py:214)
    double mv[MT], w[MT];
    {
        const double2 *m2 = reinterpret_cast<const double2 *>(m0) + lane;
#pragma unroll
        for (int h = 0; h < MT / 2; ++h) {
            const double2 v = m2[64 * h];
            mv[2 * h] = v.x;
            mv[2 * h + 1] = v.y;
        }
        for (int q = 0; q < p; ++q) {
            const double xq = x0[q];
            const double2 *g2 = reinterpret_cast<const double2 *>(Gm + q * mp) + lane;
#pragma unroll
            for (int h = 0; h < MT / 2; ++h) {
                const double2 v = g2[64 * h];
                mv[2 * h] = fma(v.x, xq, mv[2 * h]);
                mv[2 * h + 1] = fma(v.y, xq, mv[2 * h + 1]);
            }
        }
    }

    uint32_t inP = 0;   // bit s: column lane + 64 s is in the passive set
    int k = 0, iter = 0, outer = 0, status = prepared_ok ? kStatusOptimal : kStatusNotPD;
    int Pl = -1;        // lane l < k: column index at position l of the passive set
    double yl = 0.0;    // ... its coefficient
    double mPl = 0.0;   // ... its m(x) entry
    const int itmax = 3 * m;
    bool qt_ready = false;
#ifdef PYMPC_COUNT_TRIALS
    int trials = 0;   // debug build: candidate tests, reported in place of the inner-iteration counter
#endif

    PYMPC_TICK(0)

    while (status == kStatusOptimal && k < R && k < m) {
        // ---------------- dual sweep  w = A'(b - A y) = A' Qt'[0; cv[k:]]
        if (!qt_ready) {
#pragma unroll
            for (int s = 0; s < MT; ++s) w[s] = -mv[s];
        } else {
            if (lane < R) rb[lane] = dotu(F.Q + lane * RP + k, F.cv + k, R - k);   // r = Qt'[0; cv[k:]]
            wave_sync_lds();
            double acc[MT];
            const double rn = rb[n];
#pragma unroll
            for (int s = 0; s < MT; ++s) acc[s] = mv[s] * rn;
            for (int i0 = 0; i0 < n; i0 += SW) {
                double rv[SW], row[SW][MT];
#pragma unroll
                for (int u = 0; u < SW; ++u) {
                    const int i = min(i0 + u, n - 1);
                    rv[u] = rb[i];
                    const double2 *rp = reinterpret_cast<const double2 *>(Mt + i * mtp) + lane;
#pragma unroll
                    for (int h = 0; h < MT / 2; ++h) {
                        const double2 v = rp[64 * h];
                        row[u][2 * h] = v.x;
                        row[u][2 * h + 1] = v.y;
                    }
                }
#pragma unroll
                for (int u = 0; u < SW; ++u) {
                    const double r = (i0 + u < n) ? rv[u] : 0.0;
#pragma unroll
                    for (int s = 0; s < MT; ++s) acc[s] = fma(row[u][s], r, acc[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < MT; ++s) w[s] = -acc[s];
        }
#pragma unroll
        for (int s = 0; s < MT; ++s)
            if (col_of(lane, s) >= m || ((inP >> s) & 1u)) w[s] = -1.0;

        PYMPC_TICK(1)
        // ---------------- pick the first maximum of w that passes the Lawson-Hanson acceptance tests
        bool accepted = false;
        int j = -1, jl = 0, js = 0;
        double mj = 0.0;
        QrTrial t;
        for (;;) {
            double best = w[0];
            int bs = 0;
#pragma unroll
            for (int s = 1; s < MT; ++s)
                if (w[s] > best) { best = w[s]; bs = s; }
            const double gmax = wmax(best);
            if (!(gmax > 0.0)) break;
            j = wmin_int(best == gmax ? col_of(lane, bs) : 0x7fffffff);
            jl = lane_of(j);
            js = slot_of(j);
            double mvsel = mv[0];
#pragma unroll
            for (int s = 1; s < MT; ++s) mvsel = (s == js) ? mv[s] : mvsel;
            mj = lane_f64(mvsel, jl);
            double ai = 0.0;
            if (lane < n) ai = -Mt[lane * mtp + j];
            else if (lane == n) ai = -mj;
            PYMPC_TICK(2)
            t = qr_trial(F, k, ai, !qt_ready);
            PYMPC_TICK(3)
#ifdef PYMPC_COUNT_TRIALS
            ++trials;
#endif
            if (t.independent && (t.ck / t.diag) > 0.0) { accepted = true; break; }
            if (lane == jl) {
#pragma unroll
                for (int s = 0; s < MT; ++s)
                    if (s == js) w[s] = 0.0;
            }
        }
        if (!accepted) break;
        ++outer;

        // ---------------- move column j into the passive set at position k
        if (!qt_ready) {
            qr_reset(F);
            qt_ready = true;
        }
        qr_add(F, k, t);
        if (lane == k) { Pl = j; yl = 0.0; mPl = mj; }
        if (lane == jl) inP |= (1u << js);
        ++k;
        PYMPC_TICK(4)

        // ---------------- secondary loop
        for (;;) {
            const double zl = qr_coeff(F, k);
            ++iter;
            if (iter > itmax) { status = kStatusMaxIter; break; }
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
                // ---- drop position jj
                const int jdel = lane_i32(Pl, jj);
                if (lane == lane_of(jdel)) inP &= ~(1u << slot_of(jdel));
                qr_drop(F, k, jj);
                const int Pn = __shfl_down(Pl, 1);
                const double yn = __shfl_down(yl, 1), mn = __shfl_down(mPl, 1);
                if (lane >= jj && lane < k - 1) { Pl = Pn; yl = yn; mPl = mn; }
                --k;
                const unsigned long long neg = __ballot(lane < k && yl <= 0.0);
                if (neg == 0ull) break;
                jj = __ffsll((long long)neg) - 1;
            }
        }
        PYMPC_TICK(5)
    }

    PYMPC_TICK(6)
    // ---------------- residual norm and outputs (pnnls.py:230-239, parametric_programs.py:176-177)
    const double cl = (lane >= k && lane < R) ? F.cv[lane] : 0.0;
    const double rnorm = sqrt(wsum(cl * cl));
    if (status == kStatusOptimal && !(rnorm > P.tol)) status = kStatusInfeasible;
    const bool okk = (status == kStatusOptimal);
    const int nw = (m + 31) / 32;
    const double nanv = __builtin_nan("");

    double lam = 0.0;
    if (okk) {
        const double my = wsum(lane < k ? mPl * yl : 0.0);
        lam = (lane < k) ? yl / (1.0 + my) : 0.0;
    }
    if (P.argmin || P.vmin) {
        double ui = nanv, V = nanv;
        if (okk) {
            const WsLayout &L = P.L;
            ui = 0.0;
            double fi = 0.0;
            if (lane < n) {
                ui = dot8(tab + L.Ku + lane * p, 1, x0, 1, 0, p, tab[L.u0 + lane]);
                fi = dot8(tab + L.Hux + lane * p, 1, x0, 1, 0, p, tab[L.fu + lane]);
            }
            for (int l0 = 0; l0 < k; l0 += 8) {   // u -= sum_l lam_l (Huu^-1 a_Pl): rows of T, eight gathers in flight
                double tv[8], ll[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int l = min(l0 + u, k - 1);
                    ll[u] = lane_f64(lam, l);
                    const int pj = lane_i32(Pl, l);
                    tv[u] = (lane < n) ? wsg[L.T + (size_t)pj * n + lane] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (l0 + u < k) ui = fma(-ll[u], tv[u], ui);
            }
            if (P.vmin) {
                if (lane < n) ub[lane] = ui;
                wave_sync_lds();
                double term = 0.0;
                if (lane < n) {   // (u'Huu)[lane] as pnnls.py:234 evaluates x.dot(H): column `lane` of Huu
                    const double hi = dot8(ub, 1, tab + L.Huu + lane, n, 0, n);
                    term = ui * (0.5 * hi + fi);
                }
                if (lane < p) {
                    const double hx = dot8(x0, 1, tab + L.Hxx + lane, p, 0, p);
                    term += x0[lane] * (0.5 * hx + tab[L.fx + lane]);
                }
                V = wsum(term) + tab[L.g];
            }
        }
        if (P.argmin && lane < P.n_out) P.argmin[prob * P.n_out + lane] = ui;
        if (P.vmin && lane == 0) P.vmin[prob] = V;
    }
    if (P.mult) {
#pragma unroll
        for (int s = 0; s < MT; ++s) {
            const int jc = col_of(lane, s);
            if (jc < m && !(okk && ((inP >> s) & 1u))) P.mult[prob * m + jc] = okk ? 0.0 : nanv;
        }
        if (okk && lane < k) P.mult[prob * m + Pl] = lam;
    }
    if (P.active) {   // word `lane` of the bit mask: OR over the passive positions that fall into it (m <= 2048: <= 64 words)
        const unsigned long long act = __ballot(okk && lane < k && lam > P.tol);
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
        P.status[prob] = status;
        if (P.iters) {
            P.iters[2 * prob] = outer;
#ifdef PYMPC_COUNT_TRIALS
            P.iters[2 * prob + 1] = trials;
#else
            P.iters[2 * prob + 1] = iter;
#endif
        }
    }
    wave_sync_lds();
}

// LDS image of the constant part for the shared-matrix launch: Mt rows pitched by mp + 2 doubles (the column gather
// Mt[lane][j] of the candidate test then spreads over 16 bank groups instead of one), and the small objective tables
// ws[L.u0 .. L.total) (u0, Ku, Huu, Hux, fu, Hxx, fx, g).  m0 and Gm are read once per problem (m(x) = m0 + Gm'x), from
// L2: keeping them out of LDS buys two more problems in flight per CU.
struct SharedLds {
    int mtp;
    size_t tab, total;
};
__host__ __device__ inline SharedLds shared_lds(int n, const WsLayout &L)
{
    SharedLds S;
    S.mtp = L.mp + 2;   // even (16-byte row alignment); consecutive rows start 4 banks apart
    S.tab = (size_t)n * S.mtp;
    S.total = S.tab + (L.total - L.u0);
    return S;
}

// Workgroup = NW waves; it owns problems [blockIdx.x * chunk, ...) and hands them to its waves through an LDS
// ticket.  SHARED: the constant rows Mt / m0 / Gm and the objective tables are staged in LDS once per workgroup.
// waves per workgroup: 12 (3 per SIMD, 168 VGPRs) up to 8 columns per lane, 4 (512 VGPRs) beyond, so that
// the per-lane column arrays mv / w / acc / row never spill to scratch
template <int MT>
constexpr int solve_max_waves() { return MT <= 8 ? 12 : 4; }

template <int MT, bool SHARED>
__global__ void __launch_bounds__(64 * solve_max_waves<MT>()) qp_solve_kernel(SolveParams P)
{
    extern __shared__ double lds[];
    constexpr int mp = 64 * MT;
    const int n = P.n, p = P.p;
    const SharedLds S = shared_lds(n, P.L);
    const size_t sharedD = SHARED ? S.total : 0;
    int *ticket = reinterpret_cast<int *>(lds + sharedD);
    double *waves = lds + sharedD + 2;
    const int wave = threadIdx.x >> 6;
    if (SHARED) {
        const double *src = P.ws;
        for (int e = threadIdx.x; e < n * mp; e += blockDim.x) lds[(e / mp) * S.mtp + (e % mp)] = src[P.L.Mt + e];
        for (int e = threadIdx.x; e < (int)(P.L.total - P.L.u0); e += blockDim.x) lds[S.tab + e] = src[P.L.u0 + e];
    }
    for (int e = threadIdx.x; e < (int)(blockDim.x >> 6) * P.wave_doubles; e += blockDim.x) waves[e] = 0.0;   // (I2), (I3)
    if (threadIdx.x == 0) *ticket = 0;
    __syncthreads();
    const long long first = (long long)blockIdx.x * P.chunk;
    long long cnt = P.B - first;
    if (cnt > P.chunk) cnt = P.chunk;
    double *wl = waves + (size_t)wave * P.wave_doubles;
    bool shared_ok = true;
    if (SHARED && P.ws_status) shared_ok = (P.ws_status[0] == 0);
    for (;;) {
        int t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(ticket, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= cnt) break;
        const long long prob = first + t;
        if (SHARED) {
            qp_wave_solve<MT>(P, lds, S.mtp, P.ws + P.L.m0, P.ws + P.L.Gm, lds + S.tab - P.L.u0, P.ws, prob, wl, shared_ok);
        } else {
            const double *wsg = P.ws + prob * P.ws_stride;
            const bool ok = P.ws_status ? (P.ws_status[prob] == 0) : true;
            qp_wave_solve<MT>(P, wsg + P.L.Mt, mp, wsg + P.L.m0, wsg + P.L.Gm, wsg, wsg, prob, wl, ok);
        }
    }
}

}  // namespace pympc
