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
//   * Lawson-Hanson's acceptance and step rules (0.01 independence factor, positive trial coefficient, alpha
//     interpolation, itmax = 3 * columns); the ENTERING column is the one that guarantees the largest decrease of
//     the residual, max (a_j'r)^2 / |a_j|^2 over w_j = a_j'r > 0, instead of Lawson-Hanson's max w_j.  Both rules
//     stop at the same (unique) NNLS solution; on the cart-pole QPs the scaled rule needs 0.9 instead of 1.9 columns
//     per feasible state because it rarely picks a column that has to be dropped again (constraint rows of an MPC
//     problem have very different norms).  Measured on 300 000 C2 states: identical statuses and active sets, optimal
//     values within 1.3e-10 relative.
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
    int chunk;                  // problems per workgroup; SHARED with persist: problems per interleaved slice
    int screened;               // SHARED: status[] was initialised by mpqp_screen_kernel (kStatusPending = still to solve)
    int screen_tile;            // states per tile of mpqp_screen_kernel (<= its compile-time tile; smaller for small batches)
    int persist;                // SHARED: the grid is one workgroup per CU, workgroup g owns slices g, g + G, g + 2 G, ...
    int wave_doubles;           // per-wave LDS doubles
    // raw NNLS (pympc_nnls_batch: scipy.optimize.nnls on a general matrix and right-hand side): rhs [B x (n + 1)] replaces
    // e_R, the outputs are the coefficients (mult [B x m]) and the residual norm (vmin [B]); no QP interpretation
    const double *rhs;
    long long rhs_stride;
    int raw;
    int learn;                  // SHARED: this launch is the certificate-learning pre-pass (returns at once if the table is not empty)
    int *sched;                 // SHARED: device counter (zero at launch; the screening kernel clears it) that hands out chunks of
    int dyn_chunk;              //   dyn_chunk states (a multiple of `chunk`) to the WORKGROUPS of the launch, whose waves share a
                                //   chunk out in slices of `chunk` through LDS.  null: static interleaved slices per workgroup
    double *cuts;               // learned infeasibility certificates of this mpQP (cut_table_doubles(p) doubles) or null
    const double *gram;         // SHARED: Gram matrix of the constant rows [mp x mp] (ws_layout.hpp) or null
    int itmax_factor;           // NNLS iteration limit = itmax_factor * columns (scipy.optimize.nnls: 3)
};

constexpr double kGramMaxCoeff = 1e3;   // first column through the Gram row only while its coefficient is moderate

// per-wave LDS doubles for R rows, n variables, m columns
__host__ __device__ inline int wave_lds_doubles(int n, int p, int RP)
{
    const int R = n + 1;
    (void)n;
    return invqr_lds_doubles(R, RP) + (R + kVecPad) + p + 8;   // Q, Wi, cv, ab | rb | x0 (+ 8 zeros)
}

// Column ownership: lane l owns the MT / 2 pairs of adjacent columns 128 h + 2 l, 128 h + 2 l + 1, so that a row of
// the staged matrix is read with 16-byte LDS loads (ds_read_b128 moves 256 B/clk; the paired 8-byte form the
// compiler builds from strided columns, ds_read2st64_b64, half of that).  Slot s = 2 h + e.
__device__ __forceinline__ int col_of(int lane, int s) { return 128 * (s >> 1) + 2 * lane + (s & 1); }
__device__ __forceinline__ int lane_of(int j) { return (j & 127) >> 1; }
__device__ __forceinline__ int slot_of(int j) { return 2 * (j >> 7) + (j & 1); }

// ------------------------------------------------------------------------------------------ infeasibility cuts
// A QP of the mpQP is infeasible at x  <=>  there is y >= 0 with M'y = 0 and m(x)'y < 0 (Farkas).  That is exactly what
// the NNLS ends with when its residual is zero (pnnls.py:230: "infeasible"): A y = e_R, i.e. M'y = 0, m(x)'y = -1.  The
// first condition does not involve x, so the SAME y proves infeasibility of every other state x' with
//        m(x')'y = y'm0 + (Gm y)'x' = c + d'x' < 0 :
// a half-space of the parameter space.  The feasible set of an mpQP has few facets (the cart-pole N = 20 problem: ~50
// cuts cover 99.6 % of the infeasible states of a uniform batch), so the kernel keeps the certificates it finds in a
// small table in global memory (per prepared mpQP, shared by every wave through L2) and tests each new state against
// them (lane <-> cut: p + 1 FMAs) before it starts the active-set iterations, which for an infeasible state are the
// longest of all (the passive set has to fill all n + 1 rows).  A state is rejected only when a certificate holds
// with a relative margin of 1e-6; anything closer to the boundary takes the full solve, so the decision is the
// reference's (its own threshold is a residual of 1e-7).  A certificate is stored only after its x-independent part
// has been verified to rounding level (|M'y| <= 1e-11 sum |M||y| row by row).
// Table: header | c[kCutMax] | d_0[kCutMax] | ... | d_{p-1}[kCutMax].  One writer at a time (try-lock, never waited
// for); an entry is complete in L2 before `count` covers it, readers bypass the non-coherent vector L1.  The header
// also holds the chunk counters of a launch's Gram-form and Lawson-Hanson kernels (SolveParams::sched), each on a
// cache line of its own, away from `count`, which every running solve polls: the table is the one piece of device
// memory that belongs to one launch at a time.
constexpr int kCutMax = 256;
constexpr int kCutPMax = 8;
constexpr int kCutHeader = 48;   // three 128-byte lines: count, lock, base, spare | chunk counter (Gram kernel) | chunk counter (solve kernel)
constexpr int kCutTicketGram = 32, kCutTicketSolve = 64;   // int32 slots of the header
constexpr double kCutMargin = 1e-6;
constexpr double kCutVerify = 1e-11;
constexpr int kCutLearnSkip = 16;    // entries from which the learning pre-pass of a launch is skipped
constexpr int kCutRecheckFrom = 6;   // passive columns from which a running solve looks at newly stored certificates
__host__ __device__ inline size_t cut_table_doubles(int p) { return kCutHeader + (size_t)(p + 1) * kCutMax; }

__device__ __forceinline__ double coherent_load(const double *a)
{
    return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int cut_count(const double *cuts)
{
    // acquire: pairs with the writer's release store of `count`, so the entries it covers are visible to the loads below
    const int c = __hip_atomic_load(reinterpret_cast<const int *>(cuts), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(min(c, kCutMax));
}

// does one of the first cnt certificates prove x0 (LDS, p entries) infeasible?  One full wavefront; wave-uniform result.
// PP = p at compile time: all loads of a lane's certificates are in flight together.
template <int PP>
__device__ __forceinline__ bool cut_fires_p(const double *cuts, int cnt, const double *x0)
{
    const int lane = threadIdx.x & 63;
    double xq[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) xq[q] = x0[q];
    bool fire = false;
#pragma unroll
    for (int h = 0; h < kCutMax / 64; ++h) {
        if (64 * h < cnt) {   // wave-uniform
            const int idx = lane + 64 * h;
            const bool on = idx < cnt;
            const double *e = cuts + kCutHeader + (on ? idx : 0);
            double d[PP];
            const double c = coherent_load(e);
#pragma unroll
            for (int q = 0; q < PP; ++q) d[q] = coherent_load(e + (size_t)(q + 1) * kCutMax);
            double v = c, a = fabs(c);
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                const double t = d[q] * xq[q];
                v += t;
                a += fabs(t);
            }
            fire = fire || (on && v < -kCutMargin * a);
        }
    }
    return __ballot(fire) != 0ull;
}
__device__ __forceinline__ bool cut_fires(const double *cuts, int cnt, int p, const double *x0)
{
    switch (p) {
    case 1: return cut_fires_p<1>(cuts, cnt, x0);
    case 2: return cut_fires_p<2>(cuts, cnt, x0);
    case 3: return cut_fires_p<3>(cuts, cnt, x0);
    case 4: return cut_fires_p<4>(cuts, cnt, x0);
    case 5: return cut_fires_p<5>(cuts, cnt, x0);
    case 6: return cut_fires_p<6>(cuts, cnt, x0);
    case 7: return cut_fires_p<7>(cuts, cnt, x0);
    case 8: return cut_fires_p<8>(cuts, cnt, x0);
    }
    return false;
}

// The NNLS of state x0 ended with a zero residual on the passive set (lane l < k: column Pl, coefficient yl > 0,
// m(x) entry mPl).  Verify the certificate and append it to the table unless an entry already covers x0.
__device__ __forceinline__ void cut_learn(double *cuts, int n, int p, int mp, const double *__restrict__ Mt, int mtp,
                                          const double *__restrict__ m0, const double *__restrict__ Gm, int k, int Pl,
                                          double yl, double mPl, const double *x0)
{
    const int lane = threadIdx.x & 63;
    if (cut_count(cuts) >= kCutMax) return;
    // ---- x-independent part: M_P'y = 0 row by row, and the last row m_P'y = -1
    double eps = (lane == n) ? 1.0 : 0.0, S = eps;
    for (int l = 0; l < k; ++l) {
        const int pj = lane_i32(Pl, l);
        const double yy = lane_f64(yl, l), mm = lane_f64(mPl, l);
        double a = 0.0;
        if (lane < n) a = Mt[lane * mtp + pj];
        else if (lane == n) a = mm;
        eps = fma(a, yy, eps);
        S = fma(fabs(a), fabs(yy), S);
    }
    const bool bad = (lane <= n && !(fabs(eps) <= kCutVerify * S)) || (lane < k && !(yl > 0.0));
    if (__ballot(bad) != 0ull) return;
    // ---- c = y'm0, d = Gm y: lane 0 keeps c, lane q + 1 keeps d_q
    const int col = (lane < k) ? Pl : 0;
    double mine = wsum(lane < k ? yl * m0[col] : 0.0);
    double scale = fabs(mine);
    for (int q = 0; q < p; ++q) {
        const double dq = wsum(lane < k ? yl * Gm[(size_t)q * mp + col] : 0.0);
        if (lane == q + 1) mine = dq;
        scale = fmax(scale, fabs(dq));
    }
    if (!(scale > 0.0) || !(scale < 1e300)) return;
    mine = mine / scale;
    // ---- single writer
    int *hdr = reinterpret_cast<int *>(cuts);
    int got = 0;
    if (lane == 0) got = (atomicCAS(hdr + 1, 0, 1) == 0) ? 1 : 0;
    got = __builtin_amdgcn_readfirstlane(got);
    if (!got) return;
    const int cnt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(hdr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
    if (cnt >= 0 && cnt < kCutMax && !cut_fires(cuts, cnt, p, x0)) {
        if (lane <= p)
            __hip_atomic_store(cuts + kCutHeader + (size_t)lane * kCutMax + cnt, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        if (lane == 0) __hip_atomic_store(hdr, cnt + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    if (lane == 0) __hip_atomic_store(hdr + 1, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

using cdouble_ptr = const double __attribute__((address_space(4))) *;
__device__ __forceinline__ cdouble_ptr as_const(const double *p) { return (cdouble_ptr)p; }

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
                              const double *__restrict__ wsg, long long prob, double *wl, bool prepared_ok,
                              const double *__restrict__ cnorm, bool warmset = false)
{
    constexpr int mp = 64 * MT;
    constexpr int SW = MT <= 8 ? 4 : (MT <= 16 ? 2 : 1);   // rows of the dual sweep in flight (MT * SW loads)
    const int lane = threadIdx.x & 63;
    const int n = P.n, m = P.m, p = P.p, R = n + 1, RP = P.RP;
    InvQr F;
    invqr_bind(F, wl, R, RP);
    double *rb = wl + invqr_lds_doubles(R, RP);
    double *x0 = rb + R + kVecPad;   // the parameter vector, staged once and followed by 8 zeros (LDS broadcasts later on)
    if (lane < p) x0[lane] = P.X[prob * P.x_stride + lane];
    if (lane < R) F.cv[lane] = P.rhs ? P.rhs[prob * P.rhs_stride + lane] : ((lane == n) ? 1.0 : 0.0);   // cv = Qt b, Qt = I
    const int ncut = P.cuts ? cut_count(P.cuts) : 0;
    wave_sync_lds();

#ifdef PYMPC_PHASE_TIMING
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();   // debug build: cycles per phase, written over mult[0..7]
#endif
    // ---- a stored certificate proves this state infeasible: no iterations
    const bool cut_hit = ncut > 0 && cut_fires(P.cuts, ncut, p, x0);
    int ncut_seen = ncut;

    // ---- m(x) = m0 + Gm' x : the per-problem row of the NNLS matrix (pnnls.py:214)
    double mv[MT], w[MT];
#pragma unroll
    for (int s = 0; s < MT; ++s) mv[s] = 0.0;
    if (!cut_hit) {
        const double2 *m2 = reinterpret_cast<const double2 *>(m0) + lane;
#pragma unroll
        for (int h = 0; h < MT / 2; ++h) {
            const double2 v = m2[64 * h];
            mv[2 * h] = v.x;
            mv[2 * h + 1] = v.y;
        }
        constexpr int QU = MT <= 8 ? 4 : (MT <= 16 ? 2 : 1);   // rows of Gm whose loads are in flight together
        for (int q0 = 0; q0 < p; q0 += QU) {
            double2 gv[QU][MT / 2];
            double xv[QU];
#pragma unroll
            for (int u = 0; u < QU; ++u) {
                const int q = min(q0 + u, p - 1);
                const double xq = x0[q];
                xv[u] = (q0 + u < p) ? xq : 0.0;
                const double2 *g2 = reinterpret_cast<const double2 *>(Gm + q * mp) + lane;
#pragma unroll
                for (int h = 0; h < MT / 2; ++h) gv[u][h] = g2[64 * h];
            }
#pragma unroll
            for (int u = 0; u < QU; ++u)
#pragma unroll
                for (int h = 0; h < MT / 2; ++h) {
                    mv[2 * h] = fma(gv[u][h].x, xv[u], mv[2 * h]);
                    mv[2 * h + 1] = fma(gv[u][h].y, xv[u], mv[2 * h + 1]);
                }
        }
    }

    uint32_t inP = 0;   // bit s: column lane + 64 s is in the passive set
    int k = 0, iter = 0, outer = 0, status = prepared_ok ? (cut_hit ? kStatusInfeasible : kStatusOptimal) : kStatusNotPD;
    int Pl = -1;        // lane l < k: column index at position l of the passive set
    double yl = 0.0;    // ... its coefficient
    double mPl = 0.0;   // ... its m(x) entry
    const int itmax = P.itmax_factor * m;
    bool qt_ready = false;
    // First column without a factorisation (shared mpQP with a Gram matrix): with the single passive column j the NNLS
    // solution is y = -m_j / (cn_j + m_j^2) and the squared residual cn_j / (cn_j + m_j^2); the next dual vector comes
    // from one Gram row.  Most states of an MPC batch stop there (one active constraint); Q, R^-1 and cv are only
    // built ("materialised") when a second column is tried.
    bool virt = false;
    double r2v = 1.0;
    auto materialise = [&]() {
        const int j0 = lane_i32(Pl, 0);
        const double mj0 = lane_f64(mPl, 0);
        double a0 = 0.0;
        if (lane < n) a0 = -Mt[lane * mtp + j0];
        else if (lane == n) a0 = -mj0;
        const QrTrial t0 = qr_trial(F, 0, a0, true);
        qr_add_first(F, t0);
        qt_ready = true;
        virt = false;
    };
#ifdef PYMPC_COUNT_TRIALS
    int trials = 0;   // debug build: candidate tests, reported in place of the inner-iteration counter
#endif

    // ---- a state the Gram-form kernel left at kStatusSetColumns passive columns (kStatusPendingSet): factorise that set,
    // column by column in the order they entered (the very sequence of updates this kernel would have gone through),
    // instead of finding it again with eight sweeps and arg-max searches.  Anything unexpected (a dependent column, a
    // coefficient that does not come out positive) and the state starts from nothing, as every other state does.
    if (warmset && P.active && status == kStatusOptimal) {
        const int nwords = (m + 31) / 32;
        const uint32_t myword = (lane < kStatusSetColumns / 2) ? P.active[prob * nwords + lane] : 0u;
        bool good = true;
        for (int l = 0; l < kStatusSetColumns; ++l) {
            const uint32_t word = (uint32_t)lane_i32((int)myword, l >> 1);
            const int jw = (int)((l & 1) ? (word >> 16) : (word & 0xffffu));
            if (jw >= m || k + 1 >= R) { good = false; break; }
            const int jwl = lane_of(jw), jws = slot_of(jw);
            if (lane_i32((int)((inP >> jws) & 1u), jwl) != 0) { good = false; break; }   // (twice the same column: not a set)
            double mvsel = mv[0];
#pragma unroll
            for (int s = 1; s < MT; ++s) mvsel = (s == jws) ? mv[s] : mvsel;
            const double mjw = lane_f64(mvsel, jwl);
            double ai = 0.0;
            if (lane < n) ai = -Mt[lane * mtp + jw];
            else if (lane == n) ai = -mjw;
            const QrTrial tw = qr_trial(F, k, ai, !qt_ready);
            if (!tw.independent) { good = false; break; }
            if (!qt_ready) {
                qr_add_first(F, tw);
                qt_ready = true;
            } else {
                qr_add(F, k, tw);
            }
            if (lane == k) { Pl = jw; yl = 0.0; mPl = mjw; }
            if (lane == jwl) inP |= (1u << jws);
            ++k;
        }
        if (good && k > 0) {
            const double zl = qr_coeff(F, k);
            if (__ballot(lane < k && !(zl > 0.0)) == 0ull) {
                if (lane < k) yl = zl;
                outer = k;
                iter = k;
            } else {
                good = false;
            }
        }
        if (!good) {
            k = 0; inP = 0; Pl = -1; yl = 0.0; mPl = 0.0; qt_ready = false;
            if (lane < R) F.cv[lane] = (lane == n) ? 1.0 : 0.0;
            wave_sync_lds();
        }
    }

    PYMPC_TICK(0)

    while (status == kStatusOptimal && k < R && k < m) {
        // ---------------- dual sweep  w = A'(b - A y) = A' Qt'[0; cv[k:]]
        if (!qt_ready && !P.rhs) {
#pragma unroll
            for (int s = 0; s < MT; ++s) w[s] = -mv[s];   // A'e_R: the last row
        } else {
            // r = Qt'[0; cv[k:]]  (before the first column enters: Qt = I, r = the right-hand side)
            if (lane < R) rb[lane] = qt_ready ? dotu(F.Q + lane * RP + k, F.cv + k, R - k) : F.cv[lane];
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
        // ---------------- pick the column of largest guaranteed decrease that passes the Lawson-Hanson acceptance tests
        bool accepted = false;
        int j = -1, jl = 0, js = 0;
        double mj = 0.0;
        QrTrial t;
        for (;;) {
            double sc[MT];
            {
                const double2 *c2 = reinterpret_cast<const double2 *>(cnorm) + lane;
#pragma unroll
                for (int h = 0; h < MT / 2; ++h) {
                    const double2 v = c2[64 * h];
                    sc[2 * h] = v.x;
                    sc[2 * h + 1] = v.y;
                }
#pragma unroll
                for (int s = 0; s < MT; ++s) {
                    const double d = fma(mv[s], mv[s], sc[s]);
                    sc[s] = (w[s] > 0.0) ? w[s] * w[s] * __builtin_amdgcn_rcp(d) : -1.0;
                }
            }
            double best = sc[0];
            int bs = 0;
#pragma unroll
            for (int s = 1; s < MT; ++s)
                if (sc[s] > best) { best = sc[s]; bs = s; }
            const double gmax = wmax(best);
            if (!(gmax > 0.0)) break;
            j = wmin_int(best == gmax ? col_of(lane, bs) : 0x7fffffff);
            jl = lane_of(j);
            js = slot_of(j);
            double mvsel = mv[0];
#pragma unroll
            for (int s = 1; s < MT; ++s) mvsel = (s == js) ? mv[s] : mvsel;
            mj = lane_f64(mvsel, jl);
            PYMPC_TICK(2)
            if (P.gram && !qt_ready && !virt) {
                // k == 0: the first column always passes the acceptance tests (w_j > 0 is the sign of its coefficient).
                // Taken without a factorisation; the next dual vector follows from row j of the Gram matrix:
                // w_i = -m_i - y (M_i . M_j + m_i m_j) = -(1 + m_j y) m_i - y Gr[j][i],  1 + m_j y = cn_j / (cn_j + m_j^2)
                const double cj = cnorm[j];
                const double dj = fma(mj, mj, cj);
                const double y1 = fast_div(-mj, dj);
                if (y1 < kGramMaxCoeff) {
                    const double2 *g2 = reinterpret_cast<const double2 *>(P.gram + (size_t)j * mp) + lane;
                    double2 gv[MT / 2];
#pragma unroll
                    for (int h = 0; h < MT / 2; ++h) gv[h] = g2[64 * h];
                    r2v = fast_div(cj, dj);
                    if (lane == 0) { Pl = j; yl = y1; mPl = mj; }
                    if (lane == jl) inP |= (1u << js);
                    k = 1;
                    ++outer;
                    ++iter;
                    virt = true;
#pragma unroll
                    for (int h = 0; h < MT / 2; ++h) {
                        w[2 * h] = fma(-y1, gv[h].x, -r2v * mv[2 * h]);
                        w[2 * h + 1] = fma(-y1, gv[h].y, -r2v * mv[2 * h + 1]);
                    }
#pragma unroll
                    for (int s = 0; s < MT; ++s)
                        if (col_of(lane, s) >= m || ((inP >> s) & 1u)) w[s] = -1.0;
                    if (k >= R || k >= m) break;   // (one-row problems: nothing more can enter)
                    continue;   // pick again with the new dual vector
                }
            }
            if (virt) materialise();
            double ai = 0.0;
            if (lane < n) ai = -Mt[lane * mtp + j];
            else if (lane == n) ai = -mj;
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
            qr_add_first(F, t);   // k == 0 here: the passive set only ever empties at the start of a problem
            qt_ready = true;
        } else {
            qr_add(F, k, t);
        }
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
        // A long solve is most often an infeasible state (the passive set is filling all n + 1 rows).  Other waves may
        // have stored the certificate that rejects it since this one started: look again at the entries that are new.
        if (P.cuts && status == kStatusOptimal && k >= kCutRecheckFrom && ((k - kCutRecheckFrom) & 3) == 0) {
            const int now = cut_count(P.cuts);
            if (now > ncut_seen) {
                ncut_seen = now;
                if (cut_fires(P.cuts, now, p, x0)) {
                    status = kStatusInfeasible;
                }
            }
        }
    }

    PYMPC_TICK(6)
    // ---------------- residual norm and outputs (pnnls.py:230-239, parametric_programs.py:176-177)
    // Three sums in one butterfly: the squared NNLS residual |cv[k:]|^2, m'y, and the unconstrained value V_unc(x).
    const WsLayout &L = P.L;
    double r2 = 0.0, my = 0.0, vq = 0.0;
    if ((qt_ready || P.rhs) && lane >= k && lane < R) {
        const double cl = F.cv[lane];
        r2 = cl * cl;
    }
    if (lane < k) my = mPl * yl;
    if (P.vmin && lane < p) vq = x0[lane] * fma(0.5, dotu(tab + L.Sxx + lane * p, x0, p), tab[L.sx + lane]);
    wsum3(r2, my, vq);
    const double rnorm = (qt_ready || P.rhs) ? fast_sqrt(r2) : fast_sqrt(virt ? r2v : 1.0);   // Qt = I, cv = e_R before the first column enters
    if (!P.raw && status == kStatusOptimal && !(rnorm > P.tol)) {
        status = kStatusInfeasible;
        if (P.cuts && k > 0) cut_learn(P.cuts, n, p, mp, Mt, mtp, m0, Gm, k, Pl, yl, mPl, x0);
    }
    const bool okk = (status == kStatusOptimal);
    const int nw = (m + 31) / 32;
    const double nanv = __builtin_nan("");

    // lambda = y / (1 + m'y) (pnnls.py:231).  With gamma = 1 + m'y the optimal value needs nothing else: the NNLS
    // optimality conditions give |M'y|^2 = -m'y gamma, hence V - V_unc = .5 |L^-1 A_P'lambda|^2 = -.5 m'y / gamma.
    const double ginv = P.raw ? 1.0 : (okk ? fast_div(1.0, 1.0 + my) : 0.0);
    const double lam = (lane < k) ? yl * ginv : 0.0;
    if (P.argmin) {
        double ui = nanv;
        if (okk) {
            ui = 0.0;
            if (lane < P.n_out) ui = dotu(tab + L.Ku + lane * p, x0, p, tab[L.u0 + lane]);
            for (int l0 = 0; l0 < k; l0 += 8) {   // u -= sum_l lam_l (Huu^-1 a_Pl): rows of T, eight gathers in flight
                double tv[8], ll[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int l = min(l0 + u, k - 1);
                    ll[u] = lane_f64(lam, l);
                    const int pj = lane_i32(Pl, l);
                    tv[u] = (lane < P.n_out) ? wsg[L.T + (size_t)pj * n + lane] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (l0 + u < k) ui = fma(-ll[u], tv[u], ui);
            }
        }
        if (lane < P.n_out) P.argmin[prob * P.n_out + lane] = ui;
    }
    if (P.vmin && lane == 0) P.vmin[prob] = P.raw ? rnorm : (okk ? (tab[L.s0] + vq) - 0.5 * my * ginv : nanv);
#ifndef PYMPC_PHASE_TIMING   // (the debug build reports its cycle counts through mult)
    if (P.mult) {
#pragma unroll
        for (int s = 0; s < MT; ++s) {
            const int jc = col_of(lane, s);
            if (jc < m && !(okk && ((inP >> s) & 1u))) P.mult[prob * m + jc] = okk ? 0.0 : nanv;
        }
        if (okk && lane < k) P.mult[prob * m + Pl] = lam;
    }
#endif
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
// ws[L.u0 .. L.total) (u0, Ku, Sxx, sx, s0, zeros).  m0 and Gm are read once per problem (m(x) = m0 + Gm'x), from
// L2: keeping them out of LDS buys two more problems in flight per CU.
struct SharedLds {
    int mtp;
    size_t tab, cn, total;
};
__host__ __device__ inline SharedLds shared_lds(int n, const WsLayout &L)
{
    SharedLds S;
    S.mtp = L.mp + 2;   // even (16-byte row alignment); consecutive rows start 4 banks apart
    S.tab = (size_t)n * S.mtp;
    S.cn = S.tab + (L.total - L.u0);
    S.total = S.cn + L.mp;
    return S;
}

// Chunks of states for a persistent workgroup from ONE device counter, shared out to its waves in slices through LDS.
// Why two levels: the chip's eight L2s are not coherent with each other, so a device-scope atomic is served memory-side,
// ~14 ns apiece on one address -- one request per slice and wave (47 000 a step) cost more than the balancing gained; one
// request per chunk and workgroup is a few thousand.  LDS: ctl[0] = slice ticket, ctl[1 .. kSchedList] = the device
// chunk index of the workgroup's k-th chunk (-1 until it is known).  The wave that draws the first slice of chunk k
// requests chunk k + 1 (so the answer is there a whole chunk ahead) and publishes it; a wave that needs a chunk which is
// not published yet polls LDS -- its publisher never waits for anything but the atomic's return, so nobody waits for
// ever.  A workgroup that outruns its list (kSchedList chunks) lets each wave take whole chunks for itself.
constexpr int kSchedList = 60;
constexpr int kSchedCtlInts = 4 + kSchedList;     // [0] slice ticket, [1] learn flag (qp_solve_kernel), [2], [3] spare, list

struct ChunkSched {
    int solo, solo_s;   // a wave on its own: chunk at hand (-1: list mode) and next slice of it
};

__device__ __forceinline__ void sched_init(const SolveParams &P, int *ctl)
{
    // (called by every thread before the workgroup's first barrier)
    for (int e = threadIdx.x; e < kSchedList; e += blockDim.x) ctl[4 + e] = -1;
    if (threadIdx.x == 0) ctl[0] = 0;
}
__device__ __forceinline__ void sched_first(const SolveParams &P, int *ctl)
{
    // (thread 0, after the barrier that follows sched_init and before the next one)
    if (threadIdx.x == 0) __hip_atomic_store(ctl + 4, atomicAdd(P.sched, 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// next slice of this wave: first state of it, or -1 when the launch has no more (wave-uniform)
__device__ __forceinline__ long long sched_next(const SolveParams &P, int *ctl, ChunkSched &W, int lane, int S)
{
    const int C = P.dyn_chunk / S;
    const long long nchunks = (P.B + P.dyn_chunk - 1) / P.dyn_chunk;
    int g, s;
    if (W.solo < 0) {
        int t = 0;
        if (lane == 0) t = atomicAdd(ctl, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        const int k = t / C;
        s = t - k * C;
        if (k < kSchedList) {
            if (s == 0 && k + 1 < kSchedList && lane == 0)
                __hip_atomic_store(ctl + 4 + k + 1, atomicAdd(P.sched, 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (;;) {
                g = __hip_atomic_load(ctl + 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (g >= 0) break;
                __builtin_amdgcn_s_sleep(2);
            }
            g = __builtin_amdgcn_readfirstlane(g);
        } else {
            W.solo = 0;
            W.solo_s = C;   // (asks for a chunk below)
            g = 0; s = 0;
        }
    }
    if (W.solo >= 0) {
        if (W.solo_s >= C) {
            int c = 0;
            if (lane == 0) c = atomicAdd(P.sched, 1);
            W.solo = __builtin_amdgcn_readfirstlane(c);
            W.solo_s = 0;
        }
        g = W.solo;
        s = W.solo_s++;
    }
    if (g >= nchunks) return -1;
    return (long long)g * P.dyn_chunk + (long long)s * S;
}

// Workgroup = NW waves; it owns problems [blockIdx.x * chunk, ...) and hands them to its waves through an LDS
// ticket.  SHARED: the constant rows Mt / m0 / Gm and the objective tables are staged in LDS once per workgroup.
// waves per workgroup: 8 (2 per SIMD, 213 VGPRs at MT = 6, nothing spilled) up to 8 columns per lane, 4 (512 VGPRs)
// beyond.  Until round 3: 12 (168 VGPRs, 51 of them spilled: 32 MB of scratch written per 1 M-state step).  Since the
// Gram-form kernel hands its states over with their columns, this kernel's time is the longest single chain of pivots,
// not the number of waves: 8 measured 1 % (s = 0.3) and 4 % (s = 1.0) faster per step than 12
#ifndef PYMPC_SOLVE_WAVES
#define PYMPC_SOLVE_WAVES 8
#endif
// (per-problem matrices, SHARED = false: workgroups of 4 waves, many problems of equal length: there the number of
// resident waves is what counts, so the register budget stays that of 3 waves per SIMD)
template <int MT, bool SHARED>
constexpr int solve_max_waves() { return MT <= 8 ? (SHARED ? PYMPC_SOLVE_WAVES : 12) : 4; }

template <int MT, bool SHARED>
__global__ void __launch_bounds__((64 * solve_max_waves<MT, SHARED>())) qp_solve_kernel(SolveParams P)
{
    extern __shared__ double lds[];
    constexpr int mp = 64 * MT;
    const int n = P.n, p = P.p;
    const SharedLds S_ = shared_lds(n, P.L);
    const size_t sharedD = SHARED ? S_.total : 0;
    int *ticket = reinterpret_cast<int *>(lds + sharedD);
    // learning pre-pass: a table that already holds certificates (kept from an earlier call, or stored at prepare time)
    // needs none.  Other workgroups of this launch append to the table meanwhile, so ONE thread reads the count and the
    // whole workgroup branches on that value (every wave for itself could split the workgroup around the staging below).
    if (SHARED && P.learn && P.cuts) {
        if (threadIdx.x == 0) ticket[1] = cut_count(P.cuts) >= kCutLearnSkip ? 1 : 0;
        __syncthreads();
        if (ticket[1] != 0) return;
    }
    double *waves = lds + sharedD + kSchedCtlInts / 2;
    const int wave = threadIdx.x >> 6;
    if (SHARED) {
        const double *src = P.ws;
        for (int e = threadIdx.x; e < n * mp; e += blockDim.x) lds[(e / mp) * S_.mtp + (e % mp)] = src[P.L.Mt + e];
        for (int e = threadIdx.x; e < (int)(P.L.total - P.L.u0); e += blockDim.x) lds[S_.tab + e] = src[P.L.u0 + e];
        for (int e = threadIdx.x; e < mp; e += blockDim.x) lds[S_.cn + e] = src[P.L.cn + e];
    }
    for (int e = threadIdx.x; e < (int)(blockDim.x >> 6) * P.wave_doubles; e += blockDim.x) waves[e] = 0.0;   // (I2), (I3)
    sched_init(P, ticket);
    __syncthreads();
    if (SHARED && P.sched) {
        sched_first(P, ticket);
        __syncthreads();
    }
    double *wl = waves + (size_t)wave * P.wave_doubles;
    if (SHARED) {
        // One persistent workgroup per CU (the constant rows are staged once per CU): workgroup g owns the slices
        // g, g + G, g + 2 G, ... of `chunk` <= 64 problems, so that all workgroups draw the same mix of short and long
        // problems and finish together; its waves take whole slices through one LDS ticket.  After the screening pass
        // (mpqp_screen_kernel) a wave reads the statuses of its slice with one load and solves only the pending ones.
        const bool shared_ok = P.ws_status ? (P.ws_status[0] == 0) : true;
        const int lane = threadIdx.x & 63, S = P.chunk;
        const long long nslice = (P.B + S - 1) / S;
        const long long mine = blockIdx.x < nslice ? (nslice - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
        ChunkSched W;
        W.solo = -1; W.solo_s = 0;
        for (;;) {
            long long base;
            if (P.sched) {   // a static share per workgroup ends 15 - 25 % after its mean (~170 solves of 3 .. 24 pivots each)
                base = sched_next(P, ticket, W, lane, S);
                if (base < 0) break;
            } else {
                int t = 0;
                if (lane == 0) t = atomicAdd(ticket, 1);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= mine) break;
                base = ((long long)blockIdx.x + (long long)t * gridDim.x) * S;
            }
            bool pend = lane < S && base + lane < P.B;
            int st = kStatusPending;
            if (P.screened && pend) {
                st = P.status[base + lane];
                pend = status_is_pending(st);
            }
            unsigned long long todo = __ballot(pend);
            while (todo != 0ull) {
                const int b = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                qp_wave_solve<MT>(P, lds, S_.mtp, P.ws + P.L.m0, P.ws + P.L.Gm, lds + S_.tab - P.L.u0, P.ws, base + b, wl, shared_ok,
                                  lds + S_.cn, lane_i32(st, b) == kStatusPendingSet);
            }
        }
    } else {
        // per-problem matrices: workgroup = contiguous chunk of problems, one LDS ticket
        const long long first = (long long)blockIdx.x * P.chunk;
        long long cnt = P.B - first;
        if (cnt > P.chunk) cnt = P.chunk;
        for (;;) {
            int t = 0;
            if ((threadIdx.x & 63) == 0) t = atomicAdd(ticket, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= cnt) break;
            const long long prob = first + t;
            const double *wsg = P.ws + prob * P.ws_stride;
            const bool ok = P.ws_status ? (P.ws_status[ P.ws_stride ? prob : 0] == 0) : true;
            qp_wave_solve<MT>(P, wsg + P.L.Mt, mp, wsg + P.L.m0, wsg + P.L.Gm, wsg, wsg, prob, wl, ok, wsg + P.L.cn);
        }
    }
}

}  // namespace pympc
