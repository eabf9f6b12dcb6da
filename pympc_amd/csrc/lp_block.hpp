// One LP per workgroup:   min f'x  s.t.  A x <= b      (equalities folded by the caller as rows (C, -C),
// /root/reference/pympc/optimization/solvers/pnnls.py:101-106).
//
// The reference states the LP's KKT conditions (zero duality gap, primal and dual feasibility) as one partially
// non-negative least squares problem with a dense (1+m+n) x 2m matrix and an (1+m+n)^2 projector (pnnls.py:109-125,
// 5-39) and hands it to scipy.optimize.nnls: O(m^3) per LP.  A zero residual there means "(x, y) is a primal-dual
// optimal pair"; a positive one means the LP is infeasible or unbounded and every output is None (pnnls.py:137).
// This kernel computes the same pair with an active-set method that never leaves the n-dimensional space
// (tests/kernel_model.py: lp_invqr is its scalar model):
//
//   phase 1  Lawson-Hanson NNLS on the dual-feasibility system  min || A'y + f ||, y >= 0  (n rows, m columns),
//            with the add-column / delete-column steps of invqr_wave.hpp.  A positive residual means that no dual
//            feasible point exists: the LP is unbounded or infeasible -> status INFEASIBLE (None in the host API).
//   phase 2  dual active-set pivots on the same factorisation: x = minimum-norm point of {A_W x = b_W}; the most
//            violated row (distance to the hyperplane) enters; if it is independent of the working rows it is
//            appended with multiplier 0, otherwise a ratio test on d = Wi (Qt a_e)[:k] picks the row that leaves;
//            no blocking row -> Farkas certificate of primal infeasibility -> status INFEASIBLE.
//            The multipliers are y_W = Wi cv[:k] with cv = Qt(-f) carried through every update.
//
// MI355X mapping: the O(m n) sweeps (dual vector in phase 1, slacks in phase 2) are spread over the T threads of
// the workgroup, row j of A -> thread j mod T, reading the TRANSPOSED matrix At [n x m] so that a wavefront reads
// 512 contiguous bytes per instruction; the matrix is shared by every LP of a redundancy-removal batch and stays
// L2-resident (10 000 x 6 doubles = 480 KB < 4 MB per XCD).  The O(n^2) factorisation lives in LDS and is updated
// by wave 0 (lane <-> row), the other waves wait at the workgroup barrier.  Arg-max reductions are DPP/shuffle
// inside a wave and one LDS slot per wave across the workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "invqr_wave.hpp"
#include "ws_layout.hpp"

namespace pympc {

constexpr int kLpNMax = 64;         // variables (rows of the dual system; one lane per row, LDS 2 x n x (n | 1) doubles)
constexpr int kLpMMax = 1 << 18;    // rows of A (working-set bitmap of m bits in LDS)
constexpr double kLpFeasTol = 1e-9; // phase 2: row j is violated when a_j x > b_j + kLpFeasTol max(1, |b_j|)
constexpr int kLpRejMax = 32;       // rejected candidates remembered per outer iteration of phase 1

// right-hand side against which phase 2 tests a row (both LP kernels score (a_j x - lp_row_rhs(b_j)) / |a_j| > 0)
__device__ __forceinline__ double lp_row_rhs(double b) { return b + kLpFeasTol * fmax(1.0, fabs(b)); }

struct LpParams {
    int n, m, RP;
    const double *At;      long long at_stride;   // [n x m] transposed rows, then [m] inverse row norms (see lp_prepare_kernel)
    const double *f;       long long f_stride;    // [n]
    const double *b;       long long b_stride;    // [m]; stride 0: shared
    const int32_t *relax_idx;                     // [B] or null: b_k[relax_idx[k]] += relax_val (polyhedron.py:439-441)
    double relax_val;
    long long B;
    double tol;
    int maxit;
    double *argmin;        // [B x n] or null
    double *vmin;          // [B] or null
    double *mult;          // [B x m] or null
    uint32_t *active;      // [B x ceil(m/32)] or null
    int32_t *status;       // [B]
    int32_t *iters;        // [B x 3] or null: phase-1 outer, phase-1 inner, phase-2 pivots
};

// At[i * m + j] = A[j * n + i];  ainv[j] = 1 / ||a_j||  (1 for zero rows: their "distance" is the raw violation)
struct LpPrepParams {
    int n, m;
    const double *A; long long a_stride;
    double *ws;      long long ws_stride;
};

__global__ void lp_prepare_kernel(LpPrepParams P)
{
    const long long k = blockIdx.y;
    const double *A = P.A + k * P.a_stride;
    double *At = P.ws + k * P.ws_stride;
    double *ainv = At + (size_t)P.n * P.m;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < P.m; j += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < P.n; ++i) {
            const double a = A[(size_t)j * P.n + i];
            At[(size_t)i * P.m + j] = a;
            s = fma(a, a, s);
        }
        ainv[j] = s > 0.0 ? 1.0 / sqrt(s) : 1.0;
    }
}

__host__ __device__ inline size_t lp_lds_doubles(int n, int m, int RP, int nwaves)
{
    const size_t nw = (size_t)(m + 31) / 32;
    // Q, Wi, cv, ab | rb, tb, yb (padded) | ctl (8) + per-wave slots (2 x 2 x nwaves) + bitmap words (2 x nw uint32) + rej
    return (size_t)invqr_lds_doubles(n, RP) + 3 * (size_t)(n + kVecPad) + 8 + 4 * (size_t)nwaves + nw + (kLpRejMax + 1) / 2 + 2;
}

// (value, index) arg-max over the workgroup: largest value, lowest index among equal values.  Every thread gets the
// result.  `slot` alternates between two LDS buffers so that one barrier per call is enough.
__device__ __forceinline__ void block_argmax(double &val, int &idx, double *slots_v, int *slots_i, int nwaves)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double gmax = wmax(val);
    const int gi = wmin_int(val == gmax ? idx : 0x7fffffff);
    if (lane == 0) { slots_v[wave] = gmax; slots_i[wave] = gi; }
    __syncthreads();
    double bv = slots_v[0];
    int bi = slots_i[0];
    for (int w = 1; w < nwaves; ++w) {
        const double v = slots_v[w];
        const int i = slots_i[w];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    val = bv; idx = bi;
}

constexpr int kLpJU = 4;   // rows of A per thread whose loads are in flight together

// acc[q] += a_j . r for the rows j = min(jb + q T, m - 1), q < kLpJU (At is [n x m]: consecutive threads read
// consecutive doubles).  NT = n when n <= 8 (all n kLpJU loads of a thread issued before the first FMA), 0 = generic n
// (two columns of At at a time).  The compiler does not pipeline these loads by itself: with a plain loop a sweep costs
// one L2 round trip per (row, column).
template <int NT>
__device__ __forceinline__ void lp_rowdots(const double *__restrict__ At, int m, int n, const double *rb, int jb, int T,
                                           double (&acc)[kLpJU])
{
    size_t jq[kLpJU];
#pragma unroll
    for (int q = 0; q < kLpJU; ++q) jq[q] = (size_t)min(jb + q * T, m - 1);
    if (NT > 0) {
        double a[NT > 0 ? NT : 1][kLpJU];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int q = 0; q < kLpJU; ++q) a[i][q] = At[(size_t)i * m + jq[q]];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const double ri = rb[i];
#pragma unroll
            for (int q = 0; q < kLpJU; ++q) acc[q] = fma(a[i][q], ri, acc[q]);
        }
    } else {
        for (int i = 0; i < n; i += 2) {
            const int i1 = min(i + 1, n - 1);
            const double r0 = rb[i], r1 = (i + 1 < n) ? rb[i1] : 0.0;
            double a0[kLpJU], a1[kLpJU];
#pragma unroll
            for (int q = 0; q < kLpJU; ++q) {
                a0[q] = At[(size_t)i * m + jq[q]];
                a1[q] = At[(size_t)i1 * m + jq[q]];
            }
#pragma unroll
            for (int q = 0; q < kLpJU; ++q) acc[q] = fma(a1[q], r1, fma(a0[q], r0, acc[q]));
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(1024) lp_solve_kernel(LpParams P)
{
    extern __shared__ double lds[];
    const int n = P.n, m = P.m, RP = P.RP;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const long long prob = blockIdx.x;
    const int nw = (m + 31) / 32;

    InvQr F;
    invqr_bind(F, lds, n, RP);
    double *rb = lds + invqr_lds_doubles(n, RP);   // phase 1: residual direction r; phase 2: x
    double *tb = rb + n + kVecPad;
    double *yb = tb + n + kVecPad;
    double *ctl_d = yb + n + kVecPad;   // [4]
    int *ctl = reinterpret_cast<int *>(ctl_d + 4);   // [8]
    double *slots_v = ctl_d + 8;                     // [2][nwaves]
    int *slots_i = reinterpret_cast<int *>(slots_v + 2 * nwaves);   // [2][nwaves] (int)
    uint32_t *inW = reinterpret_cast<uint32_t *>(slots_v + 4 * nwaves);
    uint32_t *abits = inW + nw;
    int *rej = reinterpret_cast<int *>(abits + nw + ((2 * nw) & 1));

    const double *At = P.At + prob * P.at_stride;
    const double *ainv = At + (size_t)n * m;
    const double *f = P.f + prob * P.f_stride;
    const double *b = P.b + prob * P.b_stride;
    const int rx = P.relax_idx ? P.relax_idx[prob] : -1;
    const double rv = P.relax_val;

    // ---------------- initial state: Qt = I, cv = -f, empty working set
    for (int e = tid; e < invqr_lds_doubles(n, RP) + 3 * (n + kVecPad); e += T) lds[e] = 0.0;   // Q, Wi, padded vectors
    for (int e = tid; e < nw; e += T) { inW[e] = 0u; abits[e] = 0u; }
    __syncthreads();
    if (tid < n) {
        F.Q[tid * RP + tid] = 1.0;
        F.cv[tid] = -f[tid];
    }
    __syncthreads();

    int k = 0, status = kStatusOptimal, outer = 0, inner = 0, piv = 0, flip = 0;
    int Wl = -1;         // wave 0, lane l < k: row index at position l of the working set
    double yl = 0.0;     // ... its multiplier
    const int itmax = 3 * m;
    double fnorm = 1.0;
    {
        const double fl = (lane < n) ? F.cv[lane] : 0.0;
        fnorm = fmax(1.0, sqrt(wsum(fl * fl)));
    }

    // =========================================================== phase 1: NNLS  min || A'y + f ||, y >= 0
    for (;;) {
        if (wave == 0) {
            const double cl = (lane >= k && lane < n) ? F.cv[lane] : 0.0;
            const double rn = sqrt(wsum(cl * cl));
            const int done = (k >= n || k >= m || !(rn > 1e-13 * fnorm)) ? 1 : 0;
            if (!done && lane < n) rb[lane] = dotu(F.Q + lane * RP + k, F.cv + k, n - k);   // r = Qt'[0; cv[k:]]
            if (lane == 0) { ctl[0] = done; ctl[1] = 0; }
        }
        __syncthreads();
        if (ctl[0]) break;
        bool accepted = false;
        for (;;) {
            // ---- dual sweep w_j = a_j . r over the rows that are neither working nor rejected
            const int nrej = ctl[1];
            double best = -1.0;
            int bj = 0x7fffffff;
            for (int jb = tid; jb < m; jb += T * kLpJU) {
                double acc[kLpJU] = {0.0, 0.0, 0.0, 0.0};
                lp_rowdots<NT>(At, m, n, rb, jb, T, acc);
#pragma unroll
                for (int q = 0; q < kLpJU; ++q) {
                    const int j = jb + q * T;
                    if (j >= m || ((inW[j >> 5] >> (j & 31)) & 1u)) continue;
                    double w = acc[q] * ainv[j];   // entering row by largest guaranteed decrease: (a_j.r) / |a_j|
                    for (int c = 0; c < nrej; ++c)
                        if (rej[c] == j) w = 0.0;
                    if (w > best) { best = w; bj = j; }
                }
            }
            block_argmax(best, bj, slots_v + flip * nwaves, slots_i + flip * nwaves, nwaves);
            flip ^= 1;
            if (!(best > 0.0)) break;
            const int j = bj;
            if (wave == 0) {
                const double gl = (lane < n) ? At[(size_t)lane * m + j] : 0.0;
                const QrTrial t = qr_trial(F, k, gl);
                const bool ok = t.independent && (t.ck / t.diag) > 0.0;
                if (ok) {
                    qr_add(F, k, t);
                    if (lane == k) { Wl = j; yl = 0.0; }
                    if (lane == 0) inW[j >> 5] |= 1u << (j & 31);
                    ++k;
                    ++outer;
                    // ---- secondary loop of Lawson-Hanson
                    for (;;) {
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
                } else if (lane == 0) {
                    const int q = ctl[1];
                    if (q < kLpRejMax) rej[q] = j;
                    ctl[1] = q + 1;
                }
                if (lane == 0) { ctl[2] = ok ? 1 : 0; ctl[3] = k; ctl[4] = status; }
            }
            __syncthreads();
            k = ctl[3];
            status = ctl[4];
            if (ctl[2]) { accepted = true; break; }
            if (ctl[1] > kLpRejMax) break;   // (ctl[1] keeps telling so below: candidates with a positive dual remain untried)
        }
        if (!accepted || status != kStatusOptimal) break;
    }
    __syncthreads();
    if (wave == 0 && status == kStatusOptimal) {
        const double cl = (lane >= k && lane < n) ? F.cv[lane] : 0.0;
        const double rn = sqrt(wsum(cl * cl));
        // no dual feasible point: LP unbounded or infeasible -- unless phase 1 stopped at its cap of rejected candidates:
        // then nothing is proven and the host raises (iteration limit) instead of answering None
        if (!(rn < P.tol)) status = (ctl[1] > kLpRejMax) ? kStatusMaxIter : kStatusInfeasible;
        if (lane == 0) ctl[4] = status;
    }
    __syncthreads();
    status = ctl[4];

    // =========================================================== phase 2: dual active-set pivots
    while (status == kStatusOptimal) {
        if (wave == 0) {
            // x = Qt[:k]' (Wi' b_W)
            double bw = 0.0;
            if (lane < k) bw = b[Wl] + (Wl == rx ? rv : 0.0);
            if (lane < n) F.ab[lane] = bw;
            wave_sync_lds();
            if (lane < n) tb[lane] = (lane < k) ? dots(F.Wi + lane, RP, F.ab, k) : 0.0;   // Wi' b_W (column `lane` of Wi)
            wave_sync_lds();
            if (lane < n) rb[lane] = dotu(F.Q + lane * RP, tb, k);                         // Qt[:k]' (Wi' b_W)
        }
        __syncthreads();
        double best = 0.0;
        int be = 0x7fffffff;
        for (int jb = tid; jb < m; jb += T * kLpJU) {
            double acc[kLpJU], bq[kLpJU];
#pragma unroll
            for (int q = 0; q < kLpJU; ++q) {
                const int j = min(jb + q * T, m - 1);
                bq[q] = lp_row_rhs(b[j] + (j == rx ? rv : 0.0));
                acc[q] = 0.0;
            }
            lp_rowdots<NT>(At, m, n, rb, jb, T, acc);
#pragma unroll
            for (int q = 0; q < kLpJU; ++q) {
                const int j = jb + q * T;
                if (j >= m || ((inW[j >> 5] >> (j & 31)) & 1u)) continue;
                const double sc = (acc[q] - bq[q]) * ainv[j];
                if (sc > best) { best = sc; be = j; }
            }
        }
        block_argmax(best, be, slots_v + flip * nwaves, slots_i + flip * nwaves, nwaves);
        flip ^= 1;
        if (!(best > 0.0)) break;   // primal feasible: (x, y) optimal
        const int e = be;
        if (wave == 0) {
            ++piv;
            if (piv > P.maxit) status = kStatusMaxIter;
            const double gl = (lane < n) ? At[(size_t)lane * m + e] : 0.0;
            QrTrial t = qr_trial(F, k, gl);
            // entering row independent of the working rows?  (stricter than Lawson-Hanson's 0.01 test: a row that is
            // dependent up to rounding must take the ratio-test branch)
            const bool indep = k < n && t.independent && fabs(t.diag) > 1e-10 * (t.unorm + fabs(t.diag));
            if (status == kStatusOptimal && !indep) {
                // a_e = A_W' d: ratio test keeps y_W - t d >= 0
                if (lane < n) F.ab[lane] = t.qi;
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
                    t = qr_trial(F, k, gl);
                    if (!t.independent) status = kStatusMaxIter;   // numerical breakdown (not seen in practice)
                }
            }
            if (status == kStatusOptimal) {
                qr_add(F, k, t);
                if (lane == k) Wl = e;
                if (lane == 0) inW[e >> 5] |= 1u << (e & 31);
                ++k;
            }
            if (lane == 0) { ctl[3] = k; ctl[4] = status; }
        }
        __syncthreads();
        k = ctl[3];
        status = ctl[4];
    }

    // =========================================================== outputs (pnnls.py:137-145)
    const bool ok = (status == kStatusOptimal);
    const double nanv = __builtin_nan("");
    if (wave == 0) {
        const double zl = ok ? fmax(qr_coeff(F, k), 0.0) : 0.0;
        if (lane < k) { yb[lane] = zl; reinterpret_cast<int *>(tb)[lane] = Wl; }
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
    }
    __syncthreads();
    if (P.mult) {
        for (int j = tid; j < m; j += T) {
            double v = ok ? 0.0 : nanv;
            if (ok && ((inW[j >> 5] >> (j & 31)) & 1u))
                for (int l = 0; l < k; ++l)
                    if (reinterpret_cast<int *>(tb)[l] == j) v = yb[l];
            P.mult[prob * m + j] = v;
        }
    }
    if (P.active)
        for (int e = tid; e < nw; e += T) P.active[prob * nw + e] = abits[e];
}

}  // namespace pympc
