// Strictly convex QPs of 64 .. 1023 variables: pnnls.quadratic_program (/root/reference/pympc/optimization/solvers/
// pnnls.py:149-241) around nnls_block_kernel (nnls_block.hpp), one workgroup of 256, 512 or 1024 threads per problem.
//
// The wave path (qp_prepare.hpp, qp_wave.hpp) keeps the three n x n factors of a problem in LDS and one row of the
// (n + 1)-row NNLS matrix per lane, which ends at n = 63.  Here the factors live in device memory (they stay in L2: 2 x
// 255 x 255 doubles = 1 MB) and every loop over them runs thread <-> column, coalesced:
//
//   qp_block_prepare_kernel   L = chol(H) right-looking (pnnls.py:210; numpy raises LinAlgError for a Hessian that is
//                             not positive definite: status PYMPC_NOT_PD), L^-1 by forward substitution a column per
//                             thread (:211), v = L^-1 f, H^-1 f = L^-T v, A transposed once, then the NNLS matrix
//                             [-M'; -m'] with M = A L^-T, m = b + A H^-1 f (:213-217) a constraint row per thread;
//   nnls_block_kernel         y, r = nnls([-M'; -m'], (0 .. 0, 1)) (:218);
//   qp_block_recover_kernel   r <= tol: infeasible (:230, every output NaN); else lam = y / (1 + m'y) (:231),
//                             x = -H^-1 (f + A' lam) = -L^-T L^-1 (f + A' lam) (:233), min = x'Hx / 2 + f'x (:234),
//                             active = lam > tol (:235).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nnls_block.hpp"

namespace pympc {

constexpr int kQpBlockNMax = kBlkRowsMax - 1;   // n + 1 rows of the NNLS matrix, one per thread

struct QpBlockLayout {   // per-problem workspace, offsets in doubles
    size_t L, Li, At, An, mv, total;
};
__host__ __device__ inline QpBlockLayout qp_block_layout(int n, int m)
{
    QpBlockLayout q;
    size_t o = 0;
    q.L = o; o += (size_t)n * n;
    q.Li = o; o += (size_t)n * n;
    q.At = o; o += (size_t)n * m;          // A transposed: At[k*m + r] = A[r][k]
    q.An = o; o += (size_t)(n + 1) * m;    // the NNLS matrix, row-major (n + 1) x m
    q.mv = o; o += (size_t)m;
    q.total = (o + 3) & ~(size_t)3;
    return q;
}

struct QpBlockParams {
    int n, m;
    const double *H; long long h_stride;   // [n x n] per problem or shared (stride 0)
    const double *f;                        // [B x n]
    const double *A; long long a_stride;   // [m x n] per problem or shared
    const double *b;                        // [B x m]
    double tol;
    double *ws;                             // B x qp_block_layout(n, m).total
    // scratch behind the per-problem workspaces
    double *bn;                             // [n + 1]   right-hand side (0 .. 0, 1) of every problem (written by workgroup 0)
    double *y;                              // [B x m]   NNLS solution
    double *rn;                             // [B]       NNLS residual norm
    int32_t *nnls_status;                   // [B]
    int32_t *nnls_iters;                    // [B x 2]
    int32_t *notpd;                         // [B]
    // outputs (any but status may be null)
    double *argmin, *vmin, *mult;
    uint32_t *active;
    int32_t *status, *iters;
};

template <int T>
__global__ void __launch_bounds__(T) qp_block_prepare_kernel(QpBlockParams P)
{
    constexpr int kBlkThreads = T;
    __shared__ double fs[kBlkThreads], vs[kBlkThreads], hs[kBlkThreads];
    const int t = threadIdx.x, n = P.n, m = P.m;
    const long long k = blockIdx.x;
    const QpBlockLayout Q = qp_block_layout(n, m);
    const double *H = P.H + k * P.h_stride, *A = P.A + k * P.a_stride, *f = P.f + k * (long long)n, *b = P.b + k * (long long)m;
    double *ws = P.ws + k * (long long)Q.total;
    double *L = ws + Q.L, *Li = ws + Q.Li, *At = ws + Q.At, *An = ws + Q.An, *mv = ws + Q.mv;
    for (int e = t; e < n * n; e += kBlkThreads) L[e] = H[e];
    for (size_t e = t; e < (size_t)n * m; e += kBlkThreads) {
        const int r = (int)(e / n), kx = (int)(e - (size_t)r * n);
        At[(size_t)kx * m + r] = A[e];
    }
    fs[t] = t < n ? f[t] : 0.0;
    if (k == 0 && t <= n) P.bn[t] = t == n ? 1.0 : 0.0;
    __syncthreads();
    // Cholesky, right-looking: after step j column j is final and the columns behind it carry the sums up to j
    int bad = 0;
    for (int j = 0; j < n; ++j) {
        const double d = L[(size_t)j * n + j];
        if (!(d > 0.0)) { bad = 1; break; }
        const double s = sqrt(d);
        __syncthreads();   // everybody has read the pivot
        if (t == j) L[(size_t)j * n + j] = s;
        if (t > j && t < n) L[(size_t)t * n + j] = L[(size_t)t * n + j] / s;
        __syncthreads();
        if (t > j && t < n) {   // (eight rows' loads in flight: one workgroup per problem, nothing else hides the latency)
            const double lcj = L[(size_t)t * n + j];
            int i = t;
            for (; i + 8 <= n; i += 8) {
                double lj[8], lt[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { lj[u] = L[(size_t)(i + u) * n + j]; lt[u] = L[(size_t)(i + u) * n + t]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) L[(size_t)(i + u) * n + t] = fma(-lj[u], lcj, lt[u]);
            }
            for (; i < n; ++i) L[(size_t)i * n + t] = fma(-L[(size_t)i * n + j], lcj, L[(size_t)i * n + t]);
        }
        __syncthreads();
    }
    if (t == 0) P.notpd[k] = bad;
    if (bad) {   // no NNLS matrix for this problem: a zero one, on which the NNLS kernel never iterates
        for (size_t e = t; e < (size_t)(n + 1) * m; e += kBlkThreads) An[e] = 0.0;
        return;
    }
    // L^-1, column t by forward substitution
    if (t < n) {
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            if (i >= t) {
                s = i == t ? 1.0 : 0.0;
                int q = t;
                for (; q + 8 <= i; q += 8) {
                    double lv[8], zv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { lv[u] = L[(size_t)i * n + q + u]; zv[u] = Li[(size_t)(q + u) * n + t]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = fma(-lv[u], zv[u], s);
                }
                for (; q < i; ++q) s = fma(-L[(size_t)i * n + q], Li[(size_t)q * n + t], s);
                s = s / L[(size_t)i * n + i];
            }
            Li[(size_t)i * n + t] = s;
        }
    }
    __syncthreads();
    // v = L^-1 f,  H^-1 f = L^-T v
    {
        double s = 0.0;
        if (t < n)
            for (int q = 0; q <= t; ++q) s = fma(Li[(size_t)t * n + q], fs[q], s);
        vs[t] = s;
    }
    __syncthreads();
    {
        double s = 0.0;
        if (t < n)
            for (int q = t; q < n; ++q) s = fma(Li[(size_t)q * n + t], vs[q], s);
        hs[t] = s;
    }
    __syncthreads();
    // The NNLS matrix, a constraint row per thread and kCb columns of M at a time: the kCb rows of L^-1 staged in LDS (read
    // as broadcasts), the row of A loaded eight entries at a time; the entries of L^-1 above its diagonal are stored
    // zeros, which add nothing: the sums are those of the loop q <= c, in the same order.
    constexpr int kCb = T >= 1024 ? 4 : 8;
    __shared__ double lib[kCb * T];
    for (int c0 = 0; c0 < n; c0 += kCb) {
        const int cl = c0 + kCb <= n ? c0 + kCb - 1 : n - 1, nq = cl + 1;
        __syncthreads();
        for (int e = t; e < kCb * nq; e += kBlkThreads) {
            const int u = e / nq, q = e - u * nq;
            lib[u * T + q] = c0 + u < n ? Li[(size_t)(c0 + u) * n + q] : 0.0;
        }
        __syncthreads();
        for (int r = t; r < m; r += kBlkThreads) {
            double sv[kCb];
#pragma unroll
            for (int u = 0; u < kCb; ++u) sv[u] = 0.0;
            int q = 0;
            for (; q + 8 <= nq; q += 8) {
                double av[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) av[w] = At[(size_t)(q + w) * m + r];
#pragma unroll
                for (int w = 0; w < 8; ++w)
#pragma unroll
                    for (int u = 0; u < kCb; ++u) sv[u] = fma(av[w], lib[u * T + q + w], sv[u]);
            }
            for (; q < nq; ++q) {
                const double av = At[(size_t)q * m + r];
#pragma unroll
                for (int u = 0; u < kCb; ++u) sv[u] = fma(av, lib[u * T + q], sv[u]);
            }
#pragma unroll
            for (int u = 0; u < kCb; ++u)
                if (c0 + u < n) An[(size_t)(c0 + u) * m + r] = -sv[u];
        }
    }
    for (int r = t; r < m; r += kBlkThreads) {
        double s = 0.0;
        for (int q = 0; q < n; ++q) s = fma(At[(size_t)q * m + r], hs[q], s);
        s += b[r];
        mv[r] = s;
        An[(size_t)n * m + r] = -s;
    }
}

template <int T>
__global__ void __launch_bounds__(T) qp_block_recover_kernel(QpBlockParams P)
{
    constexpr int kBlkThreads = T, kBlkWaves = T / 64;
    __shared__ double gs[kBlkThreads], wv[kBlkThreads], xs[kBlkThreads], slots[32];
    const int t = threadIdx.x, n = P.n, m = P.m;
    const long long k = blockIdx.x;
    const QpBlockLayout Q = qp_block_layout(n, m);
    const double *H = P.H + k * P.h_stride, *f = P.f + k * (long long)n;
    const double *ws = P.ws + k * (long long)Q.total;
    const double *Li = ws + Q.Li, *At = ws + Q.At, *mv = ws + Q.mv;
    double *y = P.y + k * (long long)m;
    BlockReduce<T / 64> red{slots, 0};
    const int words = (m + 31) / 32;
    int st = 0;
    if (P.notpd[k]) st = 4;
    else if (P.nnls_status[k] == 3) st = 3;
    else if (!(P.rn[k] > P.tol)) st = 1;
    if (t == 0) {
        P.status[k] = st;
        if (P.iters) { P.iters[2 * k] = st == 4 ? 0 : P.nnls_iters[2 * k]; P.iters[2 * k + 1] = st == 4 ? 0 : P.nnls_iters[2 * k + 1]; }
    }
    if (st != 0) {
        const double nanv = __builtin_nan("");
        if (P.argmin && t < n) P.argmin[k * (long long)n + t] = nanv;
        if (P.vmin && t == 0) P.vmin[k] = nanv;
        if (P.mult) for (int r = t; r < m; r += kBlkThreads) P.mult[k * (long long)m + r] = nanv;
        if (P.active) for (int w = t; w < words; w += kBlkThreads) P.active[k * (long long)words + w] = 0u;
        return;
    }
    double part = 0.0;
    for (int r = t; r < m; r += kBlkThreads) part = fma(mv[r], y[r], part);
    const double den = 1.0 + red.sum(part);
    for (int r = t; r < m; r += kBlkThreads) {
        const double lam = y[r] / den;
        y[r] = lam;
        if (P.mult) P.mult[k * (long long)m + r] = lam;
    }
    __syncthreads();
    if (P.active)
        for (int w = t; w < words; w += kBlkThreads) {
            uint32_t bits = 0u;
            for (int q = 0; q < 32 && 32 * w + q < m; ++q) bits |= (y[32 * w + q] > P.tol ? 1u : 0u) << q;
            P.active[k * (long long)words + w] = bits;
        }
    // g = f + A' lam: a wavefront per entry, lanes over the constraint rows
    for (int j = t >> 6; j < n; j += kBlkWaves) {
        double s = 0.0;
        for (int r = t & 63; r < m; r += 64) s = fma(At[(size_t)j * m + r], y[r], s);
        s = wsum(s);
        if ((t & 63) == 0) gs[j] = f[j] + s;
    }
    __syncthreads();
    {
        double s = 0.0;
        if (t < n)
            for (int q = 0; q <= t; ++q) s = fma(Li[(size_t)t * n + q], gs[q], s);
        wv[t] = s;
    }
    __syncthreads();
    double xi = 0.0;
    if (t < n) {
        double s = 0.0;
        for (int q = t; q < n; ++q) s = fma(Li[(size_t)q * n + t], wv[q], s);
        xi = -s;
        if (P.argmin) P.argmin[k * (long long)n + t] = xi;
    }
    xs[t] = xi;
    __syncthreads();
    double hx = 0.0;
    if (t < n)
        for (int q = 0; q < n; ++q) hx = fma(xs[q], H[(size_t)q * n + t], hx);
    const double quad = red.sum(hx * xi);
    const double lin = red.sum(t < n ? f[t] * xi : 0.0);
    if (P.vmin && t == 0) P.vmin[k] = 0.5 * quad + lin;
}

}  // namespace pympc
