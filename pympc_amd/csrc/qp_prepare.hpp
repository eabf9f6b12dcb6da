// One-time factorisation of the constant part of a QP (hoists pnnls.py:210-214 out of the per-state loop):
//   L = chol(H), L^-1, H^-1 = L^-T L^-1, M = A L^-T, T = A H^-1,
//   m(x) = (b + T fu) + (T Hux - Ax) x,   u_unc(x) = -H^-1 fu - H^-1 Hux x
// written in the layout of ws_layout.hpp.  One workgroup per problem (blockIdx.x); the shared-matrix mpQP is the
// B = 1 case, the per-problem batch (pympc_qp_solve_batch) the p = 0 case.  Same formulation as the reference
// (explicit L^-1 and H^-1, /root/reference/pympc/optimization/solvers/pnnls.py:211-212) so the numbers agree to
// rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "invqr_wave.hpp"
#include "ws_layout.hpp"

namespace pympc {

struct PrepParams {
    int n, m, p;
    WsLayout L;
    const double *H;   long long h_stride;    // [n x n]
    const double *Hux; long long hux_stride;  // [n x p] (p may be 0)
    const double *Hxx; long long hxx_stride;  // [p x p]
    const double *fu;  long long fu_stride;   // [n]
    const double *fx;  long long fx_stride;   // [p]
    const double *gs;  long long g_stride;    // per-problem scalar (device) or null -> g_host
    double g_host;
    const double *A;   long long a_stride;    // [m x n]
    const double *Ax;  long long ax_stride;   // [m x p]
    const double *b;   long long b_stride;    // [m]
    double *ws;        long long ws_stride;
    int32_t *ws_status;
    int tile_rows;     // constraint rows staged in LDS at a time (= threads that work on rows)
    int skip_T;        // the caller never recovers the minimiser (no argmin output): T = A H^-1 is not written (40 % of the workspace)
};

// LDS doubles of the prepare kernel: three n x (n+1) matrices (L, L^-1, H^-1), fu and Hux, and two row tiles
// (A rows in, T rows out) of tile_rows x (n | 1) doubles each: global memory is only touched with coalesced accesses
__host__ __device__ inline size_t prep_lds_doubles(int n, int p, int tile_rows)
{
    return (size_t)3 * n * (n + 1) + n + (size_t)n * p + (size_t)2 * tile_rows * (n | 1) + 2;
}

__global__ void qp_prepare_kernel(PrepParams P)
{
    extern __shared__ double lds[];
    const int n = P.n, m = P.m, p = P.p, ld = n + 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const long long k = blockIdx.x;
    const double *H = P.H + k * P.h_stride;
    const double *Hux = P.Hux ? P.Hux + k * P.hux_stride : nullptr;
    const double *Hxx = P.Hxx ? P.Hxx + k * P.hxx_stride : nullptr;
    const double *fu = P.fu + k * P.fu_stride;
    const double *fx = P.fx ? P.fx + k * P.fx_stride : nullptr;
    const double *A = P.A + k * P.a_stride;
    const double *Ax = P.Ax ? P.Ax + k * P.ax_stride : nullptr;
    const double *b = P.b + k * P.b_stride;
    double *ws = P.ws + k * P.ws_stride;
    const WsLayout &L = P.L;
    const int mp = L.mp;

    double *Ls = lds;             // n x ld, lower Cholesky factor
    double *Li = Ls + n * ld;     // n x ld, L^-1
    double *Hi = Li + n * ld;     // n x ld, H^-1
    // flag in the dynamic block (a static __shared__ would shift the dynamic base off its 8-byte alignment)
    int &notpd = *reinterpret_cast<int *>(lds + prep_lds_doubles(n, p, P.tile_rows) - 2);
    if (tid == 0) notpd = 0;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        Ls[i * ld + j] = H[i * n + j];
        Li[i * ld + j] = 0.0;
    }
    __syncthreads();
    // right-looking Cholesky on the lower triangle (numpy.linalg.cholesky, pnnls.py:210)
    for (int j = 0; j < n; ++j) {
        const double d = Ls[j * ld + j];
        if (!(d > 0.0)) {
            if (tid == 0) notpd = 1;
            break;   // uniform: every thread read the same d
        }
        const double piv = sqrt(d);
        __syncthreads();
        for (int i = j + tid; i < n; i += nt) Ls[i * ld + j] = (i == j) ? piv : Ls[i * ld + j] / piv;
        __syncthreads();
        const int r = n - j - 1;
        for (int e = tid; e < r * r; e += nt) {
            const int i = j + 1 + e / r, c = j + 1 + e % r;
            if (c <= i) Ls[i * ld + c] -= Ls[i * ld + j] * Ls[c * ld + j];
        }
        __syncthreads();
    }
    __syncthreads();
    if (P.ws_status && tid == 0) P.ws_status[k] = notpd ? kStatusNotPD : 0;
    if (notpd) return;
    // L^-1, one column per thread (pnnls.py:211)
    for (int c = tid; c < n; c += nt) {
        Li[c * ld + c] = 1.0 / Ls[c * ld + c];
        for (int r = c + 1; r < n; ++r) {
            double s = 0.0;
            for (int q = c; q < r; ++q) s = fma(Ls[r * ld + q], Li[q * ld + c], s);
            Li[r * ld + c] = -s / Ls[r * ld + r];
        }
    }
    __syncthreads();
    // H^-1 = L^-T L^-1 (pnnls.py:212)
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        double s = 0.0;
        for (int q = (i > j ? i : j); q < n; ++q) s = fma(Li[q * ld + i], Li[q * ld + j], s);
        Hi[i * ld + j] = s;
    }
    __syncthreads();
    // rows of A: M = A L^-T (pnnls.py:213), T = A H^-1, m0, Gm (pnnls.py:214 split into its affine parts).
    // Tiles of tile_rows rows: coalesced copy of the rows into LDS (pitch n | 1: conflict-free row-per-thread access),
    // one row per thread, T tile written back coalesced.  Every short sum goes through dot8 (loads issued eight at a
    // time).  H^-1 is exactly symmetric (both triangles are the same sums of the same products), so T_j = H^-1 a_j
    // reads ROWS of Hi (unit stride).
    const int np = n | 1, TR = P.tile_rows;
    double *fus = Hi + n * ld;          // [n]
    double *Hs = fus + n;               // [n x p]
    double *As = Hs + n * p;            // [TR x np]
    double *Ts = As + (size_t)TR * np;  // [TR x np]
    for (int e = tid; e < n; e += nt) fus[e] = fu[e];
    for (int e = tid; e < n * p; e += nt) Hs[e] = Hux[e];
    for (int row0 = 0; row0 < mp; row0 += TR) {
        const int rows = min(TR, m - row0);          // real rows in this tile (<= 0: only padding columns)
        __syncthreads();
        for (int e = tid; e < rows * n; e += nt) As[(e / n) * np + (e % n)] = A[(size_t)row0 * n + e];
        __syncthreads();
        const int j = row0 + tid;
        if (tid < TR && j < mp) {
            if (j < m) {
                const double *aj = As + tid * np;
                double *Tj = Ts + tid * np;
                double cn = 0.0;
                for (int i = 0; i < n; ++i) {
                    const double mij = dot8(aj, 1, Li + i * ld, 1, 0, i + 1);
                    ws[L.Mt + (size_t)i * mp + j] = mij;
                    cn = fma(mij, mij, cn);
                }
                ws[L.cn + j] = cn;
                for (int i = 0; i < n; ++i) Tj[i] = dot8(aj, 1, Hi + i * ld, 1, 0, n);
                ws[L.m0 + j] = dot8(Tj, 1, fus, 1, 0, n, b[j]);
                for (int q = 0; q < p; ++q) ws[L.Gm + (size_t)q * mp + j] = dot8(Tj, 1, Hs + q, p, 0, n, -Ax[(size_t)j * p + q]);
            } else {
                for (int i = 0; i < n; ++i) ws[L.Mt + (size_t)i * mp + j] = 0.0;
                ws[L.m0 + j] = 1.0;
                ws[L.cn + j] = 0.0;
                for (int q = 0; q < p; ++q) ws[L.Gm + (size_t)q * mp + j] = 0.0;
            }
        }
        __syncthreads();
        if (!P.skip_T)
            for (int e = tid; e < rows * n; e += nt) ws[L.T + (size_t)row0 * n + e] = Ts[(e / n) * np + (e % n)];
    }
    __syncthreads();
    // unconstrained minimiser u_unc(x) = u0 + Ku x
    for (int i = tid; i < n; i += nt) {
        double s = 0.0;
        for (int q = 0; q < n; ++q) s = fma(Hi[i * ld + q], fu[q], s);
        ws[L.u0 + i] = -s;
        for (int c = 0; c < p; ++c) {
            double t = 0.0;
            for (int q = 0; q < n; ++q) t = fma(Hi[i * ld + q], Hux[q * p + c], t);
            ws[L.Ku + i * p + c] = -t;
        }
    }
    __syncthreads();   // u0, Ku (global, written by this workgroup) are read back below
    // optimal value of the unconstrained problem: V_unc(x) = .5 f(x)'u_unc(x) + .5 x'Hxx x + fx'x + g, f(x) = Hux x + fu
    for (int e = tid; e < p * p; e += nt) {
        const int r = e / p, c = e % p;
        double s = Hxx[e];
        for (int i = 0; i < n; ++i) s = fma(Hux[i * p + r], ws[L.Ku + i * p + c], s);
        ws[L.Sxx + e] = s;
    }
    for (int r = tid; r < p; r += nt) {
        double s = fx[r];
        for (int i = 0; i < n; ++i) s = fma(Hux[i * p + r], ws[L.u0 + i], s);
        ws[L.sx + r] = s;
    }
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s = fma(fu[i], ws[L.u0 + i], s);
        ws[L.s0] = fma(0.5, s, P.gs ? P.gs[k * P.g_stride] : P.g_host);
    }
    for (size_t e = L.s0 + 1 + tid; e < L.total; e += nt) ws[e] = 0.0;
}

// The same factorisation for MANY small problems (n <= 16, one wavefront per problem): see the row loop below.
__host__ __device__ inline size_t prep_wave_lds_doubles(int n, int p) { return (size_t)3 * n * (n + 1) + n + (size_t)n * p + 2; }

template <int NN>
__global__ void __launch_bounds__(64) qp_prepare_wave_kernel(PrepParams P)
{
    extern __shared__ double lds[];
    const int n = P.n, m = P.m, p = P.p, ld = n + 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const long long k = blockIdx.x;
    const double *H = P.H + k * P.h_stride;
    const double *Hux = P.Hux ? P.Hux + k * P.hux_stride : nullptr;
    const double *Hxx = P.Hxx ? P.Hxx + k * P.hxx_stride : nullptr;
    const double *fu = P.fu + k * P.fu_stride;
    const double *fx = P.fx ? P.fx + k * P.fx_stride : nullptr;
    const double *A = P.A + k * P.a_stride;
    const double *Ax = P.Ax ? P.Ax + k * P.ax_stride : nullptr;
    const double *b = P.b + k * P.b_stride;
    double *ws = P.ws + k * P.ws_stride;
    const WsLayout &L = P.L;
    const int mp = L.mp;

    double *Ls = lds;             // n x ld, lower Cholesky factor
    double *Li = Ls + n * ld;     // n x ld, L^-1
    double *Hi = Li + n * ld;     // n x ld, H^-1
    // flag in the dynamic block (a static __shared__ would shift the dynamic base off its 8-byte alignment)
    int &notpd = *reinterpret_cast<int *>(lds + prep_wave_lds_doubles(n, p) - 2);
    if (tid == 0) notpd = 0;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        Ls[i * ld + j] = H[i * n + j];
        Li[i * ld + j] = 0.0;
    }
    __syncthreads();
    // right-looking Cholesky on the lower triangle (numpy.linalg.cholesky, pnnls.py:210)
    for (int j = 0; j < n; ++j) {
        const double d = Ls[j * ld + j];
        if (!(d > 0.0)) {
            if (tid == 0) notpd = 1;
            break;   // uniform: every thread read the same d
        }
        const double piv = sqrt(d);
        __syncthreads();
        for (int i = j + tid; i < n; i += nt) Ls[i * ld + j] = (i == j) ? piv : Ls[i * ld + j] / piv;
        __syncthreads();
        const int r = n - j - 1;
        for (int e = tid; e < r * r; e += nt) {
            const int i = j + 1 + e / r, c = j + 1 + e % r;
            if (c <= i) Ls[i * ld + c] -= Ls[i * ld + j] * Ls[c * ld + j];
        }
        __syncthreads();
    }
    __syncthreads();
    if (P.ws_status && tid == 0) P.ws_status[k] = notpd ? kStatusNotPD : 0;
    if (notpd) return;
    // L^-1, one column per thread (pnnls.py:211)
    for (int c = tid; c < n; c += nt) {
        Li[c * ld + c] = 1.0 / Ls[c * ld + c];
        for (int r = c + 1; r < n; ++r) {
            double s = 0.0;
            for (int q = c; q < r; ++q) s = fma(Ls[r * ld + q], Li[q * ld + c], s);
            Li[r * ld + c] = -s / Ls[r * ld + r];
        }
    }
    __syncthreads();
    // H^-1 = L^-T L^-1 (pnnls.py:212)
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        double s = 0.0;
        for (int q = (i > j ? i : j); q < n; ++q) s = fma(Li[q * ld + i], Li[q * ld + j], s);
        Hi[i * ld + j] = s;
    }
    __syncthreads();
    // rows of A, one per lane, held in registers (n <= NN): M = A L^-T (pnnls.py:213), T = A H^-1, m0, Gm (pnnls.py:214 split
    // into its affine parts).  Same sums in the same order as qp_prepare_kernel's dot8 calls (the zeros above the diagonal
    // of L^-1 add nothing), so both kernels write the same bits.  No row tiles in LDS: 2.7 KB instead of 14 KB per
    // wavefront for n = 10, i.e. the wavefronts per CU are no longer limited by LDS.
    double *fus = Hi + n * ld;          // [n]
    double *Hs = fus + n;               // [n x p]
    for (int e = tid; e < n; e += nt) fus[e] = fu[e];
    for (int e = tid; e < n * p; e += nt) Hs[e] = Hux[e];
    __syncthreads();
    for (int row0 = 0; row0 < mp; row0 += 64) {
        const int j = row0 + tid;
        if (j < m) {
            double a[NN], t[NN];
#pragma unroll
            for (int q = 0; q < NN; ++q) a[q] = (q < n) ? A[(size_t)j * n + q] : 0.0;
            double cn = 0.0;
            for (int i = 0; i < n; ++i) {
                double sm = 0.0;
#pragma unroll
                for (int q = 0; q < NN; ++q) sm = fma(a[q], (q < n) ? Li[i * ld + q] : 0.0, sm);
                ws[L.Mt + (size_t)i * mp + j] = sm;
                cn = fma(sm, sm, cn);
            }
            ws[L.cn + j] = cn;
#pragma unroll
            for (int i = 0; i < NN; ++i) {
                double st = 0.0;
                if (i < n) {
#pragma unroll
                    for (int q = 0; q < NN; ++q) st = fma(a[q], (q < n) ? Hi[i * ld + q] : 0.0, st);
                }
                t[i] = st;
            }
            double s0 = b[j];
#pragma unroll
            for (int i = 0; i < NN; ++i)
                if (i < n) s0 = fma(t[i], fus[i], s0);
            ws[L.m0 + j] = s0;
            for (int q = 0; q < p; ++q) {
                double sg = -Ax[(size_t)j * p + q];
#pragma unroll
                for (int i = 0; i < NN; ++i)
                    if (i < n) sg = fma(t[i], Hs[i * p + q], sg);
                ws[L.Gm + (size_t)q * mp + j] = sg;
            }
            if (!P.skip_T) {
#pragma unroll
                for (int i = 0; i < NN; ++i)
                    if (i < n) ws[L.T + (size_t)j * n + i] = t[i];
            }
        } else if (j < mp) {
            for (int i = 0; i < n; ++i) ws[L.Mt + (size_t)i * mp + j] = 0.0;
            ws[L.m0 + j] = 1.0;
            ws[L.cn + j] = 0.0;
            for (int q = 0; q < p; ++q) ws[L.Gm + (size_t)q * mp + j] = 0.0;
        }
    }
    __syncthreads();
    // unconstrained minimiser u_unc(x) = u0 + Ku x
    for (int i = tid; i < n; i += nt) {
        double s = 0.0;
        for (int q = 0; q < n; ++q) s = fma(Hi[i * ld + q], fu[q], s);
        ws[L.u0 + i] = -s;
        for (int c = 0; c < p; ++c) {
            double t = 0.0;
            for (int q = 0; q < n; ++q) t = fma(Hi[i * ld + q], Hux[q * p + c], t);
            ws[L.Ku + i * p + c] = -t;
        }
    }
    __syncthreads();   // u0, Ku (global, written by this workgroup) are read back below
    // optimal value of the unconstrained problem: V_unc(x) = .5 f(x)'u_unc(x) + .5 x'Hxx x + fx'x + g, f(x) = Hux x + fu
    for (int e = tid; e < p * p; e += nt) {
        const int r = e / p, c = e % p;
        double s = Hxx[e];
        for (int i = 0; i < n; ++i) s = fma(Hux[i * p + r], ws[L.Ku + i * p + c], s);
        ws[L.Sxx + e] = s;
    }
    for (int r = tid; r < p; r += nt) {
        double s = fx[r];
        for (int i = 0; i < n; ++i) s = fma(Hux[i * p + r], ws[L.u0 + i], s);
        ws[L.sx + r] = s;
    }
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s = fma(fu[i], ws[L.u0 + i], s);
        ws[L.s0] = fma(0.5, s, P.gs ? P.gs[k * P.g_stride] : P.g_host);
    }
    for (size_t e = L.s0 + 1 + tid; e < L.total; e += nt) ws[e] = 0.0;
}


// Gr[i][j] = sum_r Mt[r][i] Mt[r][j] (ws_layout.hpp): one thread per entry, rows of Mt read coalesced over j.
__global__ void __launch_bounds__(64) qp_gram_kernel(int n, int mp, const double *__restrict__ Mt, double *__restrict__ Gr)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y;
    double acc = 0.0;
    for (int r = 0; r < n; ++r) acc = fma(Mt[(size_t)r * mp + i], Mt[(size_t)r * mp + j], acc);
    Gr[(size_t)i * mp + j] = (i == j) ? kGramDiag : acc;   // (the diagonal is never read as a norm: cn holds those)
}

}  // namespace pympc
