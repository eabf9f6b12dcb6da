// Raw NNLS and PNNLS entry points on the wave-per-problem kernel (qp_wave.hpp).
//
//   pympc_nnls_batch   scipy.optimize.nnls(A, b) (third party; called at /root/reference/pympc/optimization/solvers/
//                      pnnls.py:36,218): min |A x - b|, x >= 0 for matrices of up to 64 rows and 2048 columns.
//   pympc_pnnls_batch  pnnls(A, B, c) of pnnls.py:5-39: min |A v + B u - c|, v >= 0, u free.  The reference projects with
//                      the dense (rows x rows) matrix I - B pinv(B) (pnnls.py:30-33); here u is eliminated by a Householder
//                      QR of B with column pivoting (rank k): with Q'B = [R; 0] the problem splits into an NNLS in v on
//                      rows k.. of Q'[A | c] (the same residual: |(I - B pinv B) z| = |(Q'z)[k:]|) and the triangular
//                      system R u = -(Q'(A v - c))[:k] (minimum-norm solution when B is rank deficient, as pinv gives).
//
// nnls_prepare_kernel writes a general matrix into the prepared-workspace layout of ws_layout.hpp (rows 0 .. R-2 as
// "Mt", the last row as "m0", both negated as the QP matrix [-M'; -m'] is; column norms in "cn"), so that the solve
// kernel runs unchanged with SolveParams::rhs / raw set.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "invqr_wave.hpp"
#include "ws_layout.hpp"

namespace pympc {

struct NnlsPrepParams {
    int rows, cols;            // of the caller's matrix; the kernel sees R = max(rows, 2) rows (a zero row is appended to a 1-row matrix)
    WsLayout L;                // ws_layout(R - 1, cols, 0)
    const double *A;           // [B x rows x cols]
    long long a_stride;
    const double *b;           // [B x rows]
    long long b_stride;
    const int32_t *skip_rows;  // per problem: the first skip_rows[k] rows count as zero (PNNLS: rows of the eliminated u), or null
    double *ws;                // per-problem prepared workspace, stride L.total
    double *rhs;               // [B x R] right-hand side for the solve kernel
};

// one workgroup per problem, threads over columns
__global__ void __launch_bounds__(256) nnls_prepare_kernel(NnlsPrepParams P)
{
    const long long k = blockIdx.x;
    const int rows = P.rows, cols = P.cols, R = rows < 2 ? 2 : rows, n = R - 1, mp = P.L.mp;
    const double *A = P.A + k * P.a_stride;
    double *ws = P.ws + k * (long long)P.L.total;
    const int skip = P.skip_rows ? P.skip_rows[k] : 0;
    for (int j = threadIdx.x; j < mp; j += blockDim.x) {
        double cn = 0.0;
        for (int i = 0; i < n; ++i) {
            const double a = (j < cols && i < rows && i >= skip) ? A[(size_t)i * cols + j] : 0.0;
            ws[P.L.Mt + (size_t)i * mp + j] = -a;
            cn = fma(a, a, cn);
        }
        const double last = (j < cols && n < rows && n >= skip) ? A[(size_t)n * cols + j] : 0.0;
        ws[P.L.m0 + j] = j < cols ? -last : 1.0;
        ws[P.L.cn + j] = cn;
    }
    for (size_t e = P.L.T + threadIdx.x; e < P.L.total; e += blockDim.x) ws[e] = 0.0;
    if ((int)threadIdx.x < R) {
        const int i = threadIdx.x;
        P.rhs[k * R + i] = (i < rows && i >= skip) ? P.b[k * P.b_stride + i] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------------ PNNLS
struct PnnlsParams {
    int rows, cols, nu;
    const double *A;   // [B x rows x cols]
    const double *Bm;  // [B x rows x nu]
    const double *c;   // [B x rows]
    double *QA;        // [B x rows x cols]  Q'A
    double *Qc;        // [B x rows]         Q'c
    double *Rf;        // [B x rows x nu]    Q'B with permuted columns (upper trapezoid in its first `rank` rows)
    int32_t *perm;     // [B x nu]           column j of Rf is column perm[j] of B
    int32_t *rank;     // [B]
    const double *v;   // [B x cols] (recover kernel)
    double *u;         // [B x nu]   (recover kernel)
    double *scr;       // recover kernel, nu > 64: [B x (nu (nu + 1) + 2 nu)] scratch in device memory (else null: LDS)
};

constexpr double kPnnlsRankTol = 1e-13;   // relative column norm below which the rest of B counts as dependent (pinv: rcond 1e-15 on singular values)

__host__ __device__ inline size_t pnnls_lds_doubles(int rows, int nu) { return (size_t)rows * (nu | 1) + rows + 8; }

// Householder QR of B with column pivoting, applied to [A | c]: one wavefront per problem; lane <-> row for the
// reflector, lane <-> column for its application (columns of A read coalesced along a row).
__global__ void __launch_bounds__(64) pnnls_reduce_kernel(PnnlsParams P)
{
    extern __shared__ double lds[];
    const long long k = blockIdx.x;
    const int rows = P.rows, cols = P.cols, nu = P.nu, ld = nu | 1, lane = threadIdx.x;
    double *Bs = lds, *vv = lds + (size_t)rows * ld;
    const double *A = P.A + k * (long long)rows * cols;
    double *QA = P.QA + k * (long long)rows * cols;
    double *Qc = P.Qc + k * (long long)rows;
    for (int e = lane; e < rows * nu; e += 64) Bs[(e / nu) * ld + (e % nu)] = P.Bm[k * (long long)rows * nu + e];
    for (size_t e = lane; e < (size_t)rows * cols; e += 64) QA[e] = A[e];
    if (lane < rows) Qc[lane] = P.c[k * (long long)rows + lane];
    int pj = lane;   // lane < nu: original index of the column now at position `lane`
    __syncthreads();
    double norm0 = 0.0;
    int rank = 0;
    const int steps = nu < rows ? nu : rows;
    for (int j = 0; j < steps; ++j) {
        // remaining column of largest norm (rows j ..)
        double s = -1.0;
        if (lane >= j && lane < nu) {
            s = 0.0;
            for (int i = j; i < rows; ++i) s = fma(Bs[i * ld + lane], Bs[i * ld + lane], s);
        }
        const double smax = wmax(s);
        if (j == 0) norm0 = smax;
        if (!(smax > kPnnlsRankTol * kPnnlsRankTol * norm0) || !(smax > 0.0)) break;
        const int piv = wmin_int(s == smax ? lane : 0x7fffffff);
        if (piv != j) {   // swap columns j and piv
            if (lane < rows) {
                const double t = Bs[lane * ld + j];
                Bs[lane * ld + j] = Bs[lane * ld + piv];
                Bs[lane * ld + piv] = t;
            }
            const int pa = lane_i32(pj, j), pb = lane_i32(pj, piv);
            if (lane == j) pj = pb;
            if (lane == piv) pj = pa;
        }
        __syncthreads();
        // reflector of column j: v = x - alpha e_j, H = I - 2 v v' / v'v
        const double xi = (lane >= j && lane < rows) ? Bs[lane * ld + j] : 0.0;
        const double sigma = sqrt(wsum(xi * xi));
        const double xj = lane_f64(xi, j);
        const double alpha = xj > 0.0 ? -sigma : sigma;
        const double vi = (lane == j) ? xi - alpha : xi;
        const double vtv = wsum(vi * vi);
        if (lane < rows) vv[lane] = vi;
        __syncthreads();
        if (vtv > 0.0) {
            const double f = 2.0 / vtv;
            if (lane > j && lane < nu) {   // remaining columns of B
                double d = 0.0;
                for (int i = j; i < rows; ++i) d = fma(vv[i], Bs[i * ld + lane], d);
                d *= f;
                for (int i = j; i < rows; ++i) Bs[i * ld + lane] = fma(-d, vv[i], Bs[i * ld + lane]);
            }
            for (int cidx = lane; cidx <= cols; cidx += 64) {   // columns of A, then c
                double *col = cidx < cols ? QA + cidx : Qc;
                const size_t st = cidx < cols ? (size_t)cols : 1;
                double d = 0.0;
                for (int i = j; i < rows; ++i) d = fma(vv[i], col[i * st], d);
                d *= f;
                for (int i = j; i < rows; ++i) col[i * st] = fma(-d, vv[i], col[i * st]);
            }
        }
        __syncthreads();
        if (lane == j) Bs[j * ld + j] = alpha;
        if (lane > j && lane < rows) Bs[lane * ld + j] = 0.0;
        __syncthreads();
        rank = j + 1;
    }
    for (int e = lane; e < rows * nu; e += 64) P.Rf[k * (long long)rows * nu + e] = Bs[(e / nu) * ld + (e % nu)];
    if (lane < nu) P.perm[k * nu + lane] = pj;
    if (lane == 0) P.rank[k] = rank;
}

// u from v: d = (Q'c - Q'A v)[:rank], then the minimum-norm solution of Rt u_perm = d with Rt = Rf[:rank] (rank x nu).
// One wavefront per problem; the small dense solves run on lane 0 (nu <= 64).
__global__ void __launch_bounds__(64) pnnls_recover_kernel(PnnlsParams P)
{
    extern __shared__ double lds[];
    const long long k = blockIdx.x;
    const int rows = P.rows, cols = P.cols, nu = P.nu, lane = threadIdx.x;
    const int rank = P.rank[k];
    const double *QA = P.QA + k * (long long)rows * cols;
    const double *Rf = P.Rf + k * (long long)rows * nu;
    // up to 64 free variables the small systems live in LDS, beyond in the caller's scratch (pitch gp)
    const int NU = P.scr ? nu : 64, gp = NU + 1;
    double *d = P.scr ? P.scr + k * ((long long)nu * (nu + 1) + 2 * nu) : lds, *G = d + NU, *t = G + (size_t)NU * gp;
    for (int r = lane; r < rank; r += 64) {
        double s = P.Qc[k * (long long)rows + r];
        for (int j = 0; j < cols; ++j) s = fma(-QA[(size_t)r * cols + j], P.v[k * (long long)cols + j], s);
        d[r] = s;
    }
    for (int j = lane; j < nu; j += 64) P.u[k * nu + j] = 0.0;
    __syncthreads();
    if (lane != 0 || rank == 0) return;
    const int32_t *perm = P.perm + k * nu;
    double *u = P.u + k * nu;
    if (rank == nu) {   // square upper triangle: back substitution
        for (int i = nu - 1; i >= 0; --i) {
            double s = d[i];
            for (int j = i + 1; j < nu; ++j) s = fma(-Rf[i * nu + j], t[j], s);
            t[i] = s / Rf[i * nu + i];
        }
        for (int j = 0; j < nu; ++j) u[perm[j]] = t[j];
        return;
    }
    // minimum norm: u_perm = Rt' (Rt Rt')^-1 d
    for (int a = 0; a < rank; ++a)
        for (int b = 0; b <= a; ++b) {
            double s = 0.0;
            for (int j = 0; j < nu; ++j) s = fma(Rf[a * nu + j], Rf[b * nu + j], s);
            G[a * gp + b] = s;
        }
    for (int a = 0; a < rank; ++a) {   // Cholesky, in place (lower)
        for (int b = 0; b <= a; ++b) {
            double s = G[a * gp + b];
            for (int c2 = 0; c2 < b; ++c2) s = fma(-G[a * gp + c2], G[b * gp + c2], s);
            G[a * gp + b] = (a == b) ? sqrt(s) : s / G[b * gp + b];
        }
    }
    for (int a = 0; a < rank; ++a) {
        double s = d[a];
        for (int b = 0; b < a; ++b) s = fma(-G[a * gp + b], t[b], s);
        t[a] = s / G[a * gp + a];
    }
    for (int a = rank - 1; a >= 0; --a) {
        double s = t[a];
        for (int b = a + 1; b < rank; ++b) s = fma(-G[b * gp + a], t[b], s);
        t[a] = s / G[a * gp + a];
    }
    for (int j = 0; j < nu; ++j) {
        double s = 0.0;
        for (int a = 0; a < rank; ++a) s = fma(Rf[a * nu + j], t[a], s);
        u[perm[j]] = s;
    }
}

}  // namespace pympc
