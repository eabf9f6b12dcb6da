// Workspace layout of one prepared QP (shared by host launch code and device kernels).
//
// The constant part of pnnls.quadratic_program (/root/reference/pympc/optimization/solvers/pnnls.py:210-214)
// is factorised once by qp_prepare_kernel and kept in HBM in this layout, all fp64:
//
//   Mt  [n  x mp]   Mt[i][j] = M[j][i],  M = Au L^-T            (rows 0..n-1 of the NNLS matrix, negated)
//   m0  [mp]        m(x) = m0 + Gm' x  = b(x) + Au Huu^-1 f(x)   (last row of the NNLS matrix, negated)
//   Gm  [p  x mp]
//   cn  [mp]        cn[j] = |M_j|^2: with m_j(x)^2 the squared norm of column j of the NNLS matrix (pivot rule of qp_wave.hpp)
//   T   [m  x n]    T = Au Huu^-1   (row j = (Huu^-1 a_j)')      (primal recovery, pnnls.py:233)
//   u0  [n], Ku [n x p]   unconstrained minimiser u_unc(x) = u0 + Ku x = -Huu^-1 f(x)
//   Sxx [p x p], sx [p], s0 [1]   optimal value of the UNCONSTRAINED problem, V_unc(x) = .5 x'Sxx x + sx'x + s0 with
//                         Sxx = Hxx + Hux'Ku, sx = fx + Hux'u0, s0 = g + .5 fu'u0.  The kernel lifts it to the value
//                         of pnnls.py:234 + parametric_programs.py:177 with the NNLS solution alone:
//                         V = V_unc(x) + .5 |M'lambda|^2 = V_unc(x) - .5 m'y / (1 + m'y)   (see qp_wave.hpp)
//   zeros up to `total` (the short dots of the kernels read up to 7 doubles past a row: they must be finite)
//
// Columns are padded to mp = 64 * MT (MT = columns per lane, even) so that every lane of a wavefront owns MT
// columns (MT / 2 pairs of adjacent columns: lane l owns 128 h + 2 l and 128 h + 2 l + 1, h = 0 .. MT/2 - 1); pad columns have Mt = 0, Gm = 0, m0 = 1 (their dual w = -1 is never selected).
// Mt, m0, Gm are contiguous so the shared-matrix kernel stages them into LDS with one linear copy.
#pragma once
#include <stddef.h>

namespace pympc {

constexpr int kStatusOptimal = 0;
constexpr int kStatusInfeasible = 1;
constexpr int kStatusMaxIter = 3;
constexpr int kStatusNotPD = 4;
constexpr int kStatusPending = -1;   // between the screening pass and the solve kernel only: never returned
// ... and every value below it: "pending, and the lane path ended on the passive columns (c0, c1), both with a positive
// coefficient, with a third column entering": the Gram-form kernel starts from that set instead of from nothing
__host__ __device__ inline int status_pending_warm(int c0, int c1) { return -2 - (c0 | (c1 << 11)); }   // columns < 2048 = kMMax
__host__ __device__ inline bool status_is_pending(int st) { return st < 0; }
// ... and one more: "pending, the Gram-form kernel stopped at kGramK passive columns (all coefficients positive, another
// column entering) and left those columns, in the order they entered, in the first words of the state's active-set
// output (two 16-bit indices per word)": the Lawson-Hanson kernel factorises them in that order and goes on from there
constexpr int kStatusPendingSet = -2 - (1 << 30);
constexpr int kStatusSetColumns = 8;   // = kGramK (qp_gram.hpp)
__host__ __device__ inline int status_warm_columns(int st) { return (st <= -2 && st != kStatusPendingSet) ? -2 - st : -1; }   // c0 | c1 << 11, or -1
constexpr double kGramDiag = 1e300;  // diagonal of the stored Gram matrix (the true one is cn): see qp_screen.hpp

constexpr int kNMax = 63;    // variables: R = n + 1 <= 64 rows, one lane per row
constexpr int kPMax = 64;    // parameters: the parameter vector is staged one entry per lane
constexpr int kMTMax = 32;   // columns per lane
constexpr int kMMax = 64 * kMTMax;

// columns-per-lane bucket for m rows (0 if unsupported)
__host__ __device__ inline int mt_bucket(int m)
{
    const int need = (m + 63) / 64;
    const int buckets[] = {2, 4, 6, 8, 12, 16, 24, 32};   // even: a lane owns PAIRS of adjacent columns (16-byte LDS reads)
    for (int b : buckets)
        if (need <= b) return b;
    return 0;
}

struct WsLayout {
    int mp;
    size_t Mt, m0, Gm, cn, T, u0, Ku, Sxx, sx, s0, total;
};

__host__ __device__ inline WsLayout ws_layout(int n, int m, int p)
{
    WsLayout L;
    L.mp = 64 * mt_bucket(m);
    const size_t mp = (size_t)L.mp;
    L.Mt = 0;
    L.m0 = L.Mt + (size_t)n * mp;
    L.Gm = L.m0 + mp;
    L.cn = L.Gm + (size_t)p * mp;
    L.T = L.cn + mp;
    L.u0 = L.T + (size_t)m * n;
    L.Ku = L.u0 + n;
    L.Sxx = L.Ku + (size_t)n * p;
    L.sx = L.Sxx + (size_t)p * p;
    L.s0 = L.sx + p;
    L.total = (L.s0 + 1 + 8 + 31) / 32 * 32;   // >= 8 zeros after s0; per-problem workspaces stay 256-byte aligned
    return L;
}

// The prepared workspace of ONE shared mpQP (pympc_mpqp_prepare) carries, after the layout above, the Gram matrix of
// the constant rows:  Gr [mp x mp],  Gr[i][j] = M_i . M_j  (row i contiguous over the columns).  With it the dual
// vector of the NNLS follows from the passive coefficients alone,  w = -(1 + m'y) m(x) - sum_l y_l Gr[P_l]  (qp_wave.hpp),
// one row per passive column instead of the n + 1 rows of the matrix.  Per-problem workspaces (pympc_qp_solve_batch,
// pympc_mpqp_multi_solve) do not have it.
__host__ __device__ inline size_t ws_gram_doubles(const WsLayout &L) { return (size_t)L.mp * L.mp; }

}  // namespace pympc
