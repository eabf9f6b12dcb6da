// NNLS and the PNNLS reduction for problems of 65 .. 1024 rows: one workgroup of 256, 512 or 1024 threads per problem (a row per thread).
//
// The wave kernel of qp_wave.hpp keeps one row per lane, which ends at 64 rows (QPs of up to 63 variables, LPs in the
// reference's KKT form of pnnls.py:109-124 with 1 + m + n <= 64).  The reference has no such limit
// (scipy.optimize.nnls, called at /root/reference/pympc/optimization/solvers/pnnls.py:36 and :218, takes any size), so
// this file states Lawson-Hanson once more with one row per THREAD of a 256-thread workgroup:
//
//   * the working matrix stays in device memory, row-major as the caller gave it: every pass over it (dual vector,
//     Householder application, Givens rotation of two rows) runs thread <-> column, coalesced, with the column-wise
//     sums kept in one thread -- no reductions in the O(rows x cols) passes;
//   * vectors over rows (right-hand side, trial solution, reflector) and over columns (x, w, the index permutation and
//     its inverse) live in LDS; scalars every thread needs come from block reductions of one barrier each;
//   * the columns of the matrix that are in the set P have a second home: a packed, column-contiguous copy of the
//     triangular factor in device memory (a slot per column).  The triangular solves stage 16 of its columns in LDS at
//     a time (coalesced) and run the reference's column-oriented back substitution on them; when a column leaves P the
//     rotations are generated on that copy (a thread per column, the chain of dependent rotations through LDS) and then
//     applied to all other columns of the matrix in one pass without barriers;
//   * control flow is uniform: every decision is taken from LDS values all threads read alike.
// The algorithm, its tolerances and its selection rules (first maximum of the dual vector in index order, 0.01
// independence factor, first minimum of the step ratio, itmax = 3 cols) are those of Lawson & Hanson's NNLS (1974,
// ch. 23), the routine behind the scipy version the reference pins.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "invqr_wave.hpp"
#include "nnls_raw.hpp"

namespace pympc {

#ifndef PYMPC_BLK_TRI_COLS
#define PYMPC_BLK_TRI_COLS 16
#endif
// One row per thread: 256 threads up to 256 rows, 512 threads (two wavefronts per SIMD, 256 VGPRs each) up to 512,
// 1024 threads (four per SIMD, 128 VGPRs each: narrower chunks and unrolls) up to 1024.
constexpr int kBlkThreadsSmall = 256;
constexpr int kBlkThreadsBig = 512;
constexpr int kBlkThreadsHuge = 1024;
constexpr int kBlkRowsMax = kBlkThreadsHuge;
__host__ __device__ inline int blk_threads(int rows)
{
    return rows <= kBlkThreadsSmall ? kBlkThreadsSmall : rows <= kBlkThreadsBig ? kBlkThreadsBig : kBlkThreadsHuge;
}
// Columns of the triangular factor staged in LDS at a time.  One problem alone runs the same with 8, 16 or 32 (the solve
// is a latency chain either way); a batch gains from the smaller footprint (256 x 500: 56 instead of 84 KB of LDS, two
// workgroups per CU instead of one: 512 problems 79 -> 68 ms).

struct NnlsBlockParams {
    int rows, cols;
    int itmax;                 // limit of secondary-loop passes (scipy: 3 cols)
    const double *A;           // [B x rows x cols] row-major (a_stride = rows*cols) or shared (a_stride 0)
    long long a_stride;
    const double *b;           // [B x rows]
    long long b_stride;
    const int32_t *skip_rows;  // per problem: the first skip_rows[k] rows count as zero (PNNLS), or null
    double *work;              // [B x rows x cols] working copy of the matrix
    double *factor;            // [B x rows x rows] the triangular factor, a slot of `rows` doubles per column in P
    double *colvec;            // [B x 3 cols] x, w, index and pos of the COLSG instances (more columns than the LDS holds)
    double *x;                 // [B x cols]
    double *rnorm;             // [B] or null
    int32_t *status;           // [B]
    int32_t *iters;            // [B x 2] (outer, secondary) or null
};

__host__ __device__ constexpr int blk_tri_cols(int threads) { return threads >= 1024 ? PYMPC_BLK_TRI_COLS / 2 : PYMPC_BLK_TRI_COLS; }
__host__ __device__ constexpr int blk_unroll(int threads) { return threads >= 1024 ? 4 : 8; }
constexpr int kBlkColsMax = 32768;   // columns with the vectors over columns in device memory (COLSG instances)
__host__ __device__ inline size_t nnls_block_lds_bytes(int rows, int cols, bool colsg = false)
{
    const int tc = blk_tri_cols(blk_threads(rows));
    if (colsg) return ((size_t)(5 + tc) * rows + 32) * 8 + ((size_t)rows * 4 + 7) / 8 * 8;
    // doubles: b, zz, u [rows] | x, w [cols] | reduction slots [32] | rotations c, s [rows] | staged factor [kBlkTriCols x rows]
    // ints: index, pos [cols], slotpos [rows]
    return ((size_t)(5 + tc) * rows + (size_t)2 * cols + 32) * 8 + (((size_t)2 * cols + rows) * 4 + 7) / 8 * 8;
}

// Block reductions over 256 threads through alternating LDS slots: reduction k writes slot k & 1, so a writer of
// reduction k + 2 has passed the barrier of k + 1 and with it every read of reduction k -- one barrier each.
// Every thread must take part in every reduction, in the same order.
template <int W>
struct BlockReduce {
    double *slot;   // 2 x W doubles (W wavefronts)
    int phase;
    __device__ __forceinline__ double *next()
    {
        double *s = slot + (phase & 1) * W;
        ++phase;
        return s;
    }
    // pairwise over the wavefronts' values: ((s0 + s1) + (s2 + s3)) + ...
    template <typename F>
    __device__ __forceinline__ static double tree(const double *s, F f)
    {
        double v[W];
#pragma unroll
        for (int i = 0; i < W; ++i) v[i] = s[i];
#pragma unroll
        for (int stride = 1; stride < W; stride *= 2)
#pragma unroll
            for (int i = 0; i + stride < W; i += 2 * stride) v[i] = f(v[i], v[i + stride]);
        return v[0];
    }
    __device__ __forceinline__ double sum(double v)
    {
        v = wsum(v);
        double *s = next();
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
        __syncthreads();
        return tree(s, [](double a, double b) { return a + b; });
    }
    __device__ __forceinline__ double max(double v)
    {
        v = wmax(v);
        double *s = next();
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
        __syncthreads();
        return tree(s, [](double a, double b) { return max_raw(a, b); });
    }
    __device__ __forceinline__ double min(double v)
    {
        v = wmin(v);
        double *s = next();
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
        __syncthreads();
        return tree(s, [](double a, double b) { return min_raw(a, b); });
    }
    __device__ __forceinline__ int min_int(int v)
    {
        v = wmin_int(v);
        int *s = reinterpret_cast<int *>(next());
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
        __syncthreads();
        int r = s[0];
#pragma unroll
        for (int i = 1; i < W; ++i) r = ::min(r, s[i]);
        return r;
    }
};

// Lawson-Hanson's G1: the rotation that takes (a, b) to (sig, 0)
__device__ __forceinline__ void blk_givens(double a, double b, double &c, double &s, double &sig)
{
    if (fabs(a) > fabs(b)) {
        const double xr = b / a, yr = sqrt(1.0 + xr * xr);
        c = copysign(1.0 / yr, a);
        s = c * xr;
        sig = fabs(a) * yr;
    } else if (b != 0.0) {
        const double xr = a / b, yr = sqrt(1.0 + xr * xr);
        s = copysign(1.0 / yr, b);
        c = s * xr;
        sig = fabs(b) * yr;
    } else {
        sig = 0.0; c = 0.0; s = 1.0;
    }
}

// The same rotation as c = a / r, s = b / r, sig = r with one reciprocal square root: the chain of dependent rotations
// of a deletion runs through this, so its latency counts.  Outside [1e-280, 1e280] (and for a = b = 0) the exact form.
__device__ __forceinline__ void blk_givens_fast(double a, double b, double &c, double &s, double &sig)
{
    const double r2 = fma(a, a, b * b);
    if (r2 > 1e-280 && r2 < 1e280) {
        const double inv = rsqrt(r2);
        c = a * inv; s = b * inv; sig = r2 * inv;
    } else {
        blk_givens(a, b, c, s, sig);
    }
}

// Column passes of one thread over rows l0 .. m of column j (stride n): the loads of kBlkUnroll rows are issued together --
// a loop of one load and one FMA pays the whole memory latency per row (one workgroup per problem: nothing else hides it).
template <int kBlkUnroll>
__device__ __forceinline__ double blk_col_dot(const double *a, int n, int j, int l0, int m, const double *vec, double sm)
{
    int l = l0;
    for (; l + kBlkUnroll <= m; l += kBlkUnroll) {
        double v[kBlkUnroll];
#pragma unroll
        for (int u = 0; u < kBlkUnroll; ++u) v[u] = a[(size_t)(l + u) * n + j];
#pragma unroll
        for (int u = 0; u < kBlkUnroll; ++u) sm = fma(v[u], vec[l + u], sm);
    }
    for (; l < m; ++l) sm = fma(a[(size_t)l * n + j], vec[l], sm);
    return sm;
}
// a[l][j] += sm vec[l] for l0 <= l < m; returns the dot product of the NEW entries with `rhs` (the next dual-vector entry
// of this column, summed in the order blk_col_dot would use: one pass over the matrix less per entering column)
template <int kBlkUnroll>
__device__ __forceinline__ double blk_col_axpy_dot(double *a, int n, int j, int l0, int m, const double *vec, double sm, const double *rhs)
{
    double w = 0.0;
    int l = l0;
    for (; l + kBlkUnroll <= m; l += kBlkUnroll) {
        double v[kBlkUnroll];
#pragma unroll
        for (int u = 0; u < kBlkUnroll; ++u) v[u] = a[(size_t)(l + u) * n + j];
#pragma unroll
        for (int u = 0; u < kBlkUnroll; ++u) {
            v[u] = fma(sm, vec[l + u], v[u]);
            a[(size_t)(l + u) * n + j] = v[u];
            w = fma(v[u], rhs[l + u], w);
        }
    }
    for (; l < m; ++l) {
        const double v = fma(sm, vec[l], a[(size_t)l * n + j]);
        a[(size_t)l * n + j] = v;
        w = fma(v, rhs[l], w);
    }
    return w;
}

// zz[0 .. nsetp) <- solution of the triangular system whose column at position ip sits in slot slotpos[ip] of the
// packed factor Rs (rows 0 .. ip); zz holds the right-hand side on entry.  Ends with a barrier.
template <int kBlkTriCols>
__device__ __forceinline__ void blk_trisolve(const double *Rs, int m, int nsetp, const int *slotpos, double *zz, double *tri)
{
    const int t = threadIdx.x;
    for (int hi = nsetp; hi > 0; hi -= kBlkTriCols) {
        const int lo = hi > kBlkTriCols ? hi - kBlkTriCols : 0, nc = hi - lo;
        // stage rows 0 .. hi of the block's columns: a thread per row, all columns' loads in flight at once (rows below
        // a column's pivot are loaded too and never used)
        if (t < hi) {
            double v[kBlkTriCols];
#pragma unroll
            for (int c = 0; c < kBlkTriCols; ++c) v[c] = c < nc ? Rs[(size_t)slotpos[lo + c] * m + t] : 0.0;
#pragma unroll
            for (int c = 0; c < kBlkTriCols; ++c)
                if (c < nc) tri[c * m + t] = v[c];
        }
        __syncthreads();
        // The diagonal block in ONE wavefront, a lane per row: its nc dependent steps then cost two cross-lane reads, a
        // multiplication and an FMA each -- no barrier, the block's entries in registers, and the division taken out
        // of the chain (reciprocals of the nc pivots at once: z * (1 / d) instead of z / d, within an ulp of it).
        if (t < 64) {
            const int row = lo + (t < nc ? t : 0);
            double z = t < nc ? zz[row] : 0.0;
            const double rd = t < nc ? 1.0 / tri[t * m + row] : 0.0;
            double tc[kBlkTriCols];
#pragma unroll
            for (int c = 0; c < kBlkTriCols; ++c) tc[c] = c < nc ? tri[c * m + row] : 0.0;
#pragma unroll
            for (int c = kBlkTriCols - 1; c >= 0; --c) {
                if (c < nc) {
                    const double zi = lane_f64(z, c) * lane_f64(rd, c);
                    if (t == c) z = zi;
                    if (t < c) z = fma(-tc[c], zi, z);
                }
            }
            if (t < nc) zz[row] = z;
        }
        __syncthreads();
        // the rows above it: the same updates in the same (descending) order, all columns of the block in one go
        if (t < lo) {
            double z = zz[t];
            for (int c = nc - 1; c >= 0; --c) z = fma(-tri[c * m + t], zz[lo + c], z);
            zz[t] = z;
        }
        __syncthreads();
    }
}

#ifdef PYMPC_BLK_PROFILE
#define BLK_TICK(slot) { const long long now_ = wall_clock64(); prof_[slot] += now_ - last_; last_ = now_; }
#else
#define BLK_TICK(slot)
#endif
// COLSG: x, w and the index permutation with its inverse live in device memory instead of LDS (24 bytes per column:
// beyond ~2000 .. 5000 columns, depending on the rows, they no longer fit beside the staged factor).
template <int T, bool COLSG>
__global__ void __launch_bounds__(T) nnls_block_kernel(NnlsBlockParams P)
{
    constexpr int kBlkThreads = T, kBlkTriCols = blk_tri_cols(T), kBlkUnroll = blk_unroll(T);
#ifdef PYMPC_BLK_PROFILE
    long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = wall_clock64();
#endif
    extern __shared__ double lds[];
    const int t = threadIdx.x, m = P.rows, n = P.cols;
    const long long k = blockIdx.x;
    double *bs = lds, *zz = bs + m, *uu = zz + m, *slots = uu + m;
    double *rotc = slots + 32, *rots = rotc + m;   // rotations of a deletion, by the row they end in
    double *tri = rots + m;
    double *colbase = COLSG ? P.colvec + k * 3 * (long long)n : tri + (size_t)kBlkTriCols * m;
    double *xs = colbase, *ws = xs + n;
    int *index = reinterpret_cast<int *>(ws + n), *pos = index + n;
    int *slotpos = COLSG ? reinterpret_cast<int *>(tri + (size_t)kBlkTriCols * m) : pos + n;
    BlockReduce<T / 64> red{slots, 0};

    double *a = P.work + k * (long long)m * n;
    double *Rs = P.factor + k * (long long)m * m;
    const double *A = P.A + k * P.a_stride;
    const int skip = P.skip_rows ? P.skip_rows[k] : 0;
    for (size_t e = t; e < (size_t)m * n; e += kBlkThreads) a[e] = (int)(e / n) >= skip ? A[e] : 0.0;
    if (t < m) { bs[t] = t >= skip ? P.b[k * P.b_stride + t] : 0.0; slotpos[t] = t; }
    for (int j = t; j < n; j += kBlkThreads) { xs[j] = 0.0; ws[j] = 0.0; index[j] = j; pos[j] = j; }
    __syncthreads();

    const double factor = 0.01;
    int iter = 0, outer = 0, mode = 0, nsetp = 0;   // P = index[0 .. nsetp), Z = index[nsetp .. n)
    bool dual_ready = false;                        // ws already holds the dual vector (from the Householder pass below)
    for (;;) {
        if (nsetp >= n || nsetp >= m) break;
        // dual vector on Z
        if (!dual_ready) {
            for (int j = t; j < n; j += kBlkThreads) {
                if (pos[j] < nsetp) continue;
                ws[j] = blk_col_dot<kBlkUnroll>(a, n, j, nsetp, m, bs, 0.0);
            }
            __syncthreads();
        }
        dual_ready = false;
        BLK_TICK(0)
        bool accepted = false;
        int j = -1, izmax = -1;
        double up = 0.0, newpiv = 0.0, cj = 0.0;
        for (;;) {
            // largest positive dual; the first one in index order among equals
            double best = 0.0;
            int bpos = 0x7fffffff;
            for (int c = t; c < n; c += kBlkThreads) {
                const int p = pos[c];
                const double w = ws[c];
                if (p >= nsetp && w > 0.0 && (w > best || (w == best && p < bpos))) { best = w; bpos = p; }
            }
            const double wmx = red.max(best);
            if (!(wmx > 0.0)) break;   // Kuhn-Tucker conditions hold
            izmax = red.min_int(best == wmx ? bpos : 0x7fffffff);
            j = index[izmax];
            // reflector of column j at the pivot row nsetp (H12 in construction mode), without touching the matrix
            cj = t < m ? a[(size_t)t * n + j] : 0.0;
            const bool tail = t >= nsetp && t < m;
            double cl = red.max(tail ? fabs(cj) : 0.0);
            if (t == nsetp) uu[t] = cj;   // the pivot entry, for everybody
            up = 0.0;
            const double sm0 = red.sum(tail && cl > 0.0 ? (cj / cl) * (cj / cl) : 0.0);   // (the barrier also publishes uu[nsetp])
            const double cjp = uu[nsetp];
            newpiv = cjp;
            if (cl > 0.0) {
                cl *= sqrt(sm0);
                if (cjp > 0.0) cl = -cl;
                up = cjp - cl;
                newpiv = cl;
            }
            const double unorm = sqrt(red.sum(t < nsetp ? cj * cj : 0.0));
            const double t1 = unorm + fabs(newpiv) * factor;
            bool ok = false;
            if (t1 - unorm > 0.0) {   // column j is sufficiently independent: would its coefficient be positive?
                const double bb = up * newpiv;
                double sm = red.sum(t == nsetp ? bs[t] * up : (t > nsetp && t < m ? bs[t] * cj : 0.0));
                if (bb >= 0.0) sm = 0.0;
                else sm /= bb;
                if (t < m) zz[t] = t == nsetp ? fma(sm, up, bs[t]) : (t > nsetp ? fma(sm, cj, bs[t]) : bs[t]);
                __syncthreads();
                ok = zz[nsetp] / newpiv > 0.0;
            }
            if (ok) { accepted = true; break; }
            if (t == 0) ws[j] = 0.0;
            __syncthreads();
        }
        BLK_TICK(1)
        if (!accepted) break;
        ++outer;
        // column j moves from Z to P
        const int piv = nsetp;
        if (t < m) { bs[t] = zz[t]; uu[t] = t > piv ? cj : 0.0; }
        if (t == 0) {
            const int other = index[piv];
            index[izmax] = other; pos[other] = izmax;
            index[piv] = j; pos[j] = piv;
            ws[j] = 0.0;
        }
        ++nsetp;
        __syncthreads();
        {
            const double bb = up * newpiv;
            if (bb < 0.0) {
                const double binv = 1.0 / bb;
                for (int c = t; c < n; c += kBlkThreads) {
                    if (pos[c] < nsetp) continue;
                    const double ap = a[(size_t)piv * n + c];
                    double sm = blk_col_dot<kBlkUnroll>(a, n, c, piv + 1, m, uu, ap * up);
                    // (bs is the new right-hand side already and rows piv + 1 .. are the rows the next dual vector sums)
                    if (sm != 0.0) {
                        sm *= binv;
                        a[(size_t)piv * n + c] = fma(sm, up, ap);
                        ws[c] = blk_col_axpy_dot<kBlkUnroll>(a, n, c, piv + 1, m, uu, sm, bs);
                    } else {
                        ws[c] = blk_col_dot<kBlkUnroll>(a, n, c, piv + 1, m, bs, 0.0);
                    }
                }
                dual_ready = true;
            }
            // its column of the factor goes to the first free slot (its entries in the matrix are stale from here on:
            // nothing reads a column of P there, and a column that leaves P is written back first)
            if (t <= piv) Rs[(size_t)slotpos[piv] * m + t] = t < piv ? cj : newpiv;
        }
        __syncthreads();
        BLK_TICK(2)
        blk_trisolve<kBlkTriCols>(Rs, m, nsetp, slotpos, zz, tri);   // zz == bs here
        BLK_TICK(3)
        // secondary loop
        for (;;) {
            ++iter;
            if (iter > P.itmax) { mode = 3; break; }
            double ratio = 2.0;
            if (t < nsetp && zz[t] <= 0.0) {
                const double xl = xs[index[t]];
                const double r = -xl / (zz[t] - xl);
                if (r == r) ratio = r;
            }
            const double alpha = red.min(ratio);
            if (alpha == 2.0) break;
            int jj = red.min_int(ratio == alpha ? t : 0x7fffffff);
            if (t < nsetp) {
                const int l = index[t];
                xs[l] = fma(alpha, zz[t] - xs[l], xs[l]);
            }
            __syncthreads();
            int i = index[jj];
            for (;;) {
                // Column i (position jj) leaves P: the columns behind it move up one position, a rotation each.
                dual_ready = false;
                const int np = nsetp, q = jj + 1 + t;          // this thread's column of the factor, if q < np
                const bool mine = q < np;
                const int sl_del = slotpos[jj];
                const int my_sl = mine ? slotpos[q] : 0, my_col = mine ? index[q] : 0;
                // it returns to Z with its entries as they stand now: rows 0 .. jj from its slot, zeros below
                if (t < m) a[(size_t)t * n + i] = t <= jj ? Rs[(size_t)sl_del * m + t] : 0.0;
                if (t == 0) xs[i] = 0.0;
                // Rotations, generated on the factor: rotation s acts on rows s - 1 and s and is fixed by the column at
                // position s, after the rotations before it.  A thread per column, eight rows loaded and stored at a time;
                // the dependent chain runs through rotc / rots in LDS, one barrier per rotation.
                {
                    double *base = Rs + (size_t)my_sl * m;
                    double carry = mine ? base[jj] : 0.0;
                    for (int s0 = jj + 1; s0 < np; s0 += 8) {
                        double v[8], out[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            v[u] = (mine && s0 + u <= q) ? base[s0 + u] : 0.0;
                            out[u] = 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int s = s0 + u;
                            if (s < np) {
                                if (mine && q == s) {
                                    double cc, ss, sig;
                                    blk_givens_fast(carry, v[u], cc, ss, sig);
                                    rotc[s] = cc; rots[s] = ss; out[u] = sig;
                                }
                                __syncthreads();
                                if (mine && q > s) {
                                    const double cc = rotc[s], ss = rots[s];
                                    out[u] = fma(cc, carry, ss * v[u]);
                                    carry = fma(-ss, carry, cc * v[u]);
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (mine && s0 + u <= q) base[s0 + u - 1] = out[u];
                    }
                }
                __syncthreads();   // (the column written back above is visible to its owner below even without a rotation)
                // ... applied to every column outside P, the leaving one included, and to the right-hand side (c == n):
                // a thread per column down the rows jj .. np - 1, no barrier
                if (np - 1 > jj)
                    for (int c = t; c <= n; c += kBlkThreads) {
                        if (c == n) {
                            double v0 = bs[jj];
                            for (int s = jj + 1; s < np; ++s) {
                                const double v1 = bs[s];
                                bs[s - 1] = fma(rotc[s], v0, rots[s] * v1);
                                v0 = fma(-rots[s], v0, rotc[s] * v1);
                            }
                            bs[np - 1] = v0;
                            continue;
                        }
                        if (pos[c] < np && c != i) continue;
                        double v0 = a[(size_t)jj * n + c];
                        int s = jj + 1;
                        for (; s + kBlkUnroll <= np; s += kBlkUnroll) {
                            double v[kBlkUnroll];
#pragma unroll
                            for (int u = 0; u < kBlkUnroll; ++u) v[u] = a[(size_t)(s + u) * n + c];
#pragma unroll
                            for (int u = 0; u < kBlkUnroll; ++u) {
                                a[(size_t)(s + u - 1) * n + c] = fma(rotc[s + u], v0, rots[s + u] * v[u]);
                                v0 = fma(-rots[s + u], v0, rotc[s + u] * v[u]);
                            }
                        }
                        for (; s < np; ++s) {
                            const double v1 = a[(size_t)s * n + c];
                            a[(size_t)(s - 1) * n + c] = fma(rotc[s], v0, rots[s] * v1);
                            v0 = fma(-rots[s], v0, rotc[s] * v1);
                        }
                        a[(size_t)(np - 1) * n + c] = v0;
                    }
                __syncthreads();
                if (mine) { index[q - 1] = my_col; pos[my_col] = q - 1; slotpos[q - 1] = my_sl; }
                if (t == 0) { index[np - 1] = i; pos[i] = np - 1; slotpos[np - 1] = sl_del; }
                --nsetp;
                __syncthreads();
                // round-off guard: whatever is left in P with a non-positive coefficient leaves too
                jj = red.min_int(t < nsetp && xs[index[t]] <= 0.0 ? t : 0x7fffffff);
                if (jj == 0x7fffffff) break;
                i = index[jj];
            }
            if (t < m) zz[t] = bs[t];
            __syncthreads();
            BLK_TICK(4)
            blk_trisolve<kBlkTriCols>(Rs, m, nsetp, slotpos, zz, tri);
            BLK_TICK(3)
        }
        if (mode != 0) break;
        if (t < nsetp) xs[index[t]] = zz[t];
        __syncthreads();
    }
#ifdef PYMPC_BLK_PROFILE
    if (t == 0 && k == 0)
        printf("nnls_block %d x %d: outer %d secondary %d | us: dual %lld select %lld apply %lld trisolve %lld givens+ratio %lld\n", m, n, outer, iter,
               prof_[0] / 100, prof_[1] / 100, prof_[2] / 100, prof_[3] / 100, prof_[4] / 100);
#endif
    const double rn = sqrt(red.sum(t >= nsetp && t < m ? bs[t] * bs[t] : 0.0));
    for (int c = t; c < n; c += kBlkThreads) P.x[k * (long long)n + c] = xs[c];
    if (t == 0) {
        if (P.rnorm) P.rnorm[k] = rn;
        P.status[k] = mode;
        if (P.iters) { P.iters[2 * k] = outer; P.iters[2 * k + 1] = iter; }
    }
}

// ---------------------------------------------------------------------------------------- PNNLS, 65 .. 1024 rows
// pnnls_reduce_kernel of nnls_raw.hpp with one row per thread: Householder QR of B with column pivoting applied to
// [A | c].  LDS: B (rows x (nu | 1)), the reflector, reduction slots, the permutation.
constexpr int kPnnlsNuMax = 256;   // free variables (a thread per column of B)
// BG: B itself stays in device memory (the caller's Rf region) instead of LDS -- more than 64 free variables, or
// rows x nu beyond the LDS.
__host__ __device__ inline size_t pnnls_block_lds_bytes_bg(int rows, int nu) { return ((size_t)rows + 32) * 8 + (size_t)(nu + 1) * 4; }
__host__ __device__ inline size_t pnnls_block_lds_bytes(int rows, int nu)
{
    return ((size_t)rows * (nu | 1) + rows + 32) * 8 + (size_t)(nu + 1) * 4;
}

template <int T, bool BG>
__global__ void __launch_bounds__(T) pnnls_reduce_block_kernel(PnnlsParams P)
{
    constexpr int kBlkThreads = T;
    extern __shared__ double lds[];
    const long long k = blockIdx.x;
    const int rows = P.rows, cols = P.cols, nu = P.nu, ld = BG ? nu : (nu | 1), t = threadIdx.x;
    double *Bs = BG ? P.Rf + k * (long long)rows * nu : lds;
    double *vv = BG ? lds : Bs + (size_t)rows * ld, *slots = vv + rows;
    int *perm = reinterpret_cast<int *>(slots + 32);
    BlockReduce<T / 64> red{slots, 0};
    const double *A = P.A + k * (long long)rows * cols;
    double *QA = P.QA + k * (long long)rows * cols;
    double *Qc = P.Qc + k * (long long)rows;
    for (int e = t; e < rows * nu; e += kBlkThreads) Bs[(e / nu) * ld + (e % nu)] = P.Bm[k * (long long)rows * nu + e];
    for (size_t e = t; e < (size_t)rows * cols; e += kBlkThreads) QA[e] = A[e];
    if (t < rows) Qc[t] = P.c[k * (long long)rows + t];
    if (t < nu) perm[t] = t;
    __syncthreads();
    double norm0 = 0.0;
    int rank = 0;
    const int steps = nu < rows ? nu : rows;
    for (int j = 0; j < steps; ++j) {
        double s = -1.0;
        if (t >= j && t < nu) {
            s = 0.0;
            for (int i = j; i < rows; ++i) s = fma(Bs[i * ld + t], Bs[i * ld + t], s);
        }
        const double smax = red.max(s);
        if (j == 0) norm0 = smax;
        if (!(smax > kPnnlsRankTol * kPnnlsRankTol * norm0) || !(smax > 0.0)) break;
        const int piv = red.min_int(s == smax ? t : 0x7fffffff);
        if (piv != j) {
            if (t < rows) {
                const double v = Bs[t * ld + j];
                Bs[t * ld + j] = Bs[t * ld + piv];
                Bs[t * ld + piv] = v;
            }
            if (t == 0) { const int q = perm[j]; perm[j] = perm[piv]; perm[piv] = q; }
        }
        __syncthreads();
        const double xi = (t >= j && t < rows) ? Bs[t * ld + j] : 0.0;
        const double sigma = sqrt(red.sum(xi * xi));
        const double xj = Bs[j * ld + j];
        const double alpha = xj > 0.0 ? -sigma : sigma;
        const double vi = (t == j) ? xi - alpha : xi;
        const double vtv = red.sum(vi * vi);
        if (t < rows) vv[t] = vi;
        __syncthreads();
        if (vtv > 0.0) {
            const double f = 2.0 / vtv;
            if (t > j && t < nu) {
                double d = 0.0;
                for (int i = j; i < rows; ++i) d = fma(vv[i], Bs[i * ld + t], d);
                d *= f;
                for (int i = j; i < rows; ++i) Bs[i * ld + t] = fma(-d, vv[i], Bs[i * ld + t]);
            }
            for (int cidx = t; cidx <= cols; cidx += kBlkThreads) {
                double *col = cidx < cols ? QA + cidx : Qc;
                const size_t st = cidx < cols ? (size_t)cols : 1;
                double d = 0.0;
                for (int i = j; i < rows; ++i) d = fma(vv[i], col[i * st], d);
                d *= f;
                for (int i = j; i < rows; ++i) col[i * st] = fma(-d, vv[i], col[i * st]);
            }
        }
        __syncthreads();
        if (t == j) Bs[j * ld + j] = alpha;
        if (t > j && t < rows) Bs[t * ld + j] = 0.0;
        __syncthreads();
        rank = j + 1;
    }
    if (!BG)
        for (int e = t; e < rows * nu; e += kBlkThreads) P.Rf[k * (long long)rows * nu + e] = Bs[(e / nu) * ld + (e % nu)];
    if (t < nu) P.perm[k * nu + t] = perm[t];
    if (t == 0) P.rank[k] = rank;
}

}  // namespace pympc
