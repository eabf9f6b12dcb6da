// Wave-level primitives on the small per-problem factorisation that the QP and LP kernels keep in LDS
// (tests/kernel_model.py: nnls_invqr / class InvQR are the scalar numpy models of exactly these formulas):
//
//     Qt (R x R, orthogonal)   Qt G_W = [Rk; 0]  for the k working columns G_W of a matrix with R rows
//     Wi (k x k, upper tri.)   = Rk^-1
//     cv (R)                   = Qt c            (c: right-hand side of the least-squares system)
//
// Storage (LDS, per problem):  Q = Qt' row-major with pitch RP (Q[i * RP + l] = Qt[l][i]),  Wi row-major with pitch
// RP,  vectors cv / ab of R + 8 doubles.
//
// qr_trial / qr_add append a column (one Householder reflector on Qt, one new column of Wi), qr_drop removes the
// column at position jj (Givens rotations read off row jj of Wi).  These are the Lawson-Hanson add / delete steps of
// scipy.optimize.nnls (called at /root/reference/pympc/optimization/solvers/pnnls.py:36,218) restated on Q and R^-1
// instead of on a transformed copy of the whole matrix.
//
// Every function is called by ONE full wavefront (64 lanes active); lane l owns row l of Q (= column l of Qt) and row
// l of Wi, so the mat-vecs need no cross-lane reduction.  Matrices are pitched by an odd RP: lane-strided accesses are
// bank-conflict free.
//
// Latency rules this file is written to (a wave is alone, or nearly, on its SIMD: 1-3 waves per SIMD):
//   * reductions and broadcasts never go through LDS: DPP inside rows of 16 lanes, v_readlane across rows
//     (a ds_bpermute butterfly is six dependent LDS round trips per reduction);
//   * every short dot product issues its LDS loads eight at a time before the dependent FMA chain: the compiler does
//     not pipeline loops with run-time bounds, so a plain loop pays one LDS latency per term;
//   * the kernels are VALU-issue bound at 3 waves per SIMD, so the dots carry no per-term address arithmetic, clamps
//     or masks: all but one access pattern run with unit stride along the summation index (immediate offsets from one
//     base address: that is why Q is stored, not Qt), and instead of masking the ragged end of a sum the operands are
//     kept such that the surplus terms are exact zeros:
//        (I1) Wi[r][c] = 0 for c < r (zero-initialised, rotations keep it) and for c >= k (a dropped column is zeroed),
//        (I2) cv, ab (and the callers' vectors) carry 8 zeros after their R entries,
//        (I3) every double a dot may touch is finite (the per-wave LDS block is zero-filled once per kernel);
//   * the Givens sweep of qr_drop carries the running pair in registers: no LDS round trip between rotations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pympc {

__device__ __forceinline__ void wave_sync_lds()
{
    // LDS operations of one wavefront execute in program order; this only stops the compiler from moving
    // accesses across the point where other lanes' LDS writes are consumed.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------- cross-lane
constexpr int kDppXor1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <-> 7 - i inside groups of 8
constexpr int kDppMirror = 0x140;      // row_mirror:      lane i <-> 15 - i inside rows of 16

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    // every lane has a valid source for these controls; bound_ctrl lets the compiler drop the copy into `old`
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = dpp_i32<CTRL>(__double2loint(v)), hi = dpp_i32<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// value of lane `src` (wave-uniform index) in every lane
__device__ __forceinline__ int lane_i32(int v, int src)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}
__device__ __forceinline__ double lane_f64(double v, int src)
{
    const int s = __builtin_amdgcn_readfirstlane(src);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), s), __builtin_amdgcn_readlane(__double2loint(v), s));
}

// IEEE max / min as ONE instruction: fmax() / fmin() are emitted with a canonicalising v_max x, x, x on each operand
__device__ __forceinline__ double max_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double wsum(double v)
{
    v += dpp_f64<kDppXor1>(v);
    v += dpp_f64<kDppXor2>(v);
    v += dpp_f64<kDppHalfMirror>(v);
    v += dpp_f64<kDppMirror>(v);
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}
// sum of lanes 0 .. 7 (the other lanes of the first row of 16 may hold anything; lanes >= 8 of the addend must be 0 only
// if they fall into lanes 0 .. 7's butterfly: they do not)
__device__ __forceinline__ double wsum8(double v)
{
    v += dpp_f64<kDppXor1>(v);
    v += dpp_f64<kDppXor2>(v);
    v += dpp_f64<kDppHalfMirror>(v);
    return lane_f64(v, 0);
}
// three sums at once: the three butterflies are independent, written in lockstep so that they overlap in the pipeline
__device__ __forceinline__ void wsum3(double &a, double &b, double &c)
{
#define PYMPC_WSUM3_STEP(CTRL)                            \
    {                                                     \
        const double ta = dpp_f64<CTRL>(a), tb = dpp_f64<CTRL>(b), tc = dpp_f64<CTRL>(c); \
        a += ta; b += tb; c += tc;                        \
    }
    PYMPC_WSUM3_STEP(kDppXor1)
    PYMPC_WSUM3_STEP(kDppXor2)
    PYMPC_WSUM3_STEP(kDppHalfMirror)
    PYMPC_WSUM3_STEP(kDppMirror)
#undef PYMPC_WSUM3_STEP
    const double a0 = lane_f64(a, 0), a1 = lane_f64(a, 16), a2 = lane_f64(a, 32), a3 = lane_f64(a, 48);
    const double b0 = lane_f64(b, 0), b1 = lane_f64(b, 16), b2 = lane_f64(b, 32), b3 = lane_f64(b, 48);
    const double c0 = lane_f64(c, 0), c1 = lane_f64(c, 16), c2 = lane_f64(c, 32), c3 = lane_f64(c, 48);
    a = (a0 + a1) + (a2 + a3);
    b = (b0 + b1) + (b2 + b3);
    c = (c0 + c1) + (c2 + c3);
}
__device__ __forceinline__ double wmin(double v)
{
    v = min_raw(v, dpp_f64<kDppXor1>(v));
    v = min_raw(v, dpp_f64<kDppXor2>(v));
    v = min_raw(v, dpp_f64<kDppHalfMirror>(v));
    v = min_raw(v, dpp_f64<kDppMirror>(v));
    return min_raw(min_raw(lane_f64(v, 0), lane_f64(v, 16)), min_raw(lane_f64(v, 32), lane_f64(v, 48)));
}
__device__ __forceinline__ double wmax(double v)
{
    v = max_raw(v, dpp_f64<kDppXor1>(v));
    v = max_raw(v, dpp_f64<kDppXor2>(v));
    v = max_raw(v, dpp_f64<kDppHalfMirror>(v));
    v = max_raw(v, dpp_f64<kDppMirror>(v));
    return max_raw(max_raw(lane_f64(v, 0), lane_f64(v, 16)), max_raw(lane_f64(v, 32), lane_f64(v, 48)));
}
__device__ __forceinline__ int wmin_int(int v)
{
    v = min(v, dpp_i32<kDppXor1>(v));
    v = min(v, dpp_i32<kDppXor2>(v));
    v = min(v, dpp_i32<kDppHalfMirror>(v));
    v = min(v, dpp_i32<kDppMirror>(v));
    return min(min(lane_i32(v, 0), lane_i32(v, 16)), min(lane_i32(v, 32), lane_i32(v, 48)));
}

// ------------------------------------------------------------------------------------------------- scalar chains
// The pivoting arithmetic is a chain of wave-uniform divisions and square roots computed redundantly by every lane.
// The IEEE-exact expansions hipcc emits for `/` and sqrt() are ~40 dependent instructions each (scaling, fix-up of
// denormal / huge operands); these are the Newton forms without the range fix-ups: <= 1 ulp from the rounded result
// for operands in [1e-290, 1e290], which is where a least-squares factorisation of finite data lives.  Zero, infinite
// and NaN operands take the exact path.
__device__ __forceinline__ double fast_rcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ bool fast_range(double b)
{
    const double a = fabs(b);
    return a > 1e-290 && a < 1e290;
}
__device__ __forceinline__ double fast_div(double a, double b)
{
    if (__builtin_expect(!(fast_range(b) && fabs(a) < 1e290), 0)) return a / b;
    const double r = fast_rcp(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double fast_sqrt(double a)
{
    if (__builtin_expect(!(a > 1e-290 && a < 1e290), 0)) return sqrt(a);
    double y = __builtin_amdgcn_rsq(a);            // ~ 1/sqrt(a)
    y = y * fma(-0.5 * a * y, y, 1.5);            // Newton on 1/sqrt
    double g = a * y;                              // ~ sqrt(a)
    g = fma(fma(-g, g, a), 0.5 * y, g);
    g = fma(fma(-g, g, a), 0.5 * y, g);
    return g;
}

// ------------------------------------------------------------------------------------------------- short dots
// sum_{c = 0}^{len - 1} a[c] b[c] with len rounded up to a multiple of 8: both operands unit stride, terms added in
// index order, the loads of a chunk issued together.  The surplus terms must be zeros (I1-I3).
__device__ __forceinline__ double dotu(const double *a, const double *b, int len, double acc = 0.0)
{
    for (int c = 0; c < len; c += 8) {
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[c + u];
            y[u] = b[c + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(x[u], y[u], acc);
    }
    return acc;
}

// same with a run-time stride sa on the first operand
__device__ __forceinline__ double dots(const double *a, int sa, const double *b, int len, double acc = 0.0)
{
    for (int c = 0; c < len; c += 8) {
        double x[8], y[8];
        const double *ac = a + c * sa;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = ac[u * sa];
            y[u] = b[c + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(x[u], y[u], acc);
    }
    return acc;
}

// general form for the callers' own tables (clamped indices, masked surplus): sum_{l = l0}^{l1 - 1} A[l sa] B[l sb]
__device__ __forceinline__ double dot8(const double *A, int sa, const double *B, int sb, int l0, int l1, double acc = 0.0)
{
    for (int l = l0; l < l1; l += 8) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = min(l + u, l1 - 1);
            a[u] = A[q * sa];
            b[u] = B[q * sb];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = (l + u < l1) ? b[u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(a[u], b[u], acc);
    }
    return acc;
}

constexpr int kVecPad = 8;   // zeros kept after the R entries of cv / ab (I2)

struct InvQr {
    double *Q, *Wi, *cv, *ab;   // LDS: R x RP, R x RP, R + 8, R + 8
    int R, RP;
};

__host__ __device__ inline int invqr_lds_doubles(int R, int RP) { return 2 * R * RP + 2 * (R + kVecPad); }

__device__ __forceinline__ void invqr_bind(InvQr &F, double *base, int R, int RP)
{
    F.R = R; F.RP = RP;
    F.Q = base;
    F.Wi = F.Q + R * RP;
    F.cv = F.Wi + R * RP;
    F.ab = F.cv + R + kVecPad;
}

// Q = I, Wi = 0 (start of a problem; cv is the caller's).  One wavefront.
__device__ __forceinline__ void qr_reset(const InvQr &F)
{
    const int lane = threadIdx.x & 63, R = F.R, RP = F.RP;
    for (int e = lane; e < 2 * R * RP; e += 64) F.Q[e] = 0.0;   // Q and Wi are adjacent
    wave_sync_lds();
    if (lane < R) F.Q[lane * RP + lane] = 1.0;
    wave_sync_lds();
}

struct QrTrial {
    double qi;     // lane < R: (Qt g)[lane]
    double cvl;    // lane < R: cv[lane]
    double diag, up, bb, sm, ck;
    double unorm;       // || (Qt g)[:k] ||: the part of g inside the span of the working columns
    bool independent;   // the 0.01 independence test of Lawson-Hanson passed
};

// Householder data for appending the column g (lane < R holds g[lane]) at position k.  identity: Q is still I (not
// materialised), so Qt g = g.
__device__ __forceinline__ QrTrial qr_trial(const InvQr &F, int k, double gl, bool identity = false)
{
    const int lane = threadIdx.x & 63, R = F.R, RP = F.RP;
    QrTrial t;
    double qi = (lane < R) ? gl : 0.0;
    if (!identity) {
        if (lane < R) F.ab[lane] = gl;
        wave_sync_lds();
        qi = 0.0;
        if (lane < R) qi = dots(F.Q + lane, RP, F.ab, R);   // (Qt g)[lane] = sum_l Q[l][lane] g[l]
    }
    const double cvl = (lane < R) ? F.cv[lane] : 0.0;
    const double q2 = qi * qi;
    double un2 = lane < k ? q2 : 0.0, sg2 = lane >= k ? q2 : 0.0, dotc = lane > k ? cvl * qi : 0.0;
    wsum3(un2, sg2, dotc);
    const double qk = lane_f64(qi, k);
    const double ckk = lane_f64(cvl, k);
    const double sigma = fast_sqrt(sg2), unorm = fast_sqrt(un2);
    t.qi = qi; t.cvl = cvl; t.independent = false; t.unorm = unorm;
    t.diag = t.up = t.bb = t.sm = t.ck = 0.0;
    if (sigma > 0.0) {
        t.diag = qk > 0.0 ? -sigma : sigma;
        t.up = qk - t.diag;
        const double t1 = unorm + fabs(t.diag) * 0.01;
        if (t1 - unorm > 0.0) {
            t.independent = true;
            t.bb = t.up * t.diag;
            t.sm = fast_div(ckk * t.up + dotc, t.bb);
            t.ck = ckk + t.sm * t.up;
        }
    }
    wave_sync_lds();
    return t;
}

// First column of a problem (k = 0, Q = I not materialised, trial computed with identity = true): writes
//   Q = I + v v'/bb  (the reflector itself),  Wi = 0 except Wi[0][0] = 1/diag,  cv
// directly, instead of filling Q with the identity and updating it through the general path.
__device__ __forceinline__ void qr_add_first(const InvQr &F, const QrTrial &t)
{
    const int lane = threadIdx.x & 63, R = F.R, RP = F.RP;
    const double vl = (lane == 0) ? t.up : t.qi;          // reflector entry of this lane (0 for lane >= R)
    if (lane < R) F.ab[lane] = vl;
    if (lane == 0) F.cv[0] = t.ck;
    else if (lane < R) F.cv[lane] = fma(t.sm, t.qi, t.cvl);
    for (int e = lane; e < R * RP; e += 64) F.Wi[e] = 0.0;
    wave_sync_lds();
    if (lane < R) {
        const double s = fast_div(vl, t.bb);
        double *row = F.Q + lane * RP;
        for (int c = 0; c < R; c += 8) {                  // RP >= R and the vectors carry 8 trailing zeros (I2)
            double w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = F.ab[c + u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c + u < RP) row[c + u] = fma(s, w[u], (c + u == lane) ? 1.0 : 0.0);
        }
    }
    if (lane == 0) F.Wi[0] = fast_div(1.0, t.diag);
    wave_sync_lds();
}

// Append the column of a passed trial at position k (the caller increments k).  Q must be materialised.
__device__ __forceinline__ void qr_add(const InvQr &F, int k, const QrTrial &t)
{
    const int lane = threadIdx.x & 63, R = F.R, RP = F.RP;
    if (lane < R) F.ab[lane] = (lane == k) ? t.up : t.qi;   // reflector v (entries >= k) and rho = q[:k]
    if (lane == k) F.cv[lane] = t.ck;
    else if (lane > k && lane < R) F.cv[lane] = fma(t.sm, t.qi, t.cvl);
    wave_sync_lds();
    const double rdiag = fast_div(1.0, t.diag);
    if (lane < R) {   // Qt <- (I + v v'/bb) Qt: lane = column of Qt = row of Q, entries k .. R-1
        double *row = F.Q + lane * RP + k;
        const double *v = F.ab + k;
        const int len = R - k;
        const double s = fast_div(dotu(row, v, len), t.bb);
        const int full = len & ~7;
        for (int c = 0; c < full; c += 8) {
            double q[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                q[u] = row[c + u];
                w[u] = v[c + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) row[c + u] = fma(s, w[u], q[u]);
        }
        if (full < len) {   // ragged end: clamped repeats store the same value
            double q[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = min(full + u, len - 1);
                q[u] = row[c];
                w[u] = v[c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) row[min(full + u, len - 1)] = fma(s, w[u], q[u]);
        }
    }
    if (lane < k) {   // new column of Wi = Rk^-1:  -(Wi rho) / diag   (I1: the row is zero outside [lane, k))
        const double s = dotu(F.Wi + lane * RP, F.ab, k);
        F.Wi[lane * RP + k] = -s * rdiag;
    }
    if (lane == k) F.Wi[k * RP + k] = rdiag;
    wave_sync_lds();
}

// Remove the column at position jj of k (the caller decrements k and shifts its per-position registers).
// Rotation l (l = jj .. k-2) mixes columns l, l+1 of Wi, rows l, l+1 of Qt and entries l, l+1 of cv; its coefficients
// come from row jj of Wi.  Lane r carries row r of Wi, lane c column c of Qt (= row c of Q), every lane the pair of cv,
// so the whole sweep runs out of registers with the not-yet-rotated operands prefetched eight rotations ahead.
__device__ __forceinline__ void qr_drop(const InvQr &F, int k, int jj)
{
    const int lane = threadIdx.x & 63, R = F.R, RP = F.RP;
    double *wrow = F.Wi + lane * RP;
    double *qrow = F.Q + lane * RP;
    const bool inW = lane < k, inQ = lane < R;
    double aw = inW ? wrow[jj] : 0.0;      // current value of Wi[lane][l]   (I1: zero below the diagonal)
    double q = inQ ? qrow[jj] : 0.0;       // current value of Qt[l][lane]
    double c0 = F.cv[jj];
    for (int l0 = jj; l0 < k - 1; l0 += 8) {
        double bw[8], qn[8], c1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {      // over-reads past k - 1 are finite and unused
            bw[u] = inW ? wrow[l0 + 1 + u] : 0.0;
            qn[u] = inQ ? qrow[l0 + 1 + u] : 0.0;
            c1[u] = F.cv[l0 + 1 + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int l = l0 + u;
            if (l < k - 1) {
                const double xx = lane_f64(aw, jj), yy = lane_f64(bw[u], jj);   // row jj: Wi[jj][l] (current), Wi[jj][l+1]
                const double rho = fast_sqrt(xx * xx + yy * yy);
                double cc = 1.0, ss = 0.0;
                if (rho > 0.0) {
                    const double ri = fast_div(1.0, rho);
                    cc = yy * ri;
                    ss = xx * ri;
                }
                if (inW) wrow[l] = cc * aw - ss * bw[u];
                aw = ss * aw + cc * bw[u];
                if (inQ) qrow[l] = cc * q - ss * qn[u];
                q = ss * q + cc * qn[u];
                if (lane == 0) F.cv[l] = cc * c0 - ss * c1[u];
                c0 = ss * c0 + cc * c1[u];
            }
        }
    }
    if (inQ) qrow[k - 1] = q;
    if (lane == 0) F.cv[k - 1] = c0;
    wave_sync_lds();
    // delete row jj of Wi and its last column: lane c moves column c up by one row; column k-1 is zeroed (I1)
    if (lane < k - 1) {
        double *wcol = F.Wi + lane;
        for (int i = jj; i < k - 1; i += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = wcol[(min(i + u, k - 2) + 1) * RP];
#pragma unroll
            for (int u = 0; u < 8; ++u) wcol[min(i + u, k - 2) * RP] = v[u];   // clamped repeats store the same value
        }
    }
    if (lane < k) F.Wi[lane * RP + k - 1] = 0.0;
    wave_sync_lds();
    if (lane < k) F.Wi[(k - 1) * RP + lane] = 0.0;   // the vacated last row
    wave_sync_lds();
}

// lane < k: coefficient of working column lane,  (Wi cv[:k])[lane]   (I1: the row is zero outside [lane, k))
__device__ __forceinline__ double qr_coeff(const InvQr &F, int k)
{
    const int lane = threadIdx.x & 63, RP = F.RP;
    double z = 0.0;
    if (lane < k) z = dotu(F.Wi + lane * RP, F.cv, k);
    return z;
}

}  // namespace pympc
