// Wave-level primitives on the small per-problem factorisation that the LP kernel keeps in LDS
// (tests/kernel_model.py: class InvQR is the scalar numpy model of exactly these formulas):
//
//     Qt (R x R, orthogonal)   Qt G_W = [Rk; 0]  for the k working columns G_W of a matrix with R rows
//     Wi (k x k, upper tri.)   = Rk^-1
//     cv (R)                   = Qt c            (c: right-hand side of the least-squares system)
//
// qr_trial / qr_add append a column (one Householder reflector on Qt, one new column of Wi), qr_drop removes the
// column at position jj (Givens rotations read off row jj of Wi).  These are the Lawson-Hanson add / delete steps of
// scipy.optimize.nnls (called at /root/reference/pympc/optimization/solvers/pnnls.py:36) restated on Q and R^-1
// instead of on a transformed copy of the whole matrix.
//
// Every function is called by ONE full wavefront (64 lanes active); lane l owns row / column l of the small matrices
// so the mat-vecs need no cross-lane reduction.  Matrices are pitched by an odd RP: lane-strided accesses are
// bank-conflict free.
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

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wmin(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wmax(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wmin_int(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

struct InvQr {
    double *Qt, *Wi, *cv, *ab;   // LDS: R x RP, R x RP, R, R
    int R, RP;
};

struct QrTrial {
    double qi;     // lane < R: (Qt g)[lane]
    double cvl;    // lane < R: cv[lane]
    double diag, up, bb, sm, ck;
    double unorm;       // || (Qt g)[:k] ||: the part of g inside the span of the working columns
    bool independent;   // the 0.01 independence test of Lawson-Hanson passed
};

// Householder data for appending the column g (lane < R holds g[lane]) at position k.
__device__ __forceinline__ QrTrial qr_trial(const InvQr &F, int k, double gl)
{
    const int lane = threadIdx.x & 63, R = F.R, RP = F.RP;
    QrTrial t;
    if (lane < R) F.ab[lane] = gl;
    wave_sync_lds();
    double qi = 0.0;
    if (lane < R)
        for (int l = 0; l < R; ++l) qi = fma(F.Qt[lane * RP + l], F.ab[l], qi);
    const double cvl = (lane < R) ? F.cv[lane] : 0.0;
    const double q2 = qi * qi;
    const double un2 = wsum(lane < k ? q2 : 0.0);
    const double sg2 = wsum(lane >= k ? q2 : 0.0);
    const double dotc = wsum(lane > k ? cvl * qi : 0.0);
    const double qk = __shfl(qi, k);
    const double ckk = __shfl(cvl, k);
    const double sigma = sqrt(sg2), unorm = sqrt(un2);
    t.qi = qi; t.cvl = cvl; t.independent = false; t.unorm = unorm;
    t.diag = t.up = t.bb = t.sm = t.ck = 0.0;
    if (sigma > 0.0) {
        t.diag = qk > 0.0 ? -sigma : sigma;
        t.up = qk - t.diag;
        const double t1 = unorm + fabs(t.diag) * 0.01;
        if (t1 - unorm > 0.0) {
            t.independent = true;
            t.bb = t.up * t.diag;
            t.sm = (ckk * t.up + dotc) / t.bb;
            t.ck = ckk + t.sm * t.up;
        }
    }
    wave_sync_lds();
    return t;
}

// Append the column of a passed trial at position k (the caller increments k).
__device__ __forceinline__ void qr_add(const InvQr &F, int k, const QrTrial &t)
{
    const int lane = threadIdx.x & 63, R = F.R, RP = F.RP;
    if (lane < R) F.ab[lane] = (lane == k) ? t.up : t.qi;   // reflector v (entries >= k) and rho = q[:k]
    if (lane == k) F.cv[lane] = t.ck;
    else if (lane > k && lane < R) F.cv[lane] = fma(t.sm, t.qi, t.cvl);
    wave_sync_lds();
    if (lane < R) {   // Qt <- (I + v v'/bb) Qt, lane = column of Qt
        double s = 0.0;
        for (int i = k; i < R; ++i) s = fma(F.Qt[i * RP + lane], F.ab[i], s);
        s /= t.bb;
        for (int i = k; i < R; ++i) F.Qt[i * RP + lane] = fma(s, F.ab[i], F.Qt[i * RP + lane]);
    }
    if (lane < k) {   // new column of Wi = Rk^-1:  -(Wi rho) / diag
        double s = 0.0;
        for (int c = lane; c < k; ++c) s = fma(F.Wi[lane * RP + c], F.ab[c], s);
        F.Wi[lane * RP + k] = -s / t.diag;
    }
    if (lane == k) F.Wi[k * RP + k] = 1.0 / t.diag;
    wave_sync_lds();
}

// Remove the column at position jj of k (the caller decrements k and shifts its per-position registers).
__device__ __forceinline__ void qr_drop(const InvQr &F, int k, int jj)
{
    const int lane = threadIdx.x & 63, R = F.R, RP = F.RP;
    for (int l = jj; l < k - 1; ++l) {
        const double xx = F.Wi[jj * RP + l], yy = F.Wi[jj * RP + l + 1];
        const double rho = sqrt(xx * xx + yy * yy);
        double cc = 1.0, ss = 0.0;
        if (rho > 0.0) { cc = yy / rho; ss = xx / rho; }
        const double c0 = F.cv[l], c1 = F.cv[l + 1];
        wave_sync_lds();
        if (lane <= l + 1) {   // rows of the (still upper-triangular) Wi
            const double a = (lane <= l) ? F.Wi[lane * RP + l] : 0.0;
            const double b = F.Wi[lane * RP + l + 1];
            F.Wi[lane * RP + l] = cc * a - ss * b;
            F.Wi[lane * RP + l + 1] = ss * a + cc * b;
        }
        if (lane < R) {
            const double a = F.Qt[l * RP + lane], b = F.Qt[(l + 1) * RP + lane];
            F.Qt[l * RP + lane] = cc * a - ss * b;
            F.Qt[(l + 1) * RP + lane] = ss * a + cc * b;
        }
        if (lane == 0) {
            F.cv[l] = cc * c0 - ss * c1;
            F.cv[l + 1] = ss * c0 + cc * c1;
        }
        wave_sync_lds();
    }
    for (int i = jj; i < k - 1; ++i) {   // delete row jj of Wi (and implicitly its last column)
        const double v = (lane < k - 1) ? F.Wi[(i + 1) * RP + lane] : 0.0;
        wave_sync_lds();
        if (lane < k - 1) F.Wi[i * RP + lane] = v;
    }
    wave_sync_lds();
}

// lane < k: coefficient of working column lane,  (Wi cv[:k])[lane]
__device__ __forceinline__ double qr_coeff(const InvQr &F, int k)
{
    const int lane = threadIdx.x & 63, RP = F.RP;
    double z = 0.0;
    if (lane < k)
        for (int c = lane; c < k; ++c) z = fma(F.Wi[lane * RP + c], F.cv[c], z);
    return z;
}

}  // namespace pympc
