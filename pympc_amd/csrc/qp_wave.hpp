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

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS operations of one wavefront execute in program order; this only stops the compiler from moving
    // accesses across the point where other lanes' LDS writes are consumed.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_min_int(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

// per-wave LDS doubles for R rows, n variables, m columns
__host__ __device__ inline int wave_lds_doubles(int n, int m, int RP)
{
    const int R = n + 1;
    const int nw = (m + 31) / 32;
    return 2 * R * RP + 3 * R + n + (nw + 1) / 2 + 1;
}

// Solve problem `prob`.  Mt / m0 / Gm point to LDS (shared-matrix launch) or to this problem's global workspace.
template <int MT>
__device__ void qp_wave_solve(const SolveParams &P, const double *__restrict__ Mt, const double *__restrict__ m0,
                              const double *__restrict__ Gm, const double *__restrict__ wsg, long long prob,
                              double *wl, bool prepared_ok)
{
    constexpr int mp = 64 * MT;
    const int lane = threadIdx.x & 63;
    const int n = P.n, m = P.m, p = P.p, R = n + 1, RP = P.RP;
    double *Qt = wl;
    double *Wi = Qt + R * RP;
    double *cv = Wi + R * RP;
    double *ab = cv + R;
    double *rb = ab + R;
    double *ub = rb + R;
    uint32_t *bits = reinterpret_cast<uint32_t *>(ub + n);
    const double *x0 = P.X ? P.X + prob * P.x_stride : nullptr;

    // ---- m(x) = m0 + Gm' x : the per-problem row of the NNLS matrix (pnnls.py:214)
    double mv[MT], w[MT];
#pragma unroll
    for (int s = 0; s < MT; ++s) mv[s] = m0[lane + 64 * s];
    for (int q = 0; q < p; ++q) {
        const double xq = x0[q];
#pragma unroll
        for (int s = 0; s < MT; ++s) mv[s] = fma(Gm[q * mp + lane + 64 * s], xq, mv[s]);
    }

    uint32_t inP = 0;   // bit s: column lane + 64 s is in the passive set
    int k = 0, iter = 0, outer = 0, status = prepared_ok ? kStatusOptimal : kStatusNotPD;
    int Pl = -1;        // lane l < k: column index at position l of the passive set
    double yl = 0.0;    // ... its coefficient
    double mPl = 0.0;   // ... its m(x) entry
    const int itmax = 3 * m;
    bool qt_ready = false;

    if (lane < R) cv[lane] = (lane == n) ? 1.0 : 0.0;   // cv = Qt e_R with Qt = I
    wave_lds_sync();

    while (status == kStatusOptimal && k < R && k < m) {
        // ---------------- dual sweep  w = A'(b - A y) = A' Qt'[0; cv[k:]]
        if (!qt_ready) {
#pragma unroll
            for (int s = 0; s < MT; ++s) w[s] = -mv[s];
        } else {
            double ri = 0.0;
            if (lane < R)
                for (int l = k; l < R; ++l) ri = fma(Qt[l * RP + lane], cv[l], ri);
            if (lane < R) rb[lane] = ri;
            wave_lds_sync();
            double acc[MT];
            const double rn = rb[n];
#pragma unroll
            for (int s = 0; s < MT; ++s) acc[s] = mv[s] * rn;
            for (int i = 0; i < n; ++i) {
                const double rv = rb[i];
                const double *row = Mt + i * mp + lane;
#pragma unroll
                for (int s = 0; s < MT; ++s) acc[s] = fma(row[64 * s], rv, acc[s]);
            }
#pragma unroll
            for (int s = 0; s < MT; ++s) w[s] = -acc[s];
        }
#pragma unroll
        for (int s = 0; s < MT; ++s)
            if (lane + 64 * s >= m || ((inP >> s) & 1u)) w[s] = -1.0;

        // ---------------- pick the first maximum of w that passes the Lawson-Hanson acceptance tests
        bool accepted = false;
        int j = -1, jl = 0, js = 0;
        double qi = 0.0, cvl = 0.0, diag = 0.0, up = 0.0, sm = 0.0, ck = 0.0, bb = 0.0, mj = 0.0;
        for (;;) {
            double best = w[0];
            int bs = 0;
#pragma unroll
            for (int s = 1; s < MT; ++s)
                if (w[s] > best) { best = w[s]; bs = s; }
            const double gmax = wave_max(best);
            if (!(gmax > 0.0)) break;
            j = wave_min_int(best == gmax ? lane + 64 * bs : 0x7fffffff);
            jl = j & 63;
            js = j >> 6;
            double mvsel = mv[0];
#pragma unroll
            for (int s = 1; s < MT; ++s) mvsel = (s == js) ? mv[s] : mvsel;
            mj = __shfl(mvsel, jl);
            double ai = 0.0;
            if (lane < n) ai = -Mt[lane * mp + j];
            else if (lane == n) ai = -mj;
            if (!qt_ready) {
                qi = ai;
            } else {
                if (lane < R) ab[lane] = ai;
                wave_lds_sync();
                qi = 0.0;
                if (lane < R)
                    for (int l = 0; l < R; ++l) qi = fma(Qt[lane * RP + l], ab[l], qi);
            }
            cvl = (lane < R) ? cv[lane] : 0.0;
            const double q2 = qi * qi;
            const double un2 = wave_sum(lane < k ? q2 : 0.0);
            const double sg2 = wave_sum(lane >= k ? q2 : 0.0);
            const double dotc = wave_sum(lane > k ? cvl * qi : 0.0);
            const double qk = __shfl(qi, k);
            const double ckk = __shfl(cvl, k);
            const double sigma = sqrt(sg2), unorm = sqrt(un2);
            bool ok = false;
            if (sigma > 0.0) {
                diag = qk > 0.0 ? -sigma : sigma;
                up = qk - diag;
                const double t1 = unorm + fabs(diag) * 0.01;
                if (t1 - unorm > 0.0) {
                    bb = up * diag;
                    sm = (ckk * up + dotc) / bb;
                    ck = ckk + sm * up;
                    ok = (ck / diag) > 0.0;
                }
            }
            if (ok) { accepted = true; break; }
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
            for (int e = lane; e < R * RP; e += 64) Qt[e] = ((e / RP) == (e % RP)) ? 1.0 : 0.0;
            qt_ready = true;
        }
        if (lane < R) ab[lane] = (lane == k) ? up : qi;   // reflector v (entries >= k) and rho = q[:k]
        if (lane == k) cv[lane] = ck;
        else if (lane > k && lane < R) cv[lane] = fma(sm, qi, cvl);
        wave_lds_sync();
        if (lane < R) {   // Qt <- (I + v v'/bb) Qt, lane = column of Qt
            double s = 0.0;
            for (int i = k; i < R; ++i) s = fma(Qt[i * RP + lane], ab[i], s);
            s /= bb;
            for (int i = k; i < R; ++i) Qt[i * RP + lane] = fma(s, ab[i], Qt[i * RP + lane]);
        }
        if (lane < k) {   // new column of Wi = Rk^-1:  -(Wi rho) / diag
            double s = 0.0;
            for (int t = 0; t < k; ++t) {
                const double wv = Wi[lane * RP + t];
                s = (t >= lane) ? fma(wv, ab[t], s) : s;
            }
            Wi[lane * RP + k] = -s / diag;
        }
        if (lane == k) {
            Wi[k * RP + k] = 1.0 / diag;
            Pl = j;
            yl = 0.0;
            mPl = mj;
        }
        if (lane == jl) inP |= (1u << js);
        ++k;
        wave_lds_sync();

        // ---------------- secondary loop
        double zl = 0.0;
        for (;;) {
            zl = 0.0;
            if (lane < k)
                for (int t = 0; t < k; ++t) {
                    const double wv = Wi[lane * RP + t];
                    const double c = cv[t];
                    zl = (t >= lane) ? fma(wv, c, zl) : zl;
                }
            ++iter;
            if (iter > itmax) { status = kStatusMaxIter; break; }
            const bool bad = (lane < k) && (zl <= 0.0);
            if (__ballot(bad) == 0ull) {
                if (lane < k) yl = zl;
                break;
            }
            const double t = bad ? yl / (yl - zl) : 1.0e300;
            const double alpha = wave_min(t);
            int jj = wave_min_int((bad && t == alpha) ? lane : 64);
            if (lane < k) yl = fma(alpha, zl - yl, yl);
            for (;;) {
                // ---- drop position jj: Givens rotations that zero row jj of Wi (columns jj .. k-2)
                const int jdel = __shfl(Pl, jj);
                if (lane == (jdel & 63)) inP &= ~(1u << (jdel >> 6));
                for (int l = jj; l < k - 1; ++l) {
                    const double xx = Wi[jj * RP + l], yy = Wi[jj * RP + l + 1];
                    const double rho = sqrt(xx * xx + yy * yy);
                    double cc = 1.0, ss = 0.0;
                    if (rho > 0.0) { cc = yy / rho; ss = xx / rho; }
                    if (lane <= l + 1) {   // rows of the (still upper-triangular) Wi
                        const double a = (lane <= l) ? Wi[lane * RP + l] : 0.0;
                        const double b = Wi[lane * RP + l + 1];
                        Wi[lane * RP + l] = cc * a - ss * b;
                        Wi[lane * RP + l + 1] = ss * a + cc * b;
                    }
                    if (lane < R) {
                        const double a = Qt[l * RP + lane], b = Qt[(l + 1) * RP + lane];
                        Qt[l * RP + lane] = cc * a - ss * b;
                        Qt[(l + 1) * RP + lane] = ss * a + cc * b;
                    }
                    const double c0 = cv[l], c1 = cv[l + 1];
                    wave_lds_sync();
                    if (lane == 0) {
                        cv[l] = cc * c0 - ss * c1;
                        cv[l + 1] = ss * c0 + cc * c1;
                    }
                    wave_lds_sync();
                }
                for (int i = jj; i < k - 1; ++i) {   // delete row jj of Wi (and implicitly its last column)
                    const double v = (lane < k - 1) ? Wi[(i + 1) * RP + lane] : 0.0;
                    wave_lds_sync();
                    if (lane < k - 1) Wi[i * RP + lane] = v;
                }
                wave_lds_sync();
                const int Pn = __shfl_down(Pl, 1);
                const double yn = __shfl_down(yl, 1), mn = __shfl_down(mPl, 1);
                if (lane >= jj && lane < k - 1) { Pl = Pn; yl = yn; mPl = mn; }
                --k;
                const unsigned long long neg = __ballot(lane < k && yl <= 0.0);
                if (neg == 0ull) break;
                jj = __ffsll((long long)neg) - 1;
            }
        }
    }

    // ---------------- residual norm and outputs (pnnls.py:230-239, parametric_programs.py:176-177)
    const double cl = (lane >= k && lane < R) ? cv[lane] : 0.0;
    const double rnorm = sqrt(wave_sum(cl * cl));
    if (status == kStatusOptimal && !(rnorm > P.tol)) status = kStatusInfeasible;
    const bool okk = (status == kStatusOptimal);
    const int nw = (m + 31) / 32;
    const double nanv = __builtin_nan("");

    double lam = 0.0;
    if (okk) {
        const double my = wave_sum(lane < k ? mPl * yl : 0.0);
        lam = (lane < k) ? yl / (1.0 + my) : 0.0;
    }
    if (P.argmin || P.vmin) {
        double ui = nanv, V = nanv;
        if (okk) {
            const WsLayout &L = P.L;
            ui = 0.0;
            double fi = 0.0;
            if (lane < n) {
                ui = wsg[L.u0 + lane];
                fi = wsg[L.fu + lane];
                for (int q = 0; q < p; ++q) {
                    ui = fma(wsg[L.Ku + lane * p + q], x0[q], ui);
                    fi = fma(wsg[L.Hux + lane * p + q], x0[q], fi);
                }
            }
            for (int l = 0; l < k; ++l) {
                const double ll = __shfl(lam, l);
                const int pj = __shfl(Pl, l);
                if (lane < n) ui = fma(-ll, wsg[L.T + (size_t)pj * n + lane], ui);
            }
            if (P.vmin) {
                if (lane < n) ub[lane] = ui;
                wave_lds_sync();
                double term = 0.0;
                if (lane < n) {
                    double hi = 0.0;
                    for (int l = 0; l < n; ++l) hi = fma(wsg[L.Huu + lane * n + l], ub[l], hi);
                    term = ui * (0.5 * hi + fi);
                }
                if (lane < p) {
                    double hx = 0.0;
                    for (int q = 0; q < p; ++q) hx = fma(wsg[L.Hxx + lane * p + q], x0[q], hx);
                    term += x0[lane] * (0.5 * hx + wsg[L.fx + lane]);
                }
                V = wave_sum(term) + wsg[L.g];
            }
        }
        if (P.argmin && lane < P.n_out) P.argmin[prob * P.n_out + lane] = ui;
        if (P.vmin && lane == 0) P.vmin[prob] = V;
    }
    if (P.mult) {
#pragma unroll
        for (int s = 0; s < MT; ++s) {
            const int jc = lane + 64 * s;
            if (jc < m && !(okk && ((inP >> s) & 1u))) P.mult[prob * m + jc] = okk ? 0.0 : nanv;
        }
        if (okk && lane < k) P.mult[prob * m + Pl] = lam;
    }
    if (P.active) {
        for (int wd = lane; wd < nw; wd += 64) bits[wd] = 0u;
        wave_lds_sync();
        if (okk && lane < k && lam > P.tol) atomicOr(&bits[Pl >> 5], 1u << (Pl & 31));
        wave_lds_sync();
        for (int wd = lane; wd < nw; wd += 64) P.active[prob * nw + wd] = bits[wd];
    }
    if (lane == 0) {
        P.status[prob] = status;
        if (P.iters) {
            P.iters[2 * prob] = outer;
            P.iters[2 * prob + 1] = iter;
        }
    }
    wave_lds_sync();
}

// Workgroup = NW waves; it owns problems [blockIdx.x * chunk, ...) and hands them to its waves through an LDS
// ticket.  SHARED: the constant rows Mt / m0 / Gm are staged in LDS once per workgroup.
// waves per workgroup: 8 (256 VGPRs per lane) up to 8 columns per lane, 4 (512 VGPRs) beyond, so that the
// per-lane column arrays mv / w / acc never spill to scratch
template <int MT>
constexpr int solve_max_waves() { return MT <= 8 ? 8 : 4; }

template <int MT, bool SHARED>
__global__ void __launch_bounds__(64 * solve_max_waves<MT>()) qp_solve_kernel(SolveParams P)
{
    extern __shared__ double lds[];
    constexpr int mp = 64 * MT;
    const int n = P.n, p = P.p;
    const size_t sharedD = SHARED ? (size_t)(n + 1 + p) * mp : 0;
    int *ticket = reinterpret_cast<int *>(lds + sharedD);
    double *waves = lds + sharedD + 2;
    const int wave = threadIdx.x >> 6;
    if (SHARED) {
        const double2 *src = reinterpret_cast<const double2 *>(P.ws);
        double2 *dst = reinterpret_cast<double2 *>(lds);
        for (size_t e = threadIdx.x; e < sharedD / 2; e += blockDim.x) dst[e] = src[e];
    }
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
            qp_wave_solve<MT>(P, lds + P.L.Mt, lds + P.L.m0, lds + P.L.Gm, P.ws, prob, wl, shared_ok);
        } else {
            const double *wsg = P.ws + prob * P.ws_stride;
            const bool ok = P.ws_status ? (P.ws_status[prob] == 0) : true;
            qp_wave_solve<MT>(P, wsg + P.L.Mt, wsg + P.L.m0, wsg + P.L.Gm, wsg, prob, wl, ok);
        }
    }
}

}  // namespace pympc
