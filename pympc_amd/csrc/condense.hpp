// Batched condensing of the optimal control problem of a PWA system, one mode sequence per wavefront
// (/root/reference/pympc/control/controllers.py:682-749 condense_optimal_control_problem,
//  /root/reference/pympc/dynamics/discrete_time_systems.py:676-751 condense_pwa_system / productory).
//
// For the mode sequence z(0..N-1) the reference builds x_bar = A_bar x0 + B_bar u_bar + c_bar with an O(N^2) Python
// loop of small products, then Huu = R_bar + B_bar'Q_bar B_bar, Hux, Hxx, fu, fx, g and Au = G_bar + F_bar B_bar,
// Ax = F_bar A_bar, b = h_bar - F_bar c_bar with block-diagonal Q_bar, F_bar, G_bar materialised densely.
//
// Here the three prediction matrices are one matrix  S = [A_bar | B_bar | c_bar]  ((N+1) nx rows, W = nx + N nu + 1
// columns) whose block rows obey   S_0 = [I 0 0],   S_{t+1} = A_z(t) S_t + [0 | B_z(t) at block column t | c_z(t)],
// and everything else is a contraction of S with a matrix that only couples ONE block row:
//        Gram = sum_t S_t' Q_t S_t   (W x W;  Hxx, Hux, Huu - R_bar, fx, fu, 2 g are its blocks),
//        E_t  = F_z(t) S_t           (rows of [Ax | Au - G | h - b] of stage t),   E_N = F_N S_N.
// Every product has inner dimension nx, so it maps onto v_mfma_f64_16x16x4_f64 (the one genuinely dense contraction
// of the path): 16 output rows x 16 output columns per instruction, k = 4 state components per step.
//
// MI355X mapping: one wavefront (= one workgroup) per mode sequence; S lives in LDS (84 x 32 doubles for the
// cart-pole N = 20); the accumulator tile of Q_t S_t is used directly as the B operand of the Gram product (for the
// f64 16x16x4 form register r of the accumulator holds row (lane >> 4) + 4 r, which is exactly the k index the B
// operand of k-step r wants), so it never touches LDS.  Outputs go straight from accumulator registers to HBM, 16
// consecutive doubles per row segment; the kernel is bound by those writes.
//
// Stages may have different numbers of domain rows; the caller pads every mode's (F, G, h) to rmax rows with
// F = 0, G = 0, h = 1 (a constraint 0 <= 1 that can never become active), so that all problems share one shape.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pympc {

constexpr int kCondNxMax = 16;      // states (inner dimension, <= 4 k-steps)
constexpr int kCondCtMax = 5;       // column tiles of S: W = nx + N nu + 1 <= 80
constexpr int kStatusBadMode = 5;   // a mode index outside [0, nm)

struct CondParams {
    int nx, nu, N, nm, rmax, rN;
    int n, m, W, CT, pitch;        // n = N nu, m = N rmax + rN, W = nx + n + 1, CT = ceil(W / 16), pitch of S in LDS
    const int32_t *modes;          // [B x N]
    const double *As, *Bs, *cs;    // [nm x nx x nx], [nm x nx x nu], [nm x nx]
    const double *Q, *R, *P;       // [nx x nx], [nu x nu], [nx x nx]
    const double *Fd, *Gd, *hd;    // [nm x rmax x nx], [nm x rmax x nu], [nm x rmax]
    const double *FN, *hN;         // [rN x nx], [rN]
    long long B;
    double *Huu, *Hux, *Hxx, *fu, *fx, *g, *Au, *Ax, *b;   // [B x ...] contiguous per problem
    double *Sbar;                  // [B x (N+1) nx x W] or null: [A_bar | B_bar | c_bar]
    int32_t *status;               // [B]
};

using cond_d4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ cond_d4 mfma_f64(double a, double b, cond_d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// No wait states and no register constraints around the MFMA chains: the compiler's own hazard handling is correct for
// straight-line code (csrc/probes/mfma_f64_hazard.hip: 3 x 4096 chains read back exactly).  What made rounds 1 and 2 add
// pads and then an empty `asm volatile("" : "+a"(acc))` was a hazard at a CONTROL-FLOW JOIN (profiles/r03/mfma_settle_isa.txt):
// with the column-tile count CT a run-time value the Gram loop was a chain of `if (I < CT) mfma`; the register allocator
// kept acc[0] in VGPRs across iterations and bounced it through a[24:31]; the join behind the branch that skips the second
// MFMA is reached either 13 wait states after the first MFMA (branch taken, CT == 1) or much later (fall-through), and
// hipcc 7.2 sized the pad before the first v_accvgpr_read there (a31: the high half of accumulator element 3) for the
// fall-through path only: s_nop 11 instead of the 17 it emits everywhere else.  Exactly the "element 3 stale, small
// problems only" symptom of the three failing parity tests.  The empty asm only moved the allocation away from that
// shape.  The fix is structural: CT is a template parameter, every loop over column tiles is unrolled at compile time
// and no branch separates an MFMA from the first read of its accumulator.
__device__ __forceinline__ void cond_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int KS, int CT>   // k-steps: ceil(nx / 4); column tiles of S: ceil(W / 16)
__global__ void __launch_bounds__(64) condense_kernel(CondParams P)
{
    extern __shared__ double S[];   // [(N+1) nx][pitch]
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const long long prob = blockIdx.x;
    const int nx = P.nx, nu = P.nu, N = P.N, n = P.n, W = P.W, pitch = P.pitch;
    const int32_t *z = P.modes + prob * N;

    bool bad = false;
    for (int t = lane; t < N; t += 64) bad |= (z[t] < 0 || z[t] >= P.nm);
    if (__ballot(bad) != 0ull) {
        if (lane == 0) {
            P.status[prob] = kStatusBadMode;
            P.g[prob] = __builtin_nan("");
        }
        return;
    }

    for (int e = lane; e < (N + 1) * nx * pitch; e += 64) S[e] = 0.0;
    cond_lds_sync();
    if (lane < nx) S[lane * pitch + lane] = 1.0;
    cond_lds_sync();

    // ---------------- prediction recursion  S_{t+1} = A_z S_t + [0 | B_z | c_z]   (discrete_time_systems.py:711-727)
    for (int t = 0; t < N; ++t) {
        const int zt = z[t];
        const double *Az = P.As + (size_t)zt * nx * nx;
        const double *Bz = P.Bs + (size_t)zt * nx * nu;
        const double *cz = P.cs + (size_t)zt * nx;
        double a[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kk = 4 * s + lk;
            a[s] = (li < nx && kk < nx) ? Az[li * nx + kk] : 0.0;
        }
        for (int ct = 0; ct < CT; ++ct) {
            cond_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int kk = 4 * s + lk;
                const double bv = kk < nx ? S[(t * nx + kk) * pitch + 16 * ct + li] : 0.0;
                acc = mfma_f64(a[s], bv, acc);
            }
            const int col = 16 * ct + li;
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                const int row = lk + 4 * r;
                if (row < nx && col < W) {
                    double v = acc[r];
                    const int cu = col - nx - t * nu;
                    if (cu >= 0 && cu < nu) v += Bz[row * nu + cu];
                    if (col == W - 1) v += cz[row];
                    S[((t + 1) * nx + row) * pitch + col] = v;
                }
            }
        }
        cond_lds_sync();
    }
    if (P.Sbar) {
        double *out = P.Sbar + (size_t)prob * (N + 1) * nx * W;
        for (int e = lane; e < (N + 1) * nx * W; e += 64) out[e] = S[(e / W) * pitch + (e % W)];
    }

    // ---------------- cost: Gram = sum_t S_t' (Q_t S_t)   (controllers.py:716-731)
    double *Huu = P.Huu + (size_t)prob * n * n, *Hux = P.Hux + (size_t)prob * n * nx, *Hxx = P.Hxx + (size_t)prob * nx * nx;
    double *fu = P.fu + (size_t)prob * n, *fx = P.fx + (size_t)prob * nx;
#pragma unroll 1
    for (int J = 0; J < CT; ++J) {
        cond_d4 acc[CT];
#pragma unroll
        for (int I = 0; I < CT; ++I) acc[I] = cond_d4{0.0, 0.0, 0.0, 0.0};
        for (int t = 0; t <= N; ++t) {
            const double *Qt = (t == N) ? P.P : P.Q;
            cond_d4 qs = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int kk = 4 * s + lk;
                const double qa = (li < nx && kk < nx) ? Qt[li * nx + kk] : 0.0;
                const double bv = kk < nx ? S[(t * nx + kk) * pitch + 16 * J + li] : 0.0;
                qs = mfma_f64(qa, bv, qs);
            }
            // register s of qs holds row (lane >> 4) + 4 s of Q_t S_t: the B operand of k-step s, in place
#pragma unroll
            for (int I = 0; I < CT; ++I) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int kk = 4 * s + lk;
                    const double av = kk < nx ? S[(t * nx + kk) * pitch + 16 * I + li] : 0.0;
                    acc[I] = mfma_f64(av, qs[s], acc[I]);
                }
            }
        }
        const int gj = 16 * J + li;
#pragma unroll
        for (int I = 0; I < CT; ++I) {
            {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = 16 * I + lk + 4 * r;
                    if (gi >= W || gj >= W) continue;
                    const double v = acc[I][r];
                    const int ui = gi - nx, uj = gj - nx;
                    if (gi < nx) {
                        if (gj < nx) Hxx[gi * nx + gj] = v;
                        else if (gj == W - 1) fx[gi] = v;
                    } else if (gi < W - 1) {
                        if (gj < nx) Hux[ui * nx + gj] = v;
                        else if (gj == W - 1) fu[ui] = v;
                        else Huu[ui * n + uj] = v + ((ui / nu == uj / nu) ? P.R[(ui % nu) * nu + (uj % nu)] : 0.0);
                    } else if (gj == W - 1) {
                        P.g[prob] = 0.5 * v;
                    }
                }
            }
        }
    }

    // ---------------- constraints: E_t = F_z(t) S_t, E_N = F_N S_N   (controllers.py:733-747)
    double *Au = P.Au + (size_t)prob * P.m * n, *Ax = P.Ax + (size_t)prob * P.m * nx, *bo = P.b + (size_t)prob * P.m;
    for (int t = 0; t <= N; ++t) {
        const bool term = (t == N);
        const int zt = term ? 0 : z[t];
        const int rows = term ? P.rN : P.rmax;
        const double *F = term ? P.FN : P.Fd + (size_t)zt * P.rmax * nx;
        const double *G = term ? nullptr : P.Gd + (size_t)zt * P.rmax * nu;
        const double *h = term ? P.hN : P.hd + (size_t)zt * P.rmax;
        for (int rt = 0; rt * 16 < rows; ++rt) {
            double a[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int kk = 4 * s + lk, row = 16 * rt + li;
                a[s] = (row < rows && kk < nx) ? F[(size_t)row * nx + kk] : 0.0;
            }
            for (int ct = 0; ct < CT; ++ct) {
                cond_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int kk = 4 * s + lk;
                    const double bv = kk < nx ? S[(t * nx + kk) * pitch + 16 * ct + li] : 0.0;
                    acc = mfma_f64(a[s], bv, acc);
                }
                    const int col = 16 * ct + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + lk + 4 * r;
                    if (row >= rows || col >= W) continue;
                    const size_t grow = (size_t)t * P.rmax + row;
                    const double v = acc[r];
                    if (col < nx) {
                        Ax[grow * nx + col] = v;
                    } else if (col < W - 1) {
                        const int cu = col - nx;
                        double gv = 0.0;
                        if (!term && cu / nu == t) gv = G[row * nu + cu % nu];
                        Au[grow * n + cu] = v + gv;
                    } else {
                        bo[grow] = h[row] - v;
                    }
                }
            }
        }
    }
    if (lane == 0) P.status[prob] = 0;
}

}  // namespace pympc
