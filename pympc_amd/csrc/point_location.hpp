// Explicit-MPC point location + affine law evaluation (one state per lane).
//
// Reference semantics (/root/reference/pympc/optimization/parametric_programs.py:442-463): scan the critical
// regions in list order and take the FIRST one whose polyhedron contains x, where contains is
// max(A x - b) <= tol (/root/reference/pympc/geometry/polyhedron.py:596, tol = 1e-7); then
// u = ux x + u0 (parametric_programs.py:364) and V = .5 x'Vxx x + Vx'x + V0 (:398); uncovered points give None.
//
// MI355X mapping: the streaming side is HBM-bound (p doubles in, 4 + 8 nu + 8 bytes out per point, coalesced),
// the table side is kept out of HBM: all region rows (A | b) are staged once per workgroup in LDS and every lane
// of a wavefront scans the SAME region row at the same time, so each ds_read is a broadcast (no bank conflicts).
// A region is abandoned as soon as every still-searching lane of the wave has violated one of its rows, and the
// scan stops when every lane has found its region; neighbouring grid points mostly share regions, so the wave
// rarely does more rows than its slowest lane needs.
// Optional pruning that keeps the first-hit semantics exactly: `box` holds, per region, an axis-aligned box that
// contains every point the row test can accept (the host computes it with 2 p LPs over {A x <= b + tol}, inflated);
// a region whose box contains none of the wave's still-searching points is skipped without touching its rows.
// The boxes are tested 64 at a time (lane <-> region) against the axis-aligned hull of the wave's 64 points, so a wave
// of neighbouring grid points visits the handful of regions around it instead of walking the whole list.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "invqr_wave.hpp"

namespace pympc {

struct PlParams {
    int nreg, p, nu, total_rows, table_in_lds, box_in_lds;
    const double *A, *b;
    const int32_t *row_ptr;
    const double *ux, *u0, *Vxx, *Vx, *V0;
    const double *box;   // [nreg x 2 p] (lower bounds, then upper bounds) or null
    const double *X;
    long long B;
    double tol;
    int32_t *idx;
    double *u, *V;
};

template <int PD>
__global__ void point_location_kernel(PlParams P)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const double *At = P.A, *bt = P.b;
    if (P.table_in_lds) {
        const int na = P.total_rows * PD;
        for (int e = tid; e < na; e += blockDim.x) lds[e] = P.A[e];
        for (int e = tid; e < P.total_rows; e += blockDim.x) lds[na + e] = P.b[e];
        __syncthreads();
        At = lds;
        bt = lds + na;
    }
    const double *box = P.box;
    if (P.box && P.box_in_lds) {
        double *bl = lds + (P.table_in_lds ? P.total_rows * (PD + 1) : 0);
        for (int e = tid; e < P.nreg * 2 * PD; e += blockDim.x) bl[e] = P.box[e];
        __syncthreads();
        box = bl;
    }
    const double nanv = __builtin_nan("");
    const long long stride = (long long)gridDim.x * blockDim.x;
    // every lane of a wave runs the same number of grid-stride rounds (wave-uniform loop bound)
    const long long base0 = (long long)blockIdx.x * blockDim.x + (tid & ~63);
    for (long long base = base0; base < P.B; base += stride) {
        const long long pt = base + (tid & 63);
        const bool valid = pt < P.B;
        double x[PD];
#pragma unroll
        for (int q = 0; q < PD; ++q) x[q] = valid ? P.X[pt * PD + q] : 0.0;
        int found = -1;
        // axis-aligned hull of the wave's points: a region whose box misses it holds none of them
        double wlo[PD], whi[PD];
        if (box) {
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                wlo[q] = wmin(valid ? x[q] : 1.0e300);
                whi[q] = wmax(valid ? x[q] : -1.0e300);
            }
        }
        for (int r0 = 0; r0 < P.nreg; r0 += 64) {
            if (__ballot(valid && found < 0) == 0ull) break;
            // candidate regions of this block of 64, lane <-> region
            unsigned long long cand = ~0ull;
            if (box) {
                const int rl = r0 + (tid & 63);
                bool hit = rl < P.nreg;
                const double *lo = box + (size_t)(hit ? rl : 0) * 2 * PD, *hi = lo + PD;
#pragma unroll
                for (int q = 0; q < PD; ++q) hit = hit && (lo[q] <= whi[q]) && (hi[q] >= wlo[q]);
                cand = __ballot(hit);
            } else if (P.nreg - r0 < 64) {
                cand = (1ull << (P.nreg - r0)) - 1ull;
            }
            while (cand != 0ull) {
                const int r = r0 + __ffsll((long long)cand) - 1;   // list order: the first containing region wins
                cand &= cand - 1ull;
                bool inside = valid && found < 0;
                if (__ballot(inside) == 0ull) break;
                if (box) {
                    const double *lo = box + (size_t)r * 2 * PD, *hi = lo + PD;
#pragma unroll
                    for (int q = 0; q < PD; ++q) inside = inside && (x[q] >= lo[q]) && (x[q] <= hi[q]);
                    if (__ballot(inside) == 0ull) continue;
                }
                const int i1 = P.row_ptr[r + 1];
                for (int i = P.row_ptr[r]; i < i1; ++i) {
                    double s = -bt[i];
#pragma unroll
                    for (int q = 0; q < PD; ++q) s = fma(At[i * PD + q], x[q], s);
                    inside = inside && (s <= P.tol);
                    if ((i & 3) == 3 && __ballot(inside) == 0ull) break;
                }
                if (inside) found = r;
            }
        }
        if (valid) {
            P.idx[pt] = found;
            double V = nanv;
            if (found >= 0) {
                double qd = 0.0, lin = 0.0;
#pragma unroll
                for (int i = 0; i < PD; ++i) {
                    double s = 0.0;
#pragma unroll
                    for (int q = 0; q < PD; ++q) s = fma(x[q], P.Vxx[((size_t)found * PD + q) * PD + i], s);
                    qd = fma(s, x[i], qd);
                    lin = fma(P.Vx[(size_t)found * PD + i], x[i], lin);
                }
                V = .5 * qd + lin + P.V0[found];
            }
            P.V[pt] = V;
            for (int i = 0; i < P.nu; ++i) {
                double s = nanv;
                if (found >= 0) {
                    s = P.u0[(size_t)found * P.nu + i];
#pragma unroll
                    for (int q = 0; q < PD; ++q) s = fma(P.ux[((size_t)found * P.nu + i) * PD + q], x[q], s);
                }
                P.u[pt * P.nu + i] = s;
            }
        }
    }
}

}  // namespace pympc
