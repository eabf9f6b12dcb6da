// Explicit-MPC point location + affine law evaluation (one state per lane).
//
// Reference semantics (/root/reference/pympc/optimization/parametric_programs.py:442-463): scan the critical
// regions in list order and take the FIRST one whose polyhedron contains x, where contains is
// max(A x - b) <= tol (/root/reference/pympc/geometry/polyhedron.py:596, tol = 1e-7); then
// u = ux x + u0 (parametric_programs.py:364) and V = .5 x'Vxx x + Vx'x + V0 (:398); uncovered points give None.
//
// MI355X mapping: the streaming side is HBM-bound (p doubles in, 4 + 8 nu + 8 bytes out per point, coalesced),
// the table side is kept out of HBM and out of the vector memory path: every lane of a wavefront scans the SAME
// region row at the same time, so the rows (A | b) are fetched with scalar loads through the scalar data cache / L2.
// A region is abandoned as soon as every still-searching lane of the wave has violated one of its rows, and the
// scan stops when every lane has found its region; neighbouring grid points mostly share regions, so the wave
// rarely does more rows than its slowest lane needs.
// Optional pruning that keeps the first-hit semantics exactly: `box` holds, per region, an axis-aligned box that
// contains every point the row test can accept (the host computes it with 2 p LPs over {A x <= b + tol}, inflated);
// a region whose box contains none of the wave's still-searching points is skipped without touching its rows.
// The boxes are tested 64 at a time (lane <-> region) against the axis-aligned hull of the wave's 64 points, so a wave
// of neighbouring grid points visits the handful of regions around it instead of walking the whole list.
// Critical regions are often thin slabs whose boxes overlap most of the domain, so the host may add a uniform cell
// grid over the parameter space with, per cell, the bit set of regions that no row separates from the cell
// (grid_masks); the wave ORs the bit sets of the few cells its hull touches (lane <-> cell).  Same contract as the
// boxes: bits may only be missing for regions that contain no point of the cell.
// With the grid the host may also pass grid_sure, one int32 per cell: r >= 0 when every point of the cell lies strictly
// inside region r and no earlier region can touch the cell, -1 when no region can touch it, -2 otherwise.  A point whose
// cell is decided needs no scan at all (one 4-byte gather from a table that stays in L2): point_cells_kernel settles
// those and marks the rest, and only the wavefronts with a marked point -- those that touch a region boundary -- scan,
// with the hull of their marked points.  A cell that a boundary crosses can still be decided without the scan: over one
// small cell nearly every row of a candidate region holds everywhere, so the host lists, per such cell and in region
// order, the candidates with only the rows that CROSS the cell (grid_sure = -3 - offset into grid_lists: [ncand,
// (region, nrows, row ...) ...]); the point's lane walks its cell's list -- a handful of row evaluations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "invqr_wave.hpp"
#include "qp_wave.hpp"   // cdouble_ptr / as_const

namespace pympc {

struct PlParams {
    int nreg, p, nu, total_rows;
    const double *A, *b;
    const int32_t *row_ptr;
    const double *ux, *u0, *Vxx, *Vx, *V0;
    const double *box;   // [nreg x 2 p] (lower bounds, then upper bounds) or null
    const double *grid_geom;              // [4 p]: lower corner of the cell grid, 1 / cell size per axis, then the box outside
                                          //   which a point lies in no region: its upper and its lower corner; or null
    const unsigned long long *grid_masks; // [grid_n^p x ceil(nreg / 64)]: bit r of a cell = region r can touch the cell
    const int32_t *grid_sure;             // [grid_n^p]: decided cells (see above) or null
    const int32_t *grid_lists;            // decision lists of the cells with grid_sure <= -3 (pool of int32) or null
    int grid_n;                           // cells per axis
    const double *X;
    long long B;
    double tol;
    int32_t *idx;
    double *u, *V;
};

constexpr int kPlRows = 8;   // region rows tested per batch

using cint_ptr = const int __attribute__((address_space(4))) *;

// bitwise OR of a 64-bit value over the wavefront (every lane gets the result)
__device__ __forceinline__ unsigned long long wor64(unsigned long long v)
{
    int lo = (int)(v & 0xffffffffull), hi = (int)(v >> 32);
#define PYMPC_WOR_STEP(CTRL) { lo |= dpp_i32<CTRL>(lo); hi |= dpp_i32<CTRL>(hi); }
    PYMPC_WOR_STEP(kDppXor1)
    PYMPC_WOR_STEP(kDppXor2)
    PYMPC_WOR_STEP(kDppHalfMirror)
    PYMPC_WOR_STEP(kDppMirror)
#undef PYMPC_WOR_STEP
    lo = (lane_i32(lo, 0) | lane_i32(lo, 16)) | (lane_i32(lo, 32) | lane_i32(lo, 48));
    hi = (lane_i32(hi, 0) | lane_i32(hi, 16)) | (lane_i32(hi, 32) | lane_i32(hi, 48));
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}

// Region boxes and row ranges of the first 64 regions, lane <-> region: they do not depend on the points and are fetched
// once per wavefront.
template <int PD>
struct PlAhead {
    double lo[PD], hi[PD];
    int rp0, rp1;
};

// The scan of one batch of 64 points (lane <-> point; `search` marks the lanes that still need their region): first region
// in list order whose rows all hold, -1 when there is none.
template <int PD>
__device__ __forceinline__ int pl_scan(const PlParams &P, const PlAhead<PD> &ahead, const double (&x)[PD], bool search,
                                       bool xnum, int lane)
{
    // The table is read-only for the kernel's lifetime and every row access is wave-uniform: through the constant
    // address space those become scalar loads (one s_load per batch of rows into SGPRs, the FMAs take them as scalar
    // operands) instead of 64-lane vector loads of one address, which were what bounded this kernel.
    const cdouble_ptr At = as_const(P.A), bt = as_const(P.b), box = as_const(P.box);
    int found = -1;
    // axis-aligned hull of the wave's (undecided) points: a region whose box misses it holds none of them
    double wlo[PD], whi[PD];
    constexpr bool kCorners = PD <= 4;   // hull-corner tests below: 2^p corners per row
    if (box || P.grid_masks || kCorners) {
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            wlo[q] = wmin(search ? x[q] : 1.0e300);
            whi[q] = wmax(search ? x[q] : -1.0e300);
        }
    }
    // cells of the grid that the hull touches: lane l < ncell looks one of them up
    int ncell = 0;
    long long mycell = 0;
    if (P.grid_masks) {
        int c0[PD], cn[PD];
        long long tot = 1;
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            const double g0 = P.grid_geom[q], gi = P.grid_geom[PD + q], gmax = (double)(P.grid_n - 1);
            const double a = fmin(fmax(floor((wlo[q] - g0) * gi), 0.0), gmax);
            const double b = fmin(fmax(floor((whi[q] - g0) * gi), 0.0), gmax);
            c0[q] = (int)a;
            cn[q] = (int)b - (int)a + 1;
            tot *= cn[q];
        }
        if (tot <= 64) {
            ncell = (int)tot;
            int l = min(lane, ncell - 1);
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                mycell = mycell * P.grid_n + c0[q] + l % cn[q];
                l /= cn[q];
            }
        }
    }
    const int nwords = (P.nreg + 63) >> 6;
    for (int r0 = 0; r0 < P.nreg; r0 += 64) {
        if (__ballot(search && found < 0) == 0ull) break;
        // candidate regions of this block of 64, lane <-> region
        unsigned long long cand = ~0ull;
        if (ncell > 0) {
            unsigned long long mk = 0ull;
            if (lane < ncell) mk = P.grid_masks[mycell * nwords + (r0 >> 6)];
            cand = wor64(mk);
            if (cand == 0ull) continue;
        }
        // lane <-> region: box and row range of region r0 + lane (the first block's were fetched ahead)
        int rp0 = ahead.rp0, rp1 = ahead.rp1;
        double blo[PD], bhi[PD];
#pragma unroll
        for (int q = 0; q < PD; ++q) { blo[q] = ahead.lo[q]; bhi[q] = ahead.hi[q]; }
        if (r0 > 0) {
            const int rl = min(r0 + lane, P.nreg - 1);
            rp0 = P.row_ptr[rl];
            rp1 = P.row_ptr[rl + 1];
            if (box) {
#pragma unroll
                for (int q = 0; q < PD; ++q) {
                    blo[q] = P.box[(size_t)rl * 2 * PD + q];
                    bhi[q] = P.box[(size_t)rl * 2 * PD + PD + q];
                }
            }
        }
        if (box) {
            bool hit = r0 + lane < P.nreg;
#pragma unroll
            for (int q = 0; q < PD; ++q) hit = hit && (blo[q] <= whi[q]) && (bhi[q] >= wlo[q]);
            cand &= __ballot(hit);
        } else if (P.nreg - r0 < 64) {
            cand &= (1ull << (P.nreg - r0)) - 1ull;
        }
        while (cand != 0ull) {
            const int r = __builtin_amdgcn_readfirstlane(r0 + __ffsll((long long)cand) - 1);   // list order: the first containing region wins
            cand &= cand - 1ull;
            bool inside = search && found < 0;
            if (__ballot(inside) == 0ull) break;
            if (kCorners) {
                // The wave's points are neighbours of a grid and a region is convex.  Lane <-> row (64 rows at a time), each
                // row evaluated at the 2^p corners of the points' hull: a row violated at every corner is violated by every
                // point (the region holds none of them), a row satisfied at every corner is satisfied by every point and
                // needs no look at the points.  Both only decide with a margin far above the rounding of the row expression
                // (1e-12 relative to its terms), so skipping the satisfied rows leaves the per-point result as it is.  What
                // remains are the few rows that cross the hull: those are tested point by point (lane <-> point again) with
                // the row taken from the lane that holds it -- no further memory round trip.
                const int i0 = lane_i32(rp0, r - r0), i1 = lane_i32(rp1, r - r0);
                bool any_out = false;
                for (int c0 = i0; c0 < i1; c0 += 64) {
                    const int i = c0 + lane;
                    const bool on = i < i1;
                    const double *ar = P.A + (size_t)(on ? i : i0) * PD;
                    const double bb = P.b[on ? i : i0];
                    double a[PD], mag = fabs(bb);
#pragma unroll
                    for (int q = 0; q < PD; ++q) {
                        a[q] = ar[q];
                        mag = fma(fabs(a[q]), fmax(fabs(wlo[q]), fabs(whi[q])), mag);
                    }
                    const double marg = 1e-12 * mag;
                    bool in_all = true, out_all = true;
#pragma unroll
                    for (int c = 0; c < (1 << PD); ++c) {
                        double sc = -bb;
#pragma unroll
                        for (int q = 0; q < PD; ++q) sc = fma(a[q], ((c >> q) & 1) ? whi[q] : wlo[q], sc);
                        in_all = in_all && (sc <= P.tol - marg);
                        out_all = out_all && (sc >= P.tol + marg);
                    }
                    if (__ballot(on && out_all) != 0ull) {
                        any_out = true;
                        break;
                    }
                    unsigned long long need = __ballot(on && !in_all);
                    while (need != 0ull) {
                        const int l = __ffsll((long long)need) - 1;
                        need &= need - 1ull;
                        double sv = -lane_f64(bb, l);
#pragma unroll
                        for (int q = 0; q < PD; ++q) sv = fma(lane_f64(a[q], l), x[q], sv);
                        inside = inside && (sv <= P.tol);
                    }
                    if (__ballot(inside) == 0ull) break;
                }
                if (!any_out && inside && xnum) found = r;
                continue;
            }
            if (box) {
                const cdouble_ptr lo = box + (size_t)r * 2 * PD, hi = lo + PD;
#pragma unroll
                for (int q = 0; q < PD; ++q) inside = inside && (x[q] >= lo[q]) && (x[q] <= hi[q]);
                if (__ballot(inside) == 0ull) continue;
            }
            // rows in batches of kPlRows: all loads of a batch are issued before the first FMA (a row-at-a-time loop
            // pays one memory round trip per row); the clamped repeats of the last row test the same condition again
            const int i1 = lane_i32(rp1, r - r0);
            for (int i = lane_i32(rp0, r - r0); i < i1; i += kPlRows) {
                double a[kPlRows][PD], bb[kPlRows];
#pragma unroll
                for (int u = 0; u < kPlRows; ++u) {
                    const int ii = min(i + u, i1 - 1);
                    bb[u] = bt[ii];
#pragma unroll
                    for (int q = 0; q < PD; ++q) a[u][q] = At[ii * PD + q];
                }
#pragma unroll
                for (int u = 0; u < kPlRows; ++u) {
                    double s = -bb[u];
#pragma unroll
                    for (int q = 0; q < PD; ++q) s = fma(a[u][q], x[q], s);
                    inside = inside && (s <= P.tol);
                }
                if (__ballot(inside) == 0ull) break;
            }
            if (inside) found = r;
        }
    }
    return found;
}

// u = ux x + u0, V = .5 x'Vxx x + Vx'x + V0 of the region found (parametric_programs.py:364, :398), NaN without one.
template <int PD>
__device__ __forceinline__ void pl_law(const PlParams &P, const double (&x)[PD], int found, bool valid, long long pt)
{
    const double nanv = __builtin_nan("");
    // every point of the wave in ONE region (the common case on a grid): the region's law comes through scalar loads
    const unsigned long long vb = __ballot(valid);
    if (vb == 0ull) return;
    const int f0 = __builtin_amdgcn_readlane(found, __ffsll((long long)vb) - 1);
    if (f0 >= 0 && __ballot(valid && found != f0) == 0ull) {
        if (valid) {
            const cdouble_ptr Vxx = as_const(P.Vxx) + (size_t)f0 * PD * PD, Vx = as_const(P.Vx) + (size_t)f0 * PD;
            double qd = 0.0, lin = 0.0;
#pragma unroll
            for (int i = 0; i < PD; ++i) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < PD; ++q) s = fma(x[q], Vxx[q * PD + i], s);
                qd = fma(s, x[i], qd);
                lin = fma(Vx[i], x[i], lin);
            }
            P.idx[pt] = f0;
            P.V[pt] = .5 * qd + lin + as_const(P.V0)[f0];
            const cdouble_ptr u0 = as_const(P.u0) + (size_t)f0 * P.nu, ux = as_const(P.ux) + (size_t)f0 * P.nu * PD;
            for (int i = 0; i < P.nu; ++i) {
                double s = u0[i];
#pragma unroll
                for (int q = 0; q < PD; ++q) s = fma(ux[i * PD + q], x[q], s);
                P.u[pt * P.nu + i] = s;
            }
        }
    } else if (valid) {
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

// Pass 1 of a look-up with decided cells: every point whose cell decides it gets its region and law here; the others
// are marked idx = -2 for the scan kernel.  Nothing here depends on the region list: one coalesced load of the point, one
// 4-byte gather from the cell table (L2-resident: grid_n^p x 4 bytes), the law, three coalesced stores.  A wavefront
// takes several batches of 64 consecutive points per round so that enough bytes are in flight (8 TB/s x ~2 us of loaded
// latency = 16 MB over the chip, i.e. ~2 KB per resident wavefront at 8 waves per SIMD).
template <int PD>
constexpr int pl_batches() { return PD <= 2 ? 4 : (PD <= 4 ? 2 : 1); }

template <int PD>
__global__ void __launch_bounds__(256) point_cells_kernel(PlParams P)
{
    constexpr int U = pl_batches<PD>();
    const int lane = threadIdx.x & 63;
    const long long chunk = 64LL * U;
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6) * chunk;
    const long long base0 = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * chunk;
    const cdouble_ptr geom = as_const(P.grid_geom);
    double g0[PD], gi[PD], g1[PD], gl[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) { g0[q] = geom[q]; gi[q] = geom[PD + q]; g1[q] = geom[2 * PD + q]; gl[q] = geom[3 * PD + q]; }
    const double gmax = (double)(P.grid_n - 1);
    for (long long base = base0; base < P.B; base += stride) {
        double x[U][PD];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pc = min(base + u * 64 + lane, P.B - 1);     // (clamped: no divergent guards around the loads)
#pragma unroll
            for (int q = 0; q < PD; ++q) x[u][q] = P.X[pc * PD + q];
        }
        int sure[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // a point with a NaN coordinate lies in no region (contains: max(A x - b) <= tol is False, polyhedron.py:596);
            // outside the box the host gives (the hull of the region boxes, when all are bounded) in no region either (both
            // comparisons fail on NaN)
            bool in = true;
            int cell = 0;
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                in = in && (x[u][q] >= gl[q]) && (x[u][q] <= g1[q]);
                const double c = fmin(fmax(floor((x[u][q] - g0[q]) * gi[q]), 0.0), gmax);
                cell = cell * P.grid_n + (int)c;
            }
            sure[u] = P.grid_sure[cell];
            if (!in) sure[u] = -1;
        }
        if (P.grid_lists) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (sure[u] <= -3) {
                    // decision list of the cell: first candidate region, in list order, whose crossing rows all hold (rows
                    // evaluated as pl_scan evaluates them); the last entry may have no rows (it holds the whole cell)
                    const int32_t *lp = P.grid_lists + (-3 - sure[u]);
                    const int nc = lp[0];
                    int pos = 1, f = -1;
                    for (int c = 0; c < nc && f < 0; ++c) {
                        const int r = lp[pos], nr = lp[pos + 1];
                        bool in = true;
                        for (int j = 0; j < nr; ++j) {
                            const int row = lp[pos + 2 + j];
                            double s = -P.b[row];
#pragma unroll
                            for (int q = 0; q < PD; ++q) s = fma(P.A[(size_t)row * PD + q], x[u][q], s);
                            in = in && (s <= P.tol);
                        }
                        pos += 2 + nr;
                        if (in) f = r;
                    }
                    sure[u] = f;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) sure[u] = max(sure[u], -2);   // (no lists given: such a cell is left to the scan)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pt = base + u * 64 + lane;
            const bool valid = pt < P.B;
            if (valid && sure[u] == -2) P.idx[pt] = -2;
            pl_law<PD>(P, x[u], sure[u], valid && sure[u] > -2, pt);
        }
    }
}

// The scan kernel.  MARKED: point_cells_kernel ran before and only the points it marked (idx == -2) are looked up; a
// wavefront whose 64 points are all settled moves on after one 4-byte load per point.
template <int PD, bool MARKED>
__global__ void point_location_kernel(PlParams P)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // every lane of a wave runs the same number of grid-stride rounds (wave-uniform loop bound)
    const long long base0 = (long long)blockIdx.x * blockDim.x + (tid & ~63);
    // The scan is bound by the chain of dependent memory round trips per batch of 64 points (points -> cells -> rows
    // -> law), so everything that does not depend on the points is fetched ahead: the next batch's points (its marks
    // when MARKED), and (lane <-> region) the boxes and row ranges of the first 64 regions.
    PlAhead<PD> ahead;
    ahead.rp0 = ahead.rp1 = 0;
#pragma unroll
    for (int q = 0; q < PD; ++q) { ahead.lo[q] = 0.0; ahead.hi[q] = 0.0; }
    if (P.nreg > 0) {   // (an empty table has no row ranges and no boxes to fetch: every point comes out uncovered)
        const int rl = min(lane, P.nreg - 1);
        ahead.rp0 = P.row_ptr[rl];
        ahead.rp1 = P.row_ptr[rl + 1];
        if (P.box) {
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                ahead.lo[q] = P.box[(size_t)rl * 2 * PD + q];
                ahead.hi[q] = P.box[(size_t)rl * 2 * PD + PD + q];
            }
        }
    }
    double xn[PD];
    int markn = -1;
    if (MARKED) {
        if (base0 + lane < P.B) markn = P.idx[base0 + lane];
    } else {
#pragma unroll
        for (int q = 0; q < PD; ++q) xn[q] = (base0 + lane < P.B) ? P.X[(base0 + lane) * PD + q] : 0.0;
    }
    for (long long base = base0; base < P.B; base += stride) {
        const long long pt = base + lane;
        const bool valid = pt < P.B;
        const long long nx = pt + stride;
        double x[PD];
        bool search = valid;
        if (MARKED) {
            search = valid && markn == -2;
            markn = -1;
            if (nx < P.B) markn = P.idx[nx];
            if (__ballot(search) == 0ull) continue;
            const long long pc = valid ? pt : P.B - 1;
#pragma unroll
            for (int q = 0; q < PD; ++q) x[q] = P.X[pc * PD + q];
        } else {
#pragma unroll
            for (int q = 0; q < PD; ++q) x[q] = xn[q];
#pragma unroll
            for (int q = 0; q < PD; ++q) xn[q] = (nx < P.B) ? P.X[nx * PD + q] : 0.0;
        }
        // a point with a NaN coordinate lies in no region (contains: max(A x - b) <= tol is False, polyhedron.py:596); the
        // hull of the scan is built from the other points (v_min / v_max drop NaN operands), so it must not inherit theirs
        bool xnum = true;
#pragma unroll
        for (int q = 0; q < PD; ++q) xnum = xnum && (x[q] == x[q]);
        const int found = pl_scan<PD>(P, ahead, x, search, xnum, lane);
        pl_law<PD>(P, x, found, search, pt);
    }
}

}  // namespace pympc
