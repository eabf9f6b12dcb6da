// C ABI of the gfx950 batched QP path (declarations and reference citations: include/pympc_hip.h).
// Plain HIP runtime only: callers hand in device pointers and a stream; nothing here allocates or synchronises.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>

#include "../../include/pympc_hip.h"
#include "condense.hpp"
#include "lp_block.hpp"
#include "lp_tile.hpp"
#include "nnls_raw.hpp"
#include "nnls_block.hpp"
#include "qp_block.hpp"
#include "point_location.hpp"
#include "qp_prepare.hpp"
#include "qp_wave.hpp"
#include "qp_screen.hpp"
#include "qp_gram.hpp"
#include "ws_layout.hpp"

using namespace pympc;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// compute units of the current device (0 if the query fails: callers fall back to one chunk per workgroup)
int cu_count()
{
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev != cached_dev) {
        int v = 0;
        cached = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? v : 0;
        cached_dev = dev;
    }
    return cached;
}

// workgroups of a persistent launch: one per compute unit, minus the units pympc_hip_tune("reserved_cus") keeps free
// (for the copy kernels of a caller that overlaps transfers with the solve)
int persistent_cus();

constexpr size_t kLdsBytes = 160 * 1024;   // MI355X: 160 KiB LDS per CU, a single workgroup may own all of it

// Launch-policy knobs (pympc_hip_tune): thresholds that pick between kernels of the same method, read at launch time.
// They never change what is computed, only which kernel computes it (and, for nnls_itmax_factor, where the iteration
// limit of scipy.optimize.nnls sits), so tests can drive every kernel at every size.
struct Tune {
    std::atomic<long long> lp_tile_min_rows{kLpTileMinRows};     // lp_tile_kernel from this many rows ...
    std::atomic<long long> lp_tile_min_batch{kLpTileMinBatch};   // ... and this many LPs against one shared matrix
    std::atomic<long long> gram_min_batch{8192};                 // mpqp_gram_kernel from this many states
    std::atomic<long long> learn_min_batch{131072};              // certificate-learning pre-pass from this many states
    std::atomic<long long> nnls_itmax_factor{3};                 // NNLS iteration limit = factor * columns (scipy: 3)
    std::atomic<long long> reserved_cus{0};                      // compute units the persistent launches leave free
    std::atomic<long long> prepare_wave{1};                      // 1: qp_prepare_wave_kernel for batches of small problems, 0: the tile kernel
    std::atomic<long long> dynamic_chunks{1};                    // 1: chunks of states from a device counter (Gram / solve kernels), 0: static shares
};
Tune g_tune;

int persistent_cus()
{
    int ncu = cu_count();
    if (ncu <= 0) ncu = 256;
    const long long keep = g_tune.reserved_cus.load();
    if (keep > 0 && keep < ncu) ncu -= (int)keep;
    return ncu;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PYMPC_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

// Kernels that take more than 64 KiB of dynamic LDS need the attribute once per (kernel, device); the largest size
// granted so far is remembered so that steady-state launches make no runtime call for it.
constexpr int kMaxDevices = 64;
struct LdsGrant { const void *fn; int dev; size_t bytes; };
template <typename K>
int allow_lds(K kernel, size_t bytes, size_t static_bytes = 0)
{
    if (bytes + static_bytes <= 64 * 1024) return 0;   // (bytes: dynamic LDS; static_bytes: what the kernel declares itself)
    static std::mutex mu;
    static LdsGrant grants[128];
    static int ngrants = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    const void *fn = reinterpret_cast<const void *>(kernel);
    LdsGrant *slot = nullptr;
    if (dev >= 0) {
        std::lock_guard<std::mutex> lock(mu);
        for (int i = 0; i < ngrants; ++i)
            if (grants[i].fn == fn && grants[i].dev == dev) { slot = &grants[i]; break; }
        if (slot && slot->bytes >= bytes) return 0;
    }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(PYMPC_ERR_LAUNCH, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    if (dev >= 0) {
        std::lock_guard<std::mutex> lock(mu);
        if (!slot && ngrants < 128) { slot = &grants[ngrants++]; slot->fn = fn; slot->dev = dev; slot->bytes = 0; }
        if (slot && slot->bytes < bytes) slot->bytes = bytes;
    }
    return 0;
}

template <int MT, bool SHARED>
int launch_solve_mt(const SolveParams &P, int nwaves, size_t lds_bytes, long long nwg, hipStream_t st)
{
    auto kernel = qp_solve_kernel<MT, SHARED>;
    if (int rc = allow_lds(kernel, lds_bytes)) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)nwg), dim3(64 * nwaves), lds_bytes, st, P);
    return check_launch("qp_solve_kernel");
}

template <bool SHARED>
int launch_solve(int mt, const SolveParams &P, int nwaves, size_t lds_bytes, long long nwg, hipStream_t st)
{
    switch (mt) {
    case 2: return launch_solve_mt<2, SHARED>(P, nwaves, lds_bytes, nwg, st);
    case 4: return launch_solve_mt<4, SHARED>(P, nwaves, lds_bytes, nwg, st);
    case 6: return launch_solve_mt<6, SHARED>(P, nwaves, lds_bytes, nwg, st);
    case 8: return launch_solve_mt<8, SHARED>(P, nwaves, lds_bytes, nwg, st);
    case 12: return launch_solve_mt<12, SHARED>(P, nwaves, lds_bytes, nwg, st);
    case 16: return launch_solve_mt<16, SHARED>(P, nwaves, lds_bytes, nwg, st);
    case 24: return launch_solve_mt<24, SHARED>(P, nwaves, lds_bytes, nwg, st);
    case 32: return launch_solve_mt<32, SHARED>(P, nwaves, lds_bytes, nwg, st);
    }
    return fail(PYMPC_ERR_UNSUPPORTED, "no kernel for %d columns per lane", mt);
}

// Launch geometry of the per-problem prepare workgroup: one thread per constraint row of a tile (up to 256 rows,
// fewer when two row tiles of that size would not fit beside the factor matrices in ~96 KiB of LDS).
struct PrepGeom { int threads, tile_rows; size_t lds_bytes; int wave_nn; };   // wave_nn > 0: qp_prepare_wave_kernel<wave_nn>
constexpr long long kPrepSmallBatch = 1024;
PrepGeom prep_geom(int n, int m, int p, long long B = 1)
{
    PrepGeom G;
    int tr = ((m + 63) / 64) * 64;
    if (tr > 256) tr = 256;
    // many problems with small Hessians: one wavefront per problem (its barriers cost nothing, more workgroups per CU)
    if (B >= kPrepSmallBatch && n * n < 1024) tr = 64;
    while (tr > 64 && prep_lds_doubles(n, p, tr) * 8 > 96 * 1024) tr -= 64;
    G.tile_rows = tr;
    G.threads = (n * n >= 1024 && tr < 256) ? 256 : tr;
    G.lds_bytes = prep_lds_doubles(n, p, tr) * 8;
    G.wave_nn = 0;
    // many problems of up to 16 variables: one wavefront per problem with its constraint rows in registers
    // (qp_prepare_wave_kernel: no row tiles in LDS, so the wavefronts per CU are not limited by it)
    if (B >= kPrepSmallBatch && n <= 16 && g_tune.prepare_wave.load() != 0) {
        G.wave_nn = (n + 3) / 4 * 4;
        G.threads = 64;
        G.tile_rows = 64;
        G.lds_bytes = prep_wave_lds_doubles(n, p) * 8;
    }
    return G;
}

int launch_prepare(const PrepGeom &G, const PrepParams &Q, long long B, hipStream_t st)
{
    switch (G.wave_nn) {
#define PYMPC_PREP_CASE(NN)                                                                                   \
    case NN:                                                                                                  \
        if (int rc = allow_lds(qp_prepare_wave_kernel<NN>, G.lds_bytes)) return rc;                           \
        hipLaunchKernelGGL(qp_prepare_wave_kernel<NN>, dim3((unsigned)B), dim3(64), G.lds_bytes, st, Q);      \
        return check_launch("qp_prepare_wave_kernel");
        PYMPC_PREP_CASE(4) PYMPC_PREP_CASE(8) PYMPC_PREP_CASE(12) PYMPC_PREP_CASE(16)
#undef PYMPC_PREP_CASE
    }
    if (int rc = allow_lds(qp_prepare_kernel, G.lds_bytes)) return rc;
    hipLaunchKernelGGL(qp_prepare_kernel, dim3((unsigned)B), dim3(G.threads), G.lds_bytes, st, Q);
    return check_launch("qp_prepare_kernel");
}

int check_shape(int n, int m, int p)
{
    if (n < 1 || m < 1 || p < 0) return fail(PYMPC_ERR_ARG, "bad shape n=%d m=%d p=%d", n, m, p);
    if (n > kNMax) return fail(PYMPC_ERR_UNSUPPORTED, "n=%d exceeds the wave kernel's limit of %d variables", n, kNMax);
    if (m > kMMax) return fail(PYMPC_ERR_UNSUPPORTED, "m=%d exceeds the wave kernel's limit of %d rows", m, kMMax);
    if (p > kPMax) return fail(PYMPC_ERR_UNSUPPORTED, "p=%d exceeds the wave kernel's limit of %d parameters", p, kPMax);
    return 0;
}

// the prepare workgroup keeps its factor matrices and Hux in LDS: refuse shapes that do not fit instead of failing at launch
int check_prep(const PrepGeom &G, int n, int m, int p)
{
    if (G.lds_bytes > kLdsBytes)
        return fail(PYMPC_ERR_UNSUPPORTED, "factorisation of a %dx%d QP with p=%d parameters needs %zu B of LDS (160 KiB available)",
                    n, m, p, G.lds_bytes);
    return 0;
}

// Learning pre-pass of a large shared-matrix batch with a certificate table: kLearnStates states, taken at a constant
// stride over the batch, are solved first by the wave kernel alone (no screening pass: its three serial sweeps cost
// more latency than the few states are worth), at most three states per wavefront, so that the table holds the
// certificates of the batch's infeasible regions before the main launch starts.  Without it every wave of the main
// launch meets its first infeasible state with an empty table and solves it in full (a serial chain of ~24 pivots)
// at the same time as all the others.  The pre-pass writes no outputs: its statuses go to the head of the caller's
// status array, which the main launch overwrites.  With a table that is already warm its infeasible states are
// rejected at once and it costs one short launch.
#ifndef PYMPC_LEARN_STATES
#define PYMPC_LEARN_STATES 8192
#endif
constexpr long long kLearnStates = PYMPC_LEARN_STATES;
constexpr size_t kSolveCtlBytes = kSchedCtlInts * sizeof(int);   // qp_solve_kernel's control block in LDS (ticket, flags, chunk list)
// states per chunk a workgroup requests from the launch's device counter (SolveParams::sched)
#ifndef PYMPC_DYN_CHUNK
#define PYMPC_DYN_CHUNK 256
#endif

// Common launch geometry of the solve kernel.
int solve_common(bool shared, int n, int m, int p, SolveParams &P, hipStream_t st, bool learn_pass = false)
{
    const int R = n + 1;
    P.n = n; P.m = m; P.p = p;
    P.RP = R | 1;
    P.L = ws_layout(n, m, p);
    P.wave_doubles = wave_lds_doubles(n, p, P.RP);
    const int mt = P.L.mp / 64;
    const size_t wave_bytes = (size_t)P.wave_doubles * 8;
    P.itmax_factor = (int)g_tune.nnls_itmax_factor.load();
    if (shared && !learn_pass && P.cuts && p >= 1 && p <= kCutPMax && P.x_stride == p && P.B >= g_tune.learn_min_batch.load()) {
        SolveParams A = P;
        const long long step = P.B / kLearnStates;
        A.B = kLearnStates;
        A.x_stride = (long long)p * step;
        A.learn = 1;
        A.argmin = nullptr; A.vmin = nullptr; A.mult = nullptr; A.active = nullptr; A.iters = nullptr;
        if (int rc = solve_common(true, n, m, p, A, st, true)) return rc;
    }
    // a matrix too large to be staged beside four problems stays in global memory (one workspace, ws_stride 0, read
    // through L2 by every wave) instead of being refused
    if (shared && shared_lds(n, P.L).total * 8 + kSolveCtlBytes + 4 * wave_bytes > kLdsBytes) shared = false;
    if (!shared || p < 1 || p > kCutPMax) P.cuts = nullptr;   // certificates belong to ONE mpQP with a parameter
    const size_t shared_bytes = shared ? shared_lds(n, P.L).total * 8 : 0;
    if (shared_bytes + kSolveCtlBytes + wave_bytes > kLdsBytes)
        return fail(PYMPC_ERR_UNSUPPORTED, "QP %dx%d (p=%d) does not fit the 160 KiB LDS (%zu + %zu B)", n, m, p,
                    shared_bytes, wave_bytes);
    int nwaves = (int)((kLdsBytes - shared_bytes - kSolveCtlBytes) / wave_bytes);
    // without a staged matrix keep workgroups small so several fit a CU and balance at wave granularity
    const int max_waves = mt <= 8 ? PYMPC_SOLVE_WAVES : 4;   // = solve_max_waves<MT, true>() (launch bounds of the kernel)
    const int cap = shared ? max_waves : 4;
    if (nwaves > cap) nwaves = cap;
    if (shared && nwaves > 4) nwaves &= ~3;   // a multiple of the 4 SIMDs
    if ((long long)nwaves > P.B) nwaves = (int)P.B;
    if (nwaves < 1) nwaves = 1;
    const size_t lds_bytes = shared_bytes + kSolveCtlBytes + wave_bytes * nwaves;
    long long chunk = nwaves, nwg;
    if (shared) {
        // one persistent workgroup per CU over interleaved slices (qp_solve_kernel); slices of 8 problems once the batch
        // fills the chip (measured at 1M states with the screening pass ahead: 4 and 8 equal, 16 is 2 % and 32 is 5 % slower),
        // smaller ones below that so that every wave still has work
        int ncu = persistent_cus();
        chunk = learn_pass ? 1 : 8;
        while (chunk > 1 && P.B < chunk * nwaves * ncu) chunk >>= 1;
        const long long nslice = (P.B + chunk - 1) / chunk;
        const long long want = (nslice + nwaves - 1) / nwaves;   // workgroups that still give every wave a slice
        nwg = want < ncu ? want : ncu;
        if (nwg < 1) nwg = 1;
        P.persist = 1;
        P.chunk = (int)chunk;
        P.sched = nullptr;
        // screening pass: states with a feasible unconstrained minimiser (and those a stored certificate rejects) finish here
        P.screened = 0;
        if (!learn_pass && p >= 1 && p <= 8 && P.x_stride >= p && P.B <= 0x7fffffffLL) {
            P.screened = 1;
            // persistent workgroups, two per CU, each walking over whole tiles of states (qp_screen.hpp); the grid is
            // evened out so that every workgroup gets the same number of tiles (+- 1)
            long long tile = (long long)kScreenThreads * (p <= 4 ? 2 : 1);   // = kScreenThreads * screen_states_per_lane<p>()
            if (P.B < (long long)kScreenWgPerCu * ncu * tile) {   // a small batch: smaller tiles (whole wavefronts), so that it still spreads over the chip
                const long long per = (P.B + (long long)kScreenWgPerCu * ncu - 1) / ((long long)kScreenWgPerCu * ncu);
                const long long t64 = (per + 63) / 64 * 64;
                if (t64 < tile) tile = t64 < 64 ? 64 : t64;
            }
            P.screen_tile = (int)tile;
            const long long ntile = (P.B + tile - 1) / tile;
            const long long rounds = (ntile + (long long)kScreenWgPerCu * ncu - 1) / ((long long)kScreenWgPerCu * ncu);
            const long long sg = (ntile + rounds - 1) / rounds;
#define PYMPC_SCREEN_CASE(PD) case PD: hipLaunchKernelGGL(mpqp_screen_kernel<PD>, dim3((unsigned)sg), dim3(kScreenThreads), 0, st, P); break;
            switch (p) {
                PYMPC_SCREEN_CASE(1) PYMPC_SCREEN_CASE(2) PYMPC_SCREEN_CASE(3) PYMPC_SCREEN_CASE(4)
                PYMPC_SCREEN_CASE(5) PYMPC_SCREEN_CASE(6) PYMPC_SCREEN_CASE(7) PYMPC_SCREEN_CASE(8)
            }
#undef PYMPC_SCREEN_CASE
            if (int rc = check_launch("mpqp_screen_kernel")) return rc;
            // with a certificate table (whose header holds the counters, cleared by the screening kernel) and a batch
            // that gives every workgroup several chunks, the two kernels below draw their states from a device counter
            const bool dynamic = P.cuts != nullptr && g_tune.dynamic_chunks.load() != 0 && P.B >= 8LL * PYMPC_DYN_CHUNK * ncu;
            // Gram-form pass (qp_gram.hpp): what the lane path left pending and ends with up to kGramK active constraints
            // is finished here, without the matrix in LDS; the rest stays pending for the solve kernel below
            // (batches too small to keep the chip busy pay for its launch and staging more than it saves the solve kernel)
            if (P.gram && !P.rhs && !P.raw && mt <= 8 && P.B >= g_tune.gram_min_batch.load()) {
                SolveParams G = P;
                int gslice = kGramSlice;   // states per ticket; smaller while the batch does not give every wave a slice
                while (gslice > 1 && P.B < (long long)gslice * (kGramThreads / 64) * ncu) gslice >>= 1;
                G.chunk = gslice;
                if (dynamic) {
                    G.sched = reinterpret_cast<int *>(P.cuts) + kCutTicketGram;
                    G.dyn_chunk = PYMPC_DYN_CHUNK;
                }
                const size_t gbytes = gram_lds_doubles(p, P.L) * sizeof(double);   // <= 10 * 512 * 8 + tables: below 64 KiB
                const long long gslices = (P.B + gslice - 1) / gslice;
                const long long gwant = (gslices + kGramThreads / 64 - 1) / (kGramThreads / 64);
                const long long gg = gwant < ncu ? gwant : ncu;
                switch (mt) {
                case 2: hipLaunchKernelGGL(mpqp_gram_kernel<2>, dim3((unsigned)gg), dim3(kGramThreads), gbytes, st, G); break;
                case 4: hipLaunchKernelGGL(mpqp_gram_kernel<4>, dim3((unsigned)gg), dim3(kGramThreads), gbytes, st, G); break;
                case 6: hipLaunchKernelGGL(mpqp_gram_kernel<6>, dim3((unsigned)gg), dim3(kGramThreads), gbytes, st, G); break;
                case 8: hipLaunchKernelGGL(mpqp_gram_kernel<8>, dim3((unsigned)gg), dim3(kGramThreads), gbytes, st, G); break;
                }
                if (int rc = check_launch("mpqp_gram_kernel")) return rc;
            }
            if (dynamic) {
                P.sched = reinterpret_cast<int *>(P.cuts) + kCutTicketSolve;
                P.dyn_chunk = PYMPC_DYN_CHUNK;
            }
        }
    } else {
        P.chunk = (int)chunk;
        nwg = (P.B + chunk - 1) / chunk;
    }
    if (nwg > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
    return shared ? launch_solve<true>(mt, P, nwaves, lds_bytes, nwg, st)
                  : launch_solve<false>(mt, P, nwaves, lds_bytes, nwg, st);
}

// ---- device scratch of the entry point that allocates nothing (pympc_mpqp_solve_batch_static): the table of learned
// infeasibility certificates, sized for the largest parameter count the certificates support.  Every GPU the library
// runs on holds its own copy until the process ends.  In host code the array's NAME is only a handle for
// hipGetSymbolAddress: the device address is looked up once per device and kept in g_scratch[dev].
constexpr size_t kScratchDoubles = kCutHeader + (size_t)(kCutPMax + 1) * kCutMax;   // 2 352 doubles = 18 816 B
static __device__ __align__(256) double scratch[kScratchDoubles];
double *g_scratch[kMaxDevices] = {};
std::mutex g_scratch_mu;

int scratch_address(double **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(PYMPC_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail(PYMPC_ERR_UNSUPPORTED, "device ordinal %d beyond %d", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch[dev]) {
        void *addr = nullptr;
        e = hipGetSymbolAddress(&addr, HIP_SYMBOL(scratch));   // with device `dev` current: that GPU's copy
        if (e != hipSuccess || !addr)
            return fail(PYMPC_ERR_LAUNCH, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
        g_scratch[dev] = static_cast<double *>(addr);
    }
    *out = g_scratch[dev];
    return 0;
}

// table header: int32 count, lock, base (entries [0, base) were stored at set-up: pympc_mpqp_cuts_mark), spare
__global__ void cuts_mark_kernel(double *cuts, int restore)
{
    int *hdr = reinterpret_cast<int *>(cuts);
    if (threadIdx.x == 0) {
        if (restore) { hdr[0] = min(hdr[0], hdr[2]); hdr[1] = 0; }
        else hdr[2] = min(hdr[0], kCutMax);
    }
}

}  // namespace

extern "C" {

int pympc_hip_abi_version(void) { return 8; }   // 8: pympc_point_location takes grid_sure / grid_lists / grid_scan (decided cells), grid_geom is [3 p]

const char *pympc_hip_last_error(void) { return g_err; }

int pympc_hip_tune(const char *key, long long value, long long *previous)
{
    if (!key) return fail(PYMPC_ERR_ARG, "pympc_hip_tune: null key");
    struct { const char *name; std::atomic<long long> *slot; long long lo, hi; } knobs[] = {
        {"lp_tile_min_rows", &g_tune.lp_tile_min_rows, 1, 1LL << 40},
        {"lp_tile_min_batch", &g_tune.lp_tile_min_batch, 1, 1LL << 40},
        {"gram_min_batch", &g_tune.gram_min_batch, 1, 1LL << 40},
        {"learn_min_batch", &g_tune.learn_min_batch, 1, 1LL << 40},
        {"nnls_itmax_factor", &g_tune.nnls_itmax_factor, 0, 1000},
        {"reserved_cus", &g_tune.reserved_cus, 0, 128},
        {"dynamic_chunks", &g_tune.dynamic_chunks, 0, 1},
        {"prepare_wave", &g_tune.prepare_wave, 0, 1},
    };
    for (auto &k : knobs)
        if (strcmp(key, k.name) == 0) {
            if (previous) *previous = k.slot->load();
            if (value < 0) return 0;   // query only
            if (value < k.lo || value > k.hi) return fail(PYMPC_ERR_ARG, "pympc_hip_tune(%s): %lld outside [%lld, %lld]", key, value, k.lo, k.hi);
            k.slot->store(value);
            return 0;
        }
    return fail(PYMPC_ERR_ARG, "pympc_hip_tune: unknown key '%s'", key);
}

void pympc_hip_limits(int *n_max, int *m_max)
{
    if (n_max) *n_max = kNMax;
    if (m_max) *m_max = kMMax;
}

size_t pympc_mpqp_workspace_doubles(int n, int m, int p)
{
    if (n < 1 || m < 1 || p < 0 || n > kNMax || m > kMMax || p > kPMax) return 0;
    if (prep_geom(n, m, p).lds_bytes > kLdsBytes) return 0;
    const WsLayout L = ws_layout(n, m, p);
    return L.total + ws_gram_doubles(L);
}

int pympc_mpqp_prepare(int n, int m, int p, const double *Huu, const double *Hux, const double *Hxx, const double *fu,
                       const double *fx, double g, const double *Au, const double *Ax, const double *b, double *ws,
                       int32_t *ws_status, void *stream)
{
    if (int rc = check_shape(n, m, p)) return rc;
    if (!Huu || !fu || !Au || !b || !ws || !ws_status || (p > 0 && (!Hux || !Hxx || !fx || !Ax)))
        return fail(PYMPC_ERR_ARG, "pympc_mpqp_prepare: null pointer");
    PrepParams P;
    memset(&P, 0, sizeof(P));
    P.n = n; P.m = m; P.p = p;
    P.L = ws_layout(n, m, p);
    P.H = Huu; P.Hux = Hux; P.Hxx = Hxx; P.fu = fu; P.fx = fx; P.gs = nullptr; P.g_host = g;
    P.A = Au; P.Ax = Ax; P.b = b; P.ws = ws; P.ws_status = ws_status;
    const PrepGeom G = prep_geom(n, m, p);
    if (int rc = check_prep(G, n, m, p)) return rc;
    P.tile_rows = G.tile_rows;
    if (int rc = allow_lds(qp_prepare_kernel, G.lds_bytes)) return rc;
    hipLaunchKernelGGL(qp_prepare_kernel, dim3(1), dim3(G.threads), G.lds_bytes, (hipStream_t)stream, P);
    if (int rc = check_launch("qp_prepare_kernel")) return rc;
    // Gram matrix of the constant rows behind the layout (ws_layout.hpp)
    const int mp = P.L.mp;
    hipLaunchKernelGGL(qp_gram_kernel, dim3(mp / 64, mp), dim3(64), 0, (hipStream_t)stream, n, mp, ws + P.L.Mt, ws + P.L.total);
    return check_launch("qp_gram_kernel");
}

size_t pympc_mpqp_cuts_doubles(int p)
{
    if (p < 1 || p > kCutPMax) return 0;
    return cut_table_doubles(p);
}

int pympc_mpqp_cuts_reset(int p, double *cuts, void *stream)
{
    if (p < 1 || p > kCutPMax) return fail(PYMPC_ERR_UNSUPPORTED, "infeasibility cuts need 1 <= p <= %d parameters, got %d", kCutPMax, p);
    if (!cuts) return fail(PYMPC_ERR_ARG, "pympc_mpqp_cuts_reset: null pointer");
    hipError_t e = hipMemsetAsync(cuts, 0, cut_table_doubles(p) * sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return fail(PYMPC_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    return 0;
}

int pympc_mpqp_cuts_mark(int p, double *cuts, void *stream)
{
    if (p < 1 || p > kCutPMax) return fail(PYMPC_ERR_UNSUPPORTED, "infeasibility cuts need 1 <= p <= %d parameters, got %d", kCutPMax, p);
    if (!cuts) return fail(PYMPC_ERR_ARG, "pympc_mpqp_cuts_mark: null pointer");
    hipLaunchKernelGGL(cuts_mark_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cuts, 0);
    return check_launch("cuts_mark_kernel");
}

int pympc_mpqp_cuts_restore(int p, double *cuts, void *stream)
{
    if (p < 1 || p > kCutPMax) return fail(PYMPC_ERR_UNSUPPORTED, "infeasibility cuts need 1 <= p <= %d parameters, got %d", kCutPMax, p);
    if (!cuts) return fail(PYMPC_ERR_ARG, "pympc_mpqp_cuts_restore: null pointer");
    hipLaunchKernelGGL(cuts_mark_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cuts, 1);
    return check_launch("cuts_mark_kernel");
}

int pympc_mpqp_cuts_learn(int n, int m, int p, const double *ws, double *cuts, const double *X, int64_t B, double tol,
                          int32_t *status, void *stream)
{
    if (int rc = check_shape(n, m, p)) return rc;
    if (p < 1 || p > kCutPMax) return fail(PYMPC_ERR_UNSUPPORTED, "infeasibility cuts need 1 <= p <= %d parameters, got %d", kCutPMax, p);
    if (B < 0) return fail(PYMPC_ERR_ARG, "pympc_mpqp_cuts_learn: negative batch size");
    if (B == 0) return 0;
    if (!ws || !cuts || !X || !status) return fail(PYMPC_ERR_ARG, "pympc_mpqp_cuts_learn: null pointer");
    if (B > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
    SolveParams P;
    memset(&P, 0, sizeof(P));
    P.ws = ws; P.X = X; P.x_stride = p; P.B = B; P.tol = tol; P.status = status;
    P.cuts = cuts;
    P.gram = ws + ws_layout(n, m, p).total;
    // the bare wave kernel (no screening, no outputs), one state per ticket: the launch shape of the learning pre-pass
    return solve_common(true, n, m, p, P, (hipStream_t)stream, true);
}

int pympc_mpqp_solve_batch_cuts(int n, int m, int p, const double *ws, double *cuts, const double *X, int64_t B, double tol,
                                int n_out, double *argmin, double *vmin, double *mult, uint32_t *active, int32_t *status,
                                int32_t *iters, void *stream)
{
    if (int rc = check_shape(n, m, p)) return rc;
    if (B < 0) return fail(PYMPC_ERR_ARG, "pympc_mpqp_solve_batch: negative batch size");
    if (n_out < 0 || n_out > n) return fail(PYMPC_ERR_ARG, "n_out=%d outside [0, %d]", n_out, n);
    if (B == 0) return 0;   // an empty batch has no pointers to check
    if (!ws || !status || (p > 0 && !X)) return fail(PYMPC_ERR_ARG, "pympc_mpqp_solve_batch: null pointer");
    SolveParams P;
    memset(&P, 0, sizeof(P));
    P.ws = ws; P.ws_stride = 0; P.ws_status = nullptr;
    P.X = X; P.x_stride = p; P.B = B; P.tol = tol; P.n_out = n_out;
    P.argmin = n_out > 0 ? argmin : nullptr;
    P.vmin = vmin; P.mult = mult; P.active = active; P.status = status; P.iters = iters;
    P.cuts = cuts;
    P.gram = ws + ws_layout(n, m, p).total;
    return solve_common(true, n, m, p, P, (hipStream_t)stream);
}

int pympc_mpqp_solve_batch(int n, int m, int p, const double *ws, const double *X, int64_t B, double tol, int n_out,
                           double *argmin, double *vmin, double *mult, uint32_t *active, int32_t *status,
                           int32_t *iters, void *stream)
{
    return pympc_mpqp_solve_batch_cuts(n, m, p, ws, nullptr, X, B, tol, n_out, argmin, vmin, mult, active, status, iters,
                                       stream);
}

int pympc_mpqp_solve_batch_static(int n, int m, int p, const double *ws, const double *X, int64_t B, double tol, int n_out,
                                  double *argmin, double *vmin, double *mult, uint32_t *active, int32_t *status,
                                  int32_t *iters, void *stream)
{
    if (p < 1 || p > kCutPMax)
        return fail(PYMPC_ERR_UNSUPPORTED,
                    "pympc_mpqp_solve_batch_static: the static certificate table holds 1 <= p <= %d parameters, got %d "
                    "(%zu bytes of device scratch per GPU)", kCutPMax, p, kScratchDoubles * sizeof(double));
    double *table = nullptr;
    if (int rc = scratch_address(&table)) return rc;
    // the table keeps the last call's certificates, which belong to that call's mpQP: start from an empty one
    hipError_t e = hipMemsetAsync(table, 0, cut_table_doubles(p) * sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return fail(PYMPC_ERR_LAUNCH, "hipMemsetAsync(static scratch): %s", hipGetErrorString(e));
    return pympc_mpqp_solve_batch_cuts(n, m, p, ws, table, X, B, tol, n_out, argmin, vmin, mult, active, status, iters, stream);
}

// ---- raw NNLS / PNNLS (nnls_raw.hpp)
size_t pympc_nnls_workspace_doubles(int rows, int cols, int64_t B)
{
    const int R = rows < 2 ? 2 : rows;
    if (rows < 1 || cols < 1 || rows > kBlkRowsMax || cols > kBlkColsMax || B < 0) return 0;
    // nnls_block_kernel (more than 64 rows, or more columns than the wave kernel takes): a working copy of each matrix,
    // its triangular factor, and the vectors over columns in case they do not fit the LDS
    if (R - 1 > kNMax || cols > kMMax) return ((size_t)rows * cols + (size_t)rows * rows + 3 * (size_t)cols) * (size_t)B + 32;
    return (ws_layout(R - 1, cols, 0).total + (size_t)R) * (size_t)B + 32;
}

namespace {
// 65 .. 256 rows: one workgroup per problem (nnls_block.hpp)
int launch_nnls_block(int rows, int cols, int64_t B, const double *A, int64_t a_stride, const double *b, int64_t b_stride, const int32_t *skip_rows,
                      double *work, double *x, double *rnorm, int32_t *status, int32_t *iters, hipStream_t st)
{
    NnlsBlockParams Q;
    memset(&Q, 0, sizeof(Q));
    Q.rows = rows; Q.cols = cols;
    Q.itmax = (int)std::min<long long>(g_tune.nnls_itmax_factor.load() * (long long)cols, 0x7fffffffLL);
    Q.A = A; Q.a_stride = a_stride; Q.b = b; Q.b_stride = b_stride; Q.skip_rows = skip_rows;
    Q.work = work; Q.factor = work + (size_t)rows * cols * (size_t)B; Q.colvec = Q.factor + (size_t)rows * rows * (size_t)B;
    Q.x = x; Q.rnorm = rnorm; Q.status = status; Q.iters = iters;
    const bool colsg = nnls_block_lds_bytes(rows, cols) > kLdsBytes;   // the vectors over columns do not fit: device memory
    const size_t lds = nnls_block_lds_bytes(rows, cols, colsg);
    if (lds > kLdsBytes) return fail(PYMPC_ERR_UNSUPPORTED, "NNLS %d x %d needs %zu B of LDS (160 KiB available)", rows, cols, lds);
    const int T = blk_threads(rows);
#define PYMPC_LAUNCH_NNLS_BLOCK(TT, CG)                                                                          \
    do {                                                                                                         \
        if (int rc = allow_lds(nnls_block_kernel<TT, CG>, lds)) return rc;                                       \
        hipLaunchKernelGGL((nnls_block_kernel<TT, CG>), dim3((unsigned)B), dim3(TT), lds, st, Q);                \
    } while (0)
    if (T == kBlkThreadsSmall) { if (colsg) PYMPC_LAUNCH_NNLS_BLOCK(kBlkThreadsSmall, true); else PYMPC_LAUNCH_NNLS_BLOCK(kBlkThreadsSmall, false); }
    else if (T == kBlkThreadsBig) { if (colsg) PYMPC_LAUNCH_NNLS_BLOCK(kBlkThreadsBig, true); else PYMPC_LAUNCH_NNLS_BLOCK(kBlkThreadsBig, false); }
    else { if (colsg) PYMPC_LAUNCH_NNLS_BLOCK(kBlkThreadsHuge, true); else PYMPC_LAUNCH_NNLS_BLOCK(kBlkThreadsHuge, false); }
#undef PYMPC_LAUNCH_NNLS_BLOCK
    return check_launch("nnls_block_kernel");
}
}  // namespace

int pympc_nnls_batch(int rows, int cols, int64_t B, const double *A, int64_t a_stride, const double *b, double *ws, double *x,
                     double *rnorm, int32_t *status, int32_t *iters, void *stream)
{
    if (rows < 1 || cols < 1 || B < 0) return fail(PYMPC_ERR_ARG, "pympc_nnls_batch: bad shape %d x %d", rows, cols);
    const int R = rows < 2 ? 2 : rows;
    if (rows > kBlkRowsMax) return fail(PYMPC_ERR_UNSUPPORTED, "rows=%d exceeds the limit of %d rows", rows, kBlkRowsMax);
    if (cols > kBlkColsMax) return fail(PYMPC_ERR_UNSUPPORTED, "cols=%d exceeds the limit of %d columns", cols, kBlkColsMax);
    if (a_stride != 0 && a_stride < (int64_t)rows * cols) return fail(PYMPC_ERR_ARG, "a_stride %lld < rows*cols", (long long)a_stride);
    if (B == 0) return 0;
    if (!A || !b || !ws || !x || !status) return fail(PYMPC_ERR_ARG, "pympc_nnls_batch: null pointer");
    if (B > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
    if (R - 1 > kNMax || cols > kMMax)
        return launch_nnls_block(rows, cols, B, A, a_stride, b, rows, nullptr, ws, x, rnorm, status, iters, (hipStream_t)stream);
    const WsLayout L = ws_layout(R - 1, cols, 0);
    NnlsPrepParams Q;
    memset(&Q, 0, sizeof(Q));
    Q.rows = rows; Q.cols = cols; Q.L = L; Q.A = A; Q.a_stride = a_stride; Q.b = b; Q.b_stride = rows;
    Q.ws = ws; Q.rhs = ws + L.total * (size_t)B;
    hipLaunchKernelGGL(nnls_prepare_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, Q);
    if (int rc = check_launch("nnls_prepare_kernel")) return rc;
    SolveParams P;
    memset(&P, 0, sizeof(P));
    P.ws = ws; P.ws_stride = (long long)L.total; P.B = B; P.tol = 0.0; P.n_out = 0;
    P.vmin = rnorm; P.mult = x; P.status = status; P.iters = iters;
    P.rhs = Q.rhs; P.rhs_stride = R; P.raw = 1;
    return solve_common(false, R - 1, cols, 0, P, (hipStream_t)stream);
}

namespace {
// doubles at the head of a PNNLS workspace that belong to its NNLS: the block kernel's when any of the three sizes sends
// the problem there (more than 64 free variables do so even when the NNLS alone would fit the wave kernel)
size_t pnnls_nnls_doubles(int rows, int cols, int nu, int64_t B)
{
    if (rows > 64 || cols > kMMax || nu > 64) {
        if (rows > kBlkRowsMax || cols > kBlkColsMax) return 0;
        return ((size_t)rows * cols + (size_t)rows * rows + 3 * (size_t)cols) * (size_t)B + 32;
    }
    return pympc_nnls_workspace_doubles(rows, cols, B);
}
}  // namespace

size_t pympc_pnnls_workspace_doubles(int rows, int cols, int nu, int64_t B)
{
    if (rows < 1 || rows > kBlkRowsMax || cols < 1 || cols > kBlkColsMax || nu < 0 || nu > kPnnlsNuMax || B < 0) return 0;
    const size_t nn = pnnls_nnls_doubles(rows, cols, nu, B);
    if (nn == 0) return 0;
    const size_t scr = nu > 64 ? (size_t)nu * (nu + 1) + 2 * (size_t)nu : 0;   // small systems of the recover kernel
    return nn + ((size_t)rows * cols + rows + (size_t)rows * nu + nu + 2 + scr) * (size_t)B + 32;
}

int pympc_pnnls_batch(int rows, int cols, int nu, int64_t B, const double *A, const double *Bm, const double *c, double *ws,
                      double *v, double *u, double *rnorm, int32_t *status, int32_t *iters, void *stream)
{
    if (rows < 1 || cols < 1 || nu < 0 || B < 0) return fail(PYMPC_ERR_ARG, "pympc_pnnls_batch: bad shape");
    if (rows > kBlkRowsMax || nu > kPnnlsNuMax || cols > kBlkColsMax)
        return fail(PYMPC_ERR_UNSUPPORTED, "PNNLS %d x (%d + %d) exceeds the limits of %d rows, %d free and %d sign-constrained variables",
                    rows, cols, nu, kBlkRowsMax, kPnnlsNuMax, kBlkColsMax);
    if (B == 0) return 0;
    if (!A || !c || !ws || !v || !status || (nu > 0 && (!Bm || !u))) return fail(PYMPC_ERR_ARG, "pympc_pnnls_batch: null pointer");
    if (nu == 0) return pympc_nnls_batch(rows, cols, B, A, (int64_t)rows * cols, c, ws, v, rnorm, status, iters, stream);
    if (B > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
    const size_t nn = pnnls_nnls_doubles(rows, cols, nu, B);
    PnnlsParams R;
    memset(&R, 0, sizeof(R));
    R.rows = rows; R.cols = cols; R.nu = nu; R.A = A; R.Bm = Bm; R.c = c;
    R.QA = ws + nn;
    R.Qc = R.QA + (size_t)rows * cols * B;
    R.Rf = R.Qc + (size_t)rows * B;
    R.perm = reinterpret_cast<int32_t *>(R.Rf + (size_t)rows * nu * B);
    R.rank = R.perm + (size_t)nu * B;
    R.scr = nu > 64 ? reinterpret_cast<double *>(R.rank + (((size_t)B + 1) & ~(size_t)1)) : nullptr;
    R.v = v; R.u = u;
    hipStream_t st = (hipStream_t)stream;
    if (rows > 64 || cols > kMMax || nu > 64) {   // one row per thread of a workgroup instead of one per lane (nnls_block.hpp)
        const bool bg = nu > 64 || pnnls_block_lds_bytes(rows, nu) > kLdsBytes;   // B in device memory instead of LDS
        const size_t ldsb = bg ? pnnls_block_lds_bytes_bg(rows, nu) : pnnls_block_lds_bytes(rows, nu);
        const int T = blk_threads(rows);
#define PYMPC_LAUNCH_PNNLS_REDUCE(TT, BGV)                                                                       \
    do {                                                                                                         \
        if (int rc = allow_lds(pnnls_reduce_block_kernel<TT, BGV>, ldsb)) return rc;                             \
        hipLaunchKernelGGL((pnnls_reduce_block_kernel<TT, BGV>), dim3((unsigned)B), dim3(TT), ldsb, st, R);      \
    } while (0)
        if (T == kBlkThreadsSmall) { if (bg) PYMPC_LAUNCH_PNNLS_REDUCE(kBlkThreadsSmall, true); else PYMPC_LAUNCH_PNNLS_REDUCE(kBlkThreadsSmall, false); }
        else if (T == kBlkThreadsBig) { if (bg) PYMPC_LAUNCH_PNNLS_REDUCE(kBlkThreadsBig, true); else PYMPC_LAUNCH_PNNLS_REDUCE(kBlkThreadsBig, false); }
        else { if (bg) PYMPC_LAUNCH_PNNLS_REDUCE(kBlkThreadsHuge, true); else PYMPC_LAUNCH_PNNLS_REDUCE(kBlkThreadsHuge, false); }
#undef PYMPC_LAUNCH_PNNLS_REDUCE
        if (int rc = check_launch("pnnls_reduce_block_kernel")) return rc;
        if (int rc = launch_nnls_block(rows, cols, B, R.QA, (int64_t)rows * cols, R.Qc, rows, R.rank, ws, v, rnorm, status, iters, st)) return rc;
        hipLaunchKernelGGL(pnnls_recover_kernel, dim3((unsigned)B), dim3(64), (64 + 64 * 65 + 64) * 8, st, R);
        return check_launch("pnnls_recover_kernel");
    }
    const size_t lds1 = pnnls_lds_doubles(rows, nu) * 8;
    hipLaunchKernelGGL(pnnls_reduce_kernel, dim3((unsigned)B), dim3(64), lds1, st, R);
    if (int rc = check_launch("pnnls_reduce_kernel")) return rc;
    // NNLS in v on the rows the reflectors left free of u (pnnls.py:36)
    const int Rr = rows < 2 ? 2 : rows;
    const WsLayout L = ws_layout(Rr - 1, cols, 0);
    NnlsPrepParams Q;
    memset(&Q, 0, sizeof(Q));
    Q.rows = rows; Q.cols = cols; Q.L = L; Q.A = R.QA; Q.a_stride = (long long)rows * cols; Q.b = R.Qc; Q.b_stride = rows;
    Q.skip_rows = R.rank; Q.ws = ws; Q.rhs = ws + L.total * (size_t)B;
    hipLaunchKernelGGL(nnls_prepare_kernel, dim3((unsigned)B), dim3(256), 0, st, Q);
    if (int rc = check_launch("nnls_prepare_kernel")) return rc;
    SolveParams P;
    memset(&P, 0, sizeof(P));
    P.ws = ws; P.ws_stride = (long long)L.total; P.B = B; P.tol = 0.0; P.n_out = 0;
    P.vmin = rnorm; P.mult = v; P.status = status; P.iters = iters;
    P.rhs = Q.rhs; P.rhs_stride = Rr; P.raw = 1;
    if (int rc = solve_common(false, Rr - 1, cols, 0, P, st)) return rc;
    // u = -pinv(B) (A v - c) (pnnls.py:37)
    hipLaunchKernelGGL(pnnls_recover_kernel, dim3((unsigned)B), dim3(64), (64 + 64 * 65 + 64) * 8, st, R);
    return check_launch("pnnls_recover_kernel");
}

size_t pympc_qp_workspace_doubles(int n, int m, int64_t B)
{
    if (n < 1 || m < 1 || n > kQpBlockNMax || m > kBlkColsMax || B < 0) return 0;
    if (n > kNMax || m > kMMax) {   // qp_block.hpp: factors and NNLS matrix per problem, the NNLS kernel's working copy, y, scalars
        const size_t per = qp_block_layout(n, m).total + (size_t)(n + 1) * m + (size_t)(n + 1) * (n + 1) + 3 * (size_t)m + (size_t)m + 4;
        return per * (size_t)B + (size_t)n + 2 + 32;
    }
    // per-problem prepared workspace + one int32 prepare status per problem (stored after the workspaces)
    return ws_layout(n, m, 0).total * (size_t)B + ((size_t)B + 1) / 2 + 32;
}

namespace {
// 64 .. 255 variables: one workgroup per problem (qp_block.hpp)
int qp_block_solve(int n, int m, int64_t B, const double *H, int64_t h_stride, const double *f, const double *A, int64_t a_stride,
                   const double *b, double tol, double *ws, double *argmin, double *vmin, double *mult, uint32_t *active,
                   int32_t *status, int32_t *iters, hipStream_t st)
{
    const QpBlockLayout L = qp_block_layout(n, m);
    QpBlockParams P;
    memset(&P, 0, sizeof(P));
    P.n = n; P.m = m; P.H = H; P.h_stride = h_stride; P.f = f; P.A = A; P.a_stride = a_stride; P.b = b; P.tol = tol;
    P.ws = ws;
    double *work = ws + L.total * (size_t)B;               // [B x (n + 1) x m]
    P.y = work + ((size_t)(n + 1) * m + (size_t)(n + 1) * (n + 1) + 3 * (size_t)m) * (size_t)B;   // [B x m] (behind the NNLS kernel's matrices, factors and column vectors)
    P.rn = P.y + (size_t)m * (size_t)B;                    // [B]
    P.bn = P.rn + (size_t)B;                               // [n + 1] (+ 1 pad)
    P.nnls_status = reinterpret_cast<int32_t *>(P.bn + (((size_t)n + 2) & ~(size_t)1));   // [B], then iters [2B], notpd [B]
    P.nnls_iters = P.nnls_status + (size_t)B;
    P.notpd = P.nnls_iters + 2 * (size_t)B;
    P.argmin = argmin; P.vmin = vmin; P.mult = mult; P.active = active; P.status = status; P.iters = iters;
    const int T = blk_threads(n + 1);
    if (T == kBlkThreadsSmall) hipLaunchKernelGGL(qp_block_prepare_kernel<kBlkThreadsSmall>, dim3((unsigned)B), dim3(kBlkThreadsSmall), 0, st, P);
    else if (T == kBlkThreadsBig) hipLaunchKernelGGL(qp_block_prepare_kernel<kBlkThreadsBig>, dim3((unsigned)B), dim3(kBlkThreadsBig), 0, st, P);
    else hipLaunchKernelGGL(qp_block_prepare_kernel<kBlkThreadsHuge>, dim3((unsigned)B), dim3(kBlkThreadsHuge), 0, st, P);
    if (int rc = check_launch("qp_block_prepare_kernel")) return rc;
    if (int rc = launch_nnls_block(n + 1, m, B, ws + L.An, (int64_t)L.total, P.bn, 0, nullptr, work, P.y, P.rn, P.nnls_status,
                                   P.nnls_iters, st))
        return rc;
    if (T == kBlkThreadsSmall) hipLaunchKernelGGL(qp_block_recover_kernel<kBlkThreadsSmall>, dim3((unsigned)B), dim3(kBlkThreadsSmall), 0, st, P);
    else if (T == kBlkThreadsBig) hipLaunchKernelGGL(qp_block_recover_kernel<kBlkThreadsBig>, dim3((unsigned)B), dim3(kBlkThreadsBig), 0, st, P);
    else hipLaunchKernelGGL(qp_block_recover_kernel<kBlkThreadsHuge>, dim3((unsigned)B), dim3(kBlkThreadsHuge), 0, st, P);
    return check_launch("qp_block_recover_kernel");
}
}  // namespace

int pympc_qp_solve_batch(int n, int m, int64_t B, const double *H, int64_t h_stride, const double *f, const double *A,
                         int64_t a_stride, const double *b, double tol, double *ws, double *argmin, double *vmin,
                         double *mult, uint32_t *active, int32_t *status, int32_t *iters, void *stream)
{
    if (n >= 1 && m >= 1 && (n > kNMax || m > kMMax) && n <= kQpBlockNMax && m <= kBlkColsMax) {
        if (B < 0) return fail(PYMPC_ERR_ARG, "pympc_qp_solve_batch: negative batch size");
        if (B == 0) return 0;
        if (!H || !f || !A || !b || !ws || !status) return fail(PYMPC_ERR_ARG, "pympc_qp_solve_batch: null pointer");
        if (B > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
        return qp_block_solve(n, m, B, H, h_stride, f, A, a_stride, b, tol, ws, argmin, vmin, mult, active, status, iters,
                              (hipStream_t)stream);
    }
    if (int rc = check_shape(n, m, 0)) return rc;
    if (B < 0) return fail(PYMPC_ERR_ARG, "pympc_qp_solve_batch: negative batch size");
    if (B == 0) return 0;
    if (!H || !f || !A || !b || !ws || !status) return fail(PYMPC_ERR_ARG, "pympc_qp_solve_batch: null pointer");
    if (B > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
    const WsLayout L = ws_layout(n, m, 0);
    int32_t *ws_status = reinterpret_cast<int32_t *>(ws + L.total * (size_t)B);
    PrepParams Q;
    memset(&Q, 0, sizeof(Q));
    Q.n = n; Q.m = m; Q.p = 0; Q.L = L;
    Q.H = H; Q.h_stride = h_stride;
    Q.fu = f; Q.fu_stride = n;
    Q.A = A; Q.a_stride = a_stride;
    Q.b = b; Q.b_stride = m;
    Q.g_host = 0.0;
    Q.ws = ws; Q.ws_stride = (long long)L.total; Q.ws_status = ws_status;
    const PrepGeom G = prep_geom(n, m, Q.p, B);
    if (int rc = check_prep(G, n, m, Q.p)) return rc;
    Q.tile_rows = G.tile_rows;
    if (int rc = launch_prepare(G, Q, B, (hipStream_t)stream)) return rc;
    SolveParams P;
    memset(&P, 0, sizeof(P));
    P.ws = ws; P.ws_stride = (long long)L.total; P.ws_status = ws_status;
    P.X = nullptr; P.B = B; P.tol = tol; P.n_out = n;
    P.argmin = argmin; P.vmin = vmin; P.mult = mult; P.active = active; P.status = status; P.iters = iters;
    return solve_common(false, n, m, 0, P, (hipStream_t)stream);
}

size_t pympc_mpqp_multi_workspace_doubles(int n, int m, int p, int64_t B)
{
    if (n < 1 || m < 1 || p < 0 || n > kNMax || m > kMMax || p > kPMax || B < 0) return 0;
    if (prep_geom(n, m, p, B).lds_bytes > kLdsBytes) return 0;
    return ws_layout(n, m, p).total * (size_t)B + ((size_t)B + 1) / 2 + 32;
}

int pympc_mpqp_multi_solve(int n, int m, int p, int64_t B, const double *Huu, const double *Hux, const double *Hxx,
                           const double *fu, const double *fx, const double *g, const double *Au, const double *Ax,
                           const double *b, const double *X, int64_t x_stride, double tol, double *ws, double *argmin,
                           double *vmin, double *mult, uint32_t *active, int32_t *status, int32_t *iters, void *stream)
{
    if (int rc = check_shape(n, m, p)) return rc;
    if (B < 0) return fail(PYMPC_ERR_ARG, "pympc_mpqp_multi_solve: negative batch size");
    if (x_stride != 0 && x_stride < p) return fail(PYMPC_ERR_ARG, "x_stride %lld < p", (long long)x_stride);
    if (B == 0) return 0;
    if (!Huu || !fu || !Au || !b || !g || !ws || !status || (p > 0 && (!Hux || !Hxx || !fx || !Ax || !X)))
        return fail(PYMPC_ERR_ARG, "pympc_mpqp_multi_solve: null pointer");
    if (B > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
    const WsLayout L = ws_layout(n, m, p);
    int32_t *ws_status = reinterpret_cast<int32_t *>(ws + L.total * (size_t)B);
    PrepParams Q;
    memset(&Q, 0, sizeof(Q));
    Q.n = n; Q.m = m; Q.p = p; Q.L = L;
    Q.H = Huu; Q.h_stride = (long long)n * n;
    Q.Hux = Hux; Q.hux_stride = (long long)n * p;
    Q.Hxx = Hxx; Q.hxx_stride = (long long)p * p;
    Q.fu = fu; Q.fu_stride = n;
    Q.fx = fx; Q.fx_stride = p;
    Q.gs = g; Q.g_stride = 1;
    Q.A = Au; Q.a_stride = (long long)m * n;
    Q.Ax = Ax; Q.ax_stride = (long long)m * p;
    Q.b = b; Q.b_stride = m;
    Q.ws = ws; Q.ws_stride = (long long)L.total; Q.ws_status = ws_status;
    Q.skip_T = argmin == nullptr;   // (the workspaces of this call are only read by the solve launch below)
    const PrepGeom G = prep_geom(n, m, Q.p, B);
    if (int rc = check_prep(G, n, m, Q.p)) return rc;
    Q.tile_rows = G.tile_rows;
    if (int rc = launch_prepare(G, Q, B, (hipStream_t)stream)) return rc;
    SolveParams P;
    memset(&P, 0, sizeof(P));
    P.ws = ws; P.ws_stride = (long long)L.total; P.ws_status = ws_status;
    P.X = p > 0 ? X : nullptr; P.x_stride = x_stride; P.B = B; P.tol = tol; P.n_out = n;
    P.argmin = argmin; P.vmin = vmin; P.mult = mult; P.active = active; P.status = status; P.iters = iters;
    return solve_common(false, n, m, p, P, (hipStream_t)stream);
}

void pympc_condense_limits(int *nx_max, int *w_max)
{
    if (nx_max) *nx_max = kCondNxMax;
    if (w_max) *w_max = 16 * kCondCtMax;
}

int pympc_condense_batch(int nx, int nu, int N, int nm, int rmax, int rN, const int32_t *modes, int64_t B,
                         const double *As, const double *Bs, const double *cs, const double *Q, const double *R,
                         const double *P, const double *Fd, const double *Gd, const double *hd, const double *FN,
                         const double *hN, double *Huu, double *Hux, double *Hxx, double *fu, double *fx, double *g,
                         double *Au, double *Ax, double *b, double *Sbar, int32_t *status, void *stream)
{
    if (nx < 1 || nu < 1 || N < 1 || nm < 1 || rmax < 0 || rN < 0 || B < 0)
        return fail(PYMPC_ERR_ARG, "pympc_condense_batch: bad shape");
    const int W = nx + N * nu + 1;
    if (nx > kCondNxMax) return fail(PYMPC_ERR_UNSUPPORTED, "nx=%d exceeds the condensing kernel's limit of %d states", nx, kCondNxMax);
    if (W > 16 * kCondCtMax)
        return fail(PYMPC_ERR_UNSUPPORTED, "nx + N nu + 1 = %d exceeds the condensing kernel's limit of %d columns", W, 16 * kCondCtMax);
    if (!modes || !As || !Bs || !cs || !Q || !R || !P || (rmax > 0 && (!Fd || !Gd || !hd)) || (rN > 0 && (!FN || !hN)) ||
        !Huu || !Hux || !Hxx || !fu || !fx || !g || !status || (N * rmax + rN > 0 && (!Au || !Ax || !b)))
        return fail(PYMPC_ERR_ARG, "pympc_condense_batch: null pointer");
    if (B == 0) return 0;
    if (B > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
    CondParams C;
    memset(&C, 0, sizeof(C));
    C.nx = nx; C.nu = nu; C.N = N; C.nm = nm; C.rmax = rmax; C.rN = rN;
    C.n = N * nu; C.m = N * rmax + rN; C.W = W; C.CT = (W + 15) / 16;
    C.pitch = (16 * C.CT) | 16;   // odd multiple of 16 doubles: consecutive rows start 32 banks apart
    C.modes = modes; C.As = As; C.Bs = Bs; C.cs = cs; C.Q = Q; C.R = R; C.P = P;
    C.Fd = Fd; C.Gd = Gd; C.hd = hd; C.FN = FN; C.hN = hN; C.B = B;
    C.Huu = Huu; C.Hux = Hux; C.Hxx = Hxx; C.fu = fu; C.fx = fx; C.g = g; C.Au = Au; C.Ax = Ax; C.b = b;
    C.Sbar = Sbar; C.status = status;
    const size_t lds = (size_t)(N + 1) * nx * C.pitch * 8;
    if (lds > kLdsBytes) return fail(PYMPC_ERR_UNSUPPORTED, "prediction matrices (%zu B) do not fit the 160 KiB LDS", lds);
    hipStream_t st = (hipStream_t)stream;
#define PYMPC_COND_CASE(KS, CTV)                                                                  \
    case 8 * KS + CTV: {                                                                          \
        if (int rc = allow_lds(condense_kernel<KS, CTV>, lds)) return rc;                         \
        hipLaunchKernelGGL((condense_kernel<KS, CTV>), dim3((unsigned)B), dim3(64), lds, st, C);  \
        break;                                                                                    \
    }
#define PYMPC_COND_KS(KS) PYMPC_COND_CASE(KS, 1) PYMPC_COND_CASE(KS, 2) PYMPC_COND_CASE(KS, 3) PYMPC_COND_CASE(KS, 4) PYMPC_COND_CASE(KS, 5)
    switch (8 * ((nx + 3) / 4) + C.CT) {
        PYMPC_COND_KS(1) PYMPC_COND_KS(2) PYMPC_COND_KS(3) PYMPC_COND_KS(4)
    default: return fail(PYMPC_ERR_UNSUPPORTED, "no condensing kernel for nx=%d, %d column tiles", nx, C.CT);
    }
#undef PYMPC_COND_KS
#undef PYMPC_COND_CASE
    return check_launch("condense_kernel");
}

size_t pympc_lp_workspace_doubles(int n, int m, int64_t B, int a_shared)
{
    if (n < 1 || m < 1 || n > kLpNMax || m > kLpMMax || B < 0) return 0;
    return (size_t)(n + 1) * m * (a_shared ? 1 : (size_t)B) + 32;
}

void pympc_lp_limits(int *n_max, int *m_max)
{
    if (n_max) *n_max = kLpNMax;
    if (m_max) *m_max = kLpMMax;
}

int pympc_lp_solve_batch(int n, int m, int64_t B, const double *f, const double *A, int64_t a_stride, const double *b,
                         int64_t b_stride, const int32_t *relax_idx, double relax_val, double tol, double *ws,
                         double *argmin, double *vmin, double *mult, uint32_t *active, int32_t *status, int32_t *iters,
                         void *stream)
{
    if (n < 1 || m < 1 || B < 0) return fail(PYMPC_ERR_ARG, "pympc_lp_solve_batch: bad shape n=%d m=%d", n, m);
    if (n > kLpNMax) return fail(PYMPC_ERR_UNSUPPORTED, "n=%d exceeds the LP kernel's limit of %d variables", n, kLpNMax);
    if (m > kLpMMax) return fail(PYMPC_ERR_UNSUPPORTED, "m=%d exceeds the LP kernel's limit of %d rows", m, kLpMMax);
    if (a_stride != 0 && a_stride < (int64_t)m * n) return fail(PYMPC_ERR_ARG, "a_stride %lld < m*n", (long long)a_stride);
    if (b_stride != 0 && b_stride < m) return fail(PYMPC_ERR_ARG, "b_stride %lld < m", (long long)b_stride);
    if (B == 0) return 0;
    if (!f || !A || !b || !ws || !status) return fail(PYMPC_ERR_ARG, "pympc_lp_solve_batch: null pointer");
    if (B > 0x7fffffffLL) return fail(PYMPC_ERR_UNSUPPORTED, "batch too large");
    hipStream_t st = (hipStream_t)stream;
    const long long nA = a_stride == 0 ? 1 : B;
    if (nA > 65535) return fail(PYMPC_ERR_UNSUPPORTED, "more than 65535 distinct constraint matrices in one call");
    LpPrepParams Q;
    Q.n = n; Q.m = m; Q.A = A; Q.a_stride = a_stride; Q.ws = ws; Q.ws_stride = (long long)(n + 1) * m;
    const int pt = 256;
    int pgx = (m + pt - 1) / pt;
    if (pgx > 1024) pgx = 1024;
    hipLaunchKernelGGL(lp_prepare_kernel, dim3(pgx, (unsigned)nA), dim3(pt), 0, st, Q);
    if (int rc = check_launch("lp_prepare_kernel")) return rc;

    LpParams P;
    memset(&P, 0, sizeof(P));
    P.n = n; P.m = m; P.RP = n | 1;
    P.At = ws; P.at_stride = a_stride == 0 ? 0 : (long long)(n + 1) * m;
    P.f = f; P.f_stride = n;
    P.b = b; P.b_stride = b_stride;
    P.relax_idx = relax_idx; P.relax_val = relax_val;
    P.B = B; P.tol = tol;
    P.maxit = 10 * (m + n) + 100;
    P.argmin = argmin; P.vmin = vmin; P.mult = mult; P.active = active; P.status = status; P.iters = iters;
    // many LPs against ONE large matrix and right-hand side (redundancy removal): one LP per wavefront, every loaded
    // row shared by the eight LPs of a workgroup (lp_tile.hpp)
    if (a_stride == 0 && b_stride == 0 && n <= 8 && m >= g_tune.lp_tile_min_rows.load() && B >= g_tune.lp_tile_min_batch.load()) {
        const size_t lds_t = lp_tile_lds_doubles(n, m, P.RP) * 8;
        if (lds_t <= kLdsBytes) {
            int ncu = cu_count();
            if (ncu <= 0) ncu = 256;
            long long nwg = (B + kLpTileWaves - 1) / kLpTileWaves;
            if (nwg > ncu) nwg = ncu;
#define PYMPC_LPT_CASE(NT)                                                                                   \
    case NT: {                                                                                               \
        if (int rc = allow_lds(lp_tile_kernel<NT>, lds_t)) return rc;                                        \
        hipLaunchKernelGGL(lp_tile_kernel<NT>, dim3((unsigned)nwg), dim3(64 * kLpTileWaves), lds_t, st, P);  \
        break;                                                                                               \
    }
            switch (n) {
                PYMPC_LPT_CASE(1) PYMPC_LPT_CASE(2) PYMPC_LPT_CASE(3) PYMPC_LPT_CASE(4)
                PYMPC_LPT_CASE(5) PYMPC_LPT_CASE(6) PYMPC_LPT_CASE(7) PYMPC_LPT_CASE(8)
            }
#undef PYMPC_LPT_CASE
            return check_launch("lp_tile_kernel");
        }
    }
    // rows per thread ~8-20: one wavefront for small LPs, up to 8 for the 10k-facet polytopes
    int threads = ((m / 16 + 63) / 64) * 64;
    if (threads < 64) threads = 64;
    if (threads > 512) threads = 512;
    const size_t lds_bytes = lp_lds_doubles(n, m, P.RP, threads / 64) * 8;
    if (lds_bytes > kLdsBytes) return fail(PYMPC_ERR_UNSUPPORTED, "LP %dx%d does not fit the 160 KiB LDS", n, m);
#define PYMPC_LP_CASE(NT)                                                                            \
    case NT: {                                                                                       \
        if (int rc = allow_lds(lp_solve_kernel<NT>, lds_bytes)) return rc;                           \
        hipLaunchKernelGGL(lp_solve_kernel<NT>, dim3((unsigned)B), dim3(threads), lds_bytes, st, P); \
        break;                                                                                       \
    }
    switch (n <= 8 ? n : 0) {
        PYMPC_LP_CASE(0) PYMPC_LP_CASE(1) PYMPC_LP_CASE(2) PYMPC_LP_CASE(3) PYMPC_LP_CASE(4)
        PYMPC_LP_CASE(5) PYMPC_LP_CASE(6) PYMPC_LP_CASE(7) PYMPC_LP_CASE(8)
    }
#undef PYMPC_LP_CASE
    return check_launch("lp_solve_kernel");
}

int pympc_point_location(int nreg, int total_rows, int p, int nu, const double *A, const double *b,
                         const int32_t *row_ptr, const double *ux, const double *u0, const double *Vxx,
                         const double *Vx, const double *V0, const double *box, const double *grid_geom, int grid_n,
                         const uint64_t *grid_masks, const int32_t *grid_sure, const int32_t *grid_lists, int grid_scan,
                         const double *X, int64_t B, double tol, int32_t *idx, double *u, double *V, void *stream)
{
    if (nreg < 0 || total_rows < 0 || p < 1 || nu < 0 || B < 0) return fail(PYMPC_ERR_ARG, "pympc_point_location: bad shape");
    if (p > 8) return fail(PYMPC_ERR_UNSUPPORTED, "point location supports up to 8 parameters, got %d", p);
    if (B == 0) return 0;
    if (!X || !idx || !V || (nu > 0 && !u) || !row_ptr ||
        (nreg > 0 && (!A || !b || !Vxx || !Vx || !V0 || (nu > 0 && (!ux || !u0)))))
        return fail(PYMPC_ERR_ARG, "pympc_point_location: null pointer");
    PlParams P;
    memset(&P, 0, sizeof(P));
    P.nreg = nreg; P.p = p; P.nu = nu; P.total_rows = total_rows;
    P.A = A; P.b = b; P.row_ptr = row_ptr; P.ux = ux; P.u0 = u0; P.Vxx = Vxx; P.Vx = Vx; P.V0 = V0;
    P.box = box; P.X = X; P.B = B; P.tol = tol; P.idx = idx; P.u = u; P.V = V;
    if (grid_geom && grid_masks && grid_n > 0) {
        double cells = 1.0;
        for (int q = 0; q < p; ++q) cells *= grid_n;
        if (cells > 1e8) return fail(PYMPC_ERR_ARG, "pympc_point_location: grid of %d^%d cells", grid_n, p);
        P.grid_geom = grid_geom; P.grid_n = grid_n;
        P.grid_masks = reinterpret_cast<const unsigned long long *>(grid_masks);
        P.grid_sure = grid_sure;
        P.grid_lists = grid_sure ? grid_lists : nullptr;
    }
    const size_t lds = 0;
    const int threads = 256;
    hipStream_t st = (hipStream_t)stream;
    const bool marked = P.grid_sure != nullptr;
    // beyond a few waves of workgroups every workgroup runs the same number of grid-stride rounds
    auto balanced = [](long long nwg) {
        const long long cap = 256LL * 8 * 4;
        if (nwg > cap) { const long long rounds = (nwg + cap - 1) / cap; nwg = (nwg + rounds - 1) / rounds; }
        return (unsigned)nwg;
    };
    const unsigned nwg = balanced((B + threads - 1) / threads);
#define PYMPC_PL_CASE(PD)                                                                         \
    case PD: {                                                                                    \
        if (marked) {                                                                             \
            const long long per = (long long)threads * pl_batches<PD>();                          \
            hipLaunchKernelGGL(point_cells_kernel<PD>, dim3(balanced((B + per - 1) / per)), dim3(threads), 0, st, P); \
            if (grid_scan || !P.grid_lists)                                                       \
                hipLaunchKernelGGL((point_location_kernel<PD, true>), dim3(nwg), dim3(threads), lds, st, P); \
        } else {                                                                                  \
            hipLaunchKernelGGL((point_location_kernel<PD, false>), dim3(nwg), dim3(threads), lds, st, P); \
        }                                                                                         \
        break;                                                                                    \
    }
    switch (p) {
        PYMPC_PL_CASE(1) PYMPC_PL_CASE(2) PYMPC_PL_CASE(3) PYMPC_PL_CASE(4)
        PYMPC_PL_CASE(5) PYMPC_PL_CASE(6) PYMPC_PL_CASE(7) PYMPC_PL_CASE(8)
    }
#undef PYMPC_PL_CASE
    return check_launch("point_location_kernel");
}

}  // extern "C"
