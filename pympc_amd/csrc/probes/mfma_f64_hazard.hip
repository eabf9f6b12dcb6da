// Reproducer for the wait-state pad behind csrc/condense.hpp::mfma_f64_settle.
//
// One wavefront computes C = A B (16 x 16, inner dimension 4 KS) with a chain of v_mfma_f64_16x16x4_f64 and reads the
// accumulator with plain VALU / store instructions IMMEDIATELY after the last MFMA, once with the explicit wait
// states of mfma_f64_settle between chain and reader (pad = 1) and once with what the compiler emits by itself (pad = 0).
// The result is compared on the host with the same product in fp64 (exact for the small integers used).
//   extern "C" int pympc_probe_mfma_f64(int pad, int ks, int repeats, long long *wrong_entries, int *wrong_register_mask)
// returns 0 when it ran; wrong_entries = number of entries (over all repeats) that differ from the exact product,
// wrong_register_mask = bit r set if accumulator register r (rows 4 r .. 4 r + 3) was ever wrong.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

using d4 = __attribute__((ext_vector_type(4))) double;

template <bool PAD>
__global__ void __launch_bounds__(64) probe_kernel(const double *A, const double *B, double *C, int ks, int repeats)
{
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    for (int rep = 0; rep < repeats; ++rep) {
        const double *Ar = A + (size_t)rep * 16 * 4 * ks, *Br = B + (size_t)rep * 16 * 4 * ks;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        // stale values the accumulator registers would show if they were read too early
        asm volatile("" : "+v"(acc));
        for (int s = 0; s < ks; ++s) {
            const double a = Ar[li * 4 * ks + 4 * s + lk];          // A[row li][k]
            const double b = Br[(4 * s + lk) * 16 + li];            // B[k][col li]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(acc));
        // first non-MFMA readers: four stores (row 4 lk + r ... layout of the 16x16 f64 accumulator: register r of lane
        // (li, lk) holds C[4 r + lk][li])
        double *Cr = C + (size_t)rep * 256;
        Cr[(0 + lk) * 16 + li] = acc.x;
        Cr[(4 + lk) * 16 + li] = acc.y;
        Cr[(8 + lk) * 16 + li] = acc.z;
        Cr[(12 + lk) * 16 + li] = acc.w;
    }
}

extern "C" int pympc_probe_mfma_f64(int pad, int ks, int repeats, long long *wrong_entries, int *wrong_register_mask)
{
    if (ks < 1 || ks > 8 || repeats < 1) return -1;
    const size_t na = (size_t)repeats * 16 * 4 * ks;
    std::vector<double> A(na), B(na), C((size_t)repeats * 256, -777.0), R((size_t)repeats * 256, 0.0);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)((int)(st % 17) - 8); };
    for (auto &v : A) v = rnd();
    for (auto &v : B) v = rnd();
    for (int rep = 0; rep < repeats; ++rep)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0.0;
                for (int k = 0; k < 4 * ks; ++k) s += A[(size_t)rep * 64 * ks + i * 4 * ks + k] * B[(size_t)rep * 64 * ks + k * 16 + j];
                R[(size_t)rep * 256 + i * 16 + j] = s;
            }
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    if (hipMalloc(&dA, na * 8) != hipSuccess || hipMalloc(&dB, na * 8) != hipSuccess || hipMalloc(&dC, C.size() * 8) != hipSuccess) return -2;
    hipMemcpy(dA, A.data(), na * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), na * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    if (pad) hipLaunchKernelGGL(probe_kernel<true>, dim3(1), dim3(64), 0, 0, dA, dB, dC, ks, repeats);
    else hipLaunchKernelGGL(probe_kernel<false>, dim3(1), dim3(64), 0, 0, dA, dB, dC, ks, repeats);
    const hipError_t e = hipDeviceSynchronize();
    hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dC);
    if (e != hipSuccess) return -3;
    long long wrong = 0;
    int mask = 0;
    for (size_t q = 0; q < C.size(); ++q)
        if (C[q] != R[q]) { ++wrong; mask |= 1 << (int)(((q % 256) / 16) / 4); }
    if (wrong_entries) *wrong_entries = wrong;
    if (wrong_register_mask) *wrong_register_mask = mask;
    return 0;
}
