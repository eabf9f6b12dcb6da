// fp64 FMA issue rate: W waves per SIMD, C independent chains per lane
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ void __launch_bounds__(256) k(double *out, int iters, double s)
{
    double a[C];
    for (int c = 0; c < C; ++c) a[c] = threadIdx.x * 1e-3 + c;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = fma(a[c], s, 1e-9);
    long long t1 = __builtin_readcyclecounter();
    double r = 0;
    for (int c = 0; c < C; ++c) r += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + (double)(t1 - t0) * 1e-30;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[1 << 20] = t1 - t0;
}
template <int C> void run(double *d, int wg_per_cu)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<C><<<256 * wg_per_cu, 256>>>(d, 100, 0.999);
    hipEventRecord(e0); k<C><<<256 * wg_per_cu, 256>>>(d, iters, 0.999); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, (long long *)d + (1 << 20), 8, hipMemcpyDeviceToHost);
    double flops = 2.0 * C * iters * 256.0 * 256 * wg_per_cu;
    printf("chains %d, waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.2f cycles per wave-FMA (wave view)\n", C, wg_per_cu, ms, flops / ms * 1e-9,
           (double)cyc / ((double)C * iters));
}
int main()
{
    double *d; hipMalloc(&d, ((1 << 20) + 16) * 8 + 256 * 256 * 8 * 8);
    run<1>(d, 1); run<2>(d, 1); run<4>(d, 1); run<8>(d, 1); run<8>(d, 2); run<8>(d, 4); run<4>(d, 4); run<1>(d, 8);
    return 0;
}
