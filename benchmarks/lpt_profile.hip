// cycle breakdown of lp_tile_kernel<6> on the C3 polytope (m rows, 6 variables, one LP per facet)
#define PYMPC_LPT_PROFILE 1
#include "../pympc_amd/csrc/lp_tile.hpp"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace pympc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 10000, n = 6;
    std::vector<double> A((size_t)m * n), b(m), f((size_t)m * n);
    srand(1);
    for (int j = 0; j < m; ++j) {
        double s = 0;
        for (int i = 0; i < n; ++i) {
            double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
            A[(size_t)j * n + i] = sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2);
            s += A[(size_t)j * n + i] * A[(size_t)j * n + i];
        }
        for (int i = 0; i < n; ++i) { A[(size_t)j * n + i] /= sqrt(s); f[(size_t)j * n + i] = -A[(size_t)j * n + i]; }
        b[j] = 1.0 + rand() / (double)RAND_MAX;
    }
    std::vector<int> rx(m);
    for (int j = 0; j < m; ++j) rx[j] = j;
    double *dA, *db, *df, *ws, *vmin; int *drx, *st, *it;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&db, m * 8)); CK(hipMalloc(&df, f.size() * 8));
    CK(hipMalloc(&ws, ((size_t)(n + 1) * m + 32) * 8)); CK(hipMalloc(&vmin, m * 8));
    CK(hipMalloc(&drx, m * 4)); CK(hipMalloc(&st, m * 4)); CK(hipMalloc(&it, m * 12));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), m * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(df, f.data(), f.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(drx, rx.data(), m * 4, hipMemcpyHostToDevice));
    LpPrepParams Q; Q.n = n; Q.m = m; Q.A = dA; Q.a_stride = 0; Q.ws = ws; Q.ws_stride = (long long)(n + 1) * m;
    hipLaunchKernelGGL(lp_prepare_kernel, dim3((m + 255) / 256, 1), dim3(256), 0, 0, Q);
    LpParams P; memset(&P, 0, sizeof(P));
    P.n = n; P.m = m; P.RP = n | 1; P.At = ws; P.f = df; P.f_stride = n; P.b = db; P.relax_idx = drx; P.relax_val = 1.0;
    P.B = m; P.tol = 1e-7; P.maxit = 10 * (m + n) + 100; P.vmin = vmin; P.status = st; P.iters = it;
    const size_t lds = lp_tile_lds_doubles(n, m, P.RP) * 8;
    int nwg = (m + 7) / 8; if (nwg > 256) nwg = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long z[8] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lpt_prof), z, sizeof(z)));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(lp_tile_kernel<6>, dim3(nwg), dim3(512), lds, 0, P);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_lpt_prof), sizeof(z)));
        const double nwv = nwg * 8.0;
        printf("m=%d  %.3f ms  rounds/wave %.1f (busy %.1f)  cycles/round: setup %.0f wait1 %.0f sweep %.0f argmax %.0f wait2 %.0f pivot %.0f\n",
               m, ms, z[0] / nwv, z[7] / nwv, (double)z[1] / z[0], (double)z[2] / z[0], (double)z[3] / z[0],
               (double)z[4] / z[0], (double)z[5] / z[0], (double)z[6] / z[0]);
    }
    std::vector<int> hst(m), hit(3 * m);
    CK(hipMemcpy(hst.data(), st, m * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hit.data(), it, m * 12, hipMemcpyDeviceToHost));
    long long bad = 0, piv = 0, out = 0;
    for (int j = 0; j < m; ++j) { bad += hst[j] != 0; piv += hit[3 * j + 2]; out += hit[3 * j]; }
    printf("status != 0: %lld, mean phase-1 columns %.2f, mean pivots %.2f\n", bad, out / (double)m, piv / (double)m);
    return 0;
}
