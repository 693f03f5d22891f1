// standalone check + timing of factor32_dpp against lookahead_factor32 (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "plba_factor32_dev.h"
using namespace plba;

template <int MODE>
__global__ __launch_bounds__(256) void k_test(DevBuf d, const double* A, double* outKeep, unsigned long long* cyc, int reps) {
    __shared__ __attribute__((aligned(16))) double sC[32 * LS];
    __shared__ __attribute__((aligned(16))) double sK[32 * LS];
    __shared__ __attribute__((aligned(16))) Look32 S;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long tot = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int idx = threadIdx.x; idx < 1024; idx += 256) sC[(idx >> 5) * LS + (idx & 31)] = A[idx];
        if (MODE == 0) look32_reset(S, threadIdx.x); else factor32_reset(S, threadIdx.x);
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == 0) lookahead_factor32<false, true>(d, 0, sC, S, wv, lane, sK);
        else factor32_dpp<true>(d, 0, sC, *reinterpret_cast<Factor32Lds*>(&S), wv, lane, sK);
        __syncthreads();
        tot += __builtin_readcyclecounter() - t0;
    }
    for (int idx = threadIdx.x; idx < 1024; idx += 256) outKeep[idx] = sK[(idx >> 5) * LS + (idx & 31)];
    if (threadIdx.x == 0) { cyc[0] = tot / reps;
#ifdef PLBA_F32STAMPS
        if (MODE == 1) for (int k = 0; k < 8; ++k) cyc[1 + k] = g_f32stamp[k] - g_f32stamp[0];
#endif
    }
}

int main() {
    const int n = 32;
    std::vector<double> A(n * n), M(n * n);
    srand(7);
    for (auto& v : M) v = (rand() / (double)RAND_MAX - 0.5);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += M[i * n + k] * M[j * n + k]; A[i * n + j] = s * std::pow(10.0, (i + j) / 16.0) + (i == j ? 3.0 * std::pow(10.0, i / 8.0) : 0.0); }
    double *dA, *dL, *dI, *dK; Ctrl* dc; unsigned long long* dcy;
    hipMalloc(&dA, 8192); hipMalloc(&dL, 8192); hipMalloc(&dI, 8192); hipMalloc(&dK, 8192); hipMalloc(&dc, sizeof(Ctrl)); hipMalloc(&dcy, 128);
    hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice);
    DevBuf d; memset(&d, 0, sizeof d);
    d.Lfac = dL; d.Linv32 = dI; d.ld = 32; d.ctrl = dc;
    for (int mode = 0; mode < 2; ++mode) {
        Ctrl c; memset(&c, 0, sizeof c); c.solver_ok = 1;
        hipMemcpy(dc, &c, sizeof c, hipMemcpyHostToDevice);
        hipMemset(dL, 0, 8192); hipMemset(dI, 0, 8192);
        if (mode == 0) hipLaunchKernelGGL(k_test<0>, dim3(1), dim3(256), 0, 0, d, dA, dK, dcy, 20);
        else hipLaunchKernelGGL(k_test<1>, dim3(1), dim3(256), 0, 0, d, dA, dK, dcy, 20);
        hipError_t e = hipDeviceSynchronize();
        std::vector<double> L(n * n), I(n * n), K(n * n); unsigned long long cy = 0;
        hipMemcpy(L.data(), dL, 8192, hipMemcpyDeviceToHost); hipMemcpy(I.data(), dI, 8192, hipMemcpyDeviceToHost); hipMemcpy(K.data(), dK, 8192, hipMemcpyDeviceToHost);
        hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, sizeof c, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0, e3 = 0, up = 0, sc = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
            double s = 0, t = 0;
            for (int k = 0; k < n; ++k) { s += L[i * n + k] * L[j * n + k]; t += I[i * n + k] * L[k * n + j]; }
            e1 = std::fmax(e1, std::fabs(s - A[i * n + j]) / std::sqrt(A[i * n + i] * A[j * n + j]));
            e2 = std::fmax(e2, std::fabs(t - (i == j ? 1.0 : 0.0)));
            e3 = std::fmax(e3, std::fabs(I[i * n + j] - K[i * n + j]));
            if (j > i) up = std::fmax(up, std::fmax(std::fabs(L[i * n + j]), std::fabs(I[i * n + j])));
            sc = std::fmax(sc, std::fabs(A[i * n + j]));
        }
        { unsigned long long st[16]; hipMemcpy(st, dcy, 128, hipMemcpyDeviceToHost); if (mode == 1) { printf("stamps:"); for (int k = 1; k < 9; ++k) printf(" %llu", st[k]); printf("\n"); } }
        printf("mode %d (%s): err %s, solver_ok %d, |L L^T - A| (scaled) %.2e, |Linv L - I| %.2e, |Linv - kept| %.2e, upper part %.2e, cycles per tile %llu\n", mode, mode ? "factor32_dpp" : "lookahead_factor32",
               hipGetErrorString(e), (int)c.solver_ok, e1, e2, e3, up, cy);
    }
    return 0;
}
