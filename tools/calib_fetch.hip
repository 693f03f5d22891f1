// calib_fetch.hip — micro-kernels with KNOWN byte counts in the access widths of the fused landmark kernels (k_lm_schur / k_lm_trial),
// to calibrate rocprofv3's FETCH_SIZE on gfx950 (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
// streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  VERDICT r04 item 6.
// Every kernel reads `n` elements once (the buffer is far larger than the 256 MiB Infinity Cache and was written by a previous launch),
// folds them into one value per thread so that the loads cannot be dropped, and writes 8 bytes per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <typename T> __device__ __forceinline__ double as_d(T v) { return (double)v; }
__device__ __forceinline__ double as_d(double2 v) { return v.x + v.y; }

// contiguous stream, sizeof(T) bytes per lane and load: 1 (slot masks lm_ws8), 4 (index tables), 8 (weights, measurements as doubles), 16 (double2)
template <typename T>
__global__ void k_calib_stream(const T* __restrict__ src, size_t n, double* __restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += as_d(src[i]);
    __shared__ double s[256];
    s[threadIdx.x] = acc; __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < (int)blockDim.x; ++k) t += s[k]; out[blockIdx.x] = t; }
}
// the landmark estimates as the fused passes read them: 8-byte loads of the first 24 bytes of 48-byte slots, slots in a random order
// (landmark groups are sorted by keyframe, not by slot) — `idx` holds the slot of every element
__global__ void k_calib_slot24(const double* __restrict__ src, const int32_t* __restrict__ idx, size_t nslots, double* __restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (size_t)gridDim.x * blockDim.x) {
        const double* p = src + (size_t)idx[i] * 6;
        acc += p[0] + p[1] + p[2];
    }
    __shared__ double s[256];
    s[threadIdx.x] = acc; __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < (int)blockDim.x; ++k) t += s[k]; out[blockIdx.x] = t; }
}
__global__ void k_calib_fill(double* dst, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = 1.0; }

extern "C" {
// runs every pattern `reps` times on `bytes` bytes of source data; returns 0 on success
int calib_run(size_t bytes, int reps) {
    double* src = nullptr; double* out = nullptr; int32_t* idx = nullptr;
    const size_t nslots = bytes / 48;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 4096 * 8) != hipSuccess || hipMalloc(&idx, nslots * 4) != hipSuccess) return 1;
    {   // a fixed pseudo-random permutation of the slots (multiplicative hash, odd multiplier modulo a power of two would repeat: use LCG order over nslots)
        int32_t* h = (int32_t*)malloc(nslots * 4);
        size_t x = 12345;
        for (size_t i = 0; i < nslots; ++i) { x = (x * 6364136223846793005ull + 1442695040888963407ull); h[i] = (int32_t)((x >> 17) % nslots); }
        if (hipMemcpy(idx, h, nslots * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
        free(h);
    }
    const dim3 grid(2048), block(256);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_calib_fill, grid, block, 0, 0, src, bytes / 8);
        hipLaunchKernelGGL(k_calib_stream<double2>, grid, block, 0, 0, (const double2*)src, bytes / 16, out);
        hipLaunchKernelGGL(k_calib_fill, grid, block, 0, 0, src, bytes / 8);
        hipLaunchKernelGGL(k_calib_stream<double>, grid, block, 0, 0, (const double*)src, bytes / 8, out);
        hipLaunchKernelGGL(k_calib_fill, grid, block, 0, 0, src, bytes / 8);
        hipLaunchKernelGGL(k_calib_stream<int32_t>, grid, block, 0, 0, (const int32_t*)src, bytes / 4, out);
        hipLaunchKernelGGL(k_calib_fill, grid, block, 0, 0, src, bytes / 8);
        hipLaunchKernelGGL(k_calib_stream<uint8_t>, grid, block, 0, 0, (const uint8_t*)src, bytes, out);
        hipLaunchKernelGGL(k_calib_fill, grid, block, 0, 0, src, bytes / 8);
        hipLaunchKernelGGL(k_calib_slot24, grid, block, 0, 0, (const double*)src, idx, nslots, out);
    }
    const hipError_t e = hipDeviceSynchronize();
    (void)hipFree(src); (void)hipFree(out); (void)hipFree(idx);
    return e == hipSuccess ? 0 : 3;
}
}
