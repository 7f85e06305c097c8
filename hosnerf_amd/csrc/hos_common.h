// Internal helpers shared by the HIP translation units of libhosrender.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hosrender.h"

#define HOS_WAVE 64

#define HOS_CHECK_PTR(p) \
    do { if ((p) == nullptr) return HOS_E_ARG; } while (0)
#define HOS_CHECK_ALIGN16(p) \
    do { if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) return HOS_E_ALIGN; } while (0)

static inline int hos_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HOS_OK : static_cast<int>(e);
}

static inline int hos_cdiv(int a, int b) { return (a + b - 1) / b; }
unsigned int* hos_range_flag_ptr();      // hos_gemm.hip: the device word registered with hos_set_range_flag (or NULL)

// ---- wave-level primitives (64 lanes) ------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}
// inclusive suffix sum (sum over lanes >= lane)
__device__ __forceinline__ float wave_incl_rscan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float n = __shfl_down(v, o, 64);
        if (lane + o < 64) v += n;
    }
    return v;
}

__device__ __forceinline__ float softplus_f(float x) {
    // torch.nn.Softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
