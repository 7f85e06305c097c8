// Flat-buffer optimiser: all parameters of a model live in ONE contiguous fp32 buffer in HBM
// (same for grads and both Adam moments), so a training step is one sum-of-squares reduction,
// one RCCL all-reduce (multi-GPU) and one Adam launch -- instead of ~50 per-tensor launches of
// torch.optim.Adam + clip_grad_norm_ (M1:536-569, S1/run.py:155).
// HBM-bound: 16 B read + 12 B written per parameter.
#include "hos_common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr, float b1, float b2,
                                      float eps, float bc1, float rsbc2) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    // torch.optim.Adam: p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
    const float denom = sqrtf(v) * rsbc2 + eps;
    p -= (lr / bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float bc1, float rsbc2,
                                                   float grad_scale, const float* __restrict__ sumsq, float max_norm,
                                                   const float* __restrict__ hyper) {
    if (hyper != nullptr) { lr = hyper[0]; bc1 = hyper[1]; rsbc2 = hyper[2]; }   // per-step values from device memory (graph replay)
    float gs = grad_scale;
    if (sumsq != nullptr && max_norm > 0.f) {
        // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
        const float total = sqrtf(sumsq[0]) * fabsf(grad_scale);
        gs *= fminf(max_norm / (total + 1e-6f), 1.f);
    }
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        adam1(pp.x, gg.x * gs, mm.x, vv.x, lr, b1, b2, eps, bc1, rsbc2);
        adam1(pp.y, gg.y * gs, mm.y, vv.y, lr, b1, b2, eps, bc1, rsbc2);
        adam1(pp.z, gg.z * gs, mm.z, vv.z, lr, b1, b2, eps, bc1, rsbc2);
        adam1(pp.w, gg.w * gs, mm.w, vv.w, lr, b1, b2, eps, bc1, rsbc2);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        adam1(p[i], g[i] * gs, m[i], v[i], lr, b1, b2, eps, bc1, rsbc2);
    }
}

}  // namespace

extern "C" int hos_sumsq(const float* g, int64_t n, float* sumsq, hos_stream_t stream) {
    if (!g || !sumsq || n <= 0) return HOS_E_ARG;
    HOS_CHECK_ALIGN16(g);
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g, n, sumsq);
    return hos_launch_status();
}

extern "C" int hos_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int step, float grad_scale, const float* sumsq,
                             float max_norm, hos_stream_t stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return HOS_E_ARG;
    HOS_CHECK_ALIGN16(p); HOS_CHECK_ALIGN16(g); HOS_CHECK_ALIGN16(m); HOS_CHECK_ALIGN16(v);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, lr,
                       beta1, beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, sumsq, max_norm, (const float*)nullptr);
    return hos_launch_status();
}

extern "C" int hos_adam_step_dyn(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                                 float beta1, float beta2, float eps, float grad_scale, const float* sumsq,
                                 float max_norm, hos_stream_t stream) {
    if (!p || !g || !m || !v || !hyper || n <= 0) return HOS_E_ARG;
    HOS_CHECK_ALIGN16(p); HOS_CHECK_ALIGN16(g); HOS_CHECK_ALIGN16(m); HOS_CHECK_ALIGN16(v);
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, 0.f,
                       beta1, beta2, eps, 1.f, 1.f, grad_scale, sumsq, max_norm, hyper);
    return hos_launch_status();
}

extern "C" int hos_version(void) { return 100; }

extern "C" int hos_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : HOS_E_NODEVICE;
}

extern "C" const char* hos_error_string(int code) {
    switch (code) {
        case HOS_OK: return "ok";
        case HOS_E_ARG: return "invalid argument (null pointer or non-positive size)";
        case HOS_E_ALIGN: return "pointer / leading dimension not 16-byte aligned";
        case HOS_E_SHAPE: return "unsupported shape";
        case HOS_E_NODEVICE: return "no HIP device";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown error";
    }
}
