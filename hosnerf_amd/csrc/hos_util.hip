// Small multi-buffer helpers (round 4): the launches that were left to torch in a captured training step -- fills of
// accumulation targets, stacks of per-frame tensors, the sums autograd forms when a tensor feeds several consumers, a
// device-side `any` -- as ONE launch per group of buffers.  The pointer / size tables travel by value in the kernel arguments
// (no host memory is read after the call returns, so the calls are capture-safe).
#include "hos_common.h"

namespace {

constexpr int UN = 8;
struct SegTable { float* dst[UN]; const float* src[UN]; long n[UN]; long start[UN + 1]; int count; };

// segment s: dst[s][0..n) = src[s] ? src[s][i] : 0
__global__ __launch_bounds__(256) void copy_or_zero_n_kernel(const SegTable t) {
    const long total = t.start[t.count];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int s = 0;
#pragma unroll
        for (int k = 1; k < UN; ++k) if (k < t.count && i >= t.start[k]) s = k;
        const long j = i - t.start[s];
        t.dst[s][j] = t.src[s] ? t.src[s][j] : 0.f;
    }
}

struct AddTable { const float* src[UN]; int count; };
__global__ __launch_bounds__(256) void add_n_kernel(const AddTable t, long n, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = t.src[0][i];
#pragma unroll
        for (int k = 1; k < UN; ++k) if (k < t.count) s += t.src[k][i];
        out[i] = s;
    }
}

// flag[0] = any(|x[i]| < thr), one workgroup (n is a few thousand: the ray directions of a batch)
__global__ __launch_bounds__(1024) void any_abs_below_kernel(const float* __restrict__ x, long n, float thr, int* __restrict__ flag) {
    __shared__ int hit;
    if (threadIdx.x == 0) hit = 0;
    __syncthreads();
    bool f = false;
    for (long i = threadIdx.x; i < n; i += 1024) f |= fabsf(x[i]) < thr;
    if (__builtin_amdgcn_ballot_w64(f) != 0 && (threadIdx.x & 63) == 0) atomicOr(&hit, 1);
    __syncthreads();
    if (threadIdx.x == 0) flag[0] = hit;
}

// gb[n] += db[n];  g_embed[e] += sum_n db[n] W[n][c0 + e]   (n < N, e < E <= 64): the state embedding is one vector per call, so
// its gradient is the bias gradient through its columns of the weight (M:295-296, N:177-230).  One workgroup of 16 waves, each
// walking every 16th row with eight loads in flight (the first version -- 4 waves, one dependent load per iteration over 256
// rows -- took 136 us per call for 256 KB: pure latency); fixed summation order.
__global__ __launch_bounds__(1024) void state_embed_grad_kernel(const float* __restrict__ db, const float* __restrict__ W, int ldw, int c0, int N,
                                                                int E, float* __restrict__ gb, float* __restrict__ g_embed) {
    __shared__ float part[16][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    float s = 0.f;
    if (e < E) {
        const float* w = W + c0 + e;
        int n = q;
        for (; n + 7 * 16 < N; n += 8 * 16) {
            float v[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = w[(size_t)(n + 16 * u) * ldw]; d[u] = db[n + 16 * u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += d[u] * v[u];
        }
        for (; n < N; n += 16) s += db[n] * w[(size_t)n * ldw];
    }
    part[q][e] = s;
    if (gb != nullptr)
        for (int n = threadIdx.x; n < N; n += 1024) gb[n] += db[n];
    __syncthreads();
    if (q == 0 && e < E) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][e];
        g_embed[e] += t;
    }
}

}  // namespace

// n <= 8 segments in one launch: dst[s][0..count[s]) = src[s] (NULL: zeros).  Fills of accumulation targets, stacks of per-frame
// tensors (`torch.stack` of the prologue's inputs, the backward of a per-frame unbind).  No reference counterpart: plumbing.
extern "C" int hos_copy_or_zero_n(int n, float* const* dst, const float* const* src, const long long* count, hos_stream_t stream) {
    if (n <= 0 || n > UN || !dst || !count) return HOS_E_ARG;
    SegTable t{};
    long pos = 0;
    for (int s = 0; s < n; ++s) {
        if (!dst[s] || count[s] < 0) return HOS_E_ARG;
        t.dst[s] = dst[s]; t.src[s] = src ? src[s] : nullptr; t.n[s] = (long)count[s]; t.start[s] = pos;
        pos += (long)count[s];
    }
    t.start[n] = pos; t.count = n;
    for (int s = n + 1; s <= UN; ++s) t.start[s] = pos;
    if (pos == 0) return HOS_OK;
    long blocks = (pos + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy_or_zero_n_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), t);
    return hos_launch_status();
}

// out[i] = sum_k src[k][i], 2 <= n <= 8 (the gradient of a tensor that feeds several consumers, formed in one pass).
extern "C" int hos_add_n(int n, const float* const* src, long long count, float* out, hos_stream_t stream) {
    if (n < 1 || n > UN || !src || !out || count <= 0) return HOS_E_ARG;
    AddTable t{};
    for (int k = 0; k < n; ++k) { if (!src[k]) return HOS_E_ARG; t.src[k] = src[k]; }
    t.count = n;
    long blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_n_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), t, (long)count, out);
    return hos_launch_status();
}

// flag[0] = 1 if any |x[i]| < thr else 0 (M:1526: `if (torch.abs(rays_d) < 1e-5).any()` of the stage-3 re-projection, kept on the device).
extern "C" int hos_any_abs_below(const float* x, long long n, float thr, int32_t* flag, hos_stream_t stream) {
    if (!x || !flag || n <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(any_abs_below_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), x, (long)n, thr, flag);
    return hos_launch_status();
}

// gb [N] += db (NULL: skip);  g_embed [E] += db [N] . W [N, ldw][:, c0 : c0 + E].  E <= 64.  Reference: the state embedding is
// concatenated to every sample's encoding (M:295-296), so its gradient is the column block of the first / skip layer's weight
// applied to that layer's bias gradient.
extern "C" int hos_state_embed_grad(const float* db, const float* W, int ldw, int c0, int N, int E, float* gb, float* g_embed, hos_stream_t stream) {
    if (!db || !W || !g_embed || N <= 0 || E <= 0 || c0 < 0) return HOS_E_ARG;
    if (E > 64) return HOS_E_SHAPE;
    hipLaunchKernelGGL(state_embed_grad_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), db, W, ldw, c0, N, E, gb, g_embed);
    return hos_launch_status();
}

// Diagnostics: buf[slot] = the constant 100 MHz counter when this point of the stream is reached (one thread).  A kernel node like
// any other, so it can sit inside a captured step: the un-profiled timeline of the two-stream step (scripts/diag_overlap.py;
// rocprofv3's per-node interception stretches the host-side replay and with it the picture).
namespace {
__global__ void debug_stamp_kernel(long long* __restrict__ buf, int slot) { buf[slot] = (long long)wall_clock64(); }
}
extern "C" int hos_clear_last_error(void) { return static_cast<int>(hipGetLastError()); }

extern "C" int hos_debug_stamp(long long* buf, int slot, hos_stream_t stream) {
    if (!buf || slot < 0) return HOS_E_ARG;
    hipLaunchKernelGGL(debug_stamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), buf, slot);
    return hos_launch_status();
}
