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


// ---- round 4: ONE norm launch + ONE update launch per training step, whatever the number of flat buffers / learning-rate ranges,
// and the fp16 range guard consumed on the device.
constexpr int OPT_SPANS = 32;      // (round 5: 8 -> 16 for the sharded decoder; round 6: -> 32, every lazily updated span is a span of its own)
constexpr int SUMSQ_BLOCKS = 1024;
struct SumsqSpans { const float* g[OPT_SPANS]; long start4[OPT_SPANS + 1]; int count; };      // spans in float4 units (n % 4 == 0)

// partial[b] = sum of squares of block b's share of all spans: plain stores in a fixed order (no atomics, nothing to zero first,
// bit-reproducible); the update kernel adds the SUMSQ_BLOCKS partials itself.
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const SumsqSpans t, float* __restrict__ partial) {
    float acc = 0.f;
    const long total4 = t.start4[t.count];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        int s = 0;
#pragma unroll
        for (int k = 1; k < OPT_SPANS; ++k) if (k < t.count && i >= t.start4[k]) s = k;
        const float4 v = reinterpret_cast<const float4*>(t.g[s])[i - t.start4[s]];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

struct AdamSpans {
    float* p[OPT_SPANS]; const float* g[OPT_SPANS]; float* m[OPT_SPANS]; float* v[OPT_SPANS];
    const float* hyper[OPT_SPANS];                       // device {lr, 1-b1^t, 1/sqrt(1-b2^t)} of the span, or NULL: the host values below
    const float* lazy[OPT_SPANS];                        // device {t, active, 1-b1^t, 1/sqrt(1-b2^t), ..} rows (8 floats each) of a lazily updated span, or NULL
    int lazy_row4[OPT_SPANS];                            // 0: one state row for the span; else the span is ROWS of this many float4 with one state row each
    float lr[OPT_SPANS], bc1[OPT_SPANS], rsbc2[OPT_SPANS];
    long start4[OPT_SPANS + 1]; int count;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamSpans t, float b1, float b2, float eps, float grad_scale,
                                                         const float* __restrict__ partial, int n_partial, float max_norm,
                                                         const unsigned int* __restrict__ guard, unsigned int* __restrict__ skipped) {
    // fp16 range guard: a hidden activation of this step's forward left the exact hi/lo range (the forward epilogues OR the word):
    // THIS step must not reach the parameters -- and only this one (round 5; the word used to stay set until a host poll, so a loop
    // that never polled lost every later step).  The decision is BLOCK-UNIFORM (round 6): thread 0 alone reads the word and
    // hands it to the workgroup through LDS behind a barrier; only then does it take its ticket.  The LAST ticket holder (counter
    // skipped[1], self-resetting, so a replayed graph behaves the same) counts the skip in skipped[0] and clears the word: by then
    // thread 0 of every workgroup has read it -- no wave of any workgroup reads the word itself, so none can see the cleared value
    // and run Adam (or reach the clip barrier below without its wave 0) -- and the next step's forward epilogues are behind this
    // launch in stream order.  `skipped == nullptr`: the word is left set (an earlier launch of a step that issues several).
    __shared__ unsigned int guard_word;
    if (threadIdx.x == 0) guard_word = guard != nullptr ? __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    __syncthreads();
    if (guard_word != 0u) {
        if (threadIdx.x == 0 && skipped != nullptr) {
            const unsigned ticket = atomicAdd(skipped + 1, 1u);
            if (ticket == gridDim.x - 1) {
                atomicAdd(skipped, 1u);
                skipped[1] = 0u;
                __hip_atomic_store(const_cast<unsigned int*>(guard), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    float gs = grad_scale;
    if (partial != nullptr && max_norm > 0.f) {
        __shared__ float red[4];
        float s = 0.f;
        for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];      // fixed order per thread, fixed tree below
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float total = sqrtf(red[0] + red[1] + red[2] + red[3]) * fabsf(grad_scale);
        gs *= fminf(max_norm / (total + 1e-6f), 1.f);      // torch.nn.utils.clip_grad_norm_
    }
    const long total4 = t.start4[t.count];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        int s = 0;
#pragma unroll
        for (int k = 1; k < OPT_SPANS; ++k) if (k < t.count && i >= t.start4[k]) s = k;
        const long j = i - t.start4[s];
        float lr = t.lr[s], bc1 = t.bc1[s], rsbc2 = t.rsbc2[s];
        if (t.hyper[s] != nullptr) { lr = t.hyper[s][0]; bc1 = t.hyper[s][1]; rsbc2 = t.hyper[s][2]; }
        if (t.lazy[s] != nullptr) {          // torch.optim.Adam skips a parameter whose .grad is None and counts ITS steps only
            const float* lz = t.lazy[s] + (t.lazy_row4[s] > 0 ? (j / t.lazy_row4[s]) * 8 : 0);
            if (lz[1] == 0.f) continue;
            bc1 = lz[2]; rsbc2 = lz[3];
        }
        float4* p4 = reinterpret_cast<float4*>(t.p[s]) + j;
        float4* m4 = reinterpret_cast<float4*>(t.m[s]) + j;
        float4* v4 = reinterpret_cast<float4*>(t.v[s]) + j;
        float4 pp = *p4, mm = *m4, vv = *v4;
        const float4 gg = reinterpret_cast<const float4*>(t.g[s])[j];
        adam1(pp.x, gg.x * gs, mm.x, vv.x, lr, b1, b2, eps, bc1, rsbc2);
        adam1(pp.y, gg.y * gs, mm.y, vv.y, lr, b1, b2, eps, bc1, rsbc2);
        adam1(pp.z, gg.z * gs, mm.z, vv.z, lr, b1, b2, eps, bc1, rsbc2);
        adam1(pp.w, gg.w * gs, mm.w, vv.w, lr, b1, b2, eps, bc1, rsbc2);
        *p4 = pp; *m4 = mm; *v4 = vv;
    }
}

// LAZY_BLOCKS workgroups per lazily updated span: state[s] = {t, active, 1-b1^t, 1/sqrt(1-b2^t), <flag scratch>, <ticket scratch>, -, -}
// (8 floats).  active = the span's (already reduced) gradient is not identically zero; then t += 1 and the bias corrections are those
// of ITS t-th update.  Every workgroup ORs what it found into the flag word and takes a ticket; the last one decides and re-arms both
// scratch words (self-resetting: a replayed graph behaves the same).
constexpr int LAZY_BLOCKS = 16;
struct LazySpans { const float* g[OPT_SPANS]; float* state[OPT_SPANS]; long count[OPT_SPANS]; };
__global__ __launch_bounds__(256) void adam_lazy_prepare_kernel(const LazySpans t, float b1, float b2, const unsigned int* __restrict__ guard) {
    const int s = blockIdx.x / LAZY_BLOCKS, part = blockIdx.x % LAZY_BLOCKS;
    const float4* g4 = reinterpret_cast<const float4*>(t.g[s]);
    const long n4 = t.count[s] >> 2;                      // (spans are float4-aligned multiples of 4)
    int any = 0;
    for (long i = (long)part * 256 + threadIdx.x; i < n4; i += (long)LAZY_BLOCKS * 256) {
        const float4 v = g4[i];
        any |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    }
    float* st = t.state[s];
    unsigned int* flag = reinterpret_cast<unsigned int*>(st + 4);
    unsigned int* ticket = reinterpret_cast<unsigned int*>(st + 5);
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
    __syncthreads();                                       // every wave's atomicOr is issued before this workgroup's ticket
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1u) == LAZY_BLOCKS - 1) {
            __threadfence();
            const bool nonzero = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            const bool poisoned = guard != nullptr && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (nonzero && !poisoned) {          // (a step the range guard is about to skip updates nothing and counts nothing)
                const float n = st[0] + 1.f;
                st[0] = n; st[1] = 1.f;
                st[2] = 1.f - powf(b1, n);
                st[3] = 1.f / sqrtf(1.f - powf(b2, n));
            } else {
                st[1] = 0.f;
            }
            *flag = 0u; *ticket = 0u;
        }
    }
}

}  // namespace

// Lazily updated spans (round 6): torch.optim.Adam -- the reference's optimiser under Lightning, whose zero_grad sets gradients to
// None (torch 2.0.1 default) -- SKIPS a parameter that took no part in a step (no moment decay, no movement) and keeps a step count
// PER PARAMETER for the bias corrections.  Such parameters exist: the state embeddings of the states a step's frame is not in
// (M:224-296 / N:179-246: one state per call), the pose decoder before its kick-in iteration (N:589-605).  Their flat gradient is
// identically zero in such a step, which is what this launch tests (after the all-reduce: every rank decides the same).
// n <= 32 spans (float4-aligned, count % 4 == 0); state[s]: 8 floats {t, active, 1-beta1^t, 1/sqrt(1-beta2^t), two scratch words, 2 unused},
// zero-initialised by the caller, consumed by hos_adam_multi_lazy.  guard: the range-guard word (NULL: off) -- a poisoned step counts nothing.
extern "C" int hos_adam_lazy_prepare(int n, const float* const* g, const long long* count, float* const* state, float beta1, float beta2,
                                     const unsigned int* guard, hos_stream_t stream) {
    if (n <= 0 || n > OPT_SPANS || !g || !count || !state) return HOS_E_ARG;
    LazySpans t{};
    for (int s = 0; s < n; ++s) {
        if (!g[s] || !state[s] || count[s] <= 0) return HOS_E_ARG;
        if ((count[s] & 3) || ((uintptr_t)g[s] & 15u) || ((uintptr_t)state[s] & 3u)) return HOS_E_ALIGN;
        t.g[s] = g[s]; t.state[s] = state[s]; t.count[s] = (long)count[s];
    }
    hipLaunchKernelGGL(adam_lazy_prepare_kernel, dim3(n * LAZY_BLOCKS), dim3(256), 0, static_cast<hipStream_t>(stream), t, beta1, beta2, guard);
    return hos_launch_status();
}

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

extern "C" int hos_sumsq_blocks(void) { return SUMSQ_BLOCKS; }

// partial[0 .. hos_sumsq_blocks()) = per-block sums of squares over n <= 8 spans (every count % 4 == 0, 16-byte aligned): the
// gradient norm of `Trainer(gradient_clip_val=..., "norm")` (S1/run.py:155, 3rd_.../run.py:188-189) over ALL flat buffers of a step
// in one launch, deterministic, nothing to zero beforehand.  Consumed by hos_adam_multi.
extern "C" int hos_sumsq_partials(int n, const float* const* g, const long long* count, float* partial, hos_stream_t stream) {
    if (n <= 0 || n > OPT_SPANS || !g || !count || !partial) return HOS_E_ARG;
    SumsqSpans t{};
    long pos = 0;
    for (int s = 0; s < n; ++s) {
        if (!g[s] || count[s] <= 0) return HOS_E_ARG;
        if ((count[s] & 3) || ((uintptr_t)g[s] & 15u)) return HOS_E_ALIGN;
        t.g[s] = g[s]; t.start4[s] = pos; pos += (long)(count[s] >> 2);
    }
    for (int s = n; s <= OPT_SPANS; ++s) t.start4[s] = pos;
    t.count = n;
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, static_cast<hipStream_t>(stream), t, partial);
    return hos_launch_status();
}

// torch.optim.Adam over n <= 16 spans of flat buffers in ONE launch (M1:536-569, optimizer.py:19-60: the reference's per-parameter
// groups are contiguous ranges here).  Span s: p/g/m/v[s][0 .. count[s]) with count % 4 == 0; its step scalars come from device
// memory (hyper[s] = {lr, 1-beta1^t, 1/sqrt(1-beta2^t)}, graph replay) or, where hyper[s] is NULL, from lr[s] and `step`.
// partial (NULL: no clipping): hos_sumsq_partials' output, summed here; coefficient min(max_norm / (sqrt(sum) * |grad_scale| + 1e-6), 1).
// guard (NULL: off): the range-guard word (hos_set_range_flag); non-zero -> NOTHING is updated by this launch, skipped[0] is
// incremented and the word is cleared for the next step.  skipped: two zero-initialised words {count, ticket scratch} (NULL: the word
// is left set, nothing is counted).
extern "C" int hos_adam_multi(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* count,
                              const float* const* hyper, const float* lr, int step, float beta1, float beta2, float eps, float grad_scale,
                              const float* partial, float max_norm, const unsigned int* guard, unsigned int* skipped, hos_stream_t stream) {
    return hos_adam_multi_lazy(n, p, g, m, v, count, hyper, nullptr, nullptr, lr, step, beta1, beta2, eps, grad_scale, partial, max_norm, guard, skipped, stream);
}

// hos_adam_multi with lazily updated spans: lazy[s] (NULL entries / NULL table: a plain span) = the span's state row written by
// hos_adam_lazy_prepare earlier on the same stream: inactive -> the span is not touched; active -> its own bias corrections.
// lazy_row[s] (NULL table / 0: one row for the whole span) > 0: the span consists of consecutive ROWS of lazy_row[s] floats (a
// multiple of 4 that divides count[s]) with consecutive 8-float state rows -- the [n_states, 64] block of state embeddings is ONE span.
extern "C" int hos_adam_multi_lazy(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* count,
                                   const float* const* hyper, const float* const* lazy, const int* lazy_row, const float* lr, int step, float beta1, float beta2,
                                   float eps, float grad_scale, const float* partial, float max_norm, const unsigned int* guard,
                                   unsigned int* skipped, hos_stream_t stream) {
    if (n <= 0 || n > OPT_SPANS || !p || !g || !m || !v || !count) return HOS_E_ARG;
    AdamSpans t{};
    long pos = 0;
    const double bc1 = step >= 1 ? 1.0 - pow((double)beta1, (double)step) : 1.0;
    const double bc2 = step >= 1 ? 1.0 - pow((double)beta2, (double)step) : 1.0;
    for (int s = 0; s < n; ++s) {
        if (!p[s] || !g[s] || !m[s] || !v[s] || count[s] <= 0) return HOS_E_ARG;
        if ((count[s] & 3) || (((uintptr_t)p[s] | (uintptr_t)g[s] | (uintptr_t)m[s] | (uintptr_t)v[s]) & 15u)) return HOS_E_ALIGN;
        const float* h = hyper ? hyper[s] : nullptr;
        if (!h && (!lr || step < 1)) return HOS_E_ARG;
        t.p[s] = p[s]; t.g[s] = g[s]; t.m[s] = m[s]; t.v[s] = v[s]; t.hyper[s] = h;
        t.lazy[s] = lazy ? lazy[s] : nullptr;
        t.lazy_row4[s] = 0;
        if (t.lazy[s] && lazy_row && lazy_row[s] > 0) {
            if ((lazy_row[s] & 3) || count[s] % lazy_row[s]) return HOS_E_ALIGN;
            t.lazy_row4[s] = lazy_row[s] >> 2;
        }
        t.lr[s] = lr ? lr[s] : 0.f; t.bc1[s] = (float)bc1; t.rsbc2[s] = (float)(1.0 / sqrt(bc2));
        t.start4[s] = pos; pos += (long)(count[s] >> 2);
    }
    for (int s = n; s <= OPT_SPANS; ++s) t.start4[s] = pos;
    t.count = n;
    long blocks = (pos + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), t, beta1, beta2, eps, grad_scale,
                       partial, SUMSQ_BLOCKS, max_norm, guard, skipped);
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
