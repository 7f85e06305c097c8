// Device-side selection of the cycle-consistency set (N:505-536): the reference picks the sample points
// with `fg_likelihood_mask > 0.005` by boolean indexing -- a data-dependent shape, i.e. a device->host round trip in the
// middle of every training step and the one thing that keeps the step out of a hipGraph.  Here the selection is an
// order-preserving stream compaction into FIXED-CAPACITY buffers with the row count left in device memory:
//
//   hos_compact_rows      sel = ascending indices i with mask[i] > thr (what torch.nonzero returns); out_a[j] = src_a[sel[j]],
//                         out_b[j] = src_b[sel[j]] for j < count; rows >= count are zero-filled; *count = number selected
//   hos_scatter_rows      dst[sel[j]] = src[j] for j < count, every other row of dst zero (the gather's gradient)
//
// Two launches: per-block counts + an exclusive scan by the last block to finish, then the scatter pass.
#include "hos_common.h"

namespace {

constexpr int CT = 256;            // threads
constexpr int CPT = 4;             // elements per thread
constexpr int CB = CT * CPT;       // elements per block
constexpr int CMAXB = 4096;        // blocks (capacity 4 Mi rows)

__global__ __launch_bounds__(CT) void compact_count_kernel(const float* mask, float thr, long P, int* blk_off, unsigned int* ticket,
                                                          int* count) {
    __shared__ int sh[CT / 64];
    __shared__ bool last;
    const int tid = threadIdx.x;
    const long base = (long)blockIdx.x * CB + tid * CPT;
    int c = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) c += (base + k < P && mask[base + k] > thr) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((tid & 63) == 0) sh[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < CT / 64; ++w) tot += sh[w];
        blk_off[blockIdx.x] = tot;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // exclusive scan of the block counts (<= 4096 values) by this block: 16 per thread, then across threads
    __shared__ int tsum[CT];
    const int nb = gridDim.x, per = (nb + CT - 1) / CT;
    int loc = 0;
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < nb) loc += ((volatile int*)blk_off)[b];
    }
    tsum[tid] = loc;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < CT; ++t) { const int v = tsum[t]; tsum[t] = run; run += v; }
        *count = run;
        *ticket = 0u;
    }
    __syncthreads();
    int run = tsum[tid];
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < nb) { const int v = ((volatile int*)blk_off)[b]; blk_off[b] = run; run += v; }
    }
}

__global__ __launch_bounds__(CT) void compact_scatter_kernel(const float* mask, float thr, long P, const int* blk_off, const int* count,
                                                            const float* src_a, const float* src_b, int* sel, float* out_a, float* out_b) {
    __shared__ int sh[CT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long base = (long)blockIdx.x * CB + tid * CPT;
    bool f[CPT];
    int c = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) { f[k] = base + k < P && mask[base + k] > thr; c += f[k] ? 1 : 0; }
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int pos = blk_off[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) pos += sh[w];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        if (f[k]) {
            const long i = base + k;
            if (sel) sel[pos] = (int)i;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                if (out_a) out_a[3L * pos + e] = src_a[3 * i + e];
                if (out_b) out_b[3L * pos + e] = src_b[3 * i + e];
            }
            ++pos;
        }
    }
    // zero the tail rows [count, P) so that fixed-capacity consumers see finite values
    const int n = *count;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const long i = base + k;
        if (i >= n && i < P) {
            if (sel) sel[i] = -1;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                if (out_a) out_a[3 * i + e] = 0.f;
                if (out_b) out_b[3 * i + e] = 0.f;
            }
        }
    }
}

// (a kernel instead of hipMemsetAsync: see zero2d_kernel in hos_gemm.hip -- memset nodes of a captured graph)
__global__ __launch_bounds__(256) void zero_kernel(float* __restrict__ p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}

__global__ __launch_bounds__(CT) void scatter_rows_kernel(const float* src, const int* sel, const int* count, long P, float* dst) {
    const long i = (long)blockIdx.x * CT + threadIdx.x;
    if (i >= P) return;
    if (i < *count) {
        const long d = sel[i];
#pragma unroll
        for (int e = 0; e < 3; ++e) dst[3 * d + e] = src[3 * i + e];
    }
}

}  // namespace

extern "C" long long hos_compact_workspace_ints(void) { return CMAXB + 4; }

extern "C" int hos_compact_rows(const float* mask, float thr, const float* src_a, const float* src_b, int64_t P, int32_t* count,
                                int32_t* sel, float* out_a, float* out_b, int32_t* workspace, hos_stream_t stream) {
    if (!mask || !count || !workspace || P <= 0) return HOS_E_ARG;
    if ((out_a && !src_a) || (out_b && !src_b)) return HOS_E_ARG;
    const long nb = (P + CB - 1) / CB;
    if (nb > CMAXB) return HOS_E_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(CT), 0, s, mask, thr, (long)P, workspace,
                       reinterpret_cast<unsigned int*>(workspace + CMAXB), count);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nb), dim3(CT), 0, s, mask, thr, (long)P, workspace, count, src_a, src_b,
                       sel, out_a, out_b);
    return hos_launch_status();
}

extern "C" int hos_scatter_rows(const float* src, const int32_t* sel, const int32_t* count, int64_t P, float* dst, hos_stream_t stream) {
    if (!src || !sel || !count || !dst || P <= 0) return HOS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {
        long blocks = (P * 3 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, (long)P * 3);
    }
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((P + CT - 1) / CT)), dim3(CT), 0, s, src, sel, count, (long)P, dst);
    return hos_launch_status();
}
