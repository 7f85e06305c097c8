// One-column heads on planes: out[m] = softplus( sum_k A[m][k] * w[k] + bias + p0 )  for a matrix stored as fp16 (hi, lo) planes.
//
// The density heads of the proposal MLPs (Linear(256, 1), M:158-160 / M:325) and the density column of the NeRF MLP's 257-wide
// head are GEMMs with ONE output column.  Through the planes GEMM they cost a whole 256 x 128 tile per 256 rows (127 us for
// [262144, 1, 256], ~250 us for the remainder launch of [131072, 257, 1024]) although the work is one pass over A: 268 / 537 MB.
// Here the products are fp32 FMAs of the hi and lo values with the fp32 weight row (the weight is not split: exact operand) and
// the lanes of a row are summed with xor-shuffles in a fixed order (bit-reproducible).  Roofline: HBM, 4 B per element of A.
// planes_rowdot_coalesced_kernel: a wave reads 1 KB of a row per load instruction (81 us average over the three launches of a
// stage-3 step = 4.4 TB/s; the first version -- a row shared by a few lanes, one 128-byte line per lane and step -- ran at 127 us
// and was removed in round 5).  E = _Float16 (proposal MLPs) or __bf16 (the NeRF MLP's bf16-only forward, round 5).
#include "hos_common.h"
#include <cstdlib>

namespace {


// The 64 lanes of a wave read 64 CONSECUTIVE 16-byte pieces of a row (1 KB per load instruction; a 32-column
// block = 128-byte line = 4 hi pieces then 4 lo pieces, eight columns each), every lane multiplies its eight values by their
// fp32 weights -- hi and lo pieces are separate terms of the same sum -- and the 64 partial sums of a row are added with a
// fixed-order butterfly.  RU rows per pass keep RU x (pieces / 64) loads in flight per lane.  (The lane-per-line form above
// touches 64 different lines per load instruction: 2.7-2.9 TB/s.)
template <int RU, typename E>
__global__ __launch_bounds__(256) void planes_rowdot_coalesced_kernel(const uint16_t* __restrict__ A, int lda, int nblk,
                                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                                      float p0, int softplus, long M, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
    const float add = (bias != nullptr ? bias[0] : 0.f) + p0;
    const int pieces = nblk * 8;                              // 16-byte pieces per row
    for (long r0 = wave_id * RU; r0 < M; r0 += nwaves * RU) {
        float acc[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) acc[u] = 0.f;
        for (int g = lane; g < pieces; g += 64) {
            const int blk = g >> 3, q = g & 7;
            const float4* wp = reinterpret_cast<const float4*>(w + blk * 32 + (q & 3) * 8);
            const float4 w0 = wp[0], w1 = wp[1];
            uint4 v[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const long row = r0 + u < M ? r0 + u : M - 1;  // clamped for the loads; never stored
                v[u] = *reinterpret_cast<const uint4*>(A + (size_t)row * (2 * (size_t)lda) + (size_t)g * 8);
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                typedef E e8 __attribute__((ext_vector_type(8)));
                const e8 h = __builtin_bit_cast(e8, v[u]);
                float a = acc[u];
                a = fmaf((float)h[0], w0.x, a); a = fmaf((float)h[1], w0.y, a); a = fmaf((float)h[2], w0.z, a); a = fmaf((float)h[3], w0.w, a);
                a = fmaf((float)h[4], w1.x, a); a = fmaf((float)h[5], w1.y, a); a = fmaf((float)h[6], w1.z, a); a = fmaf((float)h[7], w1.w, a);
                acc[u] = a;
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            float a = acc[u];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
            if (lane == 0 && r0 + u < M) {
                const float v = a + add;
                out[r0 + u] = softplus ? softplus_f(v) : v;
            }
        }
    }
}

}  // namespace

namespace {
template <typename E>
int rowdot_launch(const void* A, int lda, int K, const float* w, const float* bias, float p0, int softplus, int64_t M, float* out,
                  hos_stream_t stream) {
    if (!A || !w || !out || M <= 0 || K <= 0) return HOS_E_ARG;
    if ((K & 31) || (lda & 31) || K > lda) return HOS_E_SHAPE;
    if ((((uintptr_t)A) | ((uintptr_t)w)) & 15u) return HOS_E_ALIGN;
    long blocks = (M + 15) / 16;                    // 4 rows per wave and pass
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((planes_rowdot_coalesced_kernel<4, E>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t*>(A), lda, K / 32, w, bias, p0, softplus, (long)M, out);
    return hos_launch_status();
}
}  // namespace

// out[M] = act( A[M, :K] . w[:K] + bias[0] + p0 ), A = fp16 planes [M][lda] (K % 32 == 0, K <= lda), w fp32 [K] (16-byte aligned),
// bias a device scalar (may be NULL); act = torch.nn.Softplus if softplus != 0.  Replaces hos_linearp_fwd with N = 1 /
// HOS_EPI_DENSITY and the density column of HOS_EPI_NERF_HEAD (the reference's density heads: mipnerf360/model.py:158-160, 325).
extern "C" int hos_planes_rowdot(const void* A, int lda, int K, const float* w, const float* bias, float p0, int softplus, int64_t M,
                                 float* out, hos_stream_t stream) {
    return rowdot_launch<_Float16>(A, lda, K, w, bias, p0, softplus, M, out, stream);
}
// The same head on bf16 planes (the activations hos_linearp_fwd_b writes).
extern "C" int hos_planes_rowdot_b(const void* A, int lda, int K, const float* w, const float* bias, float p0, int softplus, int64_t M,
                                   float* out, hos_stream_t stream) {
    return rowdot_launch<__bf16>(A, lda, K, w, bias, p0, softplus, M, out, stream);
}
