// Shared by the fp32-MFMA (hos_gemm.hip) and the bf16x3 split-precision (hos_gemm3.hip) tile kernels:
// argument block and the fused epilogue.  The 32x32 MFMA C/D layout is dtype independent on gfx950:
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) for accumulator register r.
#pragma once
#include "hos_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Mode { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

struct GemmArgs {
    const float* A0; int lda0; int kt0;   // A segment 0, number of K tiles in it
    const float* A1; int lda1;            // optional A segment 1 (fwd skip-concat)
    const float* B;  int ldb;
    float* C; int ldc;
    int M, N;                // store extents (rows i, cols j)
    int Mload, Nload;        // operand extents (may include zero padding)
    int red_limit;           // valid reduction rows for reduction-row operands (WGRAD with M % 32 != 0)
    int nk;                  // K tiles in total
    int kt_per_split;
    int tiles_m, tiles_n;
    const float* bias;
    const float* mask; int ldmask;
    float* aux; int aux_col; float p0, p1;
    float* db;
    int accumulate;
    int epi;
    int ablate;              // debug only (HOS_GEMM_ABLATE): 1 = skip global loads, 2 = skip convert/store, 4 = skip MFMA
    int pf_dist;             // split kernels: software L2 prefetch distance in K tiles (0 = off; HOS_GEMM_PF)
    const int* m_dev;        // FWD only, optional: live row count in device memory; row tiles at or past it are skipped
    unsigned int* range_flag;   // FWD, optional: set to 1 when a hidden activation leaves the exactly-representable fp16 hi/lo range
};

// Forward activations travel as fp16 (hi, lo) pairs: exact to 2^-22 for |x| <= 65504, hi saturates there and lo carries the
// residual up to |x| = 131008 (2^-11 relative), beyond that the pair saturates.  Producers flag anything above this limit
// (hos_set_range_flag) so the host can re-run the layer stack in exact fp32 MFMA mode.
#define HOS_RANGE_LIMIT 6.0e4f


// One element of the fused forward epilogue.  Returns false when the value went to `aux` instead of C.
__device__ __forceinline__ bool fwd_epilogue_value(const GemmArgs& a, float& v, int row, int col) {
    switch (a.epi) {
        case HOS_EPI_RELU: v = fmaxf(v, 0.f); break;
        case HOS_EPI_DENSITY: a.aux[row] = softplus_f(v + a.p0); return false;
        case HOS_EPI_RGB: v = sigmoid_f(v) * (1.f + 2.f * a.p0) - a.p0; break;
        case HOS_EPI_NERF_HEAD:
            if (col == a.aux_col) { a.aux[row] = softplus_f(v + a.p0); return false; }
            break;
        case HOS_EPI_SIGMOID_RELU4: v = (col < 3) ? sigmoid_f(v) : fmaxf(v, 0.f); break;
        case HOS_EPI_RESIDUAL: v += a.mask[(size_t)row * a.ldmask + col]; break;
        default: break;
    }
    return true;
}

// Exchange inside a quad of lanes (lane ^ 1, lane ^ 2) as v_mov_b32_dpp quad_perm -- a VALU move -- instead of __shfl_xor, which
// hipcc lowers to ds_bpermute_b32: 16 round trips through the LDS crossbar per 32 x 32 tile in the quad transposes below.
__device__ __forceinline__ float quad_xor1(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
}

// Epilogue of one 32x32 accumulator tile.
// The MFMA C/D layout gives a lane ONE column and 16 rows, i.e. 4-byte stores (16 store instructions of 2x128 B
// per tile) -- measured at ~1 TB/s, 45 % of the whole forward GEMM.  Each group of four accumulator registers is
// therefore transposed 4x4 across the four lanes of a quad (two xor-shuffle stages), after which a lane owns four
// CONSECUTIVE columns of one row: 16-byte stores, 4 instructions of 8x128 B per tile, and float4 bias / mask loads.
// bias_reg (FWD, optional): the four bias values of this lane's columns already in registers.  A persistent kernel passes
// them so that its epilogue issues NO global load: vmcnt retires in order, so a bias load issued behind the prefetch of a
// later tile waits for that whole prefetch (hos_thin.hip forward: 179 -> 1xx us per [262144,256,256] layer).
// relu_bits (FWD, optional): receives one bit per element this lane stores -- bit 15 - (4 g + k) = (row row0 + q + 8 g +
// 4 (lane>>5), column col0 + (lane & 28) + k) came out > 0 -- the ReLU mask of the backward pass in the ownership a lane has
// AFTER the quad transpose (hos_thin.hip keeps it as 2 bytes per lane and tile instead of re-reading the fp32 activations).
// Two VALU per element: the sign of 0 - v is shifted in from the right (v_sub, v_alignbit).
template <int MODE>
__device__ __forceinline__ void gemm_epilogue_tile(const GemmArgs& a, const f32x16& acc, int row0, int col0, int lane,
                                                   const float4* bias_reg = nullptr, uint32_t* relu_bits = nullptr) {
    const int l31 = lane & 31, lhi = lane >> 5;
    if constexpr (MODE == MODE_WGRAD) {
        const int col = col0 + l31;
        if (col >= a.N) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (row < a.M)
                __hip_atomic_fetch_add(a.C + (size_t)row * a.ldc + col, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    } else {
        const int q = l31 & 3;
        const int colb = col0 + (l31 & ~3);
        const bool c_vec = ((reinterpret_cast<uintptr_t>(a.C) & 15u) == 0) && ((a.ldc & 3) == 0);
        const bool m_vec = a.mask != nullptr && ((reinterpret_cast<uintptr_t>(a.mask) & 15u) == 0) && ((a.ldmask & 3) == 0);
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        bool big = false;
        if (MODE == MODE_FWD && bias_reg != nullptr) bias4 = *bias_reg;
        else if (MODE == MODE_FWD && a.bias != nullptr) {
            if (colb + 0 < a.N) bias4.x = a.bias[colb + 0];
            if (colb + 1 < a.N) bias4.y = a.bias[colb + 1];
            if (colb + 2 < a.N) bias4.z = a.bias[colb + 2];
            if (colb + 3 < a.N) bias4.w = a.bias[colb + 3];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = acc[4 * g + 0], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
            {   // 4x4 transpose inside the quad: afterwards (v0..v3) = row q, columns colb .. colb+3
                const float s0 = (q & 1) ? v0 : v1, s1 = (q & 1) ? v2 : v3;
                const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                if (q & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                const float t0 = (q & 2) ? v0 : v2, t1 = (q & 2) ? v1 : v3;
                const float u0 = quad_xor2(t0), u1 = quad_xor2(t1);
                if (q & 2) { v0 = u0; v1 = u1; } else { v2 = u0; v3 = u1; }
            }
            const int row = row0 + q + 8 * g + 4 * lhi;
            if (row >= a.M || colb >= a.N) {
                if (MODE == MODE_FWD && relu_bits != nullptr) *relu_bits <<= 4;
                continue;
            }
            float v[4] = {v0, v1, v2, v3};
            const bool full = colb + 3 < a.N;
            if constexpr (MODE == MODE_FWD) {
                v[0] += bias4.x; v[1] += bias4.y; v[2] += bias4.z; v[3] += bias4.w;
                const bool simple = (a.epi == HOS_EPI_NONE || a.epi == HOS_EPI_RELU);
                if (relu_bits != nullptr) {
                    uint32_t rb = *relu_bits;
#pragma unroll
                    for (int k = 0; k < 4; ++k) rb = __builtin_amdgcn_alignbit(rb, __float_as_uint(0.f - v[k]), 31);
                    *relu_bits = rb;
                }
                if (simple && full && c_vec) {
                    const float pre = (v[0] + v[1]) + (v[2] + v[3]);        // NaN iff a pre-activation is NaN: fmaxf below would hide it
                    if (a.epi == HOS_EPI_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    big |= (fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) > HOS_RANGE_LIMIT) | (pre != pre);
                    *reinterpret_cast<float4*>(a.C + (size_t)row * a.ldc + colb) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int col = colb + k;
                        if (col >= a.N) break;
                        float x = v[k];
                        if (fwd_epilogue_value(a, x, row, col)) a.C[(size_t)row * a.ldc + col] = x;
                        if (simple) big |= !(fabsf(x) <= HOS_RANGE_LIMIT);
                    }
                }
            } else {   // MODE_DGRAD
                if (a.mask != nullptr) {
                    if (full && m_vec) {
                        const float4 mk = *reinterpret_cast<const float4*>(a.mask + (size_t)row * a.ldmask + colb);
                        if (!(mk.x > 0.f)) v[0] = 0.f;
                        if (!(mk.y > 0.f)) v[1] = 0.f;
                        if (!(mk.z > 0.f)) v[2] = 0.f;
                        if (!(mk.w > 0.f)) v[3] = 0.f;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (colb + k < a.N && !(a.mask[(size_t)row * a.ldmask + colb + k] > 0.f)) v[k] = 0.f;
                    }
                }
                float* dst = a.C + (size_t)row * a.ldc + colb;
                if (full && c_vec) {
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (a.accumulate) { const float4 p = *reinterpret_cast<const float4*>(dst); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
                    *reinterpret_cast<float4*>(dst) = o;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (colb + k < a.N) dst[k] = a.accumulate ? (dst[k] + v[k]) : v[k];
                }
            }
        }
        if (MODE == MODE_FWD && a.range_flag != nullptr && __builtin_amdgcn_ballot_w64(big) != 0 && lane == 0) atomicOr(a.range_flag, 1u);
    }
}

#define HOS_GEMM_FP32 0
#define HOS_GEMM_BF16X3 1
// implemented in hos_gemm3.hip; fills tiles_m/tiles_n/kt_per_split itself
int hos_gemm3_launch(GemmArgs a, int mode, int splits, hipStream_t stream);
