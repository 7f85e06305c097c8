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
};


template <int MODE>
__device__ __forceinline__ void gemm_epilogue_tile(const GemmArgs& a, const f32x16& acc, int row0, int col0, int lane) {
    const int l31 = lane & 31, lhi = lane >> 5;
    const int col = col0 + l31;
    if (col >= a.N) return;
    float bcol = 0.f;
    if (MODE == MODE_FWD && a.bias != nullptr) bcol = a.bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row >= a.M) continue;
        float v = acc[r];
        if constexpr (MODE == MODE_FWD) {
            v += bcol;
            switch (a.epi) {
                case HOS_EPI_RELU: v = fmaxf(v, 0.f); break;
                case HOS_EPI_DENSITY: a.aux[row] = softplus_f(v + a.p0); continue;
                case HOS_EPI_RGB: v = sigmoid_f(v) * (1.f + 2.f * a.p0) - a.p0; break;
                case HOS_EPI_NERF_HEAD:
                    if (col == a.aux_col) { a.aux[row] = softplus_f(v + a.p0); continue; }
                    break;
                case HOS_EPI_SIGMOID_RELU4: v = (col < 3) ? sigmoid_f(v) : fmaxf(v, 0.f); break;
                case HOS_EPI_RESIDUAL: v += a.mask[(size_t)row * a.ldmask + col]; break;
                default: break;
            }
            a.C[(size_t)row * a.ldc + col] = v;
        } else if constexpr (MODE == MODE_DGRAD) {
            if (a.mask != nullptr && !(a.mask[(size_t)row * a.ldmask + col] > 0.f)) v = 0.f;
            float* dst = a.C + (size_t)row * a.ldc + col;
            *dst = a.accumulate ? (*dst + v) : v;
        } else {
            __hip_atomic_fetch_add(a.C + (size_t)row * a.ldc + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#define HOS_GEMM_FP32 0
#define HOS_GEMM_BF16X3 1
// implemented in hos_gemm3.hip; fills tiles_m/tiles_n/kt_per_split itself
int hos_gemm3_launch(GemmArgs a, int mode, int splits, hipStream_t stream);
