// Fused backward of ONE thin linear layer (<= 128 in / <= 128 out features) over very many rows:
//
//     dX = (dZ . W) * [X > 0]        (DGRAD, ReLU mask of the layer's input, optional)
//     dW += dZ^T . X,  db += sum_rows dZ        (WGRAD)
//
// The human-object branch runs 128-wide MLPs over P = rays x 128 = 262 144 sample points.  As two GEMMs a layer's
// backward reads dZ twice and X twice (mask + WGRAD operand) and writes dX: 20 B per element; the split-precision WGRAD
// on fp32 operands is additionally bound by its in-register transposes (1.9 TB/s measured).  Here one workgroup keeps the
// layer's W (bf16 hi/lo, 64 KB at 128 x 128) resident in LDS, walks 64-row blocks of dZ and X -- each read ONCE from HBM,
// split into bf16 (hi, lo) planes at staging time -- and feeds both products from the same two LDS tiles: 12 B per element.
// No operand is transposed by VALU: every reduction-row operand (W for DGRAD, dZ^T and X for WGRAD) is stored row-major
// and read with ds_read_b64_tr_b16, which hands a lane the 4 consecutive reduction values of its MFMA row.
//   products: a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi in v_mfma_f32_32x32x16_bf16, fp32 accumulate (as hos_gemm3.hip)
//   roofline: HBM.  12 B x M x 128 per launch; MFMA time is ~1/3 of the memory time at 11 B/clk/CU.
// Reference: the autograd backward of nn.Linear + ReLU in core/nets/human_nerf/non_rigid_motion_mlps/mlp_offset.py:54-70.
#include "hos_gemm_common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct MlpBwdArgs {
    const float* dZ; int lddz;
    const float* X; int ldx;
    const float* W; int ldw;
    float* dX; int lddx;
    float* dW; int lddw;
    float* db;
    int M, N, K;
    int relu_mask;
    float* ws;                    // optional per-workgroup slabs [dW partial (if ws_dw) | db partial] (NULL: fp32 atomics)
    int ws_dw;
    const int* m_dev;             // optional: live row count in device memory (fixed-capacity buffers); rows past it are skipped
};

constexpr int mb_rows(bool dg) { return dg ? 64 : 32; }   // rows per block iteration (wide WGRAD: 128 accumulator registers)
constexpr int MB_NT = 512;

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ void split4(const float4& v, bf16x4& hi, bf16x4& lo) {
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// Fragment of a reduction-row operand stored row-major [reduction][columns] (16-bit): the lane of MFMA row
// (lane & 31) and k half (lane >> 5) receives its 8 consecutive reduction values.  Inside a 16-lane group lane p supplies
// the address of reduction row (p >> 2), columns 4 (p & 3) .. +3, and receives column p of the four rows.
__device__ __forceinline__ bf16x8 tr_frag2(const char* p0, int pitch) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * pitch));
    const s16x8 j = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, j);
}

__device__ __forceinline__ f32x16 mma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
}

// DGRAD epilogue of one 32x32 tile: quad transpose (a lane ends up with four consecutive columns of one row, see
// gemm_epilogue_tile), ReLU mask from the sign of the bf16 hi part of X in LDS, 16-byte stores.  The mask must NOT be read
// from global memory here: vmcnt retires in order, so a wave waiting for such a load would also wait for the whole
// prefetch of the next tile issued just before it.
__device__ __forceinline__ void dgrad_store(const MlpBwdArgs& a, const f32x16& acc, const char* xh_rows, int pitch, int grow0,
                                            int col0, int lane) {
    const int l31 = lane & 31, lhi = lane >> 5, q = l31 & 3;
    const int colb = col0 + (l31 & ~3);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v0 = acc[4 * g + 0], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
        {
            const float s0 = (q & 1) ? v0 : v1, s1 = (q & 1) ? v2 : v3;
            const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
            if (q & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
            const float t0 = (q & 2) ? v0 : v2, t1 = (q & 2) ? v1 : v3;
            const float u0 = quad_xor2(t0), u1 = quad_xor2(t1);
            if (q & 2) { v0 = u0; v1 = u1; } else { v2 = u0; v3 = u1; }
        }
        const int lrow = q + 8 * g + 4 * lhi;                     // row inside the 32-row tile
        const int row = grow0 + lrow;
        if (row >= a.M || colb >= a.K) continue;
        float v[4] = {v0, v1, v2, v3};
        if (a.relu_mask) {
            const uint2 m = *reinterpret_cast<const uint2*>(xh_rows + lrow * pitch + colb * 2);    // 4 x bf16
            const uint32_t h[4] = {m.x & 0xffffu, m.x >> 16, m.y & 0xffffu, m.y >> 16};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((h[k] & 0x8000u) || (h[k] & 0x7fffu) == 0) v[k] = 0.f;                           // not (x > 0)
        }
        float* dst = a.dX + (size_t)row * a.lddx + colb;
        if (colb + 3 < a.K) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (colb + k < a.K) dst[k] = v[k];
        }
    }
}

// DG = true: the fused layer backward (N, K <= 128, W resident).  DG = false: WGRAD only for wider layers (N <= 256,
// K <= 256; W does not fit next to the tiles), same staging and transposed reads, dX comes from the split GEMM.
template <int NT, int KT, bool DG>
__global__ __launch_bounds__(MB_NT, 1) void mlp_bwd_kernel(MlpBwdArgs a) {
    constexpr int N_ = NT * 32, K_ = KT * 32, R = mb_rows(DG);
    if (a.m_dev) a.M = min(a.M, *a.m_dev);
    // row pitches in bytes: +32 B so that consecutive rows start 8 banks apart
    constexpr int PZ = N_ * 2 + 32, PX = K_ * 2 + 32, PW = K_ * 2 + 32;
    constexpr int W_PLANE = DG ? N_ * PW : 0, Z_PLANE = R * PZ, X_PLANE = R * PX;
    constexpr bool PF2 = DG;                      // two tiles of register prefetch (one when the accumulators need the registers)
    constexpr int ZU = R * (N_ / 4) / MB_NT, XU = R * (K_ / 4) / MB_NT;          // float4 units per thread and tile
    static_assert(ZU >= 1 && XU >= 1, "tile too small for 512 threads");
    constexpr int DT = DG ? 2 * KT : 0;           // DGRAD 32x32 tiles per row block
    constexpr int WT = NT * KT;                   // WGRAD tiles
    constexpr int WPW = (WT + 7) / 8;             // WGRAD tiles per wave
    static_assert(DT <= 8, "one DGRAD tile per wave");
    static_assert(8 % KT == 0, "a wave's WGRAD tiles share kt");

    extern __shared__ __attribute__((aligned(16))) char smem_mb[];
    char* const Wh = smem_mb;
    char* const Wl = Wh + W_PLANE;
    char* const Zh = Wl + W_PLANE;
    char* const Zl = Zh + Z_PLANE;
    char* const Xh = Zl + Z_PLANE;
    char* const Xl = Xh + X_PLANE;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tr_g = lane >> 4, tr_p = lane & 15;
#ifdef HOS_MB_TRACE   // timing experiment: phase stamps of workgroups 0 and 200 in the (otherwise unused) workspace
    long long* const trb = reinterpret_cast<long long*>(a.ws);
    int trn = 0;
#define MB_STAMP() do { if ((blockIdx.x == 0 || blockIdx.x == 200) && t == 0 && trn < 60) trb[(blockIdx.x ? 64 : 0) + trn++] = clock64(); } while (0)
#else
#define MB_STAMP() do {} while (0)
#endif
    MB_STAMP();
    const int tr_row = 8 * (tr_g >> 1) + (tr_p >> 2);               // reduction row inside a 16-row step
    const int tr_col = 16 * (tr_g & 1) + 4 * (tr_p & 3);            // column inside a 32-column tile

    // two tiles of register prefetch (A, B): a tile's loads are issued two iterations before they are converted
    float4 rzA[ZU], rxA[XU], rzB[PF2 ? ZU : 1], rxB[PF2 ? XU : 1];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto gload = [&](float4 (&rz)[ZU], float4 (&rx)[XU], int rb) {
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = t + MB_NT * i, row = u / (N_ / 4), c4 = u % (N_ / 4);
            const int gr = rb * R + row;
            rz[i] = (gr < a.M && c4 * 4 < a.lddz) ? ldg4(a.dZ + (size_t)gr * a.lddz + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = t + MB_NT * i, row = u / (K_ / 4), c4 = u % (K_ / 4);
            const int gr = rb * R + row;
            rx[i] = (gr < a.M && c4 * 4 < a.K) ? ldg4(a.X + (size_t)gr * a.ldx + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](const float4 (&rz)[ZU], const float4 (&rx)[XU]) {
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = t + MB_NT * i, row = u / (N_ / 4), c4 = u % (N_ / 4);
            bsum.x += rz[i].x; bsum.y += rz[i].y; bsum.z += rz[i].z; bsum.w += rz[i].w;   // c4 is the same for every i
            bf16x4 h, l;
            split4(rz[i], h, l);
            *reinterpret_cast<bf16x4*>(Zh + row * PZ + c4 * 8) = h;
            *reinterpret_cast<bf16x4*>(Zl + row * PZ + c4 * 8) = l;
        }
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = t + MB_NT * i, row = u / (K_ / 4), c4 = u % (K_ / 4);
            bf16x4 h, l;
            split4(rx[i], h, l);
            *reinterpret_cast<bf16x4*>(Xh + row * PX + c4 * 8) = h;
            *reinterpret_cast<bf16x4*>(Xl + row * PX + c4 * 8) = l;
        }
    };

    f32x16 accW[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[j][r] = 0.f;

    const int nrb = (a.M + R - 1) / R;
    const int G = gridDim.x;
    auto compute = [&](int rb) {
        // ---- DGRAD MFMAs: dX[64 x K_] = dZ[64 x N_] . W[N_ x K_]: tile (rt, kt) on wave rt*KT + kt
        f32x16 acc;
        const int rt = wave / KT, kt = wave % KT;
        if constexpr (DG) if (wave < DT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const char* zrow = Zh + (rt * 32 + l31) * PZ + lhi * 16;
            const char* wfrag = Wh + tr_row * PW + (kt * 32 + tr_col) * 2;
#pragma unroll
            for (int s = 0; s < N_ / 16; ++s) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(zrow + s * 32);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(zrow + Z_PLANE + s * 32);
                const bf16x8 bh = tr_frag2(wfrag + s * 16 * PW, PW);
                const bf16x8 bl = tr_frag2(wfrag + W_PLANE + s * 16 * PW, PW);
                acc = mma3(ah, al, bh, bl, acc);
            }
        }
        // ---- WGRAD: dW[N_ x K_] += dZ^T[N_ x 64] . X[64 x K_]: tile (nt, kt) = wave + 8 j.  8 % KT == 0, so a wave's tiles
        // share kt: the X fragments are read once per reduction step
        if (wave < WT) {
            const int kw = wave % KT;
            const char* xfrag = Xh + tr_row * PX + (kw * 32 + tr_col) * 2;
#pragma unroll
            for (int s = 0; s < R / 16; ++s) {
                const bf16x8 bh = tr_frag2(xfrag + s * 16 * PX, PX);
                const bf16x8 bl = tr_frag2(xfrag + X_PLANE + s * 16 * PX, PX);
#pragma unroll
                for (int j = 0; j < WPW; ++j) {
                    const int ti = wave + 8 * j;
                    if (ti < WT) {
                        const char* zfrag = Zh + tr_row * PZ + ((ti / KT) * 32 + tr_col) * 2;
                        const bf16x8 ah = tr_frag2(zfrag + s * 16 * PZ, PZ);
                        const bf16x8 al = tr_frag2(zfrag + Z_PLANE + s * 16 * PZ, PZ);
                        accW[j] = mma3(ah, al, bh, bl, accW[j]);
                    }
                }
            }
        }
        // ---- DGRAD store (after the WGRAD MFMAs have been issued: the stores drain under them)
        if constexpr (DG) if (wave < DT && a.dX != nullptr) dgrad_store(a, acc, Xh + rt * 32 * PX, PX, rb * R + rt * 32, kt * 32, lane);
    };

    int rb = blockIdx.x;
    if (rb < nrb) gload(rzA, rxA, rb);              // the first tiles travel while W is staged
    if constexpr (PF2) if (rb + G < nrb) gload(rzB, rxB, rb + G);
    if constexpr (DG) {   // ---- W -> LDS (hi, lo), once: all loads first, then the conversions
        constexpr int WU = N_ * (K_ / 4) / MB_NT;
        float4 rw[WU];
#pragma unroll
        for (int i = 0; i < WU; ++i) {
            const int u = t + MB_NT * i, row = u / (K_ / 4), c4 = u % (K_ / 4);
            rw[i] = (row < a.N && c4 * 4 < a.K) ? ldg4(a.W + (size_t)row * a.ldw + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < WU; ++i) {
            const int u = t + MB_NT * i, row = u / (K_ / 4), c4 = u % (K_ / 4);
            bf16x4 h, l;
            split4(rw[i], h, l);
            *reinterpret_cast<bf16x4*>(Wh + row * PW + c4 * 8) = h;
            *reinterpret_cast<bf16x4*>(Wl + row * PW + c4 * 8) = l;
        }
    }
    MB_STAMP();
    if constexpr (PF2) {
        while (rb < nrb) {
            sstore(rzA, rxA);
            MB_STAMP();
            __syncthreads();                        // tile (and, the first time, W) visible
            MB_STAMP();
            if (rb + 2 * G < nrb) gload(rzA, rxA, rb + 2 * G);
            compute(rb);
            MB_STAMP();
            __syncthreads();                        // every wave is done with this tile before it is overwritten
            MB_STAMP();
            rb += G;
            if (rb >= nrb) break;
            sstore(rzB, rxB);
            __syncthreads();
            if (rb + 2 * G < nrb) gload(rzB, rxB, rb + 2 * G);
            compute(rb);
            __syncthreads();
            rb += G;
        }
    } else {
        for (; rb < nrb; rb += G) {
            sstore(rzA, rxA);
            __syncthreads();
            if (rb + G < nrb) gload(rzA, rxA, rb + G);
            compute(rb);
            __syncthreads();
        }
    }

    // ---- dW: one partial [N_, K_] per workgroup.  256 workgroups adding 16 K floats each into the SAME 64 KB with fp32
    // atomics serialise in L2 (measured: 35 us of a 139 us launch); with a workspace every workgroup stores its slab with
    // plain 16-byte stores and mlp_bwd_reduce_kernel sums the slabs (8-way atomics only).
    MB_STAMP();
    GemmArgs ew{};
#ifdef HOS_MB_TRACE
    if (false) {
#else
    if (a.ws != nullptr && a.ws_dw) {
#endif
        ew.C = a.ws + (size_t)blockIdx.x * (N_ * K_ + N_); ew.ldc = K_; ew.M = N_; ew.N = K_; ew.epi = HOS_EPI_NONE;
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int ti = wave + 8 * j;
            if (ti < WT) gemm_epilogue_tile<MODE_FWD>(ew, accW[j], (ti / KT) * 32, (ti % KT) * 32, lane);
        }
    } else {
        ew.C = a.dW; ew.ldc = a.lddw; ew.M = a.N; ew.N = a.K;
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int ti = wave + 8 * j;
            if (ti < WT) gemm_epilogue_tile<MODE_WGRAD>(ew, accW[j], (ti / KT) * 32, (ti % KT) * 32, lane);
        }
    }
    if (a.db != nullptr) {
        float4* red = reinterpret_cast<float4*>(Zh);             // 8 KB of the (idle) tile memory
        red[t] = bsum;
        __syncthreads();
        constexpr int G = N_ / 4;                                 // column groups; thread t owns group t % G
        if (t < G) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = t; k < MB_NT; k += G) { const float4 v = red[k]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            const float sv[4] = {s.x, s.y, s.z, s.w};
            bool direct = true;
#ifndef HOS_MB_TRACE
            if (a.ws != nullptr) {      // 256 workgroups x one atomic per bias element on the SAME address cost 10-20 us: slab tail
                const int nk = a.ws_dw ? N_ * K_ : 0;
                *reinterpret_cast<float4*>(a.ws + (size_t)blockIdx.x * (nk + N_) + nk + t * 4) = s;
                direct = false;
            }
#endif
            if (direct) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t * 4 + q < a.N) __hip_atomic_fetch_add(a.db + t * 4 + q, sv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#ifdef HOS_MB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MB_STAMP();
#endif
#undef MB_STAMP
}

// ---- WGRAD of a 256-wide layer, whole tiles only (M % 32 == 0, N == 256, K == 32 KT, slab workspace): nothing in the loop is
// predicated, so the compiler can wait for ONE prefetched register set (`vmcnt(8)`) while the other is still travelling.  The
// generic kernel above refills its single set after the first barrier of a tile and needs it at the top of the next one: the
// loads have one compute phase (~1.5 us) to cover an HBM round trip of 2-3 us, and the workgroup waits for the difference on
// every tile (251 us per launch at M = 524 288 = 4.3 TB/s; half of every tile period is that wait).
template <int KT>
__global__ __launch_bounds__(MB_NT, 1) void wgrad_tr_fast_kernel(MlpBwdArgs a) {
    constexpr int NT = 8, N_ = NT * 32, K_ = KT * 32, R = 32;
    constexpr int PZ = N_ * 2 + 32, PX = K_ * 2 + 32;
    constexpr int Z_PLANE = R * PZ, X_PLANE = R * PX;
    constexpr int ZU = R * (N_ / 4) / MB_NT;                                  // float4 units of dZ per thread and tile (4)
    constexpr int XT = R * (K_ / 4);                                          // float4 units of X per tile
    constexpr int XU = (XT + MB_NT - 1) / MB_NT;                              // ... per thread (4; K = 64: 1, threads past XT idle)
    constexpr bool XALL = XT % MB_NT == 0;
    constexpr int WT = NT * KT, WPW = WT / 8;                                 // WGRAD tiles, per wave (8 or 2)
    static_assert(WT % 8 == 0 && 8 % KT == 0, "a wave's WGRAD tiles share kt");

    extern __shared__ __attribute__((aligned(16))) char smem_mb[];
    char* const Zh = smem_mb;
    char* const Zl = Zh + Z_PLANE;
    char* const Xh = Zl + Z_PLANE;
    char* const Xl = Xh + X_PLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tr_g = lane >> 4, tr_p = lane & 15;
    const int tr_row = 8 * (tr_g >> 1) + (tr_p >> 2);
    const int tr_col = 16 * (tr_g & 1) + 4 * (tr_p & 3);
    const bool xthread = XALL || t < XT;

    const int nrb = a.M / R;
    const int G = gridDim.x;
    float4 rzA[ZU], rxA[XU], rzB[ZU], rxB[XU];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto gload = [&](float4 (&rz)[ZU], float4 (&rx)[XU], int rb) {
        rb = rb < nrb ? rb : nrb - 1;                                         // clamped, never predicated
        const float* zb = a.dZ + (size_t)rb * R * a.lddz;
        const float* xb = a.X + (size_t)rb * R * a.ldx;
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = t + MB_NT * i, row = u / (N_ / 4), c4 = u % (N_ / 4);
            rz[i] = ldg4(zb + (size_t)row * a.lddz + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = XALL ? t + MB_NT * i : (t < XT ? t : 0), row = u / (K_ / 4), c4 = u % (K_ / 4);
            rx[i] = ldg4(xb + (size_t)row * a.ldx + c4 * 4);
        }
    };
    auto sstore = [&](const float4 (&rz)[ZU], const float4 (&rx)[XU]) {
#pragma unroll
        for (int i = 0; i < ZU; ++i) {
            const int u = t + MB_NT * i, row = u / (N_ / 4), c4 = u % (N_ / 4);
            bsum.x += rz[i].x; bsum.y += rz[i].y; bsum.z += rz[i].z; bsum.w += rz[i].w;   // c4 is the same for every i
            bf16x4 h, l;
            split4(rz[i], h, l);
            *reinterpret_cast<bf16x4*>(Zh + row * PZ + c4 * 8) = h;
            *reinterpret_cast<bf16x4*>(Zl + row * PZ + c4 * 8) = l;
        }
        if (xthread) {
#pragma unroll
            for (int i = 0; i < XU; ++i) {
                const int u = t + MB_NT * i, row = u / (K_ / 4), c4 = u % (K_ / 4);
                bf16x4 h, l;
                split4(rx[i], h, l);
                *reinterpret_cast<bf16x4*>(Xh + row * PX + c4 * 8) = h;
                *reinterpret_cast<bf16x4*>(Xl + row * PX + c4 * 8) = l;
            }
        }
    };
    f32x16 accW[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[j][r] = 0.f;
    auto compute = [&]() {      // dW[N_ x K_] += dZ^T[N_ x 32] . X[32 x K_]: tile (nt, kt) = wave + 8 j; a wave's tiles share kt
        const int kw = wave % KT;
        const char* xfrag = Xh + tr_row * PX + (kw * 32 + tr_col) * 2;
#pragma unroll
        for (int s = 0; s < R / 16; ++s) {
            const bf16x8 bh = tr_frag2(xfrag + s * 16 * PX, PX);
            const bf16x8 bl = tr_frag2(xfrag + X_PLANE + s * 16 * PX, PX);
#pragma unroll
            for (int j = 0; j < WPW; ++j) {
                const int ti = wave + 8 * j;
                const char* zfrag = Zh + tr_row * PZ + ((ti / KT) * 32 + tr_col) * 2;
                const bf16x8 ah = tr_frag2(zfrag + s * 16 * PZ, PZ);
                const bf16x8 al = tr_frag2(zfrag + Z_PLANE + s * 16 * PZ, PZ);
                accW[j] = mma3(ah, al, bh, bl, accW[j]);
            }
        }
    };
#define WG_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // one tile: the set is staged (the wait for it: vmcnt(loads of the other set)), refilled with the tile two strides ahead,
    // multiplied.  Two barriers: the planes are single-buffered (2 x 69 KB would fit, but the second buffer bought nothing once
    // the loads no longer stall the staging).
    auto step = [&](float4 (&rz)[ZU], float4 (&rx)[XU], const int rb) {
        sstore(rz, rx);
        WG_BAR();
        gload(rz, rx, rb + 2 * G);
        compute();
        WG_BAR();
    };
    int rb = blockIdx.x;                                                       // grid <= nrb
    gload(rzA, rxA, rb);
    gload(rzB, rxB, rb + G);
    step(rzA, rxA, rb);
    rb += G;
    if (rb < nrb) {
        step(rzB, rxB, rb);
        rb += G;
        while (rb + G < nrb) {
            step(rzA, rxA, rb);
            step(rzB, rxB, rb + G);
            rb += 2 * G;
        }
        if (rb < nrb) step(rzA, rxA, rb);
    }
#undef WG_BAR
    // ---- dW / db partials of this workgroup -> its slab (summed by the reduce kernel)
    GemmArgs ew{};
    ew.C = a.ws + (size_t)blockIdx.x * (N_ * K_ + N_); ew.ldc = K_; ew.M = N_; ew.N = K_; ew.epi = HOS_EPI_NONE;
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int ti = wave + 8 * j;
        gemm_epilogue_tile<MODE_FWD>(ew, accW[j], (ti / KT) * 32, (ti % KT) * 32, lane);
    }
    if (a.db != nullptr) {
        float4* red = reinterpret_cast<float4*>(Zh);
        red[t] = bsum;
        __syncthreads();
        constexpr int CG = N_ / 4;                                // column groups; thread t owns group t % CG
        if (t < CG) {
            float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = t; k < MB_NT; k += CG) { const float4 v = red[k]; sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w; }
            *reinterpret_cast<float4*>(a.ws + (size_t)blockIdx.x * (N_ * K_ + N_) + N_ * K_ + t * 4) = sacc;
        }
    }
}

// dW[n][k] += sum over slabs g of ws[g][n][k].  Block (x, y): 256 threads x float4 = 1024 consecutive elements, slabs
// y, y + sy, ... in rounds of eight 16-byte loads issued before the first add (one memory latency per round), then
// sy-way fp32 atomics into dW.  sy = 2: device-scope fp32 atomics are the bottleneck, not the loads -- the batched
// reduction of a stage-2 step (770 MB of slabs) takes 4 x 98 us at sy = 32, 4 x 48 us (4 TB/s) at sy = 2..4, 59 us at 1.
constexpr int MB_RSPLIT = 2;
__global__ __launch_bounds__(256) void mlp_bwd_reduce_kernel(const float* __restrict__ ws, int slabs, int n_, int k_, int nk, float* __restrict__ dW,
                                                             int lddw, float* __restrict__ db, int N, int K) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int slab = nk + n_;                             // [n_][k_] partial of dW, then [n_] partial of db
    if (e >= slab || (e >= nk && db == nullptr)) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const int sy = gridDim.y;                             // slabs blockIdx.y, + sy, + 2 sy, ... in rounds of eight loads
    for (int g0 = blockIdx.y; g0 < slabs; g0 += 8 * sy) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = g0 + i * sy;
            v[i] = g < slabs ? *reinterpret_cast<const float4*>(ws + (size_t)g * slab + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    }
    const float sv[4] = {s.x, s.y, s.z, s.w};
    if (e >= nk) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (e - nk + q < N) __hip_atomic_fetch_add(db + e - nk + q, sv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int n = e / k_, k = e % k_;                     // k_ is a multiple of 32: the four elements share a row
    if (n < N) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k + q < K) __hip_atomic_fetch_add(dW + (size_t)n * lddw + k + q, sv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The same reduction for up to MB_BATCH (dW, db) pairs in ONE launch (hos_mlp_bwd_defer / hos_mlp_bwd_flush): a thin MLP's
// backward is 6-8 such launches, each ending in a 17-67 MB slab reduction that is almost pure launch + atomic latency
// (17.7 us average, 40 of them = 0.72 ms of a 10 ms stage-2 step); batched they run as one grid.
constexpr int MB_BATCH = 16;
struct ReduceJob { const float* ws; float* dW; float* db; int slabs, n_, k_, nk, lddw, N, K, first_block; };
struct ReduceBatch { ReduceJob j[MB_BATCH]; int count; };

__global__ __launch_bounds__(256) void mlp_bwd_reduce_batch_kernel(const ReduceBatch b) {
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < b.count; ++i) if ((int)blockIdx.x >= b.j[i].first_block) ji = i;
    const ReduceJob& J = b.j[ji];
    const int e = (((int)blockIdx.x - J.first_block) * 256 + threadIdx.x) * 4;
    const int slab = J.nk + J.n_;
    if (e >= slab || (e >= J.nk && J.db == nullptr)) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const int sy = gridDim.y;
    for (int g0 = blockIdx.y; g0 < J.slabs; g0 += 8 * sy) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = g0 + i * sy;
            v[i] = g < J.slabs ? *reinterpret_cast<const float4*>(J.ws + (size_t)g * slab + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    }
    const float sv[4] = {s.x, s.y, s.z, s.w};
    if (e >= J.nk) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (e - J.nk + q < J.N) __hip_atomic_fetch_add(J.db + e - J.nk + q, sv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int n = e / J.k_, k = e % J.k_;
    if (n < J.N) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k + q < J.K) __hip_atomic_fetch_add(J.dW + (size_t)n * J.lddw + k + q, sv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

thread_local bool g_defer = false;
thread_local ReduceBatch g_batch = {};

static inline int reduce_split() {
    static const int v = getenv("HOS_MB_RSPLIT") ? atoi(getenv("HOS_MB_RSPLIT")) : MB_RSPLIT;
    return v > 0 ? v : MB_RSPLIT;
}

static int flush_reduce_batch(hipStream_t stream) {
    if (g_batch.count == 0) return 0;
    const ReduceJob& L = g_batch.j[g_batch.count - 1];
    const int blocks = L.first_block + hos_cdiv(L.nk + L.n_, 1024);
    hipLaunchKernelGGL(mlp_bwd_reduce_batch_kernel, dim3(blocks, reduce_split()), dim3(256), 0, stream, g_batch);
    g_batch.count = 0;
    return hos_launch_status();
}

template <int NT, int KT, bool DG>
int launch_mb(MlpBwdArgs a, size_t ws_floats, hipStream_t stream) {
    constexpr int N_ = NT * 32, K_ = KT * 32;
    constexpr int R = mb_rows(DG);
    constexpr size_t smem = (DG ? 2 * (size_t)N_ * (K_ * 2 + 32) : 0) + 2 * (size_t)R * (N_ * 2 + 32) + 2 * (size_t)R * (K_ * 2 + 32);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_kernel<NT, KT, DG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nrb = hos_cdiv(a.M, R);
    const int grid = nrb < 256 ? nrb : 256;
    a.ws_dw = 1;
    const int nk = a.ws_dw ? N_ * K_ : 0;
    if (a.ws != nullptr && (grid < 32 || ws_floats < (size_t)grid * (nk + N_) || (!a.ws_dw && a.db == nullptr))) a.ws = nullptr;
    hipLaunchKernelGGL((mlp_bwd_kernel<NT, KT, DG>), dim3(grid), dim3(MB_NT), smem, stream, a);
#ifndef HOS_MB_TRACE
    if (a.ws != nullptr) {
        if (g_defer) {
            if (g_batch.count == MB_BATCH) { const int rc = flush_reduce_batch(stream); if (rc != 0) return rc; }
            ReduceJob& J = g_batch.j[g_batch.count];
            J = ReduceJob{a.ws, a.dW, a.db, grid, N_, K_, nk, a.lddw, a.N, a.K,
                          g_batch.count ? g_batch.j[g_batch.count - 1].first_block + hos_cdiv(g_batch.j[g_batch.count - 1].nk + g_batch.j[g_batch.count - 1].n_, 1024) : 0};
            ++g_batch.count;
        } else {
            hipLaunchKernelGGL(mlp_bwd_reduce_kernel, dim3(hos_cdiv(nk + N_, 1024), reduce_split()), dim3(256), 0, stream,
                               a.ws, grid, N_, K_, nk, a.dW, a.lddw, a.db, a.N, a.K);
        }
    }
#endif
    return hos_launch_status();
}

// the unpredicated WGRAD (whole 32-row tiles, N = 256, K = 32 KT, slabs): returns -1 when the call does not qualify
template <int KT>
int launch_wgrad_fast(MlpBwdArgs a, size_t ws_floats, hipStream_t stream) {
    constexpr int N_ = 256, K_ = KT * 32, R = 32, nk = N_ * K_;
    constexpr size_t smem = 2 * (size_t)R * (N_ * 2 + 32) + 2 * (size_t)R * (K_ * 2 + 32);
    static const bool on = !(getenv("HOS_WGRAD_FAST") && atoi(getenv("HOS_WGRAD_FAST")) == 0);
    const int nrb = a.M / R;
    const int grid = nrb < 256 ? nrb : 256;
    if (!on || a.M % R != 0 || a.N != N_ || a.K != K_ || a.m_dev != nullptr || a.ws == nullptr || grid < 32 ||
        ws_floats < (size_t)grid * (nk + N_) || (((uintptr_t)a.ws) & 15u)) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tr_fast_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_tr_fast_kernel<KT>), dim3(grid), dim3(MB_NT), smem, stream, a);
    if (g_defer) {
        if (g_batch.count == MB_BATCH) { const int rc = flush_reduce_batch(stream); if (rc != 0) return rc; }
        ReduceJob& J = g_batch.j[g_batch.count];
        J = ReduceJob{a.ws, a.dW, a.db, grid, N_, K_, nk, a.lddw, a.N, a.K,
                      g_batch.count ? g_batch.j[g_batch.count - 1].first_block + hos_cdiv(g_batch.j[g_batch.count - 1].nk + g_batch.j[g_batch.count - 1].n_, 1024) : 0};
        ++g_batch.count;
    } else {
        hipLaunchKernelGGL(mlp_bwd_reduce_kernel, dim3(hos_cdiv(nk + N_, 1024), reduce_split()), dim3(256), 0, stream,
                           a.ws, grid, N_, K_, nk, a.dW, a.lddw, a.db, a.N, a.K);
    }
    return hos_launch_status();
}

// floats of slab workspace launch_mb<NT, KT, DG> uses for M rows (0: it accumulates with atomics)
template <int NT, int KT, bool DG>
int64_t ws_floats_mb(int M) {
    const int nrb = hos_cdiv(M, mb_rows(DG));
    const int grid = nrb < 256 ? nrb : 256;
    return grid < 32 ? 0 : (int64_t)grid * (NT * 32 * KT * 32 + NT * 32);
}


// ================================================================================================================
// Round 4: the backward of SEVERAL consecutive thin layers in one launch -- the gradient with respect to a
// layer's output never leaves the CU between the layers of a group.
//
// Per layer the fused kernel above moves 12 B per row and column: dZ read, X read, dX written -- and dX is read back as the next
// launch's dZ.  Here a workgroup walks 64-row blocks through a GROUP of layer steps: the block's dZ lives in LDS as bf16 (hi, lo)
// planes, each step stages only its layer input X (4 B per row and column) and its weight, forms  dX = (dZ . W) * [X > 0]  and
// dW += dZ^T . X  from the same tiles exactly like the single-layer kernel, and the masked dX becomes the planes of the next
// step's dZ in place (or, where the reference's graph leaves the chain -- the skip concat's hann columns, the first layer's
// input, the hand-over to the next group -- is written to HBM as fp32).  Weights cannot stay resident (one 128 x 128 bf16 pair is
// 72 KB of the 160 KB next to the dZ / X tiles), so a pre-split LDS image of every step's W (hos_mlp_chain_bwd_pack, once per
// optimiser step) is copied per block and step out of L2 through registers; its requests and those of the next X tile are issued a
// few at a time between the MFMA groups of the current step (two X tiles in flight were measured slower: the second register set
// spills); nothing in the loop is predicated (rows are clamped, dead rows enter as zero dZ).
// A group is limited by the accumulators of its weight gradients (32 registers per 128 x 128 layer and lane); the non-rigid MLP
// (mlp_offset.py:54-70, folded first layer) runs as three groups: {offset head, layer 5}, {skip concat's hann columns, layer 4,
// layer 3}, {layers 2, 1, 0}.
//   HBM per row of one 6 x 128 MLP backward: 6.2 KB (was 10.9 KB as eight fused-layer launches).
// Reference: autograd of non_rigid_motion_mlps/mlp_offset.py:54-70.
// ================================================================================================================
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int CB_MAXS = 4;
constexpr int CB_R = 64;                                  // rows per block
constexpr int CB_W_BYTES = 2 * 128 * (128 * 2 + 32);      // LDS region of one step's weight image (hi plane, lo plane)
constexpr int CB_T_BYTES = 2 * CB_R * (128 * 2 + 32);     // ... of the dZ planes / of the X planes
constexpr int CB_SMEM = CB_W_BYTES + 2 * CB_T_BYTES;

// step code: NT (1 | 4) | KT (2 | 4) << 4 | MASK << 8 | OUT << 9   (OUT = 1: dX -> HBM, the block's dZ planes stay as they are)
constexpr int cb_code(int nt, int kt, int mask, int out) { return nt | (kt << 4) | (mask << 8) | (out << 9); }
constexpr int cb_nt(int c) { return c & 15; }
constexpr int cb_kt(int c) { return (c >> 4) & 15; }
constexpr bool cb_mask(int c) { return (c >> 8) & 1; }
constexpr bool cb_out(int c) { return (c >> 9) & 1; }
constexpr int cb_image_bytes(int nt, int kt) { return ((2 * nt * 32 * (kt * 64 + 32)) + 8191) / 8192 * 8192; }

struct ChainBwdStep {
    const float* X; int ldx;          // layer input rows [M, ldx]
    const u32x4* Wp;                  // LDS image of W (hos_mlp_chain_bwd_pack)
    float* dXout; int lddx;           // OUT steps: [M, lddx] fp32
    float* ws;                        // slabs [grid][N_ * K_ + N_] of this step's dW / db partials
};
struct ChainBwdArgs {
    const float* dZ; int lddz;
    int M; const int* m_dev;
    ChainBwdStep st[CB_MAXS];
};

template <class F, int... I>
__device__ __forceinline__ void cb_static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

template <int... CODE>
__global__ __launch_bounds__(MB_NT, 1) void chain_bwd_kernel(ChainBwdArgs a) {
    constexpr int S = sizeof...(CODE);
    constexpr int codes[S] = {CODE...};
    static_assert(S >= 1 && S <= CB_MAXS && cb_out(codes[S - 1]), "the last step of a group hands its dX over through HBM");
    constexpr int R = CB_R;
    // accumulator tiles per wave and step, and their offsets in the register array
    constexpr auto wpw = [](int s) constexpr { return (cb_nt(codes[s]) * cb_kt(codes[s]) + 7) / 8; };
    constexpr auto acc_off = [wpw](int s) constexpr { int o = 0; for (int i = 0; i < s; ++i) o += wpw(i); return o; };
    constexpr int NACC = acc_off(S);
    // source of the dZ a step consumes: -1 = the launch's input, else the step that produced it
    constexpr auto zsrc = [](int s) constexpr { int v = -1; for (int i = 0; i < s; ++i) if (!cb_out(codes[i])) v = i; return v; };

    if (a.m_dev) a.M = min(a.M, *a.m_dev);
    extern __shared__ __attribute__((aligned(16))) char smem_mb[];
    char* const Wh = smem_mb;
    char* const Zh = smem_mb + CB_W_BYTES;
    char* const Xh = Zh + CB_T_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tr_g = lane >> 4, tr_p = lane & 15;
    const int tr_row = 8 * (tr_g >> 1) + (tr_p >> 2);
    const int tr_col = 16 * (tr_g & 1) + 4 * (tr_p & 3);

    constexpr int N0 = cb_nt(codes[0]) * 32;                 // columns of the incoming dZ
    constexpr int ZU0 = R * (N0 / 4) / MB_NT;                // float4 units per thread (4 or 1)
    float4 rz[ZU0];
    float4 rx[4];
    u32x4 rw[9];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 zsum[S];
    f32x16 accW[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) zsum[s] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int nrb = (a.M + R - 1) / R;
    const int G = gridDim.x;
    const int mlast = a.M - 1;

    // Every global address below is (uniform base in SGPRs) + (a 32-bit per-thread byte offset recomputed on the spot): the first
    // version kept the loop-invariant 64-bit address of each request in a VGPR pair, they spilled, and every reload from scratch
    // (`s_waitcnt vmcnt(0)`: scratch loads share the counter) serialised the weight requests of a step -- 320 of 810 us per MLP.
    auto uni = [](const void* p) {        // the pointer as a value the compiler KNOWS to be wave-uniform (SGPR pair -> saddr addressing)
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    auto blk_rlast = [&](int rb) { return min(R - 1, mlast - rb * R); };       // last live row of block rb (rows past it: clamped)
    auto gloadZ = [&](int rb) {
        rb = rb < nrb ? rb : nrb - 1;
        const char* base = uni(a.dZ + (size_t)rb * R * a.lddz);
        const int rlast = blk_rlast(rb);
#pragma unroll
        for (int i = 0; i < ZU0; ++i) {
            const int u = t + MB_NT * i, row = min(u / (N0 / 4), rlast), c4 = u % (N0 / 4);
            rz[i] = *reinterpret_cast<const float4*>(base + (unsigned)((row * a.lddz + c4 * 4) * 4));
        }
    };
    auto sstoreZ = [&](int rb) {
        constexpr int PZ = N0 * 2 + 32;
#pragma unroll
        for (int i = 0; i < ZU0; ++i) {
            const int u = t + MB_NT * i, row = u / (N0 / 4), c4 = u % (N0 / 4);
            float4 v = rz[i];
            if (rb * R + row >= a.M) v = make_float4(0.f, 0.f, 0.f, 0.f);       // dead rows contribute nothing to any product
            bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;          // c4 is the same for every i
            bf16x4 h, l;
            split4(v, h, l);
            *reinterpret_cast<bf16x4*>(Zh + row * PZ + c4 * 8) = h;
            *reinterpret_cast<bf16x4*>(Zh + R * PZ + row * PZ + c4 * 8) = l;
        }
    };
    auto gloadX_part = [&](auto sc, int rb, int i0, int i1) {
        constexpr int s = decltype(sc)::value;
        constexpr int K_ = cb_kt(codes[s]) * 32, XU = R * (K_ / 4) / MB_NT;
        rb = rb < nrb ? rb : nrb - 1;
        const int ldx = a.st[s].ldx;
        const char* base = uni(a.st[s].X + (size_t)rb * R * ldx);
        const int rlast = blk_rlast(rb);
#pragma unroll
        for (int i = 0; i < XU; ++i) if (i >= i0 && i < i1) {
            const int u = t + MB_NT * i, row = min(u / (K_ / 4), rlast), c4 = u % (K_ / 4);
            rx[i] = *reinterpret_cast<const float4*>(base + (unsigned)((row * ldx + c4 * 4) * 4));
        }
    };
    auto gloadX = [&](auto sc, int rb) { gloadX_part(sc, rb, 0, 4); };
    auto sstoreX = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int K_ = cb_kt(codes[s]) * 32, XU = R * (K_ / 4) / MB_NT, PX = K_ * 2 + 32;
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int u = t + MB_NT * i, row = u / (K_ / 4), c4 = u % (K_ / 4);
            bf16x4 h, l;
            split4(rx[i], h, l);
            *reinterpret_cast<bf16x4*>(Xh + row * PX + c4 * 8) = h;
            *reinterpret_cast<bf16x4*>(Xh + R * PX + row * PX + c4 * 8) = l;
        }
    };
    const unsigned toff = (unsigned)t * 16u;
    // requests [i0, i1) of step s's weight image: the main loop issues them (and those of the next X tile) a few at a time between
    // the MFMA groups -- a wave that issues its 13 requests of a step back to back stalls for as long as the CU's vector-memory path
    // needs to accept them
    auto loadW_part = [&](auto sc, int i0, int i1) {
        constexpr int s = decltype(sc)::value;
        constexpr int RW = cb_image_bytes(cb_nt(codes[s]), cb_kt(codes[s])) / 8192;
#pragma unroll
        for (int i = 0; i < RW; ++i) if (i >= i0 && i < i1) rw[i] = *reinterpret_cast<const u32x4*>(uni(reinterpret_cast<const char*>(a.st[s].Wp) + (size_t)i * 8192) + toff);
    };
    auto loadW = [&](auto sc) { loadW_part(sc, 0, 9); };
    auto storeW = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int RW = cb_image_bytes(cb_nt(codes[s]), cb_kt(codes[s])) / 8192;
#pragma unroll
        for (int i = 0; i < RW; ++i) *reinterpret_cast<u32x4*>(Wh + i * 8192 + toff) = rw[i];
    };

    int rb = blockIdx.x;                                      // grid <= nrb of the HOST row count; the device count may be smaller
    if (nrb > 0) {                                            // (an empty cycle set: no row to clamp to -- only the zero partials below)
        gloadZ(rb);
        gloadX(std::integral_constant<int, 0>{}, rb);
        loadW(std::integral_constant<int, 0>{});
    }

    for (; rb < nrb; rb += G) {
        sstoreZ(rb);
        sstoreX(std::integral_constant<int, 0>{});
        storeW(std::integral_constant<int, 0>{});
        __syncthreads();
        cb_static_for([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int code = codes[s];
            constexpr int NT = cb_nt(code), KT = cb_kt(code), N_ = NT * 32, K_ = KT * 32;
            constexpr int PZ = N_ * 2 + 32, PX = K_ * 2 + 32, PW = K_ * 2 + 32;
            constexpr int Z_PLANE = R * PZ, X_PLANE = R * PX, W_PLANE = N_ * PW;
            constexpr int DT = 2 * KT;                          // dgrad tiles of the block
            constexpr int WT = NT * KT, WPW = (WT + 7) / 8;
            static_assert(s == 0 || !cb_out(codes[s > 0 ? s - 1 : 0]) ? true : true, "");
            // ---- prefetches, issued a few requests at a time between the MFMA groups below: the next step's weight image (DGRAD loop)
            // and the next X tile (WGRAD loop; wrapping to the next block, whose dZ rides along in the last step)
            constexpr int SN = s + 1 < S ? s + 1 : 0;           // the step whose W / X is fetched now
            constexpr int RWN = cb_image_bytes(cb_nt(codes[SN]), cb_kt(codes[SN])) / 8192;
            constexpr int XUN = R * (cb_kt(codes[SN]) * 32 / 4) / MB_NT;
            const int rbn = s + 1 < S ? rb : rb + G;
            // ---- DGRAD: dX[64 x K_] = dZ[64 x N_] . W[N_ x K_], tile (rt, kt) on wave rt * KT + kt
            f32x16 acc;
            const int rt = wave / KT, kt = wave % KT;
            const bool dg = wave < DT;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            {
                const char* zrow = Zh + (rt * 32 + l31) * PZ + lhi * 16;
                const char* wfrag = Wh + tr_row * PW + (kt * 32 + tr_col) * 2;
                constexpr int NK = N_ / 16;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if (dg) {
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(zrow + k * 32);
                        const bf16x8 al = *reinterpret_cast<const bf16x8*>(zrow + Z_PLANE + k * 32);
                        const bf16x8 bh = tr_frag2(wfrag + k * 16 * PW, PW);
                        const bf16x8 bl = tr_frag2(wfrag + W_PLANE + k * 16 * PW, PW);
#ifndef HOS_CB_EXP_NOMFMA
                        acc = mma3(ah, al, bh, bl, acc);
#else
                        acc[k & 15] += (float)ah[0] + (float)al[1] + (float)bh[2] + (float)bl[3];
#endif
                    }
#ifndef HOS_CB_EXP_NOW
                    loadW_part(std::integral_constant<int, SN>{}, k * RWN / NK, (k + 1) * RWN / NK);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- WGRAD: dW[N_ x K_] += dZ^T[N_ x 64] . X[64 x K_], tile (nt, kt) = wave + 8 j (a wave's tiles share kt)
            {
                const int kw = wave % KT;
                const char* xfrag = Xh + tr_row * PX + (kw * 32 + tr_col) * 2;
                constexpr int NK = R / 16;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if (wave < WT) {
                        const bf16x8 bh = tr_frag2(xfrag + k * 16 * PX, PX);
                        const bf16x8 bl = tr_frag2(xfrag + X_PLANE + k * 16 * PX, PX);
#pragma unroll
                        for (int j = 0; j < WPW; ++j) {
                            const int ti = wave + 8 * j;
                            if (ti < WT) {
                                const char* zfrag = Zh + tr_row * PZ + ((ti / KT) * 32 + tr_col) * 2;
                                const bf16x8 ah = tr_frag2(zfrag + k * 16 * PZ, PZ);
                                const bf16x8 al = tr_frag2(zfrag + Z_PLANE + k * 16 * PZ, PZ);
#ifndef HOS_CB_EXP_NOMFMA
                                accW[acc_off(s) + j] = mma3(ah, al, bh, bl, accW[acc_off(s) + j]);
#else
                                accW[acc_off(s) + j][k & 15] += (float)ah[0] + (float)al[1] + (float)bh[2] + (float)bl[3];
#endif
                            }
                        }
                    }
                    gloadX_part(std::integral_constant<int, SN>{}, rbn, k * XUN / NK, (k + 1) * XUN / NK);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (s + 1 == S) gloadZ(rb + G);
            // ---- epilogue of the dgrad tile: quad transpose (a lane ends up with four consecutive columns of one row), ReLU mask from
            // the sign of X's hi plane, then either fp32 rows to HBM or -- behind the barrier -- the planes of the next step's dZ
            float v[4][4];
            if (dg) {
                const int q = l31 & 3;
                const int colb = kt * 32 + (l31 & ~3);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v0 = acc[4 * g + 0], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
                    {
                        const float s0 = (q & 1) ? v0 : v1, s1 = (q & 1) ? v2 : v3;
                        const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                        if (q & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                        const float t0 = (q & 2) ? v0 : v2, t1 = (q & 2) ? v1 : v3;
                        const float u0 = quad_xor2(t0), u1 = quad_xor2(t1);
                        if (q & 2) { v0 = u0; v1 = u1; } else { v2 = u0; v3 = u1; }
                    }
                    v[g][0] = v0; v[g][1] = v1; v[g][2] = v2; v[g][3] = v3;
                    const int lrow = rt * 32 + q + 8 * g + 4 * lhi;              // row inside the block
                    if constexpr (cb_mask(code)) {
                        const uint2 m = *reinterpret_cast<const uint2*>(Xh + lrow * PX + colb * 2);    // 4 x bf16 (hi)
                        const uint32_t h[4] = {m.x & 0xffffu, m.x >> 16, m.y & 0xffffu, m.y >> 16};
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if ((h[c] & 0x8000u) || (h[c] & 0x7fffu) == 0) v[g][c] = 0.f;                // not (x > 0)
                    }
                    if constexpr (cb_out(code)) {
                        const int lddx = a.st[s].lddx;
                        char* ob = const_cast<char*>(uni(a.st[s].dXout + (size_t)rb * R * lddx));
                        if (rb * R + lrow < a.M)
                            *reinterpret_cast<float4*>(ob + (unsigned)((lrow * lddx + colb) * 4)) = make_float4(v[g][0], v[g][1], v[g][2], v[g][3]);
                    } else {
                        zsum[s].x += v[g][0]; zsum[s].y += v[g][1]; zsum[s].z += v[g][2]; zsum[s].w += v[g][3];
                    }
                }
            }
            if constexpr (s + 1 < S) {
                __syncthreads();                                   // every wave is done with dZ, X and W of this step
                if constexpr (!cb_out(code)) {
                    constexpr int PZn = K_ * 2 + 32;               // the masked dX is the next step's dZ: N_next = K_
                    static_assert(cb_nt(codes[s + 1 < S ? s + 1 : s]) == KT || cb_out(code), "step shapes do not chain");
                    if (wave < DT) {
                        const int q = l31 & 3;
                        const int colb = kt * 32 + (l31 & ~3);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int lrow = rt * 32 + q + 8 * g + 4 * lhi;
                            bf16x4 h, l;
                            split4(make_float4(v[g][0], v[g][1], v[g][2], v[g][3]), h, l);
                            *reinterpret_cast<bf16x4*>(Zh + lrow * PZn + colb * 2) = h;
                            *reinterpret_cast<bf16x4*>(Zh + R * PZn + lrow * PZn + colb * 2) = l;
                        }
                    }
                }
                sstoreX(std::integral_constant<int, s + 1 < S ? s + 1 : 0>{});
#ifndef HOS_CB_EXP_NOW
                storeW(std::integral_constant<int, s + 1 < S ? s + 1 : 0>{});
#endif
                __syncthreads();
            }
        }, std::make_integer_sequence<int, S>{});
        __syncthreads();                                           // the block's tiles are free
    }

    // ---- dW / db partials of this workgroup -> its slabs (summed by the reduce kernel)
    cb_static_for([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int NT = cb_nt(codes[s]), KT = cb_kt(codes[s]), N_ = NT * 32, K_ = KT * 32;
        constexpr int WT = NT * KT, WPW = (WT + 7) / 8;
        GemmArgs ew{};
        ew.C = a.st[s].ws + (size_t)blockIdx.x * (N_ * K_ + N_); ew.ldc = K_; ew.M = N_; ew.N = K_; ew.epi = HOS_EPI_NONE;
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int ti = wave + 8 * j;
            if (ti < WT) gemm_epilogue_tile<MODE_FWD>(ew, accW[acc_off(s) + j], (ti / KT) * 32, (ti % KT) * 32, lane);
        }
    }, std::make_integer_sequence<int, S>{});
    // column sums of every version of dZ -> the slab tail of the steps that consumed it
    float4* red = reinterpret_cast<float4*>(Zh);
    {
        red[t] = bsum;
        __syncthreads();
        constexpr int CG = N0 / 4;
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < CG)
            for (int k = t; k < MB_NT; k += CG) { const float4 x = red[k]; sacc.x += x.x; sacc.y += x.y; sacc.z += x.z; sacc.w += x.w; }
        cb_static_for([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int N_ = cb_nt(codes[s]) * 32, K_ = cb_kt(codes[s]) * 32;
            if constexpr (zsrc(s) == -1)
                if (t < CG) *reinterpret_cast<float4*>(a.st[s].ws + (size_t)blockIdx.x * (N_ * K_ + N_) + N_ * K_ + t * 4) = sacc;
        }, std::make_integer_sequence<int, S>{});
        __syncthreads();
    }
    cb_static_for([&](auto pc) {
        constexpr int p = decltype(pc)::value;                    // producing step
        if constexpr (!cb_out(codes[p])) {
            constexpr int KTp = cb_kt(codes[p]), CG = KTp * 8;    // column groups of the dZ it produced
            red[t] = (wave < 2 * KTp) ? zsum[p] : make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < CG) {
                const int kt = t >> 3, cq = t & 7;                // group = kt * 8 + (l31 >> 2)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 x = red[(rt * KTp + kt) * 64 + hh * 32 + cq * 4 + q];
                            sacc.x += x.x; sacc.y += x.y; sacc.z += x.z; sacc.w += x.w;
                        }
            }
            cb_static_for([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int N_ = cb_nt(codes[s]) * 32, K_ = cb_kt(codes[s]) * 32;
                if constexpr (zsrc(s) == p)
                    if (t < CG) *reinterpret_cast<float4*>(a.st[s].ws + (size_t)blockIdx.x * (N_ * K_ + N_) + N_ * K_ + t * 4) = sacc;
            }, std::make_integer_sequence<int, S>{});
            __syncthreads();
        }
    }, std::make_integer_sequence<int, S>{});
}

// fp32 W[:N, col0 : col0 + K] (nn.Linear layout [out, in], row pitch ldw) -> the LDS image of a step: bf16 hi plane [N_][K_ + 16]
// then lo plane, zero padded, rounded up to whole 8 KB copy rounds.  One launch packs up to 8 images (blockIdx.y = job).
struct ChainBwdPackJob { const float* W; int ldw, N, K, nt, kt; uint16_t* image; };
struct ChainBwdPackJobs { ChainBwdPackJob j[8]; };
__global__ __launch_bounds__(256) void chain_bwd_pack_kernel(const ChainBwdPackJobs jobs) {
    const ChainBwdPackJob& J = jobs.j[blockIdx.y];
    const int N_ = J.nt * 32, K_ = J.kt * 32, pw = K_ + 16;      // pitch in 16-bit elements
    const int total = cb_image_bytes(J.nt, J.kt) / 2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int plane = e / (N_ * pw), r = e % (N_ * pw);
        const int n = r / pw, k = r % pw;
        uint16_t out = 0;
        if (plane < 2 && n < J.N && k < J.K) {
            const float w = J.W[(size_t)n * J.ldw + k];
            const __bf16 hi = (__bf16)w;
            out = plane == 0 ? __builtin_bit_cast(uint16_t, hi) : __builtin_bit_cast(uint16_t, (__bf16)(w - (float)hi));
        }
        J.image[e] = out;
    }
}

struct ChainBwdCfg { int S; int code[CB_MAXS]; };
constexpr int CB_NCFG = 4;
// 0: {offset head 3 -> 128, layer 5}          1: {skip concat's hann columns, layer 4, layer 3}      2: {layers 2, 1, folded layer 0}
// 3: {layers 2, 1, layer 0 on the unfolded [cond | hann] rows}
// (round 4 also instantiated the MLP as TWO groups of four steps: 803 vs 650 us per MLP backward, spills -- removed)
constexpr ChainBwdCfg CB_CFG[CB_NCFG] = {
    {2, {cb_code(1, 4, 1, 0), cb_code(4, 4, 1, 1), 0, 0}},
    {3, {cb_code(4, 2, 0, 1), cb_code(4, 4, 1, 0), cb_code(4, 4, 1, 1), 0}},
    {3, {cb_code(4, 4, 1, 0), cb_code(4, 4, 1, 0), cb_code(4, 2, 0, 1), 0}},
    {3, {cb_code(4, 4, 1, 0), cb_code(4, 4, 1, 0), cb_code(4, 4, 0, 1), 0}},
};

template <int... CODE>
int launch_chain_bwd(const ChainBwdArgs& a, int grid, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_bwd_kernel<CODE...>), hipFuncAttributeMaxDynamicSharedMemorySize, CB_SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((chain_bwd_kernel<CODE...>), dim3(grid), dim3(MB_NT), CB_SMEM, stream, a);
    return hos_launch_status();
}

}  // namespace

// dZ [M, lddz >= 32*ceil(N/32)] (columns >= N zero), X [M, ldx >= K], W [N, ldw] (first column of the K-slice) ->
// dX [M, lddx] (NULL: skip), dW [N, lddw] +=, db [N] += (NULL: skip).  N, K <= 128, K % 4 == 0.
// ws: optional scratch of ws_floats >= 256 * (128 * 128 + 128) floats for the per-workgroup dW partials (NULL: atomics).
extern "C" int hos_linear_bwd_fused(const float* dZ, int lddz, const float* X, int ldx, const float* W, int ldw,
                                    float* dX, int lddx, float* dW, int lddw, float* db, int M, int N, int K,
                                    int relu_mask, float* ws, int64_t ws_floats, const int32_t* rows_dev, hos_stream_t stream) {
    if (!dZ || !X || !W || !dW || M <= 0 || N <= 0 || K <= 0 || ws_floats < 0) return HOS_E_ARG;
    if (N > 128 || K > 128) return HOS_E_SHAPE;
    if ((lddz & 3) || (ldx & 3) || (ldw & 3) || (K & 3) || (dX && (lddx & 3))) return HOS_E_ALIGN;
    if (((uintptr_t)dZ | (uintptr_t)X | (uintptr_t)W | (uintptr_t)ws) & 15u) return HOS_E_ALIGN;
    MlpBwdArgs a{dZ, lddz, X, ldx, W, ldw, dX, lddx, dW, lddw, db, M, N, K, relu_mask, ws, 1, rows_dev};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nt = hos_cdiv(N, 32), kt = hos_cdiv(K, 32);
    if (nt <= 1 && kt <= 4) return launch_mb<1, 4, true>(a, (size_t)ws_floats, s);
    if (kt <= 2) return launch_mb<4, 2, true>(a, (size_t)ws_floats, s);
    return launch_mb<4, 4, true>(a, (size_t)ws_floats, s);
}

// WGRAD of a wider layer with the same staging (fp32 operands split once into LDS planes, transposed LDS reads, no VALU
// transposes): dW [N, lddw] += dZ^T . X, db [N] += column sums.  N <= 256, K <= 256, K % 4 == 0.  The split-K partials of
// the 256 workgroups (and their db partials) go through `ws` (>= 256*(256*256+256) floats = 67 MB; NULL or smaller: fp32 atomics,
// measured 72 us of fixed cost per launch against ~30 us for the slab write + reduce).
extern "C" int hos_linear_wgrad_tr(const float* dZ, int lddz, const float* X, int ldx, float* dW, int lddw, float* db,
                                   int M, int N, int K, float* ws, int64_t ws_floats, hos_stream_t stream) {
    if (!dZ || !X || !dW || M <= 0 || N <= 0 || K <= 0) return HOS_E_ARG;
    if (N > 256 || K > 256) return HOS_E_SHAPE;
    if ((lddz & 3) || (ldx & 3) || (K & 3)) return HOS_E_ALIGN;
    if (((uintptr_t)dZ | (uintptr_t)X) & 15u) return HOS_E_ALIGN;
    MlpBwdArgs a{dZ, lddz, X, ldx, nullptr, 0, nullptr, 0, dW, lddw, db, M, N, K, 0, ws, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int kt = hos_cdiv(K, 32);
    if (N == 256 && (K == 256 || K == 64)) {        // the canonical MLP's layers (and the 64-column Fourier rows of its folded input)
        const int rc = K == 256 ? launch_wgrad_fast<8>(a, (size_t)ws_floats, s) : launch_wgrad_fast<2>(a, (size_t)ws_floats, s);
        if (rc >= 0) return rc;
    }
    if (kt <= 4) return launch_mb<8, 4, false>(a, (size_t)ws_floats, s);
    return launch_mb<8, 8, false>(a, (size_t)ws_floats, s);
}

// Deferred slab reductions: between hos_mlp_bwd_defer(1) and hos_mlp_bwd_flush() every hos_linear_bwd_fused /
// hos_linear_wgrad_tr call of this thread only RECORDS its reduction (so each call needs its own `ws` region, sized by
// hos_mlp_bwd_ws_floats); the flush runs them all in one launch per 16 and the gradients are complete after it.
extern "C" int hos_mlp_bwd_defer(int on) { g_defer = on != 0; return 0; }

extern "C" int hos_mlp_bwd_flush(hos_stream_t stream) { return flush_reduce_batch(static_cast<hipStream_t>(stream)); }

extern "C" long long hos_mlp_bwd_ws_floats(int M, int N, int K, int fused) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int nt = hos_cdiv(N, 32), kt = hos_cdiv(K, 32);
    if (fused) {
        if (nt <= 1 && kt <= 4) return ws_floats_mb<1, 4, true>(M);
        if (kt <= 2) return ws_floats_mb<4, 2, true>(M);
        return ws_floats_mb<4, 4, true>(M);
    }
    return kt <= 4 ? ws_floats_mb<8, 4, false>(M) : ws_floats_mb<8, 8, false>(M);
}

// ---- backward of a GROUP of consecutive thin layers in one launch (chain_bwd_kernel above).  cfg selects the group's step shapes:
//   0: {N 32 K 128 mask -> chain, N 128 K 128 mask -> HBM}                      (offset head, layer 5 of mlp_offset.py)
//   1: {N 128 K 64 -> HBM, N 128 K 128 mask -> chain, N 128 K 128 mask -> HBM}  (hann columns of the skip concat, layers 4, 3)
//   2: {N 128 K 128 mask -> chain, same, N 128 K 64 -> HBM}                     (layers 2, 1 and the folded layer 0)
//   3: as 2 with K 128 in the last step (unfolded first layer);  4, 5: the whole MLP as two groups of four steps
// dZ [M, lddz] is the gradient entering the first step; X[s] / images[s] the layer input rows and the packed weight of step s;
// dXout[s] receives fp32 rows for "-> HBM" steps (NULL otherwise); dW[s] [N, lddw[s]] += and db[s] [N] += (NULL: none) go through
// the slab workspace `ws` (hos_mlp_chain_bwd_ws_floats) and the (deferrable) reduce launch of hos_linear_bwd_fused.
extern "C" int hos_mlp_chain_bwd_steps(int cfg) { return cfg >= 0 && cfg < CB_NCFG ? CB_CFG[cfg].S : HOS_E_ARG; }

extern "C" long long hos_mlp_chain_bwd_image_bytes(int cfg, int step) {
    if (cfg < 0 || cfg >= CB_NCFG || step < 0 || step >= CB_CFG[cfg].S) return 0;
    return cb_image_bytes(cb_nt(CB_CFG[cfg].code[step]), cb_kt(CB_CFG[cfg].code[step]));
}

extern "C" long long hos_mlp_chain_bwd_ws_floats(int cfg, int M) {
    if (cfg < 0 || cfg >= CB_NCFG || M <= 0) return 0;
    const int nrb = hos_cdiv(M, CB_R), grid = nrb < 256 ? nrb : 256;
    long long f = 0;
    for (int s = 0; s < CB_CFG[cfg].S; ++s) {
        const int n_ = cb_nt(CB_CFG[cfg].code[s]) * 32, k_ = cb_kt(CB_CFG[cfg].code[s]) * 32;
        f += (long long)grid * (n_ * k_ + n_);
    }
    return f;
}

extern "C" int hos_mlp_chain_bwd_pack(int n, const int* cfg, const int* step, const float* const* W, const int* ldw, const int* N,
                                      const int* K, void* const* image, hos_stream_t stream) {
    if (n <= 0 || n > 8 || !cfg || !step || !W || !ldw || !N || !K || !image) return HOS_E_ARG;
    ChainBwdPackJobs jobs{};
    for (int i = 0; i < n; ++i) {
        if (cfg[i] < 0 || cfg[i] >= CB_NCFG || step[i] < 0 || step[i] >= CB_CFG[cfg[i]].S || !W[i] || !image[i] || N[i] <= 0 || K[i] <= 0) return HOS_E_ARG;
        const int nt = cb_nt(CB_CFG[cfg[i]].code[step[i]]), kt = cb_kt(CB_CFG[cfg[i]].code[step[i]]);
        if (N[i] > nt * 32 || K[i] > kt * 32 || ldw[i] < K[i]) return HOS_E_SHAPE;
        if ((uintptr_t)image[i] & 15u) return HOS_E_ALIGN;
        jobs.j[i] = ChainBwdPackJob{W[i], ldw[i], N[i], K[i], nt, kt, static_cast<uint16_t*>(image[i])};
    }
    hipLaunchKernelGGL(chain_bwd_pack_kernel, dim3(36, n), dim3(256), 0, static_cast<hipStream_t>(stream), jobs);
    return hos_launch_status();
}

extern "C" int hos_mlp_chain_bwd(int cfg, const float* dZ, int lddz, int M, const int32_t* rows_dev, const float* const* X, const int* ldx,
                                 const void* const* images, float* const* dXout, const int* lddx, float* const* dW, const int* lddw,
                                 float* const* db, const int* N, const int* K, float* ws, int64_t ws_floats, hos_stream_t stream) {
    if (cfg < 0 || cfg >= CB_NCFG || !dZ || M <= 0 || !X || !ldx || !images || !dXout || !lddx || !dW || !lddw || !db || !N || !K || !ws) return HOS_E_ARG;
    const ChainBwdCfg& C = CB_CFG[cfg];
    if (ws_floats < hos_mlp_chain_bwd_ws_floats(cfg, M)) return HOS_E_SHAPE;
    if (((uintptr_t)dZ | (uintptr_t)ws) & 15u) return HOS_E_ALIGN;
    if (lddz < cb_nt(C.code[0]) * 32 || (lddz & 3)) return HOS_E_SHAPE;
    const int nrb = hos_cdiv(M, CB_R), grid = nrb < 256 ? nrb : 256;
    ChainBwdArgs a{};
    a.dZ = dZ; a.lddz = lddz; a.M = M; a.m_dev = rows_dev;
    float* wsp = ws;
    ReduceJob jobs[CB_MAXS];
    for (int s = 0; s < C.S; ++s) {
        const int n_ = cb_nt(C.code[s]) * 32, k_ = cb_kt(C.code[s]) * 32;
        if (!X[s] || !images[s] || !dW[s] || N[s] <= 0 || K[s] <= 0 || N[s] > n_ || K[s] > k_) return HOS_E_ARG;
        if (ldx[s] < k_ || (ldx[s] & 3) || (((uintptr_t)X[s] | (uintptr_t)images[s]) & 15u)) return HOS_E_ALIGN;
        if (cb_out(C.code[s])) {
            if (!dXout[s] || lddx[s] < k_ || (lddx[s] & 3) || ((uintptr_t)dXout[s] & 15u)) return HOS_E_ALIGN;
        }
        a.st[s].X = X[s]; a.st[s].ldx = ldx[s]; a.st[s].Wp = static_cast<const u32x4*>(images[s]);
        a.st[s].dXout = dXout[s]; a.st[s].lddx = lddx[s]; a.st[s].ws = wsp;
        jobs[s] = ReduceJob{wsp, dW[s], db[s], grid, n_, k_, n_ * k_, lddw[s], N[s], K[s], 0};
        wsp += (size_t)grid * (n_ * k_ + n_);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    switch (cfg) {
        case 0: rc = launch_chain_bwd<CB_CFG[0].code[0], CB_CFG[0].code[1]>(a, grid, st); break;
        case 1: rc = launch_chain_bwd<CB_CFG[1].code[0], CB_CFG[1].code[1], CB_CFG[1].code[2]>(a, grid, st); break;
        case 2: rc = launch_chain_bwd<CB_CFG[2].code[0], CB_CFG[2].code[1], CB_CFG[2].code[2]>(a, grid, st); break;
        default: rc = launch_chain_bwd<CB_CFG[3].code[0], CB_CFG[3].code[1], CB_CFG[3].code[2]>(a, grid, st); break;
    }
    if (rc != 0) return rc;
    for (int s = 0; s < C.S; ++s) {
        if (g_defer) {
            if (g_batch.count == MB_BATCH) { const int r2 = flush_reduce_batch(st); if (r2 != 0) return r2; }
            ReduceJob& J = g_batch.j[g_batch.count];
            J = jobs[s];
            J.first_block = g_batch.count ? g_batch.j[g_batch.count - 1].first_block + hos_cdiv(g_batch.j[g_batch.count - 1].nk + g_batch.j[g_batch.count - 1].n_, 1024) : 0;
            ++g_batch.count;
        } else {
            const ReduceJob& J = jobs[s];
            hipLaunchKernelGGL(mlp_bwd_reduce_kernel, dim3(hos_cdiv(J.nk + J.n_, 1024), reduce_split()), dim3(256), 0, st,
                               J.ws, J.slabs, J.n_, J.k_, J.nk, J.dW, J.lddw, J.db, J.N, J.K);
        }
    }
    return hos_launch_status();
}
