// "Planes" GEMM: operands pre-split into 16-bit hi/lo planes in HBM, staged by LDS-DMA, 3 MFMAs per product.
//
// Why: the phase timeline of the on-the-fly split kernel (hos_gemm3.hip, scripts/trace_gemm.py) showed that per
// K tile the eight global-load instructions take ~1000 cycles to get through the CU's 64 B/clk vector-memory path
// and the fp32 -> (hi,lo) conversion + LDS store 1300 (alone) .. 2900 (next to a partner wave's MFMAs) cycles,
// against 1536 cycles of MFMA work -- the matrix pipe sat at 28-44 %.  Here the split is done ONCE per element by
// the producer (GEMM epilogues, hos_split_planes for weights / boundary tensors), so a K tile is
//     8 x global_load_lds_dwordx4 per wave (no VGPRs, no VALU)  +  24 x ds_read_b128  +  48 x MFMA.
//
// One kernel, NT form for all three passes (both operands have the reduction index contiguous):
//   FWD    C[m][n]  = sum_k  X[m][k]    W[n][k]      X, W   : fp16 planes (22 mantissa bits -> fp32-grade)
//   DGRAD  dX[m][k] = sum_n  dZ[m][n]   Wt[k][n]     dZ, Wt : bf16 planes (8-bit exponent: gradients of any size)
//   WGRAD  dW[n][k] = sum_m  dZt[n][m]  Xt[k][m]     dZt,Xt : bf16 planes, transposed copies written by producers
// Element (r,c) of a split matrix is hi[r*ld+c] + lo[r*ld+c]; every reduction extent is a multiple of 32.
//
// Tile 256 x BN x 32, 512 threads (8 wave64 as 4x2), LDS = 2 stages x {A hi, A lo, B hi, B lo} x (rows x 64 B)
// = 128 KB at BN=256; 16-byte chunks XOR-swizzled by (row>>2)&3 exactly like hos_gemm3.hip.  Because
// global_load_lds writes LDS linearly (wave base + lane*16), the swizzle is applied on the per-lane SOURCE address.
// Wave pair p (waves 2p, 2p+1) owns plane p of the stage; each wave issues rows/32 DMA instructions per K tile,
// interleaved between the MFMAs of the current tile.  One barrier per K tile.
#include "hos_gemm_common.h"
#include <cstdlib>

namespace {

constexpr int PBM = 256;
constexpr int PBK = 32;
constexpr int PNT = 512;
constexpr int PROWB = 64;

template <typename E> struct PVec { typedef E x8 __attribute__((ext_vector_type(8))); typedef E x4 __attribute__((ext_vector_type(4))); };
__device__ __forceinline__ f32x16 pmfma(const PVec<__bf16>::x8& a, const PVec<__bf16>::x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 pmfma(const PVec<_Float16>::x8& a, const PVec<_Float16>::x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

enum PEpi { PEPI_F32 = 0, PEPI_PLANES_FWD = 1, PEPI_PLANES_DGRAD = 2, PEPI_WGRAD = 3 };

struct PArgs {
    // operands (16-bit planes); A may have a second K segment (skip concat)
    const uint16_t* Ahi; const uint16_t* Alo; int lda; int kt0;
    const uint16_t* A1hi; const uint16_t* A1lo; int lda1;
    const uint16_t* Bhi; const uint16_t* Blo; int ldb;
    int M, N;            // output extents (rows i of A, rows j of B)
    int nk, kt_per_split, tiles_m, tiles_n;
    // fp32 output path (reuses the fused epilogues of hos_gemm_common.h)
    GemmArgs f32;
    // plane outputs
    const float* bias;   // FWD
    int relu;            // FWD: apply ReLU
    const uint16_t* mask_hi; int ldmask;    // DGRAD: fp16 hi plane of the layer input; gradient passes where > 0
    uint16_t* Yhi; uint16_t* Ylo; int ldy;        // row-major planes [M][ldy]   (fp16 for FWD, bf16 for DGRAD)
    uint16_t* YThi; uint16_t* YTlo; int ldyt;     // transposed bf16 planes [N][ldyt] (optional)
    float out_scale;
    int ablate;          // debug (HOS_GEMM_ABLATE): 1 skip DMA, 4 skip MFMA, 32 skip LDS fragment reads
};

template <typename E> __device__ __forceinline__ uint16_t to_bits(E v) { return __builtin_bit_cast(uint16_t, v); }
template <typename E> __device__ __forceinline__ float clampE(float x) { return x; }
template <> __device__ __forceinline__ float clampE<_Float16>(float x) { return fminf(fmaxf(x, -65504.f), 65504.f); }

template <typename E>
__device__ __forceinline__ void split1(float x, uint16_t& hi, uint16_t& lo) {
    const E h = (E)clampE<E>(x);
    hi = to_bits<E>(h);
    lo = to_bits<E>((E)(x - (float)h));
}

__device__ __forceinline__ void dma16(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

template <int BN, int EPI, typename EIN>
__global__ __launch_bounds__(PNT, 2) void gemmp_kernel(const PArgs a) {
    typedef typename PVec<EIN>::x8 ex8;
    constexpr int WN = 2;
    constexpr int TM = PBM / (4 * 32);       // 2
    constexpr int TN = BN / (WN * 32);       // 4 or 2
    constexpr int A_PLANE = PBM * PROWB, B_PLANE = BN * PROWB;
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    constexpr int QA = PBM / 32, QB = BN / 32;      // DMA instructions per wave per K tile for an A / B plane
    constexpr int QMAX = QA > QB ? QA : QB;

    extern __shared__ __attribute__((aligned(16))) char smemp[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nb >> 3, r = nb & 7, x = bid & 7, y = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tn_i = bid % a.tiles_n;
    const int tm_i = (bid / a.tiles_n) % a.tiles_m;
    const int split = bid / (a.tiles_n * a.tiles_m);
    const int i0 = tm_i * PBM, j0 = tn_i * BN;
    const int kt_begin = split * a.kt_per_split;
    const int kt_end = min(a.nk, kt_begin + a.kt_per_split);
    if (kt_begin >= kt_end) return;

    // ---- DMA plan of this wave: plane (wave>>1), half (wave&1) of its rows --------------------------------
    const int plane = wave >> 1;                 // 0: A hi, 1: A lo, 2: B hi, 3: B lo
    const bool isB = plane >= 2;
    const int nq = isB ? QB : QA;                // instructions per K tile
    const int rows_half = (isB ? BN : PBM) / 2;
    const int row_l = (wave & 1) * rows_half + (lane >> 2);           // + 16*q
    const int cphys = lane & 3;
    const int clog = cphys ^ ((lane >> 4) & 3);                         // logical 16-byte k chunk this lane fetches
    const int lds_plane_off = isB ? (2 * A_PLANE + (plane - 2) * B_PLANE) : plane * A_PLANE;
    const int lds_row0 = (wave & 1) * rows_half;
    const int g_row0 = isB ? j0 : i0;
    const int g_limit = isB ? a.N : a.M;

    // Source planes of this wave, selected ONCE into scalars.  (Selecting them inside the DMA lambda made hipcc
    // build a pointer table in scratch; every scratch_load result was then waited for with vmcnt(0), which drained
    // the LDS-DMA queue before EACH global_load_lds and serialised the eight requests of a tile.)
    const uint16_t* P0; const uint16_t* P1; int ld0, ld1;
    if (plane == 0)      { P0 = a.Ahi; P1 = a.A1hi; ld0 = a.lda; ld1 = a.lda1; }
    else if (plane == 1) { P0 = a.Alo; P1 = a.A1lo; ld0 = a.lda; ld1 = a.lda1; }
    else if (plane == 2) { P0 = a.Bhi; P1 = a.Bhi;  ld0 = a.ldb; ld1 = a.ldb; }
    else                 { P0 = a.Blo; P1 = a.Blo;  ld0 = a.ldb; ld1 = a.ldb; }
    const int kt0 = isB ? 0x7fffffff : a.kt0;
    char* const lds_wave = smemp + lds_plane_off + lds_row0 * PROWB;

    auto issue_dma = [&](int q, int kt, int stage) {
        const bool seg1 = kt >= kt0;                                     // A may switch to its second K segment
        const uint16_t* P = seg1 ? P1 : P0;
        const int ld = seg1 ? ld1 : ld0;
        const int k0 = (seg1 ? kt - kt0 : kt) * PBK;
        int gr = g_row0 + row_l + 16 * q;
        gr = gr < g_limit ? gr : g_limit - 1;                          // clamp: out-of-range rows are never stored
        const unsigned off = (unsigned)gr * (unsigned)ld + (unsigned)(k0 + clog * 8);      // planes are < 2^32 elements
        dma16(P + off, lds_wave + stage * STAGE + q * 16 * PROWB);      // LDS address is wave-uniform
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    float dbsum = 0.f;
    const bool do_db = (EPI == PEPI_WGRAD) && a.f32.db != nullptr && tn_i == 0;

    // ---- main loop: software-pipelined over QUARTER tiles (k half s = 0/1  x  column half yh = 0/1) -----------
    // Two A fragment sets (one per k half) and two B fragment sets (one per quarter) rotate so that every
    // quarter's ds_reads fly under the previous quarter's MFMAs; two LDS stages; the DMA runs one tile ahead:
    //     read B1=(s0,yh1)              | MFMA (A0,B0)
    //     read A1=(s1), B0=(s1,yh0)     | MFMA (A0,B1)
    //     read B1=(s1,yh1)              | MFMA (A1,B0)
    //     wait DMA(kt+1) + own reads, barrier        <- one barrier per K tile
    //     DMA(kt+2) -> stage(kt) | read A0,B0 of kt+1 | MFMA (A1,B1)
    // All waits are explicit (counted asm s_waitcnt + raw s_barrier): __syncthreads() would drain the LDS-DMA
    // queue (vmcnt(0)) at every barrier, which serialises DMA and MFMA at one workgroup per CU.
    constexpr int TH = TN / 2;
    ex8 a0h[TM], a0l[TM], a1h[TM], a1l[TM], b0h[TH], b0l[TH], b1h[TH], b1l[TH];
    auto read_a = [&](const char* base, int s, ex8 (&ah)[TM], ex8 (&al)[TM]) {
        const int c = 2 * s + lhi;
#pragma unroll
        for (int x = 0; x < TM; ++x) {
            const int row = wm * (TM * 32) + x * 32 + l31;
            const int off = row * PROWB + ((c ^ ((row >> 2) & 3)) * 16);
            ah[x] = *reinterpret_cast<const ex8*>(base + off);
            al[x] = *reinterpret_cast<const ex8*>(base + A_PLANE + off);
        }
    };
    auto read_b = [&](const char* base, int s, int yh, ex8 (&bh)[TH], ex8 (&bl)[TH]) {
        const int c = 2 * s + lhi;
#pragma unroll
        for (int y = 0; y < TH; ++y) {
            const int row = wn * (TN * 32) + (yh * TH + y) * 32 + l31;
            const int off = row * PROWB + ((c ^ ((row >> 2) & 3)) * 16);
            bh[y] = *reinterpret_cast<const ex8*>(base + 2 * A_PLANE + off);
            bl[y] = *reinterpret_cast<const ex8*>(base + 2 * A_PLANE + B_PLANE + off);
        }
    };
#define HOS_MMA(AH, AL, BH, BL, YH)                                                      \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        __builtin_amdgcn_s_setprio(1);                                                   \
        if (!(a.ablate & 4)) {                                                           \
            _Pragma("unroll") for (int x = 0; x < TM; ++x)                               \
            _Pragma("unroll") for (int y = 0; y < TH; ++y) {                             \
                acc[x][(YH) * TH + y] = pmfma(AL[x], BH[y], acc[x][(YH) * TH + y]);      \
                acc[x][(YH) * TH + y] = pmfma(AH[x], BL[y], acc[x][(YH) * TH + y]);      \
                acc[x][(YH) * TH + y] = pmfma(AH[x], BH[y], acc[x][(YH) * TH + y]);      \
            }                                                                            \
        }                                                                                \
        __builtin_amdgcn_s_setprio(0);                                                   \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
    auto issue_tile = [&](int kt, int stage) {
        if (a.ablate & 1) return;
#pragma unroll
        for (int q = 0; q < QMAX; ++q) if (q < nq) issue_dma(q, kt, stage);
    };

    issue_tile(kt_begin, 0);
    if (kt_begin + 1 < kt_end) issue_tile(kt_begin + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    read_a(smemp, 0, a0h, a0l);
    read_b(smemp, 0, 0, b0h, b0l);

    int stage = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const char* base = smemp + stage * STAGE;
        read_b(base, 0, 1, b1h, b1l);
        if constexpr (EPI == PEPI_WGRAD) {
            if (do_db && t < PBM) {      // bias gradient: row sums of the dZt tile (hi + lo planes), 32 m values per K tile
                const int sw = (t >> 2) & 3;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const ex8 h = *reinterpret_cast<const ex8*>(base + t * PROWB + ((c ^ sw) * 16));
                    const ex8 l = *reinterpret_cast<const ex8*>(base + A_PLANE + t * PROWB + ((c ^ sw) * 16));
#pragma unroll
                    for (int e = 0; e < 8; ++e) dbsum += (float)h[e] + (float)l[e];
                }
            }
        }
        HOS_MMA(a0h, a0l, b0h, b0l, 0);
        read_a(base, 1, a1h, a1l);
        read_b(base, 1, 0, b0h, b0l);
        HOS_MMA(a0h, a0l, b1h, b1l, 1);
        read_b(base, 1, 1, b1h, b1l);
        HOS_MMA(a1h, a1l, b0h, b0l, 0);
        // this wave's share of tile kt+1 has landed and its reads of `stage` are complete; after the barrier that
        // holds for every wave, so `stage` may be refilled and stage^1 read
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (kt + 2 < kt_end) issue_tile(kt + 2, stage);
        if (kt + 1 < kt_end) {
            read_a(smemp + (stage ^ 1) * STAGE, 0, a0h, a0l);
            read_b(smemp + (stage ^ 1) * STAGE, 0, 0, b0h, b0l);
        }
        HOS_MMA(a1h, a1l, b1h, b1l, 1);
        stage ^= 1;
    }
#undef HOS_MMA

    // ---------------------------------------------------------------------------------------- epilogues
    if constexpr (EPI == PEPI_F32 || EPI == PEPI_WGRAD) {
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int y = 0; y < TN; ++y) {
                if constexpr (EPI == PEPI_WGRAD) {
                    f32x16 v = acc[x][y];
                    gemm_epilogue_tile<MODE_WGRAD>(a.f32, v, i0 + wm * (TM * 32) + x * 32, j0 + wn * (TN * 32) + y * 32, lane);
                } else {
                    gemm_epilogue_tile<MODE_FWD>(a.f32, acc[x][y], i0 + wm * (TM * 32) + x * 32, j0 + wn * (TN * 32) + y * 32, lane);
                }
            }
        if constexpr (EPI == PEPI_WGRAD) {
            if (do_db && t < PBM && i0 + t < a.M)
                __hip_atomic_fetch_add(a.f32.db + i0 + t, dbsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // plane outputs: EOUT = fp16 for FWD (next layer's input), bf16 for DGRAD (next dgrad/wgrad operand)
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int y = 0; y < TN; ++y) {
                const int row0 = i0 + wm * (TM * 32) + x * 32, col0 = j0 + wn * (TN * 32) + y * 32;
                const int colL = col0 + l31;                    // this lane's column in the MFMA layout
                float bcol = 0.f;
                if (EPI == PEPI_PLANES_FWD && a.bias != nullptr && colL < a.N) bcol = a.bias[colL];
                const int q = l31 & 3;
                const int colb = col0 + (l31 & ~3);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[x][y][4 * g + j] + bcol;
                        if (EPI == PEPI_PLANES_FWD && a.relu) v[j] = fmaxf(v[j], 0.f);
                    }
                    const int rbase = row0 + 8 * g + 4 * lhi;          // rows rbase .. rbase+3, column colL
                    if constexpr (EPI == PEPI_PLANES_DGRAD) {
                        if (a.mask_hi != nullptr && colL < a.N) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (rbase + j < a.M) {
                                    const uint16_t mh = a.mask_hi[(size_t)(rbase + j) * a.ldmask + colL];
                                    if ((mh & 0x8000u) || (mh & 0x7fffu) == 0) v[j] = 0.f;      // fp16 hi plane: x > 0 ?
                                }
                            }
                        }
                    }
                    // (a) transposed bf16 planes [n][m]: four consecutive m of one n = 8 bytes per plane
                    if (a.YThi != nullptr && colL < a.N && rbase < a.M) {
                        uint16_t h[4], l[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) split1<__bf16>(v[j], h[j], l[j]);
                        const size_t o = (size_t)colL * a.ldyt + rbase;
                        if (rbase + 3 < a.M) {
                            *reinterpret_cast<uint2*>(a.YThi + o) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                            *reinterpret_cast<uint2*>(a.YTlo + o) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
                        } else {
                            for (int j = 0; j < 4 && rbase + j < a.M; ++j) { a.YThi[o + j] = h[j]; a.YTlo[o + j] = l[j]; }
                        }
                    }
                    // (b) row-major planes: quad transpose so a lane owns four consecutive columns of one row
                    float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
                    {
                        const float s0 = (q & 1) ? v0 : v1, s1 = (q & 1) ? v2 : v3;
                        const float r0 = __shfl_xor(s0, 1, 64), r1 = __shfl_xor(s1, 1, 64);
                        if (q & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                        const float t0 = (q & 2) ? v0 : v2, t1 = (q & 2) ? v1 : v3;
                        const float u0 = __shfl_xor(t0, 2, 64), u1 = __shfl_xor(t1, 2, 64);
                        if (q & 2) { v0 = u0; v1 = u1; } else { v2 = u0; v3 = u1; }
                    }
                    const int row = row0 + q + 8 * g + 4 * lhi;
                    if (a.Yhi != nullptr && row < a.M && colb < a.ldy) {
                        const float w[4] = {v0, v1, v2, v3};
                        uint16_t h[4], l[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float val = (colb + k < a.N) ? w[k] : 0.f;        // zero the padding columns
                            if (EPI == PEPI_PLANES_FWD) split1<_Float16>(val, h[k], l[k]);
                            else split1<__bf16>(val, h[k], l[k]);
                        }
                        const size_t o = (size_t)row * a.ldy + colb;
                        *reinterpret_cast<uint2*>(a.Yhi + o) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                        *reinterpret_cast<uint2*>(a.Ylo + o) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
                    }
                }
            }
    }
}

template <int BN, int EPI, typename EIN>
int launchp(PArgs& a, int splits, hipStream_t stream) {
    constexpr size_t smem = 2 * (2 * PBM * PROWB + 2 * BN * PROWB);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemmp_kernel<BN, EPI, EIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    static const int ablate = getenv("HOS_GEMM_ABLATE") ? atoi(getenv("HOS_GEMM_ABLATE")) : 0;
    a.ablate = ablate;
    a.tiles_m = hos_cdiv(a.M, PBM);
    a.tiles_n = hos_cdiv(a.N, BN);
    if (EPI == PEPI_WGRAD) {
        static const int env_splits = getenv("HOS_WGRAD_SPLITS") ? atoi(getenv("HOS_WGRAD_SPLITS")) : 0;
        if (env_splits > 0) splits = env_splits;
        if (splits <= 0) {
            const int tiles = a.tiles_m * a.tiles_n;
            splits = hos_cdiv(512, tiles);
            if (splits > a.nk / 8) splits = a.nk / 8 > 0 ? a.nk / 8 : 1;
        }
        if (splits > a.nk) splits = a.nk;
        a.kt_per_split = hos_cdiv(a.nk, splits);
        splits = hos_cdiv(a.nk, a.kt_per_split);
    } else {
        splits = 1;
        a.kt_per_split = a.nk;
    }
    hipLaunchKernelGGL((gemmp_kernel<BN, EPI, EIN>), dim3(a.tiles_m * a.tiles_n * splits), dim3(PNT), smem, stream, a);
    return hos_launch_status();
}

// fp32 [R][ld] -> 16-bit hi/lo planes, row-major (padding columns [C, ldo) zeroed) and/or transposed [C][ldt]
template <typename E>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int lds, int R, int C,
                                                           uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int ldo,
                                                           uint16_t* __restrict__ hiT, uint16_t* __restrict__ loT, int ldt) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        const float v = (r < R && c < C) ? src[(size_t)r * lds + c] : 0.f;
        tile[ty + 8 * k][tx] = v;
        if (hi != nullptr && r < R && c < ldo) {
            uint16_t h, l;
            split1<E>(v, h, l);
            hi[(size_t)r * ldo + c] = h;
            lo[(size_t)r * ldo + c] = l;
        }
    }
    if (hiT == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;       // transposed: row index c, column r
        if (c < C && r < ldt) {
            uint16_t h, l;
            split1<E>(r < R ? tile[tx][ty + 8 * k] : 0.f, h, l);
            hiT[(size_t)c * ldt + r] = h;
            loT[(size_t)c * ldt + r] = l;
        }
    }
}

inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int hos_split_planes(const float* src, int lds, int R, int C, int dtype, void* hi, void* lo, int ldo,
                                void* hiT, void* loT, int ldt, hos_stream_t stream) {
    if (!src || R <= 0 || C <= 0 || (!hi && !hiT)) return HOS_E_ARG;
    if ((hi && !lo) || (hiT && !loT)) return HOS_E_ARG;
    const int cols = hi ? (ldo > C ? ldo : C) : C;
    dim3 grid(hos_cdiv(cols, 32), hos_cdiv(hiT ? (ldt > R ? ldt : R) : R, 32));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == 0)
        hipLaunchKernelGGL(split_planes_kernel<_Float16>, grid, dim3(256), 0, s, src, lds, R, C, (uint16_t*)hi, (uint16_t*)lo, ldo,
                           (uint16_t*)hiT, (uint16_t*)loT, ldt);
    else
        hipLaunchKernelGGL(split_planes_kernel<__bf16>, grid, dim3(256), 0, s, src, lds, R, C, (uint16_t*)hi, (uint16_t*)lo, ldo,
                           (uint16_t*)hiT, (uint16_t*)loT, ldt);
    return hos_launch_status();
}

extern "C" int hos_linearp_fwd(const void* Ahi, const void* Alo, int lda, int K0, const void* A1hi, const void* A1lo,
                               int lda1, int K1, const void* Whi, const void* Wlo, int ldw, const float* bias,
                               int M, int N, int relu, void* Yhi, void* Ylo, int ldy, void* YThi, void* YTlo, int ldyt,
                               float* C, int ldc, int epilogue, float* aux, int aux_col, float p0,
                               hos_stream_t stream) {
    if (!Ahi || !Alo || !Whi || !Wlo || M <= 0 || N <= 0 || K0 <= 0 || K1 < 0) return HOS_E_ARG;
    if (K1 > 0 && (!A1hi || !A1lo)) return HOS_E_ARG;
    if ((K0 % PBK) || (K1 % PBK)) return HOS_E_SHAPE;
    if ((lda & 7) || (ldw & 7) || (K1 > 0 && (lda1 & 7))) return HOS_E_ALIGN;
    if (!al16p(Ahi) || !al16p(Alo) || !al16p(Whi) || !al16p(Wlo)) return HOS_E_ALIGN;
    const bool planes_out = (Yhi != nullptr) || (YThi != nullptr);
    if (planes_out && ((Yhi && (!Ylo || (ldy & 3))) || (YThi && (!YTlo || (ldyt & 3))))) return HOS_E_ARG;
    if (!planes_out && !C && epilogue != HOS_EPI_DENSITY) return HOS_E_ARG;
    PArgs a{};
    a.Ahi = (const uint16_t*)Ahi; a.Alo = (const uint16_t*)Alo; a.lda = lda; a.kt0 = K0 / PBK;
    a.A1hi = (const uint16_t*)A1hi; a.A1lo = (const uint16_t*)A1lo; a.lda1 = lda1;
    a.Bhi = (const uint16_t*)Whi; a.Blo = (const uint16_t*)Wlo; a.ldb = ldw;
    a.M = M; a.N = N; a.nk = (K0 + K1) / PBK;
    a.bias = bias; a.relu = relu;
    a.Yhi = (uint16_t*)Yhi; a.Ylo = (uint16_t*)Ylo; a.ldy = ldy; a.YThi = (uint16_t*)YThi; a.YTlo = (uint16_t*)YTlo; a.ldyt = ldyt;
    a.f32.C = C; a.f32.ldc = ldc; a.f32.M = M; a.f32.N = N; a.f32.bias = bias; a.f32.aux = aux; a.f32.aux_col = aux_col;
    a.f32.p0 = p0; a.f32.epi = epilogue;
    if (epilogue == HOS_EPI_RESIDUAL) { a.f32.mask = aux; a.f32.ldmask = aux_col; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = N > 128;
    if (planes_out) return wide ? launchp<256, PEPI_PLANES_FWD, _Float16>(a, 1, s) : launchp<128, PEPI_PLANES_FWD, _Float16>(a, 1, s);
    return wide ? launchp<256, PEPI_F32, _Float16>(a, 1, s) : launchp<128, PEPI_F32, _Float16>(a, 1, s);
}

extern "C" int hos_linearp_dgrad(const void* dZhi, const void* dZlo, int lddz, const void* WThi, const void* WTlo,
                                 int ldwt, int Npad, const void* mask_hi, int ldmask, int M, int K,
                                 void* dXhi, void* dXlo, int lddx, void* dXThi, void* dXTlo, int lddxt,
                                 hos_stream_t stream) {
    if (!dZhi || !dZlo || !WThi || !WTlo || M <= 0 || K <= 0 || Npad <= 0) return HOS_E_ARG;
    if (!dXhi && !dXThi) return HOS_E_ARG;
    if (Npad % PBK) return HOS_E_SHAPE;
    if ((lddz & 7) || (ldwt & 7) || (dXhi && (lddx & 3)) || (dXThi && (lddxt & 3))) return HOS_E_ALIGN;
    PArgs a{};
    a.Ahi = (const uint16_t*)dZhi; a.Alo = (const uint16_t*)dZlo; a.lda = lddz; a.kt0 = Npad / PBK;
    a.Bhi = (const uint16_t*)WThi; a.Blo = (const uint16_t*)WTlo; a.ldb = ldwt;
    a.M = M; a.N = K; a.nk = Npad / PBK;
    a.mask_hi = (const uint16_t*)mask_hi; a.ldmask = ldmask;
    a.Yhi = (uint16_t*)dXhi; a.Ylo = (uint16_t*)dXlo; a.ldy = lddx; a.YThi = (uint16_t*)dXThi; a.YTlo = (uint16_t*)dXTlo; a.ldyt = lddxt;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return K > 128 ? launchp<256, PEPI_PLANES_DGRAD, __bf16>(a, 1, s) : launchp<128, PEPI_PLANES_DGRAD, __bf16>(a, 1, s);
}

extern "C" int hos_linearp_wgrad(const void* dZThi, const void* dZTlo, int lddzt, const void* XThi, const void* XTlo,
                                 int ldxt, float* dW, int ldw, float* db, int M, int N, int K, int splits,
                                 hos_stream_t stream) {
    if (!dZThi || !dZTlo || !XThi || !XTlo || !dW || M <= 0 || N <= 0 || K <= 0) return HOS_E_ARG;
    if (M % PBK) return HOS_E_SHAPE;
    if ((lddzt & 7) || (ldxt & 7)) return HOS_E_ALIGN;
    PArgs a{};
    a.Ahi = (const uint16_t*)dZThi; a.Alo = (const uint16_t*)dZTlo; a.lda = lddzt; a.kt0 = M / PBK;
    a.Bhi = (const uint16_t*)XThi; a.Blo = (const uint16_t*)XTlo; a.ldb = ldxt;
    a.M = N; a.N = K; a.nk = M / PBK;
    a.f32.C = dW; a.f32.ldc = ldw; a.f32.M = N; a.f32.N = K; a.f32.db = db;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return K > 128 ? launchp<256, PEPI_WGRAD, __bf16>(a, splits, s) : launchp<128, PEPI_WGRAD, __bf16>(a, splits, s);
}
