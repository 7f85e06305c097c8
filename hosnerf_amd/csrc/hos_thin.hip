// Thin-layer GEMMs with REGISTER-RESIDENT weights:   C[M, N] = epi(A[M, K] . B^T),  N, K <= 256,  M = rays x samples >> N.
//
// The 256-wide canonical MLP of the human branch runs at M = 262 144 rows.  A tiled GEMM re-reads the whole 256 x 256
// weight for every 256-row tile (as many bytes again as the activations), has only eight K tiles to hide its prologue and
// epilogue behind, and -- on fp32 operands -- spends half its issue slots converting: 217 us forward / 267 us DGRAD per
// layer = 2.5-3 TB/s.  Here a workgroup is persistent, each of its 8 waves owns 32 output columns and keeps ITS slice of
// the weight as ready-made MFMA B fragments in registers for the whole launch (16 reduction steps x (hi, lo) x 4 VGPRs =
// 128 VGPRs; 8 waves x 64 lanes x 512 B = the 256 KB that do not fit LDS), and the only thing that streams is the
// activation tile: read once from HBM, split into 16-bit (hi, lo) planes while it is staged to LDS (double buffered),
// read as A fragments by all eight waves.  HBM traffic = A in + C out (+ the ReLU mask source for DGRAD).
//   FWD  : B[n][k] = W[n][k]   (nn.Linear weight rows are k-contiguous: two 16-byte loads per fragment), fp16 hi/lo
//   DGRAD: B[k][n] = W[n][k]   (column gather, once per launch), bf16 hi/lo, ReLU mask of the layer input staged as bytes
//   products: a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate (hos_gemm3.hip)
//   roofline: HBM, 8 B (FWD) / 12 B (DGRAD) per row and column.
// Reference: CanonicalMLP.forward, core/nets/human_nerf/canonical_mlps/mlp_rgb_sigma.py:49-58, and its autograd.
#include "hos_gemm_common.h"
#include <cstdlib>
#include <type_traits>

#ifndef HOS_THIN_PF2_MAXKS
#define HOS_THIN_PF2_MAXKS 16        // fast forward kernel: two tiles of register prefetch up to this many reduction steps
#endif
#ifndef HOS_THIN_R_FWD
#define HOS_THIN_R_FWD 32          // rows per forward tile (64 measured 3-5 % slower once the epilogue stopped loading the bias)
#endif

namespace {

template <typename E> struct V8 { typedef E t __attribute__((ext_vector_type(8))); typedef E q __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ f32x16 mfma_e(const V8<__bf16>::t& a, const V8<__bf16>::t& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_e(const V8<_Float16>::t& a, const V8<_Float16>::t& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <typename E> __device__ __forceinline__ float hi_of(float x) { return x; }
// (one v_med3_f32; fminf(fmaxf()) costs a canonicalising v_max in front of it: 16 more VALU per staged tile and thread)
template <> __device__ __forceinline__ float hi_of<_Float16>(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }

template <typename E>
__device__ __forceinline__ void split_pair(float x, E& hi, E& lo) {
    hi = (E)hi_of<E>(x);
    lo = (E)(x - (float)hi);
}

struct ThinArgs {
    const float* A; int lda;          // activations (FWD: layer input; DGRAD: dZ)
    const float* W; int ldw;          // nn.Linear weight [Nout_fwd, ldw]
    const float* bias;                // FWD only (may be NULL)
    float* C; int ldc;
    int M, N, K;                      // C is [M, N]; reduction length K
    int epi;                          // FWD: HOS_EPI_NONE / HOS_EPI_RELU
    const float* mask; int ldmask;    // DGRAD: ReLU mask source [M, >= N] (NULL: none)
    unsigned int* range_flag;         // FWD: see HOS_RANGE_LIMIT
    uint16_t* bits;                   // ReLU bit mask [ceil(M/32)][8 waves][64 lanes] x 16 bits (FWD: written; DGRAD: read instead of
                                      // `mask`), bit 15 - (4 g + k) of a lane = the element it owns after the quad transpose (NULL: none)
};

constexpr int TH_NT = 512;
// Row padding of the (hi, lo) planes in LDS.  A fragment read is ds_read_b128 at row (lane & 31), 16-byte piece 2 s + (lane >> 5);
// gfx950 serves it in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...; MI355X_MICROARCH.md, LDS) over 64 banks.  With a pitch
// of 16 x (odd) bytes the 16 lanes of a group sit in 16 different 16-byte slots (conflict free); the round-2 padding of 32 B
// (slot = 2 row mod 16) makes every read a 2-way conflict.  Measured at [262144, 256, 256]: DGRAD 203 -> 201 us (164 -> 160 with
// mask bits) with 16 B; the forward kernels were FASTER with 32 B (138 vs 149 us: they are bound by MFMA + VALU issue of the
// SIMD's two waves, not by LDS cycles, and the staging writes of the next tile interleave differently) -- so each keeps its own.
#ifndef HOS_THIN_PAD_DGRAD
#define HOS_THIN_PAD_DGRAD 16
#endif
#ifndef HOS_THIN_PAD_FWD
#define HOS_THIN_PAD_FWD 32
#endif
constexpr int TH_PAD_FWD = HOS_THIN_PAD_FWD;
template <bool DGRAD> constexpr int th_pad() { return DGRAD ? HOS_THIN_PAD_DGRAD : HOS_THIN_PAD_FWD; }
constexpr int TH_MP = 264;     // LDS pitch of a mask row: 256 columns + the 16-byte group a window at an unaligned column spills into
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// KS: reduction steps of 16 (K <= 16 KS).  R: rows per tile (64 FWD, 32 DGRAD: the mask tile needs the registers).
template <int KS, bool DGRAD>
__global__ __launch_bounds__(TH_NT, 1) void thin_gemm_kernel(const ThinArgs a) {
    typedef typename std::conditional<DGRAD, __bf16, _Float16>::type E;
    typedef typename V8<E>::t e8;
    typedef typename V8<E>::q e4;
    constexpr int R = DGRAD ? 32 : HOS_THIN_R_FWD;
    constexpr int KD = KS * 16;
    constexpr int P = KD * 2 + th_pad<DGRAD>();          // LDS row pitch of one plane (bytes), see th_pad
    constexpr int PLANE = R * P, BUF = 2 * PLANE;        // hi, lo
    constexpr int MSK = DGRAD ? R * TH_MP : 0;           // mask bytes [R][TH_MP] per buffer
    constexpr int AU = R * (KD / 4) / TH_NT;             // float4 units of the A tile per thread
    constexpr int MU = DGRAD ? R * 64 / TH_NT : 1;       // float4 units of the mask tile per thread (256 columns)
    static_assert(AU >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem_th[];
    char* const buf0 = smem_th;
    char* const msk0 = smem_th + 2 * BUF;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int col0 = wave * 32;                          // this wave's output columns
#ifdef HOS_TH_TRACE   // timing experiment: phase stamps of workgroup 0 / wave 0 over the bias array (results invalid)
    long long* const trb = reinterpret_cast<long long*>(const_cast<float*>(a.bias));
    int trn = 0;
#define TH_STAMP() do { if (blockIdx.x == 0 && t == 0 && trn < 120) trb[trn++] = clock64(); } while (0)
#else
#define TH_STAMP() do {} while (0)
#endif
    TH_STAMP();

    // ---- this wave's slice of the weight as B fragments (hi, lo), once
    e8 bh[KS], bl[KS];
    {
        const int j = col0 + l31;                        // output column = B row
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float w[8];
            const int k0 = 16 * s + 8 * lhi;
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = 0.f;
            if (j < a.N) {
                if constexpr (!DGRAD) {
                    if (k0 + 7 < a.K) {
                        const float4 u = ld4(a.W + (size_t)j * a.ldw + k0), v = ld4(a.W + (size_t)j * a.ldw + k0 + 4);
                        w[0] = u.x; w[1] = u.y; w[2] = u.z; w[3] = u.w; w[4] = v.x; w[5] = v.y; w[6] = v.z; w[7] = v.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) if (k0 + q < a.K) w[q] = a.W[(size_t)j * a.ldw + k0 + q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (k0 + q < a.K) w[q] = a.W[(size_t)(k0 + q) * a.ldw + j];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { E h, l; split_pair<E>(w[q], h, l); bh[s][q] = h; bl[s][q] = l; }
        }
    }

    // ReLU mask window at an unaligned column (the h part of a skip layer's concat row starts at column 127): msh = its offset
    // inside a 16-byte group.  The loads stay ALIGNED on the enclosing groups (64 of them, +1 when msh != 0: `rmx`, by the
    // first R threads) -- dwordx4 loads at 4-byte aligned addresses run at about half the rate (measured: +120 us on a 142 us
    // launch) -- the byte tile in LDS keeps the group positions and the epilogue shifts by msh (v_alignbyte).
    const int msh = (int)((reinterpret_cast<uintptr_t>(a.mask) >> 2) & 3u);
    const float* const mbase = a.mask - msh;
    float4 ra[AU], rm[MU], rmx = make_float4(0.f, 0.f, 0.f, 0.f);
    // ReLU mask as bits (a.bits, written by the forward kernel of the layer below in THIS kernel's tile / wave / lane order):
    // 2 bytes per lane and tile travel with the prefetch instead of 32 x 256 x 4 B of activations through LDS
    uint32_t rbits = 0, bits_staged = 0, bits_cur = 0;
    auto gload = [&](int tile) {
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int u = t + TH_NT * i, row = u / (KD / 4), c4 = u % (KD / 4);
            const int gr = tile * R + row;
            ra[i] = (gr < a.M && c4 * 4 < a.K) ? ld4(a.A + (size_t)gr * a.lda + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (DGRAD) {
            if (a.bits != nullptr) rbits = a.bits[(size_t)tile * TH_NT + t];
            else if (a.mask != nullptr) {
#pragma unroll
                for (int i = 0; i < MU; ++i) {
                    const int u = t + TH_NT * i, row = u >> 6, c4 = u & 63;
                    const int gr = tile * R + row;
                    rm[i] = (gr < a.M && c4 * 4 < a.N + msh) ? ld4(mbase + (size_t)gr * a.ldmask + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (msh != 0 && t < R) {
                    const int gr = tile * R + t;
                    rmx = (gr < a.M && 256 < a.N + msh) ? ld4(mbase + (size_t)gr * a.ldmask + 256) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };
    auto sstore = [&](int b) {
        char* const hi = buf0 + b * BUF;
        char* const lo = hi + PLANE;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int u = t + TH_NT * i, row = u / (KD / 4), c4 = u % (KD / 4);
            e4 h, l;
            const float xs[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { E hh, ll; split_pair<E>(xs[q], hh, ll); h[q] = hh; l[q] = ll; }
            *reinterpret_cast<e4*>(hi + row * P + c4 * 8) = h;
            *reinterpret_cast<e4*>(lo + row * P + c4 * 8) = l;
        }
        if constexpr (DGRAD) {
            if (a.bits != nullptr) bits_staged = rbits;
            else if (a.mask != nullptr) {
#pragma unroll
                for (int i = 0; i < MU; ++i) {
                    const int u = t + TH_NT * i, row = u >> 6, c4 = u & 63;
                    const uint32_t m = (rm[i].x > 0.f ? 1u : 0u) | (rm[i].y > 0.f ? 0x100u : 0u) | (rm[i].z > 0.f ? 0x10000u : 0u) |
                                       (rm[i].w > 0.f ? 0x1000000u : 0u);
                    *reinterpret_cast<uint32_t*>(msk0 + b * MSK + row * TH_MP + c4 * 4) = m;
                }
                if (msh != 0 && t < R) {
                    const uint32_t m = (rmx.x > 0.f ? 1u : 0u) | (rmx.y > 0.f ? 0x100u : 0u) | (rmx.z > 0.f ? 0x10000u : 0u) |
                                       (rmx.w > 0.f ? 0x1000000u : 0u);
                    *reinterpret_cast<uint32_t*>(msk0 + b * MSK + t * TH_MP + 256) = m;
                }
            }
        }
    };

    GemmArgs ef{};                                       // FWD epilogue (bias, ReLU, 16-byte stores)
    ef.C = a.C; ef.ldc = a.ldc; ef.M = a.M; ef.N = a.N; ef.bias = a.bias; ef.epi = a.epi; ef.range_flag = a.range_flag;
    // this lane's four bias values, once: the epilogue of a tile must not issue global loads (they would queue behind the
    // prefetch of the tile after next and wait for all of it -- vmcnt retires in order)
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!DGRAD && a.bias != nullptr) {
        const int cb = col0 + (l31 & ~3);
        if (cb + 0 < a.N) bias4.x = a.bias[cb + 0];
        if (cb + 1 < a.N) bias4.y = a.bias[cb + 1];
        if (cb + 2 < a.N) bias4.z = a.bias[cb + 2];
        if (cb + 3 < a.N) bias4.w = a.bias[cb + 3];
    }

    const int ntiles = (a.M + R - 1) / R;
    const int G = gridDim.x;
    int tile = blockIdx.x;
    TH_STAMP();
    if (tile < ntiles) { gload(tile); sstore(0); }
    if (tile + G < ntiles) gload(tile + G);              // registers are free again: the second tile starts travelling
    __syncthreads();
    TH_STAMP();
    int b = 0;
    for (; tile < ntiles; tile += G, b ^= 1) {
        const bool more = tile + G < ntiles;
        TH_STAMP();
        if constexpr (DGRAD) bits_cur = bits_staged;      // this tile's ReLU bits (staged together with its operands)
        const char* hi = buf0 + b * BUF + l31 * P + lhi * 16;
        f32x16 acc[R / 32];
#pragma unroll
        for (int rt = 0; rt < R / 32; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int rt = 0; rt < R / 32; ++rt) {
                const e8 ah = *reinterpret_cast<const e8*>(hi + rt * 32 * P + s * 32);
                const e8 al = *reinterpret_cast<const e8*>(hi + PLANE + rt * 32 * P + s * 32);
                acc[rt] = mfma_e(al, bh[s], acc[rt]);
                acc[rt] = mfma_e(ah, bl[s], acc[rt]);
                acc[rt] = mfma_e(ah, bh[s], acc[rt]);
            }
        }
        TH_STAMP();
        // Stage the next tile BEFORE this tile's stores: vmcnt retires in order, so converting the prefetched registers
        // after the epilogue would first wait for every store just issued (a full HBM write round trip per tile).
        if (more) sstore(b ^ 1);
        // ... and the staging registers are free: the tile after next travels during this tile's stores, the barrier and the
        // whole next compute phase (a load issued at the top of an iteration had ~1 k cycles of cover, measured 10 k waiting)
        if (tile + 2 * G < ntiles) gload(tile + 2 * G);
        TH_STAMP();
        if (col0 < a.N) {
#pragma unroll
            for (int rt = 0; rt < R / 32; ++rt) {
                const int row0 = tile * R + rt * 32;
                if constexpr (!DGRAD) {
                    uint32_t rb = 0;       // ReLU bits of this lane's 16 outputs (one copy of the epilogue code: the bits always
                    gemm_epilogue_tile<MODE_FWD>(ef, acc[rt], row0, col0, lane, &bias4, &rb);      // form, the store is optional)
                    if (a.bits != nullptr) a.bits[((size_t)tile * (R / 32) + rt) * TH_NT + t] = (uint16_t)rb;
                } else {
                    // quad transpose -> a lane owns four consecutive columns of one row; mask bytes from LDS; 16-byte stores
                    const int q = l31 & 3, colb = col0 + (l31 & ~3);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v0 = acc[rt][4 * g + 0], v1 = acc[rt][4 * g + 1], v2 = acc[rt][4 * g + 2], v3 = acc[rt][4 * g + 3];
                        {
                            const float s0 = (q & 1) ? v0 : v1, s1 = (q & 1) ? v2 : v3;
                            const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                            if (q & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                            const float t0 = (q & 2) ? v0 : v2, t1 = (q & 2) ? v1 : v3;
                            const float u0 = quad_xor2(t0), u1 = quad_xor2(t1);
                            if (q & 2) { v0 = u0; v1 = u1; } else { v2 = u0; v3 = u1; }
                        }
                        const int lrow = q + 8 * g + 4 * lhi, row = row0 + lrow;
                        if (row >= a.M || colb >= a.N) continue;
                        float v[4] = {v0, v1, v2, v3};
                        if (a.bits != nullptr) {
#pragma unroll
                            for (int k = 0; k < 4; ++k)       // bit -> all-ones / zero (v_bfe_i32), AND: 2 VALU per element
                                v[k] = __uint_as_float(__float_as_uint(v[k]) & (uint32_t)__builtin_amdgcn_sbfe((int)bits_cur, 15 - (4 * g + k), 1));
                        } else if (a.mask != nullptr) {
                            const char* const mp = msk0 + b * MSK + (rt * 32 + lrow) * TH_MP + colb;
                            uint32_t m = *reinterpret_cast<const uint32_t*>(mp);
                            if (msh != 0) m = __builtin_amdgcn_alignbyte(*reinterpret_cast<const uint32_t*>(mp + 4), m, (uint32_t)msh);
#pragma unroll
                            for (int k = 0; k < 4; ++k) if (!((m >> (8 * k)) & 1u)) v[k] = 0.f;
                        }
                        float* dst = a.C + (size_t)row * a.ldc + colb;
                        if (colb + 3 < a.N && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) if (colb + k < a.N) dst[k] = v[k];
                        }
                    }
                }
            }
        }
        TH_STAMP();
        // next buffer complete; this one free for the tile after next.  (hipcc compiles __syncthreads() to `s_waitcnt
        // lgkmcnt(0); s_barrier` on this target: waves of a workgroup share the CU, so the fence does not wait for global
        // memory operations.  What the waves wait for at this barrier is each other: HOS_TH_TRACE.)
        __syncthreads();
    }
    TH_STAMP();
#undef TH_STAMP
}

// ---- FWD fast path: whole tiles only (M % 32 == 0, K == 16 KS, N == 256, 16-byte aligned rows of C).
// The generic kernel above is LATENCY-bound (HOS_TH_TRACE: per 32-row tile a wave computes for ~2.9 k cycles and waits ~4.4 k --
// 2.0 k at the `vmcnt(0)` in front of the staging and 2.4 k at the barrier): every global load and store of its loop is
// predicated, so the compiler has no lower bound on how many younger operations follow a prefetch and must wait for ALL of
// them (vmcnt retires in order), i.e. also for the stores just issued and for any deeper prefetch.  One tile of loads plus one
// tile of stores in flight per CU is ~1/3 of the bandwidth-delay product, hence 3.3-3.8 TB/s.  Here nothing in the loop is
// conditional: loads are clamped instead of predicated, every tile issues exactly 4 row stores (+1 bit-mask store) per lane,
// so the compiler's own counter can wait for the prefetched set alone (`vmcnt(N > 0)`) with TWO further tiles and both
// epilogues' stores still in flight; the barrier is written out as `s_waitcnt lgkmcnt(0); s_barrier`; the quad transpose of the epilogue uses DPP
// (v_mov_b32_dpp quad_perm) instead of 16 ds_bpermute per tile; the range flag is one atomic per wave at the end.
template <int KS, bool BITS>
__global__ __launch_bounds__(TH_NT, 1) void thin_fwd_fast_kernel(const ThinArgs a) {
    typedef _Float16 E;
    typedef typename V8<E>::t e8;
    typedef typename V8<E>::q e4;
    constexpr int R = 32;
    constexpr int KD = KS * 16;
    constexpr int P = KD * 2 + TH_PAD_FWD;
    constexpr int PLANE = R * P, BUF = 2 * PLANE;
    constexpr int AU = R * (KD / 4) / TH_NT;             // float4 units of a tile per thread
    constexpr bool PF2 = KS <= HOS_THIN_PF2_MAXKS;       // two tiles of register prefetch (20 steps: the registers hold weights)
    constexpr int PFD = PF2 ? 3 : 2;
    static_assert(AU >= 1 && R * (KD / 4) % TH_NT == 0, "tile does not divide over 512 threads");

    extern __shared__ __attribute__((aligned(16))) char smem_th[];
    char* const buf0 = smem_th;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int col0 = wave * 32;

    e8 bh[KS], bl[KS];
    {
        const float* wrow = a.W + (size_t)(col0 + l31) * a.ldw + 8 * lhi;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 u = ld4(wrow + 16 * s), v = ld4(wrow + 16 * s + 4);
            const float w[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) { E h, l; split_pair<E>(w[q], h, l); bh[s][q] = h; bl[s][q] = l; }
        }
    }
    const int q4 = l31 & 3, colb = col0 + (l31 & ~3);
    const float4 bias4 = a.bias != nullptr ? ld4(a.bias + colb) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool relu = a.epi == HOS_EPI_RELU;

    const int ntiles = a.M / R;
    const int G = gridDim.x;
    float4 ra[AU], rb[PF2 ? AU : 1];
    // this thread's float4 units of a tile: unit u = t + 512 i -> (row, 16-byte column group)
    auto gload = [&](float4 (&r)[AU], int tile) {
        tile = tile < ntiles ? tile : ntiles - 1;        // clamped, never predicated (a tile past the end is loaded, not staged)
#ifdef HOS_EXP_SAMEA       // timing experiment: every tile is read from the workgroup's FIRST tile (cache hits; results invalid)
        tile = blockIdx.x;
#endif
        const float* base = a.A + (size_t)tile * R * a.lda;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int u = t + TH_NT * i, row = u / (KD / 4), c4 = u % (KD / 4);
            r[i] = ld4(base + (size_t)row * a.lda + c4 * 4);
        }
    };
    auto sstore = [&](const float4 (&r)[AU], int b) {
        char* const hi = buf0 + b * BUF;
        char* const lo = hi + PLANE;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int u = t + TH_NT * i, row = u / (KD / 4), c4 = u % (KD / 4);
            e4 h, l;
            const float xs[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { E hh, ll; split_pair<E>(xs[q], hh, ll); h[q] = hh; l[q] = ll; }
            *reinterpret_cast<e4*>(hi + row * P + c4 * 8) = h;
            *reinterpret_cast<e4*>(lo + row * P + c4 * 8) = l;
        }
    };

    int tile = blockIdx.x;                               // grid <= ntiles
    gload(ra, tile);
    sstore(ra, 0);
    gload(ra, tile + G);
    if constexpr (PF2) gload(rb, tile + 2 * G);
    __syncthreads();
    bool big = false;
    // one tile: MFMAs on buffer b; `rcur` (the tile one grid stride ahead) is staged into the other buffer and refilled with the
    // tile PFD strides ahead; epilogue.  `more`: a further tile follows (compile-time in the loop, so that no load or store of
    // the steady state sits behind a branch)
    auto body = [&](float4 (&rcur)[AU], const int tile, const int b, const bool more) {
        const char* hi = buf0 + b * BUF + l31 * P + lhi * 16;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const e8 ah = *reinterpret_cast<const e8*>(hi + s * 32);
            const e8 al = *reinterpret_cast<const e8*>(hi + PLANE + s * 32);
            acc = mfma_e(al, bh[s], acc);
            acc = mfma_e(ah, bl[s], acc);
            acc = mfma_e(ah, bh[s], acc);
        }
        if (more) sstore(rcur, b ^ 1);                  // the wait for `rcur` sits here: vmcnt(number of younger operations)
        gload(rcur, tile + PFD * G);
        // epilogue: bias, ReLU bits, ReLU, 4 x 16-byte stores -- unconditional
        uint32_t rbits = 0;
        float* crow = a.C + (size_t)(tile * R + q4 + 4 * lhi) * a.ldc + colb;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = acc[4 * g + 0], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
            {   // 4x4 transpose inside the quad (as gemm_epilogue_tile): afterwards (v0..v3) = row q4, columns colb .. colb+3
                const float s0 = (q4 & 1) ? v0 : v1, s1 = (q4 & 1) ? v2 : v3;
                const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                if (q4 & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                const float t0 = (q4 & 2) ? v0 : v2, t1 = (q4 & 2) ? v1 : v3;
                const float u0 = quad_xor2(t0), u1 = quad_xor2(t1);
                if (q4 & 2) { v0 = u0; v1 = u1; } else { v2 = u0; v3 = u1; }
            }
            float v[4] = {v0 + bias4.x, v1 + bias4.y, v2 + bias4.z, v3 + bias4.w};
            if constexpr (BITS) {        // four bits per row group, an independent chain each (bit 15 - (4 g + k) = element k of group g)
                uint32_t m = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) m = __builtin_amdgcn_alignbit(m, __float_as_uint(0.f - v[k]), 31);
                rbits |= m << (12 - 4 * g);
            }
            // a NaN pre-activation becomes 0 in fmaxf and would pass the range test below: the sum of the four pre-activations is
            // NaN exactly when one of them is (or when +inf meets -inf), and NaN != NaN sets the flag
            const float pre = (v[0] + v[1]) + (v[2] + v[3]);
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            big |= (fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) > HOS_RANGE_LIMIT) | (pre != pre);
#ifdef HOS_EXP_NOSTORE     // timing experiment: results are written to the workgroup's first tile only (cache hits; results invalid)
            *reinterpret_cast<float4*>(a.C + (size_t)(blockIdx.x * R + q4 + 4 * lhi + 8 * g) * a.ldc + colb) = make_float4(v[0], v[1], v[2], v[3]);
#else
            *reinterpret_cast<float4*>(crow + (size_t)(8 * g) * a.ldc) = make_float4(v[0], v[1], v[2], v[3]);
#endif
        }
        if constexpr (BITS) {
            // two lanes' 16-bit masks as ONE dword store by the even lane (same bytes in memory: bits[t] | bits[t + 1] << 16)
            const uint32_t odd = (uint32_t)__builtin_amdgcn_mov_dpp((int)rbits, 0xB1, 0xF, 0xF, true);
            if ((lane & 1) == 0) reinterpret_cast<uint32_t*>(a.bits)[((size_t)tile * TH_NT + t) >> 1] = (rbits & 0xffffu) | (odd << 16);
        }
        // the other buffer is complete (this wave's LDS writes have landed at lgkmcnt(0)), this one is free for the tile after
        // next.  (Written out -- it is what __syncthreads() compiles to here -- so that it can never become a wait for the loads
        // and stores just issued.)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // The first round is peeled and the loop takes whole pairs (one latch, no exit between the halves), so that the loop is
    // entered with the same pending operations its back edge carries ([loads of one set][stores][loads of the other][stores]):
    // merged with a state that has nothing pending the compiler's counter keeps only the operations of the current round and
    // waits for the previous round's prefetch again.
    int b = 0;
    if constexpr (PF2) {
        body(ra, tile, b, tile + G < ntiles);
        tile += G; b ^= 1;
        if (tile < ntiles) {
            body(rb, tile, b, tile + G < ntiles);
            tile += G; b ^= 1;
            while (tile + G < ntiles) {
                body(ra, tile, b, true);
                body(rb, tile + G, b ^ 1, tile + 2 * G < ntiles);
                tile += 2 * G;
            }
            if (tile < ntiles) body(ra, tile, b, false);
        }
    } else {
        body(ra, tile, b, tile + G < ntiles);
        tile += G; b ^= 1;
        for (; tile < ntiles; tile += G, b ^= 1) body(ra, tile, b, tile + G < ntiles);
    }
    if (a.range_flag != nullptr && __builtin_amdgcn_ballot_w64(big) != 0 && lane == 0) atomicOr(a.range_flag, 1u);
}

// ---- DGRAD fast path, same recipe: dX[M, 256] = (dY[M, 256] . W[256, w-window of 256 columns]) * [bit mask], whole tiles only.
// The ReLU mask comes as the bit mask of the forward launch (or none); 2 bytes per lane travel with each tile's prefetch.
template <bool MASK>
__global__ __launch_bounds__(TH_NT, 1) void thin_dgrad_fast_kernel(const ThinArgs a) {
    typedef __bf16 E;
    typedef typename V8<E>::t e8;
    typedef typename V8<E>::q e4;
    constexpr int KS = 16, R = 32;
    constexpr int KD = KS * 16;
    constexpr int P = KD * 2 + th_pad<true>();
    constexpr int PLANE = R * P, BUF = 2 * PLANE;
    constexpr int AU = R * (KD / 4) / TH_NT;
    constexpr int PFD = 3;                               // two tiles of register prefetch

    extern __shared__ __attribute__((aligned(16))) char smem_th[];
    char* const buf0 = smem_th;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int col0 = wave * 32;

    // a.N = 32 x (active waves): 256 output columns (a whole layer) or 64 (the Fourier part of an input row: the other six waves
    // only stage tiles -- they issue no store, so they run their own copy of the loop with its own static operation counts)
    const bool active = col0 < a.N;
    e8 bh[KS], bl[KS];
    if (active) {   // B[k][n] = W[k][col0 + n]: column gather, once per launch
        const float* wcol = a.W + (size_t)(8 * lhi) * a.ldw + col0 + l31;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) { E h, l; split_pair<E>(wcol[(size_t)(16 * s + q) * a.ldw], h, l); bh[s][q] = h; bl[s][q] = l; }
    }
    const int q4 = l31 & 3, colb = col0 + (l31 & ~3);
    const int ntiles = a.M / R;
    const int G = gridDim.x;
    float4 ra[AU], rb[AU];
    uint32_t ba = 0, bb = 0, bits_staged = 0;
    auto gload = [&](float4 (&r)[AU], uint32_t& rbits, int tile) {
        tile = tile < ntiles ? tile : ntiles - 1;        // clamped, never predicated
        const float* base = a.A + (size_t)tile * R * a.lda;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int u = t + TH_NT * i, row = u / (KD / 4), c4 = u % (KD / 4);
            r[i] = ld4(base + (size_t)row * a.lda + c4 * 4);
        }
        if constexpr (MASK) rbits = a.bits[(size_t)tile * TH_NT + t];
    };
    auto sstore = [&](const float4 (&r)[AU], const uint32_t rbits, int b) {
        char* const hi = buf0 + b * BUF;
        char* const lo = hi + PLANE;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int u = t + TH_NT * i, row = u / (KD / 4), c4 = u % (KD / 4);
            e4 h, l;
            const float xs[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { E hh, ll; split_pair<E>(xs[q], hh, ll); h[q] = hh; l[q] = ll; }
            *reinterpret_cast<e4*>(hi + row * P + c4 * 8) = h;
            *reinterpret_cast<e4*>(lo + row * P + c4 * 8) = l;
        }
        bits_staged = rbits;
    };

    int tile = blockIdx.x;
    gload(ra, ba, tile);
    sstore(ra, ba, 0);
    gload(ra, ba, tile + G);
    gload(rb, bb, tile + 2 * G);
    __syncthreads();
    auto body = [&](auto act, float4 (&rcur)[AU], uint32_t& bcur, const int tile, const int b, const bool more) {
        constexpr bool ACT = decltype(act)::value;
        const uint32_t bits_cur = bits_staged;           // this tile's ReLU bits (staged together with its operands)
        const char* hi = buf0 + b * BUF + l31 * P + lhi * 16;
        f32x16 acc;
        if constexpr (ACT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const e8 ah = *reinterpret_cast<const e8*>(hi + s * 32);
                const e8 al = *reinterpret_cast<const e8*>(hi + PLANE + s * 32);
                acc = mfma_e(al, bh[s], acc);
                acc = mfma_e(ah, bl[s], acc);
                acc = mfma_e(ah, bh[s], acc);
            }
        }
        if (more) sstore(rcur, bcur, b ^ 1);
        gload(rcur, bcur, tile + PFD * G);
        float* crow = a.C + (size_t)(tile * R + q4 + 4 * lhi) * a.ldc + colb;
        if constexpr (ACT)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = acc[4 * g + 0], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
            {   // 4x4 transpose inside the quad: afterwards (v0..v3) = row q4 + 8 g + 4 lhi, columns colb .. colb+3
                const float s0 = (q4 & 1) ? v0 : v1, s1 = (q4 & 1) ? v2 : v3;
                const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                if (q4 & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                const float t0 = (q4 & 2) ? v0 : v2, t1 = (q4 & 2) ? v1 : v3;
                const float u0 = quad_xor2(t0), u1 = quad_xor2(t1);
                if (q4 & 2) { v0 = u0; v1 = u1; } else { v2 = u0; v3 = u1; }
            }
            float v[4] = {v0, v1, v2, v3};
            if constexpr (MASK) {
#pragma unroll
                for (int k = 0; k < 4; ++k)       // bit -> all-ones / zero (v_bfe_i32), AND
                    v[k] = __uint_as_float(__float_as_uint(v[k]) & (uint32_t)__builtin_amdgcn_sbfe((int)bits_cur, 15 - (4 * g + k), 1));
            }
            *reinterpret_cast<float4*>(crow + (size_t)(8 * g) * a.ldc) = make_float4(v[0], v[1], v[2], v[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto run = [&](auto act) {
        int b = 0;
        body(act, ra, ba, tile, b, tile + G < ntiles);
        tile += G; b ^= 1;
        if (tile < ntiles) {
            body(act, rb, bb, tile, b, tile + G < ntiles);
            tile += G; b ^= 1;
            while (tile + G < ntiles) {
                body(act, ra, ba, tile, b, true);
                body(act, rb, bb, tile + G, b ^ 1, tile + 2 * G < ntiles);
                tile += 2 * G;
            }
            if (tile < ntiles) body(act, ra, ba, tile, b, false);
        }
    };
    if (active) run(std::true_type{});
    else run(std::false_type{});
}

template <bool MASK>
int launch_thin_dgrad_fast(const ThinArgs& a, hipStream_t stream) {
    constexpr size_t smem = 2 * 2 * (size_t)32 * (16 * 32 + th_pad<true>());
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_dgrad_fast_kernel<MASK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int ntiles = a.M / 32;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL((thin_dgrad_fast_kernel<MASK>), dim3(grid), dim3(TH_NT), smem, stream, a);
    return hos_launch_status();
}

template <int KS, bool BITS>
int launch_thin_fast(const ThinArgs& a, hipStream_t stream) {
    constexpr size_t smem = 2 * 2 * (size_t)32 * (KS * 32 + TH_PAD_FWD);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_fwd_fast_kernel<KS, BITS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int ntiles = a.M / 32;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL((thin_fwd_fast_kernel<KS, BITS>), dim3(grid), dim3(TH_NT), smem, stream, a);
    return hos_launch_status();
}

template <int KS, bool DGRAD>
int launch_thin(const ThinArgs& a, hipStream_t stream) {
    constexpr int R = DGRAD ? 32 : HOS_THIN_R_FWD;
    constexpr size_t smem = 2 * 2 * (size_t)R * (KS * 32 + th_pad<DGRAD>()) + (DGRAD ? 2 * (size_t)R * TH_MP : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_gemm_kernel<KS, DGRAD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int ntiles = hos_cdiv(a.M, R);
    const int grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL((thin_gemm_kernel<KS, DGRAD>), dim3(grid), dim3(TH_NT), smem, stream, a);
    return hos_launch_status();
}

// ---- The canonical MLP with its per-call state embedding folded into biases (mlp_rgb_sigma.py:49-58: the input row is
// [fourier nf | state ne], the skip layer's [fourier nf | state ne | h nh]; the state embedding is ONE vector per call).
//   pack:   W0f [n0, nfp] = W0[:, :nf] | 0,   b0f = b0 + W0[:, nf:nf+ne] . embed
//           W5f [n0, nfp + nh] = W5[:, :nf] | 0 | W5[:, nf+ne:],   b5f = b5 + W5[:, nf:nf+ne] . embed        (nfp = nf rounded up to 4)
//   unfold: the gradients of the folded layers back into the reference-shaped ones, the state columns as db (x) embed, and
//           d embed += W0[:, nf:nf+ne]^T db0 + W5[:, nf:nf+ne]^T db5 (fixed-order fp32 sums)
struct CnlFold {
    const float* W0; int ld0; const float* b0; const float* W5; int ld5; const float* b5;
    const float* embed; int n0, nf, ne, nh, nfp;
    float* W0f; float* b0f; float* W5f; float* b5f;                   // pack outputs / unfold: gradient INPUTS (gW0f, db0, gW5f, db5)
    float* gW0; float* gb0; float* gW5; float* gb5; float* g_embed;   // unfold outputs (+=)
};

__global__ __launch_bounds__(256) void cnl_fold_pack_kernel(const CnlFold a) {
    const int n = blockIdx.x, t = threadIdx.x;
    const float* w0 = a.W0 + (size_t)n * a.ld0;
    const float* w5 = a.W5 + (size_t)n * a.ld5;
    for (int k = t; k < a.nfp; k += 256) a.W0f[(size_t)n * a.nfp + k] = k < a.nf ? w0[k] : 0.f;
    const int k5 = a.nfp + a.nh;
    for (int k = t; k < k5; k += 256) a.W5f[(size_t)n * k5 + k] = k < a.nf ? w5[k] : (k < a.nfp ? 0.f : w5[a.nf + a.ne + (k - a.nfp)]);
    if (t < 2) {
        const float* w = (t ? w5 : w0) + a.nf;
        float v = t ? a.b5[n] : a.b0[n];
        for (int c = 0; c < a.ne; ++c) v = fmaf(w[c], a.embed[c], v);
        (t ? a.b5f : a.b0f)[n] = v;
    }
}

__global__ __launch_bounds__(256) void cnl_fold_unfold_kernel(const CnlFold a) {
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= a.n0) {                 // d embed: block n0 + c sums column c over the output rows of both layers
        // (256 threads take rows t, t + 256, ... in order, then a fixed-order tree: deterministic.  One thread per column
        // walking 2 x 256 strided rows serially took 209 us -- a fifth of a millisecond for 64 numbers.)
        __shared__ float part[256];
        const int c = (int)blockIdx.x - a.n0;
        float s = 0.f;
        for (int n = t; n < a.n0; n += 256) s = fmaf(a.W0[(size_t)n * a.ld0 + a.nf + c], a.b0f[n], s);
        for (int n = t; n < a.n0; n += 256) s = fmaf(a.W5[(size_t)n * a.ld5 + a.nf + c], a.b5f[n], s);
        part[t] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (t < w) part[t] += part[t + w];
            __syncthreads();
        }
        if (t == 0) a.g_embed[c] += part[0];
        return;
    }
    const int n = blockIdx.x;
    const float d0 = a.b0f[n], d5 = a.b5f[n];
    float* g0 = a.gW0 + (size_t)n * a.ld0;
    float* g5 = a.gW5 + (size_t)n * a.ld5;
    const int k5 = a.nfp + a.nh;
    for (int k = t; k < a.nf + a.ne; k += 256) {
        g0[k] += k < a.nf ? a.W0f[(size_t)n * a.nfp + k] : d0 * a.embed[k - a.nf];
        g5[k] += k < a.nf ? a.W5f[(size_t)n * k5 + k] : d5 * a.embed[k - a.nf];
    }
    for (int j = t; j < a.nh; j += 256) g5[a.nf + a.ne + j] += a.W5f[(size_t)n * k5 + a.nfp + j];
    if (t == 0) { a.gb0[n] += d0; a.gb5[n] += d5; }
}

}  // namespace

extern "C" int hos_canonical_fold_pack(const float* W0, int ld0, const float* b0, const float* W5, int ld5, const float* b5,
                                       const float* embed, int n_out, int nf, int ne, int nh,
                                       float* W0f, float* b0f, float* W5f, float* b5f, hos_stream_t stream) {
    if (!W0 || !b0 || !W5 || !b5 || !embed || !W0f || !b0f || !W5f || !b5f) return HOS_E_ARG;
    if (n_out <= 0 || nf <= 0 || ne <= 0 || nh <= 0 || ld0 < nf + ne || ld5 < nf + ne + nh) return HOS_E_SHAPE;
    CnlFold a{W0, ld0, b0, W5, ld5, b5, embed, n_out, nf, ne, nh, (nf + 3) & ~3, W0f, b0f, W5f, b5f, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(cnl_fold_pack_kernel, dim3(n_out), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hos_launch_status();
}

extern "C" int hos_canonical_fold_unfold(const float* gW0f, const float* db0, const float* gW5f, const float* db5,
                                         const float* W0, int ld0, const float* W5, int ld5, const float* embed,
                                         int n_out, int nf, int ne, int nh,
                                         float* gW0, float* gb0, float* gW5, float* gb5, float* g_embed, hos_stream_t stream) {
    if (!gW0f || !db0 || !gW5f || !db5 || !W0 || !W5 || !embed || !gW0 || !gb0 || !gW5 || !gb5 || !g_embed) return HOS_E_ARG;
    if (n_out <= 0 || nf <= 0 || ne <= 0 || ne > 256 || nh <= 0 || ld0 < nf + ne || ld5 < nf + ne + nh) return HOS_E_SHAPE;
    CnlFold a{W0, ld0, nullptr, W5, ld5, nullptr, embed, n_out, nf, ne, nh, (nf + 3) & ~3,
              const_cast<float*>(gW0f), const_cast<float*>(db0), const_cast<float*>(gW5f), const_cast<float*>(db5), gW0, gb0, gW5, gb5, g_embed};
    hipLaunchKernelGGL(cnl_fold_unfold_kernel, dim3(n_out + ne), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hos_launch_status();
}

// Y[M, N] = epi(X[M, :K] . W[:N, :K]^T + bias), N <= 256, K <= 320 (K % 4 == 0), epilogue HOS_EPI_NONE or HOS_EPI_RELU.
// relu_bits (optional, 2 * 512 * ceil(M / 32) bytes): one bit per output element = "came out > 0", in the order the backward
// kernel (hos_thin_linear_dgrad, mask_bits) consumes it; a waves's 32 columns that lie at or beyond N are not written.
extern "C" int hos_thin_linear_fwd(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy,
                                   int M, int N, int K, int epilogue, void* relu_bits, hos_stream_t stream) {
    if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return HOS_E_ARG;
    if (N > 256 || K > 320 || (epilogue != HOS_EPI_NONE && epilogue != HOS_EPI_RELU)) return HOS_E_SHAPE;
    if ((ldx & 3) || (ldw & 3) || (K & 3) || (((uintptr_t)X | (uintptr_t)W) & 15u)) return HOS_E_ALIGN;
    if (relu_bits && ((uintptr_t)relu_bits & 1u)) return HOS_E_ALIGN;
    ThinArgs a{X, ldx, W, ldw, bias, Y, ldy, M, N, K, epilogue, nullptr, 0, hos_range_flag_ptr(), static_cast<uint16_t*>(relu_bits)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // reduction steps held in registers: 4 (the folded canonical input layer, 64 columns), 8, 16, 20 (the folded skip layer:
    // [fourier 64 | h 256]; 250 VGPRs -- 24 steps for the reference-shaped 384-wide concat row do not fit two waves per SIMD)
    // whole tiles of the shapes the canonical MLP runs (256 outputs; 64 / 256 / 320 inputs): the unpredicated kernel; a ragged
    // tail of M % 32 rows goes through the generic one (HOS_THIN_FAST=0: everything does)
    static const bool fast_on = !(getenv("HOS_THIN_FAST") && atoi(getenv("HOS_THIN_FAST")) == 0);
    if (fast_on && N == 256 && M >= 32 && (K == 64 || K == 256 || K == 320) && !(ldy & 3) && !((uintptr_t)Y & 15u) &&
        (!bias || !((uintptr_t)bias & 15u)) && !((uintptr_t)relu_bits & 3u)) {
        ThinArgs f = a;
        f.M = M & ~31;
        int rc;
        if (relu_bits) rc = K == 64 ? launch_thin_fast<4, true>(f, s) : (K == 256 ? launch_thin_fast<16, true>(f, s) : launch_thin_fast<20, true>(f, s));
        else rc = K == 64 ? launch_thin_fast<4, false>(f, s) : (K == 256 ? launch_thin_fast<16, false>(f, s) : launch_thin_fast<20, false>(f, s));
        if (rc != 0 || f.M == M) return rc;
        a.A += (size_t)f.M * ldx; a.C += (size_t)f.M * ldy; a.M = M - f.M;
        if (a.bits) a.bits += (size_t)(f.M / 32) * TH_NT;
    }
    if (K <= 64) return launch_thin<4, false>(a, s);
    if (K <= 128) return launch_thin<8, false>(a, s);
    return K <= 256 ? launch_thin<16, false>(a, s) : launch_thin<20, false>(a, s);
}

// dX[M, K] = (dY[M, :Npad] . W[:Npad, :K]) * [mask > 0], K <= 256 output columns, Npad <= 256 (Npad % 4 == 0; rows of W and
// columns of dY beyond the layer's width are zero by contract).  mask: the layer's input activations [M, >= K] or NULL;
// W and mask may start at ANY column of their matrices (4-byte aligned: the h part of a skip layer's concat row starts at
// column 127; the mask's 16-byte groups around the window must be readable), dY and dX rows are 16-byte aligned.
// mask_bits (optional): the bit mask hos_thin_linear_fwd wrote for the layer's input activations (its Y [M, K]); takes
// precedence over `mask` and removes the re-read of the fp32 activations (a third of this kernel's HBM traffic).
extern "C" int hos_thin_linear_dgrad(const float* dY, int lddy, const float* W, int ldw, int Npad, const float* mask, int ldmask,
                                     const void* mask_bits, float* dX, int lddx, int M, int K, hos_stream_t stream) {
    if (!dY || !W || !dX || M <= 0 || K <= 0 || Npad <= 0) return HOS_E_ARG;
    if (K > 256 || Npad > 256) return HOS_E_SHAPE;
    if ((lddy & 3) || (Npad & 3) || (mask && (ldmask & 3)) || ((uintptr_t)dY & 15u) || (((uintptr_t)W | (uintptr_t)mask) & 3u)) return HOS_E_ALIGN;
    if (mask_bits && ((uintptr_t)mask_bits & 1u)) return HOS_E_ALIGN;
    ThinArgs a{dY, lddy, W, ldw, nullptr, dX, lddx, M, K, Npad, 0, mask_bits ? nullptr : mask, ldmask, nullptr,
               static_cast<uint16_t*>(const_cast<void*>(mask_bits))};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // whole tiles of a full 256 x 256 layer (or its 64-column Fourier window) with the bit mask or no mask: the unpredicated kernel;
    // ragged tail: generic
    static const bool fast_on = !(getenv("HOS_THIN_FAST") && atoi(getenv("HOS_THIN_FAST")) == 0);
    if (fast_on && (K == 256 || K == 64) && Npad == 256 && M >= 32 && (mask_bits || !mask) && !(lddx & 3) && !((uintptr_t)dX & 15u)) {
        ThinArgs f = a;
        f.M = M & ~31;
        const int rc = mask_bits ? launch_thin_dgrad_fast<true>(f, s) : launch_thin_dgrad_fast<false>(f, s);
        if (rc != 0 || f.M == M) return rc;
        a.A += (size_t)f.M * lddy; a.C += (size_t)f.M * lddx; a.M = M - f.M;
        if (a.bits) a.bits += (size_t)(f.M / 32) * TH_NT;
    }
    return Npad <= 128 ? launch_thin<8, true>(a, s) : launch_thin<16, true>(a, s);
}
