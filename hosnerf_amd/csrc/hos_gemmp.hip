// "Planes" GEMM: operands pre-split into 16-bit hi/lo planes in HBM, staged by LDS-DMA, 3 MFMAs per product.
//
// Why: in the on-the-fly split kernel (hos_gemm3.hip) every K tile costs, per CU, 64 KB through registers, ~160
// VALU conversions and 16 LDS stores per thread next to 1536 cycles of MFMA work per wave -- the matrix pipe sat
// at 28-44 %.  Here the split is done ONCE per element by the producer (GEMM epilogues, hos_split_planes for
// weights and boundary tensors), so a K tile is
//     8 x global_load_lds_dwordx4 per wave (no VGPRs, no VALU)  +  24 fragment reads  +  48 x MFMA.
// scripts/probe/dma_probe.hip: this DMA pattern alone streams the whole operand set of a [32768,1024,1024] GEMM in
// 66-87 us at one workgroup per CU (20-26 B/clk/CU), i.e. under the 82 us the MFMAs need.
//
// Plane storage ("interleaved planes"): a split matrix [R][ld] (ld % 32 == 0) is ONE 16-bit array [R][ld/32][2][32]:
// per row and per block of 32 columns, 32 hi values followed by the 32 lo values (128 bytes).  The 32-deep K slice
// of a row that a tile needs is then one whole 128-byte line (a request for half a line costs the memory path as much
// as a whole one: scripts/probe/dma_probe.hip streams the operand set in 66 us with 128-byte pieces, 87 us with 64).
// One kernel for the three passes; element (r,c) of a split matrix is hi + lo of its block:
//   FWD    Y[m][n]  = sum_k X[m][k]  W[n][k]     X, W   fp16 planes (22 mantissa bits: fp32-grade), k contiguous
//   DGRAD  dX[m][k] = sum_n dZ[m][n] Wt[k][n]    dZ, Wt bf16 planes (8-bit exponent: gradients of any size)
//   WGRAD  dW[n][k] = sum_m dZ[m][n] X[m][k]     dZ, X  bf16 planes, ROW-MAJOR (reduction index = row): the MFMA
//          fragments (8 consecutive reduction values per lane) are gathered with ds_read_b64_tr_b16, the gfx950
//          LDS transpose read, so no transposed copy of any activation is ever written to HBM.
//
// Tile 256 x BN x 32, 512 threads (8 wave64 as 4(M) x 2(N)), LDS = 2 stages x {A tile, B tile}:
//   * k-contiguous operands: tile [rows][hi 32 k | lo 32 k] = 128-byte rows, the eight 16-byte chunks of a row
//     XOR-swizzled by (row>>1)&7 so the ds_read_b128 fragment reads are conflict free;
//   * reduction-row operands (WGRAD): tile [32 m][R cols x {hi,lo}], the 64-byte units of a row XOR-swizzled by
//     m&3 so the four rows a transpose read touches sit in four different 64-byte bank groups.
//   global_load_lds writes LDS linearly (wave base + lane*16), so both swizzles are applied on the per-lane SOURCE
//   address.  Waves 0-3 stage the A tile (a quarter each), waves 4-7 the B tile.
// Main loop: software-pipelined over quarter tiles with explicit counted waits and raw s_barrier (see below).
// Epilogue: plane outputs go through a wave-private LDS staging buffer so that every global store instruction
// writes whole 128-byte lines (the MFMA C layout gives a lane one column of 16 rows).
#include "hos_gemm_common.h"
#include <cstdlib>

// compile-time ablation switches for timing experiments (a run-time test inside the K loop splits the basic block and
// makes hipcc fall back to lgkmcnt(0) before every MFMA group, which serialises LDS reads and MFMAs)
#ifndef HOS_ABLATE_MFMA
#define HOS_ABLATE_MFMA 0
#endif
#ifndef HOS_ABLATE_DMA
#define HOS_ABLATE_DMA 0
#endif
// Round 5, built, measured and compiled OUT by default (-DHOS_GEMMP_PERSIST=1 builds it in; scripts/build_variant.sh): PERSISTENT FWD /
// DGRAD launches of the 256-wide tile -- one workgroup per CU walks the output tiles and the K pipeline (LDS-DMA two tiles ahead) runs
// straight across tile boundaries, so a tile has no prologue: its first two K tiles are requested while the previous tile still
// multiplies, and its epilogue's stores drain under the next tile's K loop.  Isolated launch at [131072,1024,1024]: 0-2 % (706 -> 691 us
// forward, dgrad equal).  In the steps it LOSES: two-stream stage 3 31.93 vs 31.29 ms, one stream 33.03 vs 32.97, stage 1 6.42 vs 6.39,
// the 1080p frame 3958 vs 3936-3951 ms (three / two alternations on one box, profiles/r05_persist_step_ab.txt) -- one workgroup per tile
// lets the other stream's kernels take CUs tile by tile and lets the dispatcher balance the tail, a persistent launch holds its CUs
// from its first tile to its last.
#ifndef HOS_GEMMP_PERSIST
#define HOS_GEMMP_PERSIST 0
#endif

namespace {

constexpr int PBM = 256;
constexpr int PBK = 32;
constexpr int PNT = 512;
constexpr int PROWB = 64;

template <typename E> struct PVec { typedef E x8 __attribute__((ext_vector_type(8))); typedef E x4 __attribute__((ext_vector_type(4))); };
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 pmfma(const PVec<__bf16>::x8& a, const PVec<__bf16>::x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 pmfma(const PVec<_Float16>::x8& a, const PVec<_Float16>::x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

enum PEpi { PEPI_F32 = 0, PEPI_PLANES_FWD = 1, PEPI_PLANES_DGRAD = 2, PEPI_WGRAD = 3 };

struct PArgs {
    // operands (interleaved 16-bit planes, row pitch 2*ld elements); A may have a second K segment (skip concat,
    // k-contiguous form only)
    const uint16_t* A; int lda; int kt0;
    const uint16_t* A1; int lda1;
    const uint16_t* B; int ldb;
    int M, N;            // output extents (rows i of the A side, rows j of the B side)
    int nk, kt_per_split, tiles_m, tiles_n;
    GemmArgs f32;        // fp32 output path (fused epilogues of hos_gemm_common.h) and WGRAD accumulation
    const float* bias;   // FWD
    int relu;            // FWD: apply ReLU
    const uint16_t* mask; int ldmask;             // DGRAD: fp16 planes of the layer input; gradient passes where hi > 0
    uint16_t* Y; int ldy;                         // row-major planes [M][ldy]: fp16 for FWD, bf16 for DGRAD
    uint16_t* Yb; int ldyb;                       // FWD only: the same values as bf16 planes (WGRAD operand)
    uint32_t* bits; int bits_nb;                  // ReLU bit mask (FWD writes, DGRAD reads), 64-column blocks per row block
    int total;                                    // output tiles x K splits (virtual block ids)
};

template <typename E> __device__ __forceinline__ uint16_t to_bits(E v) { return __builtin_bit_cast(uint16_t, v); }
template <typename E> __device__ __forceinline__ float clampE(float x) { return x; }
template <> __device__ __forceinline__ float clampE<_Float16>(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }

// (hi, lo) of x packed as lo16 = hi bits, hi16 = lo bits
template <typename E>
__device__ __forceinline__ uint32_t split_pack(float x) {
    const E h = (E)clampE<E>(x);
    const E l = (E)(x - (float)h);
    return (uint32_t)to_bits<E>(h) | ((uint32_t)to_bits<E>(l) << 16);
}
// the same dword for a value the epilogue holds in a register: the final pack is ONE packed conversion (lower half = the
// hi part again, upper half = the residual), gfx950's v_cvt_pk_{f16,bf16}_f32 (round to nearest even like the scalar forms)
typedef float pf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t split_pack_pk(float x, _Float16) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float c = clampE<_Float16>(x);
    const pf32x2 v = {c, x - (float)(_Float16)c};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
}
__device__ __forceinline__ uint32_t split_pack_pk(float x, __bf16) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const pf32x2 v = {x, x - (float)(__bf16)x};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}
template <typename E>
__device__ __forceinline__ void split1(float x, uint16_t& hi, uint16_t& lo) {
    const uint32_t p = split_pack<E>(x);
    hi = (uint16_t)(p & 0xffffu); lo = (uint16_t)(p >> 16);
}

__device__ __forceinline__ void dma16(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}
__device__ __forceinline__ s16x4 lds_tr(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// TR = false: both operands k-contiguous (FWD, DGRAD).  TR = true: both operands reduction-row-major (WGRAD).
// PERSIST: the workgroup walks output tiles blockIdx.x, blockIdx.x + gridDim.x, ... (k-contiguous form, BN = 256, nk >= 2).
template <int BN, int EPI, typename EIN, bool TR, bool PERSIST>
__global__ __launch_bounds__(PNT, 2) void gemmp_kernel(const PArgs a) {
    static_assert(!PERSIST || (!TR && BN == 256), "persistent form: k-contiguous operands, 256-wide tile");
    typedef typename PVec<EIN>::x8 ex8;
    constexpr int WN = 2;
    constexpr int TM = PBM / (4 * 32);       // 2
    constexpr int TN = BN / (WN * 32);       // 4 or 2
    constexpr int TH = TN / 2;
    constexpr int A_TILE = PBM * 128, B_TILE = BN * 128;            // bytes: rows x (hi 64 B + lo 64 B) in either form
    constexpr int STAGE = A_TILE + B_TILE;
    constexpr int QA = PBM / 32, QB = BN / 32;      // DMA instructions per wave per K tile for the A / B tile
    constexpr int QMAX = QA > QB ? QA : QB;
    constexpr int A_PITCH = PBM * 4, B_PITCH = BN * 4;              // TR form: bytes per reduction row (hi and lo)

    extern __shared__ __attribute__((aligned(16))) char smemp[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // virtual block id -> (row tile, column tile, K split); consecutive ids of one XCD share the A panel
    const int nb = a.total;
    int tm_i, tn_i, split;
    auto decode = [&](int vb) {
        const int q = nb >> 3, r = nb & 7, x = vb & 7, y = vb >> 3;
        const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        tn_i = bid % a.tiles_n;
        tm_i = (bid / a.tiles_n) % a.tiles_m;
        split = bid / (a.tiles_n * a.tiles_m);
    };
    int vb = blockIdx.x;
    decode(vb);
    int i0 = tm_i * PBM, j0 = tn_i * BN;
    const int kt_begin = PERSIST ? 0 : split * a.kt_per_split;
    const int kt_end = PERSIST ? a.nk : min(a.nk, kt_begin + a.kt_per_split);
    if (kt_begin >= kt_end) return;

    // ---- DMA plan of this wave -----------------------------------------------------------------------------------
    const bool isB = wave >= 4;                  // waves 0-3 stage the A tile, waves 4-7 the B tile
    const int quarter = wave & 3;                // ... one quarter of it each
    const int nq = isB ? QB : QA;                // instructions per K tile
    int g_row0 = isB ? j0 : i0;                  // first output row (k-contiguous) / first column (TR) of this side
    const int g_limit = isB ? a.N : a.M;
    // Source arrays selected ONCE into scalars.  (Selecting them inside the DMA lambda made hipcc build a pointer
    // table in scratch; every scratch_load result was then waited for with vmcnt(0), which drained the LDS-DMA
    // queue before EACH global_load_lds and serialised the eight requests of a tile.)
    const uint16_t* P0; const uint16_t* P1; int ld0, ld1;
    if (!isB) { P0 = a.A; P1 = a.A1; ld0 = a.lda; ld1 = a.lda1; }
    else      { P0 = a.B; P1 = a.B;  ld0 = a.ldb; ld1 = a.ldb; }
    const int kt0 = (isB || TR) ? 0x7fffffff : a.kt0;
    const int tile_bytes = isB ? B_TILE : A_TILE;
    char* const lds_wave = smemp + (isB ? A_TILE : 0) + quarter * (tile_bytes / 4);

    // k-contiguous form: instruction q covers 8 rows x 128 B of this wave's quarter; lane -> (row, chunk position);
    // the source chunk is position ^ swizzle(row), swizzle(row) = (row>>1)&7 = (lane>>4) | ((q&1)<<2)
    const int kc_row = quarter * ((isB ? BN : PBM) / 4) + (lane >> 3);
    const int kc_pos = lane & 7, kc_swz = (lane >> 4) & 3;
    // TR form: instruction q covers 1 KB of this wave's 8 reduction rows; pitch = 4 * (columns of the tile)
    const int tr_pitch = isB ? B_PITCH : A_PITCH;

    auto issue_dma = [&](int q, int kt, int stage, int row0) {
        size_t off;            // 64-bit: 4 Mi rows x 576 columns x 2 planes already exceeds 2^32 elements (65 536-ray proposal levels)
        const uint16_t* P;
        if constexpr (!TR) {
            const bool seg1 = kt >= kt0;                                 // A may switch to its second K segment
            P = seg1 ? P1 : P0;
            const int ld = seg1 ? ld1 : ld0;
            const int kb = seg1 ? kt - kt0 : kt;                         // 32-column block of the row
            int gr = row0 + kc_row + 8 * q;
            gr = gr < g_limit ? gr : g_limit - 1;                       // clamp: out-of-range rows are never stored
            const int chunk = kc_pos ^ (kc_swz | ((q & 1) << 2));       // 0-3: hi k 0..31, 4-7: lo
            off = (size_t)gr * (size_t)(2 * ld) + (size_t)(kb * 64 + chunk * 8);
        } else {
            P = P0;
            const int pos = q * 1024 + lane * 16;                       // byte position inside this wave's quarter
            const int m = quarter * 8 + pos / tr_pitch;                 // reduction row inside the K tile
            const int rb = pos % tr_pitch;                              // byte inside the LDS row
            const int unit = (rb >> 6) ^ (m & 3);                       // source 64-byte unit (swizzle on the source)
            int gc = row0 + (unit >> 1) * 32 + ((rb >> 4) & 3) * 8;     // logical column of this 16-byte piece
            gc = gc < ((g_limit + 7) & ~7) ? gc : 0;                    // clamp: columns past the operand are never stored
            off = (size_t)(kt * PBK + m) * (size_t)(2 * ld0) + (size_t)((gc >> 5) * 64 + (unit & 1) * 32 + (gc & 31));
        }
        dma16(P + off, lds_wave + stage * STAGE + q * 1024);             // LDS address is wave-uniform
    };

    f32x16 acc[TM][TN];
    unsigned bw_ok_mask = 0xffffffffu;         // DGRAD: blocks whose ReLU bit dword exists (set per tile)
    const int l31 = lane & 31, lhi = lane >> 5;
    float dbsum[TM];
#pragma unroll
    for (int x = 0; x < TM; ++x) dbsum[x] = 0.f;
    const bool do_db = (EPI == PEPI_WGRAD) && a.f32.db != nullptr && tn_i == 0 && wn == 0;

    // ---- fragment reads: opaque asm loads + explicit counted waits ---------------------------------------------
    // hipcc's own waitcnt insertion puts lgkmcnt(0) in front of every MFMA group of this loop (it stops counting
    // once an LDS-DMA is in flight), i.e. it would also wait for the reads just issued for the NEXT quarter.  The
    // reads are therefore asm statements the compiler does not count, every MFMA group is preceded by an asm
    // s_waitcnt lgkmcnt(0) (LDS returns in order and the next quarter's reads are issued BEHIND that wait), and the
    // fragments pass through that statement as "+v" operands so no MFMA can be scheduled above its wait
    // (cdna_hip_programming.md 5.7, form (ii)).
    //   k-contiguous: lane (row = l31, k half = lhi) reads chunk (2s + lhi) [hi] / 4 + (2s + lhi) [lo] of its 128-byte
    //       row, stored at chunk position c ^ ((row>>1)&7).
    //   TR: 16-lane group g = lane>>4 reads the [4 m][16 col] block (m0 = 16 s + 8 (g>>1) (+4), col0 = 16 (g&1));
    //       lane p of the group supplies the address of row m0 + (p>>2), columns col0 + 4 (p&3) .. +3 and receives
    //       column col0 + p of the four rows: two reads give the 8 consecutive reduction values of its MFMA row.
    //       A 32-column block is two 64-byte units (hi, lo), stored at unit position u ^ (m&3).
    // Per-lane byte offsets inside stage 0 ([.][0] hi, [.][1] lo); the rest of an address is an immediate or the stage.
    constexpr int NOA = 2, NOB = TR ? TN : 2;
    unsigned offA[NOA][2], offB[NOB][2];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smemp;
    if constexpr (!TR) {
        const int swz = (l31 >> 1) & 7;
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                const int pos = (hl * 4 + 2 * sx + lhi) ^ swz;
                offA[sx][hl] = lds_base + (wm * (TM * 32) + l31) * 128 + pos * 16;
                offB[sx][hl] = lds_base + A_TILE + (wn * (TN * 32) + l31) * 128 + pos * 16;
            }
    } else {
        const int tr_p = lane & 15, tr_g = lane >> 4;
        const int m_lane = 8 * (tr_g >> 1) + (tr_p >> 2), sw = tr_p >> 2;
        const int lcol = (16 * (tr_g & 1) + 4 * (tr_p & 3)) * 2;
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
            for (int x = 0; x < TM; ++x) offA[x][hl] = lds_base + m_lane * A_PITCH + ((((wm * TM + x) * 2 + hl) ^ sw) << 6) + lcol;
#pragma unroll
            for (int y = 0; y < TN; ++y) offB[y][hl] = lds_base + A_TILE + m_lane * B_PITCH + ((((wn * TN + y) * 2 + hl) ^ sw) << 6) + lcol;
        }
    }
#define HOS_RD128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define HOS_RDTR(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
    // fragment storage: k-contiguous form uses .v directly; TR form loads two 64-bit halves and joins them after the wait
    struct Frag { ex8 v; s16x4 h0, h1; };
    Frag a0h[TM], a0l[TM], a1h[TM], a1l[TM], b0h[TH], b0l[TH], b1h[TH], b1l[TH];

    // S = k half, X / Y = tile index inside the wave tile: all compile-time so the LDS offsets are immediates
#define HOS_READ_A1(SET_H, SET_L, STG, S, X)                                                                   \
    do {                                                                                                       \
        if constexpr (!TR) {                                                                                   \
            HOS_RD128(SET_H[X].v, offA[S][0] + (STG), (X) * 32 * 128);                                         \
            HOS_RD128(SET_L[X].v, offA[S][1] + (STG), (X) * 32 * 128);                                         \
        } else {                                                                                               \
            HOS_RDTR(SET_H[X].h0, offA[X][0] + (STG), (S) * 16 * A_PITCH);                                     \
            HOS_RDTR(SET_H[X].h1, offA[X][0] + (STG), (S) * 16 * A_PITCH + 4 * A_PITCH);                       \
            HOS_RDTR(SET_L[X].h0, offA[X][1] + (STG), (S) * 16 * A_PITCH);                                     \
            HOS_RDTR(SET_L[X].h1, offA[X][1] + (STG), (S) * 16 * A_PITCH + 4 * A_PITCH);                       \
        }                                                                                                      \
    } while (0)
#define HOS_READ_A(SET_H, SET_L, STG, S) do { HOS_READ_A1(SET_H, SET_L, STG, S, 0); HOS_READ_A1(SET_H, SET_L, STG, S, 1); } while (0)
#define HOS_READ_B1(SET_H, SET_L, STG, S, YH, Y)                                                               \
    do {                                                                                                       \
        if constexpr ((Y) < TH) {                                                                              \
            if constexpr (!TR) {                                                                               \
                HOS_RD128(SET_H[Y].v, offB[S][0] + (STG), ((YH) * TH + (Y)) * 32 * 128);                       \
                HOS_RD128(SET_L[Y].v, offB[S][1] + (STG), ((YH) * TH + (Y)) * 32 * 128);                       \
            } else {                                                                                           \
                HOS_RDTR(SET_H[Y].h0, offB[((YH) * TH + (Y)) % NOB][0] + (STG), (S) * 16 * B_PITCH);           \
                HOS_RDTR(SET_H[Y].h1, offB[((YH) * TH + (Y)) % NOB][0] + (STG), (S) * 16 * B_PITCH + 4 * B_PITCH); \
                HOS_RDTR(SET_L[Y].h0, offB[((YH) * TH + (Y)) % NOB][1] + (STG), (S) * 16 * B_PITCH);           \
                HOS_RDTR(SET_L[Y].h1, offB[((YH) * TH + (Y)) % NOB][1] + (STG), (S) * 16 * B_PITCH + 4 * B_PITCH); \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#define HOS_READ_B(SET_H, SET_L, STG, S, YH) do { HOS_READ_B1(SET_H, SET_L, STG, S, YH, 0); HOS_READ_B1(SET_H, SET_L, STG, S, YH, 1); } while (0)
    // make the fragments of one A and one B set visible behind a wait
    auto tie = [&](Frag& f) {
        if constexpr (!TR) { asm volatile("" : "+v"(f.v)); }
        else {
            asm volatile("" : "+v"(f.h0), "+v"(f.h1));
            const s16x8 j = __builtin_shufflevector(f.h0, f.h1, 0, 1, 2, 3, 4, 5, 6, 7);
            f.v = __builtin_bit_cast(ex8, j);
        }
    };
    // bias gradient (WGRAD): the A fragments of the wn == 0 waves hold dZ[m][n] for 8 m per lane
    auto db_acc = [&](const Frag (&ah)[TM], const Frag (&al)[TM]) {
        if constexpr (EPI == PEPI_WGRAD) {
            if (do_db) {
#pragma unroll
                for (int x = 0; x < TM; ++x)
#pragma unroll
                    for (int e = 0; e < 8; ++e) dbsum[x] += (float)ah[x].v[e] + (float)al[x].v[e];
            }
        }
    };

    // ---- main loop: software-pipelined over QUARTER tiles (k half s = 0/1  x  column half yh = 0/1) -----------
    // Two A fragment sets (one per k half) and two B fragment sets (one per quarter) rotate so that every
    // quarter's LDS reads fly under the previous quarter's MFMAs; two LDS stages; the DMA runs one tile ahead:
    //     read B1=(s0,yh1), B waves: DMA(kt+1) | MFMA (A0,B0)
    //     read A1=(s1), B0=(s1,yh0)            | MFMA (A0,B1)
    //     read B1=(s1,yh1)                     | MFMA (A1,B0)
    //     wait DMA(kt+1) + own reads, barrier        <- one barrier per K tile
    //     A waves: DMA(kt+2) -> stage(kt) | read A0,B0 of kt+1 | MFMA (A1,B1)
    // Each SIMD holds one A wave and one B wave; the older (A) wave wins the MFMA arbitration, so the B wave is starved
    // during the first group behind the barrier anyway and issues its requests there (weights: L2 hits, short latency)
    // instead of together with the A wave in the last group, where both stalled on DMA issue at once (-3 %).
    // All waits are explicit (asm s_waitcnt + raw s_barrier): __syncthreads() would drain the LDS-DMA queue
    // (vmcnt(0)) wherever the compiler places it, which serialises DMA and MFMA at one workgroup per CU.
    auto issue_tile = [&](int kt, int stage, int row0) {
        if (HOS_ABLATE_DMA) return;
#pragma unroll
        for (int q = 0; q < QMAX; ++q) if (q < nq) issue_dma(q, kt, stage, row0);
    };

#ifdef HOS_TRACE   // block timeline: entry / loop start / loop end / exit of workgroups 0 and 300 (second round)
// (slots 0 and 3 also record the constant 100 MHz counter: shader cycles / wall time = the effective clock of the launch)
#define HOS_BSTAMP(slot) do { if ((blockIdx.x == 0 || blockIdx.x == 300) && lane == 0 && a.f32.aux) { \
        reinterpret_cast<long long*>(a.f32.aux)[256 + ((blockIdx.x ? 1 : 0) * 8 + wave) * 4 + (slot)] = clock64(); \
        if ((slot) == 0 || (slot) == 3) reinterpret_cast<long long*>(a.f32.aux)[320 + ((blockIdx.x ? 1 : 0) * 8 + wave) * 2 + ((slot) ? 1 : 0)] = wall_clock64(); } } while (0)
#else
#define HOS_BSTAMP(slot) do {} while (0)
#endif
#ifdef HOS_TRACE2  // per-tile timeline of workgroups 0 and 100, waves 0 and 4: K loop start / end, epilogue end, next tile ready
    int tile_no = 0;
#define HOS_TSTAMP(slot) do { if ((blockIdx.x == 0 || blockIdx.x == 100) && lane == 0 && (wave & 3) == 0 && a.f32.aux && tile_no < 16) \
        reinterpret_cast<long long*>(a.f32.aux)[(((blockIdx.x ? 1 : 0) * 2 + (wave >> 2)) * 16 + tile_no) * 4 + (slot)] = clock64(); } while (0)
#define HOS_WSTAMP(slot) do { if ((blockIdx.x == 0 || blockIdx.x == 100) && lane == 0 && (wave & 3) == 0 && a.f32.aux) { \
        reinterpret_cast<long long*>(a.f32.aux)[256 + (((blockIdx.x ? 1 : 0) * 2 + (wave >> 2)) * 2 + (slot)) * 2] = wall_clock64(); \
        reinterpret_cast<long long*>(a.f32.aux)[256 + (((blockIdx.x ? 1 : 0) * 2 + (wave >> 2)) * 2 + (slot)) * 2 + 1] = clock64(); } } while (0)
// every workgroup: wall clock at entry / exit and its XCC id (aux[512 + 4 b ..])
#define HOS_ASTAMP(slot) do { if (lane == 0 && wave == 0 && a.f32.aux && blockIdx.x < 4096) { \
        reinterpret_cast<long long*>(a.f32.aux)[512 + 4 * blockIdx.x + (slot)] = wall_clock64(); \
        if ((slot) == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
                           unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); \
                           reinterpret_cast<long long*>(a.f32.aux)[512 + 4 * blockIdx.x + 2] = (long long)(xcc & 0xf) | ((long long)hwid << 8); } } } while (0)
#else
#define HOS_TSTAMP(slot) do {} while (0)
#define HOS_WSTAMP(slot) do {} while (0)
#define HOS_ASTAMP(slot) do {} while (0)
#endif
    HOS_BSTAMP(0);
    HOS_WSTAMP(0);
    HOS_ASTAMP(0);
    issue_tile(kt_begin, 0, g_row0);
    if (kt_begin + 1 < kt_end) issue_tile(kt_begin + 1, 1, g_row0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    HOS_READ_A(a0h, a0l, 0u, 0);
    HOS_READ_B(b0h, b0l, 0u, 0, 0);

#ifdef HOS_TRACE   // timing experiment: lane 0 of every wave of workgroup 0 stamps the phase boundaries of K tiles 8..11
    long long* const trbuf = reinterpret_cast<long long*>(a.f32.aux);
#define HOS_STAMP(slot) do { if (blockIdx.x == 0 && first_tile && lane == 0 && trbuf && kt >= kt_begin + 8 && kt < kt_begin + 12) trbuf[(wave * 4 + (kt - kt_begin - 8)) * 8 + (slot)] = clock64(); } while (0)
#else
#define HOS_STAMP(slot) do {} while (0)
#endif
    // One quarter = NM MFMAs (3 products x TM x TH tiles) with the LDS reads of the NEXT quarter (and the DMA requests of a
    // later tile) issued BETWEEN them: `fill(i)` runs right behind MFMA i.  A wave issues
    // in order, so anything placed in front of an MFMA group delays it; placed between MFMAs it costs nothing
    // while the matrix pipe is busy.  Eight back-to-back global_load_lds stalled a wave for 600-1900 cycles
    // (the CU's vector-memory path takes ~16 cycles per request and all eight waves queue up at once).
    constexpr int NM = 3 * TM * TH;
    // (Round 5, measured and dropped -- profiles/r05_gemmp_order_ablation.txt: an A-fragment-stationary and a B-fragment-stationary
    // order of the NM MFMAs changed nothing, 752 / 745 / 683-687 us against 753 / 742 / 683 for fwd / dgrad / wgrad.)
#define HOS_ORDER_DECODE(i) const int pr = (i) / (TM * TH), x = ((i) % (TM * TH)) / TH, y = (i) % TH
#define HOS_GROUP(AH, AL, BH, BL, YH, FILL)                                                               \
    do {                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        _Pragma("unroll") for (int x = 0; x < TM; ++x) { tie(AH[x]); tie(AL[x]); }                        \
        _Pragma("unroll") for (int y = 0; y < TH; ++y) { tie(BH[y]); tie(BL[y]); }                        \
        if (YH == 0) db_acc(AH, AL);                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                    \
        _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                  \
            HOS_ORDER_DECODE(i);                                                                          \
            if (!HOS_ABLATE_MFMA) {                                                                       \
                if (pr == 0)      acc[x][(YH) * TH + y] = pmfma(AL[x].v, BH[y].v, acc[x][(YH) * TH + y]); \
                else if (pr == 1) acc[x][(YH) * TH + y] = pmfma(AH[x].v, BL[y].v, acc[x][(YH) * TH + y]); \
                else              acc[x][(YH) * TH + y] = pmfma(AH[x].v, BH[y].v, acc[x][(YH) * TH + y]); \
            }                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            FILL(i);                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                            \
        }                                                                                                 \
        __builtin_amdgcn_s_setprio(0);                                                                    \
    } while (0)

    unsigned so = 0;                     // byte offset of the stage holding the K tile being multiplied
    bool first_tile = true;
    HOS_BSTAMP(1);
  for (;;) {                             // ---- output tiles of this workgroup (one unless PERSIST) ----
    // the tile behind this one: its first K tiles are requested by the last two iterations of this tile's loop
    const int vnext = vb + (int)gridDim.x;
    const bool has_next = PERSIST && vnext < nb;
    int n_row0 = 0;
    if (has_next) { decode(vnext); n_row0 = isB ? tn_i * BN : tm_i * PBM; }
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    // What the plane epilogues read from global memory -- the bias of this lane's columns (FWD), this lane's dword of every
    // block's ReLU bit mask (DGRAD) -- is requested HERE, as asm loads the compiler does not track: they land under the K loop
    // (every iteration ends in an explicit vmcnt(0)) and the epilogue issues no load at all.  A compiler-visible load or LDS
    // access anywhere in this loop nest makes hipcc put s_waitcnt vmcnt(0) at the top of every K iteration (it cannot see the
    // asm waits and assumes the LDS-DMA writes may alias), which drains the DMA queue once per K tile.
    float bias_r[TN];
    uint32_t bw[TM][TN / 2];
    bw_ok_mask = 0xffffffffu;
    if constexpr (EPI == PEPI_PLANES_FWD) {
        const float* const bp = a.bias != nullptr ? a.bias : reinterpret_cast<const float*>(a.B);
        const int nmax = a.bias != nullptr ? a.N - 1 : 0;
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int col = j0 + wn * (TN * 32) + y * 32 + l31;
            const float* const q = bp + (col < nmax ? col : nmax);
            asm volatile("global_load_dword %0, %1, off" : "=v"(bias_r[y]) : "v"(q));
        }
    }
    if constexpr (EPI == PEPI_PLANES_DGRAD) {
        const uint32_t* const bp = a.bits != nullptr ? a.bits : reinterpret_cast<const uint32_t*>(a.B);
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int yp = 0; yp < TN / 2; ++yp) {
                const int rb = (i0 + wm * (TM * 32) + x * 32) >> 5, cbk = (j0 + wn * (TN * 32) + yp * 64) >> 6;
                const bool ok = a.bits != nullptr && rb * 32 < a.M && cbk < a.bits_nb;
                const uint32_t* const q = bp + (ok ? ((size_t)rb * a.bits_nb + cbk) * 64 + lane : (size_t)0);
                asm volatile("global_load_dword %0, %1, off" : "=v"(bw[x][yp]) : "v"(q));
                if (!ok) bw_ok_mask &= ~(1u << (x * (TN / 2) + yp));
            }
    }
    HOS_TSTAMP(0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const unsigned sn = STAGE - so;
        const bool more1 = kt + 1 < kt_end;
        const int dstage = so ? 1 : 0;
        // K tile v+1 / v+2 of this workgroup's stream: inside this output tile, or the first / second of the next one
        // (the one-tile form keeps kt + 1 / kt + 2 as plain induction values: with the selects below in the address arithmetic
        // of every request the WGRAD launch was 5 % slower, 669 -> 706 us at [1024,1024,131072], same box)
        bool dma1, dma2; int kt1, row1, kt2, row2;
        if constexpr (PERSIST) {
            const bool w1 = kt + 1 >= kt_end, w2 = kt + 2 >= kt_end;
            dma1 = !w1 || has_next; dma2 = !w2 || has_next;
            kt1 = w1 ? kt + 1 - kt_end : kt + 1; row1 = w1 ? n_row0 : g_row0;
            kt2 = w2 ? kt + 2 - kt_end : kt + 2; row2 = w2 ? n_row0 : g_row0;
        } else {
            dma1 = kt + 1 < kt_end; dma2 = kt + 2 < kt_end;
            kt1 = kt + 1; kt2 = kt + 2; row1 = row2 = g_row0;
        }
        const bool b_issue = !HOS_ABLATE_DMA && isB && dma1 && (PERSIST ? !(first_tile && kt == kt_begin) : kt > kt_begin);     // (the prologue requested tile 1)
        const bool a_issue = !HOS_ABLATE_DMA && !isB && dma2;
        HOS_STAMP(0);
        auto fill1 = [&](int i) {                // under (A0,B0): B1 = (s0, yh1); B waves: the DMA of tile v+1
            if (i == 1) HOS_READ_B1(b1h, b1l, so, 0, 1, 0);
            if (i == 3) HOS_READ_B1(b1h, b1l, so, 0, 1, 1);
            constexpr int D0 = NM / 3, ND = NM - D0;          // DMA slots: the last two thirds of the group
            if (b_issue && i >= D0) {
#pragma unroll
                for (int q = (i - D0) * QMAX / ND; q < (i - D0 + 1) * QMAX / ND; ++q) if (q < nq) issue_dma(q, kt1, dstage ^ 1, row1);
            }
        };
        HOS_GROUP(a0h, a0l, b0h, b0l, 0, fill1);
        HOS_STAMP(1);
        auto fill2 = [&](int i) {                // under (A0,B1): A1 = (s1), B0 = (s1, yh0)
            if (i == 0) HOS_READ_A1(a1h, a1l, so, 1, 0);
            if (i == 1) HOS_READ_A1(a1h, a1l, so, 1, 1);
            if (i == 2) HOS_READ_B1(b0h, b0l, so, 1, 0, 0);
            if (i == 3) HOS_READ_B1(b0h, b0l, so, 1, 0, 1);
        };
        HOS_GROUP(a0h, a0l, b1h, b1l, 1, fill2);
        HOS_STAMP(2);
        auto fill3 = [&](int i) {                // under (A1,B0): B1 = (s1, yh1)
            if (i == 1) HOS_READ_B1(b1h, b1l, so, 1, 1, 0);
            if (i == 3) HOS_READ_B1(b1h, b1l, so, 1, 1, 1);
        };
        HOS_GROUP(a1h, a1l, b0h, b0l, 0, fill3);
        HOS_STAMP(3);
        // this wave's share of tile v+1 has landed and its reads of this stage are complete; after the barrier
        // that holds for every wave, so the stage may be refilled and the other one read
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        HOS_STAMP(4);
        asm volatile("s_barrier" ::: "memory");
        HOS_STAMP(5);
        auto fill4 = [&](int i) {                // under (A1,B1): A0, B0 of tile kt+1, A waves: the DMA of tile v+2
            if (more1) {
                if (i == 0) HOS_READ_A1(a0h, a0l, sn, 0, 0);
                if (i == 1) HOS_READ_A1(a0h, a0l, sn, 0, 1);
                if (i == 2) HOS_READ_B1(b0h, b0l, sn, 0, 0, 0);
                if (i == 3) HOS_READ_B1(b0h, b0l, sn, 0, 0, 1);
            }
            constexpr int D0 = NM / 3, ND = NM - D0;
            if (a_issue && i >= D0) {
#pragma unroll
                for (int q = (i - D0) * QMAX / ND; q < (i - D0 + 1) * QMAX / ND; ++q) if (q < nq) issue_dma(q, kt2, dstage, row2);
            }
        };
        HOS_GROUP(a1h, a1l, b1h, b1l, 1, fill4);
        HOS_STAMP(7);
        so = sn;
    }
    HOS_BSTAMP(2);
    HOS_TSTAMP(1);
    // Here: every wave has passed the last iteration's barrier, so nobody reads the stage of the last K tile any more
    // (`STAGE - so`).  PERSIST: the A waves' requests for the next tile's second K tile are in flight into the A half of that
    // stage, the next tile's first K tile has landed in stage `so`; the B half of the free stage is the epilogue's staging memory.

    // ---------------------------------------------------------------------------------------- epilogues
    if constexpr (EPI == PEPI_F32 || EPI == PEPI_WGRAD) {
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int y = 0; y < TN; ++y) {
                if constexpr (EPI == PEPI_WGRAD) {
                    if (a.f32.aux != nullptr) {
                        // split-K partial tile -> slab `split` of the workspace with plain 16-byte stores; the slabs
                        // are summed into dW by wgrad_reduce_kernel (fp32 atomics ran at ~1 TB/s: 70 us for 65 MB)
                        GemmArgs w = a.f32;
                        w.C = a.f32.aux + (size_t)split * a.f32.M * a.f32.aux_col;
                        w.ldc = a.f32.aux_col; w.bias = nullptr; w.epi = HOS_EPI_NONE; w.mask = nullptr; w.aux = nullptr;
                        gemm_epilogue_tile<MODE_FWD>(w, acc[x][y], i0 + wm * (TM * 32) + x * 32, j0 + wn * (TN * 32) + y * 32, lane);
                    } else {
                        f32x16 v = acc[x][y];
                        gemm_epilogue_tile<MODE_WGRAD>(a.f32, v, i0 + wm * (TM * 32) + x * 32, j0 + wn * (TN * 32) + y * 32, lane);
                    }
                } else {
                    gemm_epilogue_tile<MODE_FWD>(a.f32, acc[x][y], i0 + wm * (TM * 32) + x * 32, j0 + wn * (TN * 32) + y * 32, lane);
                }
            }
        if constexpr (EPI == PEPI_WGRAD) {
            if (do_db) {
#pragma unroll
                for (int x = 0; x < TM; ++x) {
                    const float sum = dbsum[x] + __shfl_xor(dbsum[x], 32, 64);
                    const int n = i0 + wm * (TM * 32) + x * 32 + l31;
                    if (lhi == 0 && n < a.M) __hip_atomic_fetch_add(a.f32.db + n, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    } else {
        // Plane outputs.  Every wave owns a private piece of idle stage memory.  Per block of 32 rows x (32 YB) columns:
        // each lane packs (hi, lo) of its values into one dword and writes it at [row][col] (ds_write_b32,
        // 32 consecutive dwords per half wave: conflict free); after the wave's own writes have landed every lane
        // reads 32 bytes = eight consecutive columns of one row, keeps their hi or their lo halves and stores 16 bytes.
        // ReLU mask of the backward pass: FWD can emit one BIT per output element in ACCUMULATOR layout (a.bits: per
        // 32-row x 64-column block 64 dwords, dword = lane, bit 31 - (16 yy + r) = element (row (r&3) + 8 (r>>2) + 4 (lane>>5),
        // column 32 yy + (lane&31)) of the block), and DGRAD, whose accumulators have the same layout, reads its own dword
        // back: 8 KB per tile instead of the 256 KB of the fp16 planes of the layer input (whose fetch at ~11 B/clk/CU
        // cost ~16 us at the end of every tile: 110 us of an 817 us launch at M = 131072).
        constexpr int YB = PERSIST ? 1 : 2;                       // 32-column blocks per staging round (4 / 8 KB per wave)
        if constexpr (!PERSIST) asm volatile("s_barrier" ::: "memory");       // all waves are done with the stage memory
        // the loads issued at the top of the tile have landed (the K loop waited vmcnt(0)); make that visible to the compiler
        if constexpr (EPI == PEPI_PLANES_FWD) {
#pragma unroll
            for (int y = 0; y < TN; ++y) asm volatile("" : "+v"(bias_r[y]));
        }
        if constexpr (EPI == PEPI_PLANES_DGRAD) {
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int yp = 0; yp < TN / 2; ++yp) asm volatile("" : "+v"(bw[x][yp]));
        }
        // staging memory as raw LDS addresses; every access below is an asm statement (see the note at the top of the tile)
        const unsigned stg = lds_base + (PERSIST ? (STAGE - so) + A_TILE + wave * 4096 : wave * 8192);
        const unsigned stg_w = stg + (4 * lhi * (32 * YB) + l31) * 4;                    // + row / block immediates
        const unsigned stg_r = YB == 2 ? stg + (lane >> 4) * 256 + ((lane >> 3) & 1) * 128 + (lane & 3) * 32
                                       : stg + (lane >> 3) * 128 + (lane & 3) * 32;       // + pass * 1024
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const bool dual = (EPI == PEPI_PLANES_FWD) && a.Yb != nullptr;
        const bool first = a.Y != nullptr;
        const bool legacy_mask = (EPI == PEPI_PLANES_DGRAD) && a.bits == nullptr && a.mask != nullptr;     // never on a persistent launch
        float vmax = 0.f;          // largest hidden activation: beyond the exact fp16 hi/lo range (HOS_RANGE_LIMIT)?
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int yp = 0; yp < TN / 2; ++yp) {
                const int row0 = i0 + wm * (TM * 32) + x * 32;
                const int col0 = j0 + wn * (TN * 32) + yp * 64;
                // activation (FWD) / ReLU mask (DGRAD) once, in place in the accumulators; both formats are split from that
                uint32_t mybits = 0u;            // FWD: v > 0 of this lane's 32 values of the block, value (yy, r) at bit 31 - (16 yy + r)
                uint32_t keepbits = 0xffffffffu;
                if constexpr (EPI == PEPI_PLANES_DGRAD) {
                    if (a.bits != nullptr) keepbits = ((bw_ok_mask >> (x * (TN / 2) + yp)) & 1u) ? bw[x][yp] : 0u;
                }
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    const int col = col0 + yy * 32 + l31;
                    float bcol = 0.f;
                    if constexpr (EPI == PEPI_PLANES_FWD) bcol = a.bias != nullptr ? bias_r[2 * yp + yy] : 0.f;
                    const bool inb = col < a.N;                                        // zero the padding columns
                    // ReLU and the zeroing of padding columns as ONE clamp: [0, inf) / (-inf, inf) / [0, 0]
                    const float vlo = inb ? (a.relu ? 0.f : -__builtin_inff()) : 0.f, vhi = inb ? __builtin_inff() : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[x][2 * yp + yy][r];
                        if constexpr (EPI == PEPI_PLANES_FWD) {
                            v = __builtin_amdgcn_fmed3f(v + bcol, vlo, vhi);
                            mybits = __builtin_amdgcn_alignbit(mybits, __builtin_bit_cast(uint32_t, 0.f - v), 31);    // sign of -v: v > 0
                            if constexpr (__is_same(EIN, _Float16)) vmax = fmaxf(vmax, fabsf(v));
                        } else {
                            if (!inb) v = 0.f;
                            v = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & (uint32_t)__builtin_amdgcn_sbfe(keepbits, 31 - (yy * 16 + r), 1));
                        }
                        acc[x][2 * yp + yy][r] = v;
                    }
                }
#pragma unroll
                for (int fmt = 0; fmt < 2; ++fmt) {
                    if (fmt == 0 && !first) continue;
                    if (fmt == 1 && !dual) continue;
                    uint16_t* const Po = fmt == 0 ? a.Y : a.Yb;
                    const int ldo = fmt == 0 ? a.ldy : a.ldyb;
#pragma unroll
                    for (int y0 = 0; y0 < 2; y0 += YB) {
#pragma unroll
                        for (int yy = y0; yy < y0 + YB; ++yy)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[x][2 * yp + yy][r];
                                const uint32_t p = (EPI == PEPI_PLANES_FWD && fmt == 0) ? split_pack_pk(v, (EIN)0) : split_pack_pk(v, (__bf16)0);
                                asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(stg_w), "v"(p),
                                             "n"((((r & 3) + 8 * (r >> 2)) * (32 * YB) + (yy - y0) * 32) * 4) : "memory");
                            }
                        // 16-byte stores, whole lines per instruction: 8 lanes per (row, 32-column block) -- lanes 0-3 the hi
                        // half of the line (8 columns each), lanes 4-7 the lo half; a wave instruction writes 8 complete
                        // 128-byte lines.  Four passes (8 reads of 16 bytes) per wait.
#pragma unroll
                        for (int p0 = 0; p0 < 4 * YB; p0 += 4) {
                            u32x4 w0[4], w1[4];
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (the wave's own writes have landed)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w0[k]) : "v"(stg_r), "n"((p0 + k) * 1024) : "memory");
                                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w1[k]) : "v"(stg_r), "n"((p0 + k) * 1024 + 16) : "memory");
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(w0[k]), "+v"(w1[k]));
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int idx = (p0 + k) * 64 + lane;
                                const int rl = YB == 2 ? idx >> 4 : idx >> 3, cb = YB == 2 ? (idx >> 3) & 1 : 0, half = (idx >> 2) & 1, c8 = (idx & 3) * 8;
                                const int row = row0 + rl, col = col0 + (y0 + cb) * 32;
                                const uint32_t sel = half ? 0x07060302u : 0x05040100u;
                                uint4 o4 = make_uint4(__builtin_amdgcn_perm(w0[k].y, w0[k].x, sel), __builtin_amdgcn_perm(w0[k].w, w0[k].z, sel),
                                                      __builtin_amdgcn_perm(w1[k].y, w1[k].x, sel), __builtin_amdgcn_perm(w1[k].w, w1[k].z, sel));
                                if constexpr (EPI == PEPI_PLANES_DGRAD && !PERSIST) {
                                    if (legacy_mask && row < a.M && col < a.ldmask) {
                                        const uint4 mk = *reinterpret_cast<const uint4*>(a.mask + (size_t)row * (2 * a.ldmask) + (col >> 5) * 64 + c8);
                                        const uint32_t mw[4] = {mk.x, mk.y, mk.z, mk.w};
                                        uint32_t* ov = reinterpret_cast<uint32_t*>(&o4);
#pragma unroll
                                        for (int e = 0; e < 4; ++e) {          // 16-bit hi part of the layer input: x > 0 ?
                                            const uint32_t lo16 = mw[e] & 0xffffu, hi16 = mw[e] >> 16;
                                            uint32_t keep = 0u;
                                            if (!(lo16 & 0x8000u) && lo16 != 0u) keep |= 0x0000ffffu;
                                            if (!(hi16 & 0x8000u) && hi16 != 0u) keep |= 0xffff0000u;
                                            ov[e] &= keep;
                                        }
                                    }
                                }
                                if (row < a.M && col < ldo)
                                    *reinterpret_cast<uint4*>(Po + (size_t)row * (2 * ldo) + (col >> 5) * 64 + half * 32 + c8) = o4;
                            }
                        }
                    }
                }
                if (EPI == PEPI_PLANES_FWD && a.bits != nullptr && first && row0 < a.M && (col0 >> 6) < a.bits_nb)
                    a.bits[((size_t)(row0 >> 5) * a.bits_nb + (col0 >> 6)) * 64 + lane] = mybits;
            }
        if constexpr (EPI == PEPI_PLANES_FWD && __is_same(EIN, _Float16)) {
            if (a.f32.range_flag != nullptr && __builtin_amdgcn_ballot_w64(vmax > HOS_RANGE_LIMIT) != 0 && lane == 0)
                atomicOr(a.f32.range_flag, 1u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    HOS_TSTAMP(2);
    if (!has_next) break;
    // ---- next output tile: its first K tile is in stage `so`; the staging memory becomes a DMA target again ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    HOS_READ_A(a0h, a0l, so, 0);
    HOS_READ_B(b0h, b0l, so, 0, 0);
    vb = vnext;
    i0 = tm_i * PBM; j0 = tn_i * BN;           // (decode(vnext) above left the next tile's coordinates in tm_i / tn_i)
    g_row0 = n_row0;
    first_tile = false;
    HOS_TSTAMP(3);
#ifdef HOS_TRACE2
    ++tile_no;
#endif
  }
#ifdef HOS_TRACE2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    HOS_WSTAMP(1);
    HOS_ASTAMP(1);
#ifdef HOS_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    HOS_BSTAMP(3);
#undef HOS_BSTAMP
#undef HOS_TSTAMP
#undef HOS_WSTAMP
#undef HOS_ASTAMP
#undef HOS_GROUP
#undef HOS_ORDER_DECODE
#undef HOS_STAMP
#undef HOS_READ_A
#undef HOS_READ_A1
#undef HOS_READ_B
#undef HOS_READ_B1
#undef HOS_RD128
#undef HOS_RDTR
}

// number of compute units (persistent grids), queried once
inline int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

template <int BN, int EPI, typename EIN, bool TR, bool PERSIST>
int launchp_impl(PArgs& a, int grid, hipStream_t stream) {
    constexpr size_t smem = 2 * (PBM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemmp_kernel<BN, EPI, EIN, TR, PERSIST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemmp_kernel<BN, EPI, EIN, TR, PERSIST>), dim3(grid), dim3(PNT), smem, stream, a);
    return hos_launch_status();
}

template <int BN, int EPI, typename EIN, bool TR>
int launchp(PArgs& a, int splits, hipStream_t stream) {
    a.tiles_m = hos_cdiv(a.M, PBM);
    a.tiles_n = hos_cdiv(a.N, BN);
    if (EPI == PEPI_WGRAD) {
        a.kt_per_split = hos_cdiv(a.nk, splits);          // splits chosen by wgrad_splits()
        splits = hos_cdiv(a.nk, a.kt_per_split);
    } else {
        splits = 1;
        a.kt_per_split = a.nk;
    }
    a.total = a.tiles_m * a.tiles_n * splits;
    if constexpr (!TR && BN == 256 && HOS_GEMMP_PERSIST && (EPI == PEPI_PLANES_FWD || EPI == PEPI_PLANES_DGRAD)) {
        static const int env_persist = getenv("HOS_GEMMP_PERSIST") ? atoi(getenv("HOS_GEMMP_PERSIST")) : 1;      // (only in a -DHOS_GEMMP_PERSIST=1 build)
        const int cus = cu_count();
        const bool legacy_mask = EPI == PEPI_PLANES_DGRAD && a.bits == nullptr && a.mask != nullptr;
        if (env_persist && a.nk >= 2 && a.total > cus && !legacy_mask) return launchp_impl<BN, EPI, EIN, TR, true>(a, cus, stream);
    }
    return launchp_impl<BN, EPI, EIN, TR, false>(a, a.total, stream);
}

// element offset of (row r, column c) in an interleaved-planes array with `ld` logical columns: hi there, lo 32 further
__device__ __forceinline__ size_t pl_off(int r, int c, int ld) { return (size_t)r * (2 * ld) + (c >> 5) * 64 + (c & 31); }

// fp32 [R][lds] -> interleaved planes, row-major [R][ldo] (padding columns [C, ldo) zeroed) and/or transposed [C][ldt]
template <typename E>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int lds, int R, int C,
                                                           uint16_t* __restrict__ out, int ldo,
                                                           uint16_t* __restrict__ outT, int ldt) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        const float v = (r < R && c < C) ? src[(size_t)r * lds + c] : 0.f;
        tile[ty + 8 * k][tx] = v;
        if (out != nullptr && r < R && c < ldo) {
            uint16_t h, l;
            split1<E>(v, h, l);
            const size_t o = pl_off(r, c, ldo);
            out[o] = h;
            out[o + 32] = l;
        }
    }
    if (outT == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;       // transposed: row index c, column r
        if (c < C && r < ldt) {
            uint16_t h, l;
            split1<E>(r < R ? tile[tx][ty + 8 * k] : 0.f, h, l);
            const size_t o = pl_off(c, r, ldt);
            outT[o] = h;
            outT[o + 32] = l;
        }
    }
}

// Up to SPB_MAX transposed splits in ONE launch (blockIdx.z = job): the backward pass needs the transposed bf16 planes of every
// weight of an MLP, 5-9 small matrices per MLP call -- 29 launches of ~5 us per stage-1 step.
constexpr int SPB_MAX = 12;
struct SplitJob { const float* src; int lds, R, C; uint16_t* outT; int ldt; };
struct SplitBatch { SplitJob j[SPB_MAX]; };
__global__ __launch_bounds__(256) void split_planes_T_batch_kernel(const SplitBatch b) {
    const SplitJob J = b.j[blockIdx.z];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    if (c0 >= J.C || r0 >= (J.ldt > J.R ? J.ldt : J.R)) return;
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < J.R && c < J.C) ? J.src[(size_t)r * J.lds + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;       // transposed: row index c, column r
        if (c < J.C && r < J.ldt) {
            uint16_t h, l;
            split1<__bf16>(r < J.R ? tile[tx][ty + 8 * k] : 0.f, h, l);
            const size_t o = pl_off(c, r, J.ldt);
            J.outT[o] = h;
            J.outT[o + 32] = l;
        }
    }
}

// fp32 [R][lds] -> fp16 planes and/or bf16 planes in one pass (row-major, 4 elements per thread)
__global__ __launch_bounds__(256) void split_planes2_kernel(const float* __restrict__ src, int lds, int R, int C,
                                                            uint16_t* __restrict__ p16, int ld16,
                                                            uint16_t* __restrict__ pb, int ldb) {
    const int ldmax = ld16 > ldb ? ld16 : ldb;
    const int groups = ldmax >> 2;
    const size_t total = (size_t)R * groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / groups), c = (int)(i % groups) * 4;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (c + k < C) ? src[(size_t)r * lds + c + k] : 0.f;
        if (p16 != nullptr && c < ld16) {
            uint32_t p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = split_pack<_Float16>(v[k]);
            const size_t o = pl_off(r, c, ld16);
            *reinterpret_cast<uint2*>(p16 + o) = make_uint2((p[0] & 0xffffu) | (p[1] << 16), (p[2] & 0xffffu) | (p[3] << 16));
            *reinterpret_cast<uint2*>(p16 + o + 32) = make_uint2((p[0] >> 16) | (p[1] & 0xffff0000u), (p[2] >> 16) | (p[3] & 0xffff0000u));
        }
        if (pb != nullptr && c < ldb) {
            uint32_t p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = split_pack<__bf16>(v[k]);
            const size_t o = pl_off(r, c, ldb);
            *reinterpret_cast<uint2*>(pb + o) = make_uint2((p[0] & 0xffffu) | (p[1] << 16), (p[2] & 0xffffu) | (p[3] << 16));
            *reinterpret_cast<uint2*>(pb + o + 32) = make_uint2((p[0] >> 16) | (p[1] & 0xffff0000u), (p[2] >> 16) | (p[3] & 0xffff0000u));
        }
    }
}

// dW[n][k] += sum_s ws[s][n][k]   (ws slabs [N][wsld], wsld % 4 == 0; dW row stride ldw)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int N, int K, int wsld,
                                                           float* __restrict__ dW, int ldw) {
    const int groups = wsld >> 2;
    const size_t total = (size_t)N * groups, slab = (size_t)N * wsld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / groups), k = (int)(i % groups) * 4;
        const float* p = ws + (size_t)n * wsld + k;
        // eight independent 16-byte loads per round: one memory latency per round instead of one per slab
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j0 = 0; j0 < splits; j0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (j0 + u < splits) ? *reinterpret_cast<const float4*>(p + (size_t)(j0 + u) * slab) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        float* d = dW + (size_t)n * ldw + k;
        const float vals[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < K) d[e] += vals[e];
    }
}

// split-K factor of the weight gradient: one workgroup per CU, at least 8 K tiles per split
inline int wgrad_splits(int tiles, int nk, int requested) {
    static const int env_splits = getenv("HOS_WGRAD_SPLITS") ? atoi(getenv("HOS_WGRAD_SPLITS")) : 0;
    int splits = env_splits > 0 ? env_splits : requested;
    if (splits <= 0) {
        splits = 256 / tiles > 0 ? 256 / tiles : 1;       // never more workgroups than CUs: one over costs a whole second round
        if (splits > nk / 8) splits = nk / 8 > 0 ? nk / 8 : 1;
    }
    if (splits > nk) splits = nk;
    const int per = hos_cdiv(nk, splits);
    return hos_cdiv(nk, per);
}

inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int hos_split_planes(const float* src, int lds, int R, int C, int dtype, void* out, int ldo,
                                void* outT, int ldt, hos_stream_t stream) {
    if (!src || R <= 0 || C <= 0 || (!out && !outT)) return HOS_E_ARG;
    if ((out && (ldo & 31)) || (outT && (ldt & 31))) return HOS_E_ALIGN;
    const int cols = out ? (ldo > C ? ldo : C) : C;
    dim3 grid(hos_cdiv(cols, 32), hos_cdiv(outT ? (ldt > R ? ldt : R) : R, 32));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == 0)
        hipLaunchKernelGGL(split_planes_kernel<_Float16>, grid, dim3(256), 0, s, src, lds, R, C, (uint16_t*)out, ldo, (uint16_t*)outT, ldt);
    else
        hipLaunchKernelGGL(split_planes_kernel<__bf16>, grid, dim3(256), 0, s, src, lds, R, C, (uint16_t*)out, ldo, (uint16_t*)outT, ldt);
    return hos_launch_status();
}

extern "C" int hos_split_planes2(const float* src, int lds, int R, int C, void* p16, int ld16, void* pb, int ldb,
                                 hos_stream_t stream) {
    if (!src || R <= 0 || C <= 0 || (!p16 && !pb)) return HOS_E_ARG;
    if ((p16 && (ld16 & 31)) || (pb && (ldb & 31))) return HOS_E_ALIGN;
    const int ldmax = (p16 ? ld16 : 0) > (pb ? ldb : 0) ? ld16 : ldb;
    const size_t total = (size_t)R * (ldmax >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_planes2_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds, R, C,
                       (uint16_t*)p16, p16 ? ld16 : 0, (uint16_t*)pb, pb ? ldb : 0);
    return hos_launch_status();
}

// n <= 12 transposed bf16 splits in one launch: job i = fp32 src[i] [R[i]][lds[i]] (C[i] columns used) -> planes outT[i] [C[i]][ldt[i]]
// (columns [R, ldt) zeroed).  The arrays are host arrays, read during the call.
extern "C" int hos_split_planes_t_batch(int n, const float* const* src, const int* lds, const int* R, const int* C, void* const* outT,
                                        const int* ldt, hos_stream_t stream) {
    if (n <= 0 || n > SPB_MAX || !src || !lds || !R || !C || !outT || !ldt) return HOS_E_ARG;
    SplitBatch b{};
    int gx = 0, gy = 0;
    for (int i = 0; i < SPB_MAX; ++i) {
        const int k = i < n ? i : 0;                     // unused slots repeat job 0 (never launched: grid.z = n)
        if (!src[k] || !outT[k] || R[k] <= 0 || C[k] <= 0) return HOS_E_ARG;
        if (ldt[k] & 31) return HOS_E_ALIGN;
        b.j[i] = SplitJob{src[k], lds[k], R[k], C[k], (uint16_t*)outT[k], ldt[k]};
        if (i < n) {
            const int rows = ldt[k] > R[k] ? ldt[k] : R[k];
            gx = hos_cdiv(C[k], 32) > gx ? hos_cdiv(C[k], 32) : gx;
            gy = hos_cdiv(rows, 32) > gy ? hos_cdiv(rows, 32) : gy;
        }
    }
    hipLaunchKernelGGL(split_planes_T_batch_kernel, dim3(gx, gy, n), dim3(256), 0, static_cast<hipStream_t>(stream), b);
    return hos_launch_status();
}

namespace {
// FWD on operands of element type E: fp16 planes (22 mantissa bits per value: the proposal MLPs, whose densities steer the
// resampling and must reproduce the reference's sample indices) or bf16 planes (16 bits: the NeRF MLP, whose outputs are only
// rendered -- 2.9e-5 RGB L-inf on the reference model, SURVEY 7.1 -- ONE activation format for forward, dgrad mask and wgrad).
template <typename E>
int linearp_fwd_impl(const void* A, int lda, int K0, const void* A1, int lda1, int K1, const void* W, int ldw,
                     const float* bias, int M, int N, int relu, void* Y, int ldy, void* Yb, int ldyb, void* relu_bits,
                     float* C, int ldc, int epilogue, float* aux, int aux_col, float p0, hos_stream_t stream) {
    if (!A || !W || M <= 0 || N <= 0 || K0 <= 0 || K1 < 0) return HOS_E_ARG;
    if (K1 > 0 && !A1) return HOS_E_ARG;
    if ((K0 % PBK) || (K1 % PBK)) return HOS_E_SHAPE;
    if ((lda & 31) || (ldw & 31) || (K1 > 0 && (lda1 & 31))) return HOS_E_ALIGN;
    if (!al16p(A) || !al16p(W) || (K1 > 0 && !al16p(A1))) return HOS_E_ALIGN;
    const bool planes_out = (Y != nullptr) || (Yb != nullptr);
    if (planes_out && ((Y && (ldy & 31)) || (Yb && (ldyb & 31)))) return HOS_E_ALIGN;
    if (!planes_out && !C && epilogue != HOS_EPI_DENSITY) return HOS_E_ARG;
    if (relu_bits && (!Y || !relu || ((uintptr_t)relu_bits & 3u))) return HOS_E_ARG;
    PArgs a{};
    a.A = (const uint16_t*)A; a.lda = lda; a.kt0 = K0 / PBK;
    a.A1 = (const uint16_t*)A1; a.lda1 = lda1;
    a.B = (const uint16_t*)W; a.ldb = ldw;
    a.M = M; a.N = N; a.nk = (K0 + K1) / PBK;
    a.bias = bias; a.relu = relu;
    a.Y = (uint16_t*)Y; a.ldy = ldy; a.Yb = (uint16_t*)Yb; a.ldyb = ldyb;
    a.bits = (uint32_t*)relu_bits; a.bits_nb = hos_cdiv(ldy, 64);
    a.f32.C = C; a.f32.ldc = ldc; a.f32.M = M; a.f32.N = N; a.f32.bias = bias; a.f32.aux = aux; a.f32.aux_col = aux_col;
    a.f32.range_flag = hos_range_flag_ptr();
    a.f32.p0 = p0; a.f32.epi = epilogue;
    if (epilogue == HOS_EPI_RESIDUAL) { a.f32.mask = aux; a.f32.ldmask = aux_col; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = N > 128;
    if (planes_out) return wide ? launchp<256, PEPI_PLANES_FWD, E, false>(a, 1, s) : launchp<128, PEPI_PLANES_FWD, E, false>(a, 1, s);
    // N = 256 q + r with a short remainder (the NeRF head: 256 bottleneck columns + 1 density column): the last column tile
    // of a 256-wide launch would be a whole 256 x 256 tile for r columns -- as much MFMA work and A traffic again as the
    // first q.  The remainder goes to the 128-wide tile instead (fp32 epilogues only: their fields are all relative to the
    // first column, so the second launch is the same call on shifted pointers).  [131072,257,1024]: 587 -> 280 us.
    const int rem = N % 256;
    if (N > 256 && rem > 0 && rem <= 128 && (epilogue == HOS_EPI_NONE || epilogue == HOS_EPI_RELU || epilogue == HOS_EPI_NERF_HEAD)) {
        const int n0 = N - rem;
        PArgs b = a;
        a.N = n0; a.f32.N = n0;
        a.f32.aux_col = aux_col < n0 ? aux_col : -1;           // a column index of the other launch never matches
        b.B = a.B + (size_t)n0 * (2 * ldw);
        b.N = rem; b.f32.N = rem;
        if (bias) { b.bias = bias + n0; b.f32.bias = bias + n0; }
        if (C) b.f32.C = C + n0;
        b.f32.aux_col = aux_col >= n0 ? aux_col - n0 : -1;
        const int rc = launchp<256, PEPI_F32, E, false>(a, 1, s);
        return rc != 0 ? rc : launchp<128, PEPI_F32, E, false>(b, 1, s);
    }
    return wide ? launchp<256, PEPI_F32, E, false>(a, 1, s) : launchp<128, PEPI_F32, E, false>(a, 1, s);
}
}  // namespace

extern "C" int hos_linearp_fwd(const void* A, int lda, int K0, const void* A1, int lda1, int K1, const void* W, int ldw,
                               const float* bias, int M, int N, int relu, void* Y, int ldy, void* Yb, int ldyb, void* relu_bits,
                               float* C, int ldc, int epilogue, float* aux, int aux_col, float p0,
                               hos_stream_t stream) {
    return linearp_fwd_impl<_Float16>(A, lda, K0, A1, lda1, K1, W, ldw, bias, M, N, relu, Y, ldy, Yb, ldyb, relu_bits, C, ldc, epilogue,
                                      aux, aux_col, p0, stream);
}

// The same layer on bf16 planes throughout: A, A1, W and the ONE plane output Y (with its optional ReLU bit mask) are bf16 planes.
extern "C" int hos_linearp_fwd_b(const void* A, int lda, int K0, const void* A1, int lda1, int K1, const void* W, int ldw,
                                 const float* bias, int M, int N, int relu, void* Y, int ldy, void* relu_bits,
                                 float* C, int ldc, int epilogue, float* aux, int aux_col, float p0, hos_stream_t stream) {
    return linearp_fwd_impl<__bf16>(A, lda, K0, A1, lda1, K1, W, ldw, bias, M, N, relu, Y, ldy, nullptr, 0, relu_bits, C, ldc, epilogue,
                                    aux, aux_col, p0, stream);
}

extern "C" int hos_linearp_dgrad(const void* dZ, int lddz, const void* WT, int ldwt, int Npad, const void* mask, int ldmask,
                                 const void* mask_bits, int M, int K, void* dX, int lddx, hos_stream_t stream) {
    if (!dZ || !WT || !dX || M <= 0 || K <= 0 || Npad <= 0) return HOS_E_ARG;
    if (Npad % PBK) return HOS_E_SHAPE;
    if ((lddz & 31) || (ldwt & 31) || (lddx & 31) || ((mask || mask_bits) && (ldmask & 31))) return HOS_E_ALIGN;
    if (mask_bits && (ldmask <= 0 || ((uintptr_t)mask_bits & 3u))) return HOS_E_ARG;
    if (!al16p(dZ) || !al16p(WT)) return HOS_E_ALIGN;
    PArgs a{};
    a.A = (const uint16_t*)dZ; a.lda = lddz; a.kt0 = Npad / PBK;
    a.B = (const uint16_t*)WT; a.ldb = ldwt;
    a.M = M; a.N = K; a.nk = Npad / PBK;
    a.mask = (const uint16_t*)mask; a.ldmask = ldmask;
    a.bits = (uint32_t*)mask_bits; a.bits_nb = hos_cdiv(ldmask, 64);
    a.Y = (uint16_t*)dX; a.ldy = lddx;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return K > 128 ? launchp<256, PEPI_PLANES_DGRAD, __bf16, false>(a, 1, s) : launchp<128, PEPI_PLANES_DGRAD, __bf16, false>(a, 1, s);
}

extern "C" int hos_linearp_wgrad(const void* dZ, int lddz, const void* X, int ldx, int x_col0, float* dW, int ldw, float* db,
                                 int M, int N, int K, int splits, float* ws, long long ws_floats, hos_stream_t stream) {
    if (!dZ || !X || !dW || M <= 0 || N <= 0 || K <= 0 || x_col0 < 0) return HOS_E_ARG;
    if (M % PBK) return HOS_E_SHAPE;
    if ((lddz & 31) || (ldx & 31) || (x_col0 & 31)) return HOS_E_ALIGN;
    if (!al16p(dZ) || !al16p(X)) return HOS_E_ALIGN;
    PArgs a{};
    a.A = (const uint16_t*)dZ; a.lda = lddz; a.kt0 = M / PBK;
    a.B = (const uint16_t*)X + 2 * x_col0; a.ldb = ldx;          // column block x_col0/32 of every row
    a.M = N; a.N = K; a.nk = M / PBK;
    a.f32.C = dW; a.f32.ldc = ldw; a.f32.M = N; a.f32.N = K; a.f32.db = db;
    // 256 x 128 tiles up to K = 256: a [256,256] gradient then has two tiles x 128 splits instead of one x 256 --
    // half the atomic traffic at the same parallelism (98 -> 55 us at M = 65536)
    static const int narrow_max = getenv("HOS_WGRAD_NARROW_MAX") ? atoi(getenv("HOS_WGRAD_NARROW_MAX")) : 256;
    const bool wide = K > narrow_max;
    const int tiles = hos_cdiv(N, PBM) * hos_cdiv(K, wide ? 256 : 128);
    splits = wgrad_splits(tiles, a.nk, splits);
    // slab reduction instead of atomics when the caller lent a large enough, 16-byte aligned workspace
    const int wsld = (K + 3) & ~3;
    const bool slabs = splits > 1 && ws != nullptr && al16p(ws) && (long long)splits * N * wsld <= ws_floats;
    if (slabs) { a.f32.aux = ws; a.f32.aux_col = wsld; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = wide ? launchp<256, PEPI_WGRAD, __bf16, true>(a, splits, s) : launchp<128, PEPI_WGRAD, __bf16, true>(a, splits, s);
    if (rc != 0 || !slabs) return rc;
    const size_t total = (size_t)N * (wsld >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, splits, N, K, wsld, dW, ldw);
    return hos_launch_status();
}
