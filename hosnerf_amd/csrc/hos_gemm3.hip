// bf16x3 split-precision GEMM: fp32 in / fp32 out, products on the bf16 matrix cores.
//
//   a = a_hi + a_lo,  b = b_hi + b_lo   (bf16 round-to-nearest of x and of the residual x - hi)
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (fp32 accumulate; the dropped a_lo*b_lo term is 2^-18 relative)
//
// SURVEY 7.1 measured this mode at RGB L-inf 2.9e-5 against the reference (fp32-vs-fp64 self noise 3.1e-5),
// while plain bf16 / TF32-class inputs miss the 1e-4 budget by 10-100x.  Three v_mfma_f32_32x32x16_bf16
// replace eight v_mfma_f32_32x32x2_f32: 96 vs 512 matrix-pipe cycles per 32x32x16 block = 5.3x the fp32-MFMA
// rate; the honest roofline for this kernel is (bf16 dense peak)/3 = 833 TFLOP/s of algorithmic flops.
//
// Structure (one workgroup = 512 threads = 8 wave64 as 4(M) x 2(N), tile 256 x BN x 32):
//   * operands stay fp32 in HBM (drop-in for the fp32 kernel: same arguments, same epilogues);
//   * the split happens ONCE per tile at staging time: global float4 -> registers -> (hi, lo) bf16 -> LDS,
//     not per fragment read, so it costs ~3 VALU ops per element per workgroup;
//   * LDS holds, per operand, a hi and a lo plane [row][32 k] bf16 (64 B rows, 16-B chunks XOR-swizzled by
//     (row>>2)&3 so the ds_read_b128 fragment reads -- lane -> row, 8 consecutive k -- are conflict free);
//     double buffered: 2 x (256+BN) x 128 B = 128 KB at BN = 256, one workgroup per CU;
//   * the next tile's global loads are issued before the MFMA block of the current tile and converted/stored
//     to the other LDS buffer after it: one barrier per K tile;
//   * operands whose reduction index is the row (dgrad's W, both wgrad operands) are loaded as 4x4 micro-tiles
//     and transposed in registers, so every LDS store is still an 8-byte (4 x bf16, k-contiguous) write.
// A 256-wide tile needs (256+BN)*128 B of L2 traffic per 3072 (BN=256) matrix cycles per SIMD = 21 B/clk/CU,
// inside the ~56 B/clk/CU the L2 delivers; a 128x128 tile would need 42 B/clk/CU and starve.
#include "hos_gemm_common.h"
#include <cstdlib>

// Split element type: __bf16 (8-bit exponent: safe for gradients of any magnitude, ~2^-17 relative error per
// product) for DGRAD/WGRAD, _Float16 (11-bit mantissa: hi+lo carry 22 bits, ~2^-21 relative error -- fp32 grade)
// for the FORWARD GEMMs whose operands (features, activations, weights) are O(1e-3..1e3) by construction.
template <typename E> struct Vec { typedef E x8 __attribute__((ext_vector_type(8))); typedef E x4 __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ f32x16 mfma16(const Vec<__bf16>::x8& a, const Vec<__bf16>::x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(const Vec<_Float16>::x8& a, const Vec<_Float16>::x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

namespace {

constexpr int BM = 256;
constexpr int BK = 32;
constexpr int NT3 = 512;
constexpr int ROWB = 64;          // bytes per LDS row per plane (32 bf16)

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <typename E> __device__ __forceinline__ float hi_src(float x) { return x; }
// fp16 hi part saturates at +-65504 instead of overflowing to inf; the residual then lands in lo (exact up to 131008)
template <> __device__ __forceinline__ float hi_src<_Float16>(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }

template <typename E>
__device__ __forceinline__ void split4(const float4& v, typename Vec<E>::x4& hi, typename Vec<E>::x4& lo) {
    hi[0] = (E)hi_src<E>(v.x); hi[1] = (E)hi_src<E>(v.y); hi[2] = (E)hi_src<E>(v.z); hi[3] = (E)hi_src<E>(v.w);
    lo[0] = (E)(v.x - (float)hi[0]); lo[1] = (E)(v.y - (float)hi[1]);
    lo[2] = (E)(v.z - (float)hi[2]); lo[3] = (E)(v.w - (float)hi[3]);
}

// byte offset of the 8-byte group holding k = kq*4 .. kq*4+3 of `row` inside one plane
__device__ __forceinline__ int lds_off(int row, int kq) {
    const int chunk = (kq >> 1) ^ ((row >> 2) & 3);
    return row * ROWB + chunk * 16 + (kq & 1) * 8;
}

// ---- staging: k-contiguous operand P[i][k]  (thread -> kq = t&7, rows (t>>3) + 64 r) --------------------
template <int ROWS>
__device__ __forceinline__ void load_kc3(float4 (&v)[ROWS / 64], const float* __restrict__ P, int ld, int i0, int limit,
                                         int k0, int t) {
    const int kq = t & 7, ir = t >> 3;
#pragma unroll
    for (int r = 0; r < ROWS / 64; ++r) {
        const int gi = i0 + ir + 64 * r;
        v[r] = gi < limit ? ldg4(P + (size_t)gi * ld + k0 + kq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int ROWS, typename E>
__device__ __forceinline__ void store_kc3(const float4 (&v)[ROWS / 64], char* __restrict__ hi_plane, char* __restrict__ lo_plane, int t) {
    const int kq = t & 7, ir = t >> 3;
#pragma unroll
    for (int r = 0; r < ROWS / 64; ++r) {
        const int row = ir + 64 * r;
        typename Vec<E>::x4 h, l;
        split4<E>(v[r], h, l);
        const int off = lds_off(row, kq);
        *reinterpret_cast<typename Vec<E>::x4*>(hi_plane + off) = h;
        *reinterpret_cast<typename Vec<E>::x4*>(lo_plane + off) = l;
    }
}
// ---- staging: reduction-row operand P[red][i]: thread owns 4 red rows x 4 columns, ROWS/256 column groups ----
template <int ROWS>
__device__ __forceinline__ void load_rc3(float4 (&v)[ROWS / 256][4], const float* __restrict__ P, int ld, int i0, int limit,
                                         int k0, int t, int row_limit) {
    // lane -> (k group = t&7, column group = t>>3): a 16-lane LDS-store group then covers 8 k-groups x 2 rows
    // (2-way conflicts); the previous (t>>6, t&63) mapping put all 64 lanes of a wave on 4 bank slots (16-way).
    const int kg = t & 7, ig = t >> 3;
#pragma unroll
    for (int g = 0; g < ROWS / 256; ++g) {
        const int gi = i0 + g * 256 + ig * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = k0 + kg * 4 + r;
            v[g][r] = (gi < limit && row < row_limit) ? ldg4(P + (size_t)row * ld + gi) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
template <int ROWS, typename E>
__device__ __forceinline__ void store_rc3(const float4 (&v)[ROWS / 256][4], char* __restrict__ hi_plane, char* __restrict__ lo_plane, int t) {
    const int kg = t & 7, ig = t >> 3;
#pragma unroll
    for (int g = 0; g < ROWS / 256; ++g) {
        const float4 c0 = make_float4(v[g][0].x, v[g][1].x, v[g][2].x, v[g][3].x);
        const float4 c1 = make_float4(v[g][0].y, v[g][1].y, v[g][2].y, v[g][3].y);
        const float4 c2 = make_float4(v[g][0].z, v[g][1].z, v[g][2].z, v[g][3].z);
        const float4 c3 = make_float4(v[g][0].w, v[g][1].w, v[g][2].w, v[g][3].w);
        const float4 cols[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int row = g * 256 + ig * 4 + ii;
            typename Vec<E>::x4 h, l;
            split4<E>(cols[ii], h, l);
            const int off = lds_off(row, kg);
            *reinterpret_cast<typename Vec<E>::x4*>(hi_plane + off) = h;
            *reinterpret_cast<typename Vec<E>::x4*>(lo_plane + off) = l;
        }
    }
}

template <int BN, int MODE, typename E>
__global__ __launch_bounds__(NT3, 2) void gemm3_kernel(const GemmArgs a) {
    typedef typename Vec<E>::x8 ex8;
    typedef typename Vec<E>::x4 ex4;
    constexpr bool A_KC = (MODE != MODE_WGRAD);
    constexpr bool B_KC = (MODE == MODE_FWD);
    constexpr int WM = 4, WN = 2;
    constexpr int TM = BM / (WM * 32);       // 2
    constexpr int TN = BN / (WN * 32);       // 4 (BN=256) or 2 (BN=128)
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;

    extern __shared__ __attribute__((aligned(16))) char smem3[];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
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
    const int i0 = tm_i * BM, j0 = tn_i * BN;
    if (MODE == MODE_FWD && a.m_dev && i0 >= *a.m_dev) return;     // fixed-capacity buffer: rows past the device-side count are dead
    const int kt_begin = split * a.kt_per_split;
    const int kt_end = min(a.nk, kt_begin + a.kt_per_split);
    if (kt_begin >= kt_end) return;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

    float4 ra_kc[BM / 64], rb_kc[BN / 64];
    float4 ra_rc[BM / 256][4], rb_rc[BN >= 256 ? BN / 256 : 1][4];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_db = (MODE == MODE_WGRAD) && a.db != nullptr && tn_i == 0;

    auto gload = [&](int kt) {
        if constexpr (A_KC) {
            const float* P = a.A0; int ld = a.lda0; int k0 = kt * BK;
            if (MODE == MODE_FWD && kt >= a.kt0) { P = a.A1; ld = a.lda1; k0 = (kt - a.kt0) * BK; }
            load_kc3<BM>(ra_kc, P, ld, i0, a.Mload, k0, t);
        } else {
            load_rc3<BM>(ra_rc, a.A0, a.lda0, i0, a.Mload, kt * BK, t, a.red_limit);
            if (do_db) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { bsum.x += ra_rc[0][r].x; bsum.y += ra_rc[0][r].y; bsum.z += ra_rc[0][r].z; bsum.w += ra_rc[0][r].w; }
            }
        }
        if constexpr (B_KC) {
            load_kc3<BN>(rb_kc, a.B, a.ldb, j0, a.Nload, kt * BK, t);
        } else {
            if constexpr (BN >= 256) load_rc3<BN>(rb_rc, a.B, a.ldb, j0, a.Nload, kt * BK, t, a.red_limit);
            else {   // BN = 128: half of the threads (column groups 0..31) own a micro-tile
                const int kg = t & 7, ig = t >> 3;
                const int gi = j0 + ig * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = kt * BK + kg * 4 + r;
                    rb_rc[0][r] = (ig < 32 && gi < a.Nload && row < a.red_limit) ? ldg4(a.B + (size_t)row * a.ldb + gi) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };
    auto sstore = [&](int buf) {
        char* base = smem3 + buf * STAGE;
        char* Ah = base, *Al = base + A_PLANE, *Bh = base + 2 * A_PLANE, *Bl = base + 2 * A_PLANE + B_PLANE;
        if constexpr (A_KC) store_kc3<BM, E>(ra_kc, Ah, Al, t);
        else                store_rc3<BM, E>(ra_rc, Ah, Al, t);
        if constexpr (B_KC) store_kc3<BN, E>(rb_kc, Bh, Bl, t);
        else {
            if constexpr (BN >= 256) store_rc3<BN, E>(rb_rc, Bh, Bl, t);
            else if ((t >> 3) < 32) {
                const int kg = t & 7, ig = t >> 3;
                const float4 cols[4] = {make_float4(rb_rc[0][0].x, rb_rc[0][1].x, rb_rc[0][2].x, rb_rc[0][3].x),
                                        make_float4(rb_rc[0][0].y, rb_rc[0][1].y, rb_rc[0][2].y, rb_rc[0][3].y),
                                        make_float4(rb_rc[0][0].z, rb_rc[0][1].z, rb_rc[0][2].z, rb_rc[0][3].z),
                                        make_float4(rb_rc[0][0].w, rb_rc[0][1].w, rb_rc[0][2].w, rb_rc[0][3].w)};
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    ex4 h, l;
                    split4<E>(cols[ii], h, l);
                    const int off = lds_off(ig * 4 + ii, kg);
                    *reinterpret_cast<ex4*>(Bh + off) = h;
                    *reinterpret_cast<ex4*>(Bl + off) = l;
                }
            }
        }
    };

    // Ping-pong schedule: waves 0-3 ("A") and 4-7 ("B") share the four SIMDs pairwise (wave w and w+4 sit on the
    // same SIMD).  Per K tile there are two phases separated by barriers; in each phase one wave of a SIMD runs
    // its 48 MFMAs while its partner converts/stores its share of the NEXT tile (VALU + LDS), so the matrix
    // pipe and the VALU work overlap instead of alternating in lockstep (measured before this change:
    // MFMA busy 28 %, waves parked 49 % of the time):
    //   A:  issue loads(kt+1) | MFMA(kt)          | barrier | convert+store(kt+1) | barrier
    //   B:  convert+store(kt+1)                    | barrier | issue loads(kt+2) | MFMA(kt) | barrier
    // Every wave issues its global loads right before its own MFMA phase and consumes them right after it.
    const bool grpB = !(a.ablate & 8) && __builtin_amdgcn_readfirstlane(t >> 6) >= 4;   // ablate&8: lockstep schedule
    const int l31 = lane & 31, lhi = lane >> 5;

    auto compute = [&](int buf) {
        const char* base = smem3 + buf * STAGE;
        const char* Ah = base, *Al = base + A_PLANE, *Bh = base + 2 * A_PLANE, *Bl = base + 2 * A_PLANE + B_PLANE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = 2 * s + lhi;                  // 16-byte k chunk: k = 8c .. 8c+7
            ex8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int x = 0; x < TM; ++x) {
                const int row = wm * (TM * 32) + x * 32 + l31;
                const int off = row * ROWB + ((c ^ ((row >> 2) & 3)) * 16);
                ah[x] = *reinterpret_cast<const ex8*>(Ah + off);
                al[x] = *reinterpret_cast<const ex8*>(Al + off);
            }
#pragma unroll
            for (int y = 0; y < TN; ++y) {
                const int row = wn * (TN * 32) + y * 32 + l31;
                const int off = row * ROWB + ((c ^ ((row >> 2) & 3)) * 16);
                bh[y] = *reinterpret_cast<const ex8*>(Bh + off);
                bl[y] = *reinterpret_cast<const ex8*>(Bl + off);
            }
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    acc[x][y] = mfma16(al[x], bh[y], acc[x][y]);
                    acc[x][y] = mfma16(ah[x], bl[y], acc[x][y]);
                    acc[x][y] = mfma16(ah[x], bh[y], acc[x][y]);
                }
        }
    };

    // Software L2 prefetch.  A tile's operand loads miss L2 (the row operand is streamed from HBM, the weight
    // panel is evicted by it) and a wave has only one MFMA phase (~0.6 us) between issuing a tile's loads and
    // needing them -- less than the loaded HBM latency, which made the kernel latency-bound at ~10 B/clk/CU
    // (scripts/probe/*: the same access pattern streams at 23 B/clk/CU when enough requests are in flight).
    // Each thread therefore touches ONE dword of one 128-byte line of the tile `pf_dist` tiles ahead
    // (512 threads = the 256 + 256 lines of an A and a B tile); the value is only "used" by an empty asm one
    // iteration later, so the real loads that follow find their lines in L2.  (pf_dist = 0 degenerates to touching
    // the current, already resident tile.)
    float pfv = 0.f;
    auto prefetch = [&](int kt) {
        const int line = t & 255;
        const float* p = a.B;                  // always-valid fallback: the load below must be unconditional, or the
                                               // compiler has to drain it (vmcnt(0)) before the next tile's LDS stores
        if (t < 256) {
            if constexpr (A_KC) {
                const float* P = a.A0; int ld = a.lda0; int k0 = kt * BK;
                if (MODE == MODE_FWD && kt >= a.kt0) { P = a.A1; ld = a.lda1; k0 = (kt - a.kt0) * BK; }
                const int gi = i0 + line;
                if (gi < a.Mload) p = P + (size_t)gi * ld + k0;
            } else {
                const int row = kt * BK + (line >> 3), gi = i0 + (line & 7) * 32;
                if (row < a.red_limit && gi < a.Mload) p = a.A0 + (size_t)row * a.lda0 + gi;
            }
        } else {
            if constexpr (B_KC) {
                const int gj = j0 + line;
                if (line < BN && gj < a.Nload) p = a.B + (size_t)gj * a.ldb + kt * BK;
            } else {
                const int row = kt * BK + (line >> 3), cj = (line & 7) * 32;
                if (cj < BN && row < a.red_limit && j0 + cj < a.Nload) p = a.B + (size_t)row * a.ldb + j0 + cj;
            }
        }
        pfv = *p;
    };
    const int pf = a.pf_dist;

    gload(kt_begin);
    sstore(0);
    if (grpB && kt_begin + 1 < kt_end) gload(kt_begin + 1);
    __syncthreads();

    int buf = 0;
    const int ahead = grpB ? 2 : 1;
    // debug timeline (HOS_GEMM_ABLATE & 16): lane 0 of every wave of tile 0 stamps s_memtime at the phase
    // boundaries of K tiles 8..11 into a.aux (as long long[8 waves][4 iters][8 stamps])
    const bool trace = (a.ablate & 16) && bid == 0 && lane == 0 && a.aux != nullptr;
    long long* tr = reinterpret_cast<long long*>(a.aux);
#define HOS_STAMP(slot) do { if (trace && kt >= kt_begin + 8 && kt < kt_begin + 12) tr[(wave * 4 + (kt - kt_begin - 8)) * 8 + (slot)] = clock64(); } while (0)
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        HOS_STAMP(0);
        if (grpB) {                      // B, phase 1: convert/store its share of tile kt+1 (loaded one iteration ago)
            if (more && !(a.ablate & 2)) sstore(buf ^ 1);
            HOS_STAMP(1);
            __syncthreads();             // B's barrier #1  <->  A's barrier #1 (the common one below)
        }
        HOS_STAMP(2);
        asm volatile("" :: "v"(pfv));                                       // retire the previous prefetch
        if (kt + ahead < kt_end && !(a.ablate & 1)) gload(kt + ahead);
        prefetch(min(kt + pf, kt_end - 1));    // single call site: A fetches tile kt+1, B tile kt+2
        HOS_STAMP(3);
        if (!(a.ablate & 4)) compute(buf);                                  // single call site: A runs it in phase 1, B in phase 2
        HOS_STAMP(4);
        __syncthreads();                 // A's barrier #1 / B's barrier #2
        HOS_STAMP(5);
        if (!grpB) {                     // A, phase 2: convert/store its share of tile kt+1
            if (more && !(a.ablate & 2)) sstore(buf ^ 1);
            HOS_STAMP(6);
            __syncthreads();             // A's barrier #2  <->  B's barrier #2 (the common one)
        }
        HOS_STAMP(7);
        buf ^= 1;
    }
#undef HOS_STAMP
    // (both groups have executed exactly two barriers per iteration: no drain needed)

#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y)
            gemm_epilogue_tile<MODE>(a, acc[x][y], i0 + wm * (TM * 32) + x * 32, j0 + wn * (TN * 32) + y * 32, lane);

    if constexpr (MODE == MODE_WGRAD) {
        if (do_db) {
            // thread (kg = t&7, ig = t>>3) holds column sums of its 4 rows for columns ig*4..+3: reduce over the 8 kg lanes
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                bsum.x += __shfl_xor(bsum.x, o, 64); bsum.y += __shfl_xor(bsum.y, o, 64);
                bsum.z += __shfl_xor(bsum.z, o, 64); bsum.w += __shfl_xor(bsum.w, o, 64);
            }
            if ((t & 7) == 0 && i0 + (t >> 3) * 4 < a.M) {
                const int c0 = i0 + (t >> 3) * 4;
                const float vals[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c0 + k < a.M) __hip_atomic_fetch_add(a.db + c0 + k, vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int BN, int MODE, typename E>
int launch3(GemmArgs& a, int splits, hipStream_t stream) {
    constexpr size_t smem = 2 * (2 * BM * ROWB + 2 * BN * ROWB);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<BN, MODE, E>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    a.tiles_m = hos_cdiv(a.M, BM);
    a.tiles_n = hos_cdiv(a.N, BN);
    if (MODE == MODE_WGRAD) {
        static const int env_splits = getenv("HOS_WGRAD_SPLITS") ? atoi(getenv("HOS_WGRAD_SPLITS")) : 0;
        if (env_splits > 0) splits = env_splits;
        if (splits <= 0) {
            const int tiles = a.tiles_m * a.tiles_n;
            splits = 256 / tiles > 0 ? 256 / tiles : 1;          // one workgroup per CU: one over costs a whole second round
            if (splits > a.nk / 8) splits = a.nk / 8 > 0 ? a.nk / 8 : 1;
        }
        if (splits > a.nk) splits = a.nk;
        a.kt_per_split = hos_cdiv(a.nk, splits);
        splits = hos_cdiv(a.nk, a.kt_per_split);
    } else {
        splits = 1;
        a.kt_per_split = a.nk;
    }
    hipLaunchKernelGGL((gemm3_kernel<BN, MODE, E>), dim3(a.tiles_m * a.tiles_n * splits), dim3(NT3), smem, stream, a);
    return hos_launch_status();
}

}  // namespace

int hos_gemm3_launch(GemmArgs a, int mode, int splits, hipStream_t stream) {
    static const int ablate = getenv("HOS_GEMM_ABLATE") ? atoi(getenv("HOS_GEMM_ABLATE")) : 0;
    a.ablate = ablate;
    static const int pf_dist = getenv("HOS_GEMM_PF") ? atoi(getenv("HOS_GEMM_PF")) : 3;
    a.pf_dist = pf_dist;
    const bool wide = a.N > 128;
    switch (mode) {
        case MODE_FWD:   return wide ? launch3<256, MODE_FWD, _Float16>(a, 1, stream) : launch3<128, MODE_FWD, _Float16>(a, 1, stream);
        case MODE_DGRAD: return wide ? launch3<256, MODE_DGRAD, __bf16>(a, 1, stream) : launch3<128, MODE_DGRAD, __bf16>(a, 1, stream);
        default: {
            // 256 x 128 tiles up to N = 256: a [256,256] gradient then has two tiles x 128 splits instead of one x 256
            // (half the atomic traffic at the same parallelism; same finding as hos_gemmp.hip)
            static const int narrow_max = getenv("HOS_WGRAD_NARROW_MAX") ? atoi(getenv("HOS_WGRAD_NARROW_MAX")) : 256;
            return a.N > narrow_max ? launch3<256, MODE_WGRAD, __bf16>(a, splits, stream) : launch3<128, MODE_WGRAD, __bf16>(a, splits, stream);
        }
    }
}
