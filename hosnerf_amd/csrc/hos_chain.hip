// Forward of a whole thin MLP in ONE launch with the activations on chip across its layers (SURVEY 8(b).6 `fused_mlp_nonrigid`,
// VERDICT r1 item 2): the non-rigid motion MLPs of the human branch -- `NonRigidMotionMLP` / `NonRigidForwardMLP`,
// non_rigid_motion_mlps/mlp_offset.py:16-70: [cond 75 | hann 36] -> 5 x (128, ReLU) with the hann features re-concatenated before
// Linear #4 -> 3, xyz = x + offset -- over P = rays x 128 sample points.
//
// Layer-by-layer launches move every activation through HBM twice (written by layer l, read by layer l+1: 8 B per row and
// column and layer).  Here a wave keeps its 32 rows in REGISTERS from the embedding to the offset:
//   * the product is formed transposed, H^T = W . X^T (v_mfma_f32_32x32x16_f16, weights as the A operand, 32 sample rows as
//     the B operand), so the accumulator layout of layer l -- lane (row, half) holds 16 outputs of every 32-wide block for ITS
//     row -- already IS a B-operand layout of layer l+1 once the reduction index is permuted inside every group of 16
//     (k -> 8*(c/4) + 4*half + c%4); the weights are packed in that order once per step (hos_mlp_chain_pack), the contraction
//     does not care.  No LDS round trip, no cross-lane traffic between layers.
//   * fp16 (hi, lo) split of both operands, three MFMAs per product, fp32 accumulate: the arithmetic of the split GEMMs.
//   * every layer's output is written ONCE (fp32, for the backward pass: 4 B per row and column and layer), nothing is read
//     back; the last layer (3 outputs) is an fp32 FMA reduction on the lanes that already hold its input.
//   * the weights stream from L2 through a two-deep LDS ring, one 32-column output block (16-24 KB of fragments) at a time, by
//     LDS-DMA (global_load_lds), shared by the four waves of a workgroup; two workgroups per CU so that one's epilogue VALU
//     work runs under the other's MFMAs.
// Per row: 101 120 MAC x 3 products; HBM bytes per row = 768 read (E, PE) + 6 x 512 + 12 written.
#include <stdlib.h>

#include "hos_common.h"
#include "hos_gemm_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int CT = 256;             // threads per workgroup = 4 waves, one per SIMD
constexpr int CROWS = 128;          // rows per workgroup tile (32 per wave)
constexpr int NL = 6;               // hidden layers
constexpr int CW = 128;             // layer width
constexpr int KSMAX = 12;           // k-steps of 16: 8 (K = 128), 12 for the skip layer (128 + 64)
constexpr int CBUF = KSMAX * 2 * 1024;      // one out-block of fragments: k-steps x (hi, lo) x 1 KB
constexpr int SKIP_LAYER = 4;

// FOLD: the 75 pose-condition columns of the first layer's input are the same for every sample point of a call, so they enter
// as a per-call bias (b0 + W0[:, :75] . cond, formed by the pack kernel) and the first layer's reduction is the 36 hann features
// alone: 4 k-steps instead of 8, operand rows PE [P, 64] instead of E [P, 128] -- and since PE is also the skip layer's
// re-concatenated input, ONE 256-byte row read serves both (768 B per row without the fold).
__host__ __device__ constexpr int ks_of(int l, bool fold = false) { return l == SKIP_LAYER ? 12 : ((fold && l == 0) ? 4 : 8); }
__host__ __device__ constexpr int layer_off(int l, bool fold = false) {      // bytes, chain planes of layer l
    int o = 0;
    for (int i = 0; i < l; ++i) o += 4 * ks_of(i, fold) * 2048;
    return o;
}
constexpr int WC_BYTES = layer_off(NL);
constexpr int W0H_LD = 64;          // fp32 copy of the hann columns of W0 [128, 64] for the folded first layer's backward pass

struct ChainArgs {
    const float* E; int lde;        // [P, >=128]  first-layer rows [cond | hann | 0]
    const float* PE; int ldpe;      // [P, >=64]   hann features (skip concat)
    const float* x;                 // [P, 3] residual
    const uint16_t* Wc;             // chain planes of the hidden layers (hos_mlp_chain_pack)
    const float* aux;               // [6*128 biases | 3*128 last-layer weight | 3 last-layer bias | pad]
    float* acts[NL]; int ldact;     // [P, >=128] each: post-ReLU outputs, kept for the backward pass
    float* xyz;                     // [P, 3]
    long P; const int* p_dev;
    unsigned int* range_flag;
};
constexpr int AUX_FLOATS = NL * CW + 3 * CW + 4;
constexpr int AUX_BYTES = (AUX_FLOATS * 4 + 15) & ~15;
constexpr int SMEM_BYTES = 2 * CBUF + AUX_BYTES;

__device__ __forceinline__ void dma16(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

__device__ __forceinline__ f32x16 mfma3(const h8& ah, const h8& al, const h8& bh, const h8& bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ void split_to(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);        // one v_med3 (fminf(fmaxf()) adds a canonicalising v_max)
    lo = (_Float16)(x - (float)hi);
}

// eight fp32 values at p[0..3] and p[8..11] (the k-slots of this half-lane inside a group of 16) -> (hi, lo) fragments
__device__ __forceinline__ void load_split8(const float* p, h8& hi, h8& lo) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 8);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) { _Float16 h, l; split_to(v[c], h, l); hi[c] = h; lo[c] = l; }
}

#define CH_RD128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))

template <bool FOLD>
__global__ __launch_bounds__(CT, 2) void chain128_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wbuf = smem;                                              // 2 x CBUF weight fragments
    float* const s_aux = reinterpret_cast<float*>(smem + 2 * CBUF);       // biases, last layer
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, hh = lane >> 5;
    long P = a.P;
    if (a.p_dev) P = min(P, (long)*a.p_dev);
    for (int i = t; i < AUX_FLOATS; i += CT) s_aux[i] = a.aux[i];
    const long ntiles = (P + CROWS - 1) / CROWS;
    if ((long)blockIdx.x >= ntiles) return;

    // Weight chunk g (0..23) = (layer g / 4, out-block g % 4): this wave moves its quarter of every 4 KB round.  The layer
    // loop below is a REAL loop (not unrolled): with everything unrolled the compiler hoists ~110 loop-invariant 64-bit
    // per-lane addresses out of the tile loop and spills them; as a function of the loop counter they are recomputed instead.
    const unsigned voff = (unsigned)(wave * 1024 + lane * 16);
    // one 1 KB request of this wave for round q of a chunk (a chunk = ks / 2 rounds of 4 KB)
    auto issue_round = [&](int l, int ob, int parity, int q) {
        const int ks = ks_of(l, FOLD);
        // byte offset of layer l's planes: 8 k-steps per earlier layer, +4 behind the skip layer, -4 behind a folded layer 0
        const int lo = (l * 8 + (l > SKIP_LAYER ? 4 : 0) - ((FOLD && l > 0) ? 4 : 0)) * 4 * 2048;
#ifndef HOS_CHAIN_NO_DMA        // timing experiments only (results invalid)
        dma16(reinterpret_cast<const char*>(a.Wc) + lo + ob * ks * 2048 + voff + q * 4096, wbuf + parity * CBUF + wave * 1024 + q * 4096);
#endif
    };
    auto issue_chunk = [&](int l, int ob, int parity) {
        const int rounds = ks_of(l, FOLD) / 2;
        for (int q = 0; q < rounds; ++q) issue_round(l, ob, parity, q);
    };
    issue_chunk(0, 0, 0);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    bool big = false;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row = tile * CROWS + wave * 32 + r;
        const long lrow = row < P ? row : P - 1;                       // clamped for loads; never stored
        const bool tile_full = (tile + 1) * CROWS <= P;
        h8 bh[KSMAX], bl[KSMAX];
        if constexpr (FOLD) {
            // the hann features: operand of layer 0 AND of the skip layer -- loaded once per tile, kept in slots 8..11
            const float* pe = a.PE + (size_t)lrow * a.ldpe + 4 * hh;
#pragma unroll
            for (int s = 0; s < 4; ++s) load_split8(pe + 16 * s, bh[8 + s], bl[8 + s]);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int c = 0; c < 8; ++c) { bh[s][c] = (_Float16)0.f; bl[s][c] = (_Float16)0.f; }
        } else {
            const float* e = a.E + (size_t)lrow * a.lde + 4 * hh;
#pragma unroll
            for (int s = 0; s < 8; ++s) load_split8(e + 16 * s, bh[s], bl[s]);
#pragma unroll
            for (int s = 8; s < KSMAX; ++s)
#pragma unroll
                for (int c = 0; c < 8; ++c) { bh[s][c] = (_Float16)0.f; bl[s][c] = (_Float16)0.f; }
        }
        float part[3] = {0.f, 0.f, 0.f};
        int parity = 0;
#pragma unroll 1
        for (int l = 0; l < NL; ++l) {
            const int ks = ks_of(l, FOLD);
            if (!FOLD && l == SKIP_LAYER) {
                const float* pe = a.PE + (size_t)lrow * a.ldpe + 4 * hh;
#pragma unroll
                for (int s = 0; s < 4; ++s) load_split8(pe + 16 * s, bh[8 + s], bl[8 + s]);
            }
            f32x16 acc[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ob][i] = 0.f;
                // This wave's share of the chunk has landed.  vmcnt retires in issue order and the chunk's requests were issued
                // BEFORE the 16 activation stores of the previous layer's epilogue, so at a layer's first chunk only those may
                // still be in flight: the stores drain under the MFMAs instead of being waited for.
                // (Only in a tile whose 128 rows are all live: a wave whose rows are all past P issues NO store -- an instruction
                // with an empty EXEC mask does not count -- so "16 younger operations" would include this chunk's own requests,
                // the wave would pass the barrier before its share of the weights has landed and the whole workgroup would
                // multiply by stale LDS in the boundary tile of the compacted cycle set.)
                if (ob == 0 && l > 0 && tile_full) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                          // ... and everybody's; the other buffer is free
                // the next chunk (wrapping to chunk 0 for the next tile) goes into the other buffer, one request every second
                // k-step: back-to-back requests stall the issuing wave, and an in-order wave that waits issues no MFMA
                int nl = l, nob = ob + 1;
                if (nob == 4) { nob = 0; nl = l + 1 == NL ? 0 : l + 1; }
                const int nrounds = ks_of(nl, FOLD) / 2;
                const int npar = parity ^ 1;
                const int parity_buf = parity;
                parity ^= 1;
                // Fragments one k-step ahead of the MFMAs.  hipcc stops counting LDS operations once an LDS-DMA is in flight and
                // puts lgkmcnt(0) in front of every MFMA group -- i.e. it would also wait for the reads just issued for the NEXT
                // k-step -- so the reads are asm statements it does not count, followed by an explicit counted wait (LDS returns
                // in order); the fragments pass through the wait as "+v" operands so that no MFMA can move above it.
                const unsigned la = lds_base + (unsigned)(parity_buf * CBUF + lane * 16);
                // two register sets alternate by k-step parity: a register that an asm read is still filling must not be
                // copied (the compiler believes the asm statement has completed)
                h8 fh[2], fl[2];
                // NO fragment read may be in flight across a RUN-TIME branch: at a control-flow merge hipcc is free to copy a
                // register (v_mov) that it believes the asm statement has already written -- while the LDS is still filling
                // it.  The original loop decided `ks == 12` inside k-step 7 with the reads of step 8 / the last reads of step 7
                // outstanding; the copies hipcc placed there picked up half-written fragments a few times per 10^5 launches
                // (garbage in some rows of one wave, layer 5: found by a 600-step soak with NaN-poisoned allocations).  Steps
                // 0..7 are now straight-line code that ends with everything landed, and the four extra steps of the skip layer
                // are a second straight-line group inside ONE branch.
                if (FOLD && l == 0) {
                    // folded first layer: the four k-steps of the hann features (operand slots 8..11), same discipline
                    CH_RD128(fh[0], la, 0);
                    CH_RD128(fl[0], la, 1024);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (s < 3) {
                            CH_RD128(fh[(s + 1) & 1], la, (2 * s + 2) * 1024);
                            CH_RD128(fl[(s + 1) & 1], la, (2 * s + 3) * 1024);
                            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[s & 1]), "+v"(fl[s & 1]));
                        } else {
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[0]), "+v"(fl[0]), "+v"(fh[1]), "+v"(fl[1]));
                        }
                        acc[ob] = mfma3(fh[s & 1], fl[s & 1], bh[8 + s], bl[8 + s], acc[ob]);
                        if ((s & 1) == 0) issue_round(nl, nob, npar, s >> 1);
                    }
                    if (nrounds == 4) { issue_round(nl, nob, npar, 2); issue_round(nl, nob, npar, 3); }     // (0, 3) -> layer 1
                } else {
                CH_RD128(fh[0], la, 0);
                CH_RD128(fl[0], la, 1024);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s < 7) {
                        CH_RD128(fh[(s + 1) & 1], la, (2 * s + 2) * 1024);
                        CH_RD128(fl[(s + 1) & 1], la, (2 * s + 3) * 1024);
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[s & 1]), "+v"(fl[s & 1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[0]), "+v"(fl[0]), "+v"(fh[1]), "+v"(fl[1]));
                    }
#ifndef HOS_CHAIN_NO_MFMA
                    acc[ob] = mfma3(fh[s & 1], fl[s & 1], bh[s], bl[s], acc[ob]);
#else
                    acc[ob][s & 15] += (float)fh[s & 1][0] + (float)fl[s & 1][1];
#endif
                    if ((s & 1) == 0 && (s >> 1) < nrounds) issue_round(nl, nob, npar, s >> 1);
                }
                if (ks == 12) {
                    CH_RD128(fh[0], la, 16 * 1024);
                    CH_RD128(fl[0], la, 17 * 1024);
#pragma unroll
                    for (int s = 8; s < 12; ++s) {
                        if (s < 11) {
                            CH_RD128(fh[(s + 1) & 1], la, (2 * s + 2) * 1024);
                            CH_RD128(fl[(s + 1) & 1], la, (2 * s + 3) * 1024);
                            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[s & 1]), "+v"(fl[s & 1]));
                        } else {
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[0]), "+v"(fl[0]), "+v"(fh[1]), "+v"(fl[1]));
                        }
#ifndef HOS_CHAIN_NO_MFMA
                        acc[ob] = mfma3(fh[s & 1], fl[s & 1], bh[s], bl[s], acc[ob]);
#else
                        acc[ob][s & 15] += (float)fh[s & 1][0] + (float)fl[s & 1][1];
#endif
                        if ((s & 1) == 0 && (s >> 1) < nrounds) issue_round(nl, nob, npar, s >> 1);
                    }
                }
                if (ks == 8 && nrounds == 6) { issue_round(nl, nob, npar, 4); issue_round(nl, nob, npar, 5); }
                }
            }
            // ---- epilogue of layer l: bias, ReLU, one store for the backward pass, next layer's B fragments in place
            // lane-dependent part (4 * half) folded into the bases: every access below is base + immediate
            const float* bias = s_aux + l * CW + 4 * hh;
            float* out = a.acts[l] + (size_t)lrow * a.ldact + 4 * hh;
            const bool last = l == NL - 1;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
                for (int tq = 0; tq < 2; ++tq) {         // accumulator registers 8 tq .. 8 tq + 7 = the 8 k-slots of next k-step 2 ob + tq
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int q = 2 * tq + u;
                        const int n0 = 32 * ob + 8 * q;
                        const float4 b4 = *reinterpret_cast<const float4*>(bias + n0);
                        v[4 * u + 0] = fmaxf(acc[ob][4 * q + 0] + b4.x, 0.f);
                        v[4 * u + 1] = fmaxf(acc[ob][4 * q + 1] + b4.y, 0.f);
                        v[4 * u + 2] = fmaxf(acc[ob][4 * q + 2] + b4.z, 0.f);
                        v[4 * u + 3] = fmaxf(acc[ob][4 * q + 3] + b4.w, 0.f);
#ifndef HOS_CHAIN_NO_STORE
                        // (a lane owns ONE row here, so a store instruction touches 32 lines; staging the block through LDS for
                        // whole-line stores was measured no faster: the stores are bound by the HBM write rate -- 3 KB per row --
                        // and all CUs reach their epilogues at the same time)
                        if (row < P) *reinterpret_cast<float4*>(out + n0) = make_float4(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
#endif
                        if (last) {
                            const float* w6 = s_aux + NL * CW + 4 * hh;
#pragma unroll
                            for (int m = 0; m < 3; ++m) {
                                const float4 w4 = *reinterpret_cast<const float4*>(w6 + m * CW + n0);
                                part[m] += v[4 * u] * w4.x + v[4 * u + 1] * w4.y + v[4 * u + 2] * w4.z + v[4 * u + 3] * w4.w;
                            }
                        }
                    }
                    float mx = v[0];
#pragma unroll
                    for (int c = 1; c < 8; ++c) mx = fmaxf(mx, v[c]);
                    big |= mx > HOS_RANGE_LIMIT;
                    h8 hi, lo;
#pragma unroll
                    for (int c = 0; c < 8; ++c) { _Float16 h_, l_; split_to(v[c], h_, l_); hi[c] = h_; lo[c] = l_; }
                    bh[2 * ob + tq] = hi;
                    bl[2 * ob + tq] = lo;
                }
            }
        }
        // ---- last layer (mlp_offset.py:66-70): offset = W6 h + b6, xyz = x + offset
#pragma unroll
        for (int m = 0; m < 3; ++m) part[m] += __shfl_xor(part[m], 32, 64);
        if (hh == 0 && row < P) {
            const float* b6 = s_aux + NL * CW + 3 * CW;
#pragma unroll
            for (int m = 0; m < 3; ++m) a.xyz[row * 3 + m] = part[m] + b6[m] + a.x[row * 3 + m];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the wrapped-around prefetch of chunk 0
    if (a.range_flag != nullptr && __builtin_amdgcn_ballot_w64(big) != 0 && lane == 0) atomicOr(a.range_flag, 1u);
}

// fp32 weights (nn.Linear layout [128, ldw]) -> chain planes: [out-block 4][k-step][hi, lo][lane 64][8 halfs], the lane's eight
// values = W[32 ob + lane % 32][16 s + 8 (c / 4) + 4 (lane / 32) + c % 4], c = 0..7
struct PackArgs {
    const float* W[NL]; int ldw[NL];
    const float* b[NL];
    const float* W6; int ldw6; const float* b6;
    uint16_t* Wc; float* aux;
    // FOLD only: the condition vector [ncond] shared by all rows of the coming launches, the number of feature columns behind
    // it in W0, and the aligned fp32 copy [128, W0H_LD] of those columns (operand of the first layer's backward pass)
    const float* cond; int ncond; int nfeat; float* w0h;
};

template <bool FOLD>
__global__ __launch_bounds__(256) void chain_pack_kernel(PackArgs p) {
    const int l = blockIdx.y;
    if (l == NL) {                                   // biases and the last layer
        for (int i = blockIdx.x * 256 + threadIdx.x; i < AUX_FLOATS; i += gridDim.x * 256) {
            float v = 0.f;
            if (i < NL * CW) {
                v = p.b[i / CW][i % CW];
                if (FOLD && i < CW) {                // b0 + W0[:, :ncond] . cond, summed in column order in fp32
                    const float* w = p.W[0] + (size_t)i * p.ldw[0];
                    for (int c = 0; c < p.ncond; ++c) v = fmaf(w[c], p.cond[c], v);
                }
            }
            else if (i < NL * CW + 3 * CW) { const int j = i - NL * CW; v = p.W6[(j / CW) * p.ldw6 + j % CW]; }
            else if (i < NL * CW + 3 * CW + 3) v = p.b6[i - NL * CW - 3 * CW];
            p.aux[i] = v;
        }
        return;
    }
    const int ks = ks_of(l, FOLD);
    const int total = 4 * ks * 64 * 8;               // (ob, s, lane, c): one thread writes hi and lo
    uint16_t* dst = p.Wc + layer_off(l, FOLD) / 2;
    if (FOLD && l == 0)
        for (int e = blockIdx.x * 256 + threadIdx.x; e < CW * W0H_LD; e += gridDim.x * 256) {
            const int n = e / W0H_LD, k = e % W0H_LD;
            p.w0h[e] = k < p.nfeat ? p.W[0][(size_t)n * p.ldw[0] + p.ncond + k] : 0.f;
        }
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int c = e & 7, lane = (e >> 3) & 63, s = (e >> 9) % ks, ob = (e >> 9) / ks;
        const int n = 32 * ob + (lane & 31), k = 16 * s + 8 * (c >> 2) + 4 * (lane >> 5) + (c & 3);
        float w;
        if (FOLD && l == 0) w = k < p.nfeat ? p.W[0][(size_t)n * p.ldw[0] + p.ncond + k] : 0.f;
        else w = p.W[l][(size_t)n * p.ldw[l] + k];
        _Float16 hi, lo;
        split_to(w, hi, lo);
        const size_t o = (((size_t)(ob * ks + s) * 2) * 64 + lane) * 8 + c;
        dst[o] = __builtin_bit_cast(uint16_t, hi);
        dst[o + 64 * 8] = __builtin_bit_cast(uint16_t, lo);
    }
}


}  // namespace

extern "C" long long hos_mlp_chain_weight_bytes(void) { return WC_BYTES; }
extern "C" long long hos_mlp_chain_aux_floats(void) { return AUX_FLOATS; }

static int chain_pack_args(PackArgs& p, const float* const* weights7, const int* ldw7, const float* const* biases7, void* chain_planes,
                           float* aux, int k0) {
    if (!weights7 || !ldw7 || !biases7 || !chain_planes || !aux) return HOS_E_ARG;
    for (int l = 0; l < NL; ++l) {
        if (!weights7[l] || !biases7[l] || ldw7[l] < (l == 0 ? k0 : 16 * ks_of(l))) return HOS_E_ARG;
        p.W[l] = weights7[l]; p.ldw[l] = ldw7[l]; p.b[l] = biases7[l];
    }
    if (!weights7[NL] || !biases7[NL] || ldw7[NL] < CW) return HOS_E_ARG;
    p.W6 = weights7[NL]; p.ldw6 = ldw7[NL]; p.b6 = biases7[NL];
    p.Wc = static_cast<uint16_t*>(chain_planes); p.aux = aux;
    return 0;
}

extern "C" int hos_mlp_chain_pack(const float* const* weights7, const int* ldw7, const float* const* biases7, void* chain_planes,
                                  float* aux, hos_stream_t stream) {
    PackArgs p{};
    if (int rc = chain_pack_args(p, weights7, ldw7, biases7, chain_planes, aux, 16 * ks_of(0))) return rc;
    hipLaunchKernelGGL(chain_pack_kernel<false>, dim3(24, NL + 1), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hos_launch_status();
}

extern "C" int hos_mlp_chain_pack_fold(const float* const* weights7, const int* ldw7, const float* const* biases7, const float* cond,
                                       int ncond, int nfeat, void* chain_planes, float* aux, float* w0h, hos_stream_t stream) {
    if (!cond || !w0h || ncond <= 0 || nfeat <= 0) return HOS_E_ARG;
    if (nfeat > W0H_LD) return HOS_E_SHAPE;
    if ((uintptr_t)w0h & 15u) return HOS_E_ALIGN;
    PackArgs p{};
    if (int rc = chain_pack_args(p, weights7, ldw7, biases7, chain_planes, aux, ncond + nfeat)) return rc;
    p.cond = cond; p.ncond = ncond; p.nfeat = nfeat; p.w0h = w0h;
    hipLaunchKernelGGL(chain_pack_kernel<true>, dim3(24, NL + 1), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hos_launch_status();
}

template <bool FOLD>
static int chain128_launch(const ChainArgs& a, hipStream_t stream) {
    // HOS_CHAIN_LDS_PAD (diagnostic): extra dynamic LDS per workgroup, e.g. 26000 -> two workgroups own a CU's whole LDS and no
    // other kernel's workgroup can become co-resident on it
    static const size_t pad = getenv("HOS_CHAIN_LDS_PAD") ? (size_t)atoi(getenv("HOS_CHAIN_LDS_PAD")) : 0;
    const size_t smem = SMEM_BYTES + pad;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain128_kernel<FOLD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long ntiles = (a.P + CROWS - 1) / CROWS;
    static int max_grid = 0;
    if (max_grid == 0) { const char* e = getenv("HOS_CHAIN_GRID"); max_grid = e ? atoi(e) : 1024; if (max_grid <= 0) max_grid = 1024; }
    const int grid = (int)(ntiles < max_grid ? ntiles : max_grid);
    hipLaunchKernelGGL(chain128_kernel<FOLD>, dim3(grid), dim3(CT), smem, stream, a);
    return hos_launch_status();
}

// E == NULL selects the folded form: planes / aux from hos_mlp_chain_pack_fold, PE is the only per-row operand.
extern "C" int hos_mlp_chain128_fwd(const float* E, int lde, const float* PE, int ldpe, const float* x, const void* chain_planes,
                                    const float* aux, float* const* acts6, int ldact, float* xyz, int64_t P,
                                    const int32_t* rows_dev, hos_stream_t stream) {
    if (!PE || !x || !chain_planes || !aux || !acts6 || !xyz || P <= 0) return HOS_E_ARG;
    if ((E && lde < 128) || ldpe < 64 || ldact < 128) return HOS_E_SHAPE;
    if ((E && (lde & 3)) || (ldpe & 3) || (ldact & 3) || (((uintptr_t)E | (uintptr_t)PE | (uintptr_t)chain_planes) & 15u)) return HOS_E_ALIGN;
    ChainArgs a{};
    a.E = E; a.lde = lde; a.PE = PE; a.ldpe = ldpe; a.x = x; a.Wc = static_cast<const uint16_t*>(chain_planes); a.aux = aux;
    for (int l = 0; l < NL; ++l) {
        if (!acts6[l] || ((uintptr_t)acts6[l] & 15u)) return HOS_E_ARG;
        a.acts[l] = acts6[l];
    }
    a.ldact = ldact; a.xyz = xyz; a.P = P; a.p_dev = rows_dev; a.range_flag = hos_range_flag_ptr();
    return E ? chain128_launch<false>(a, static_cast<hipStream_t>(stream)) : chain128_launch<true>(a, static_cast<hipStream_t>(stream));
}

// Gradients of a folded first layer back into the reference-shaped W0 / b0 (called after the slab reductions have landed):
//   gW0[n, ncond + k] += gw0h[n, k] (k < nfeat);   gW0[n, c] += db[n] * cond[c] (c < ncond);   gb0[n] += db[n]
// -- every row of the launch saw the same condition vector, so its weight gradient is the outer product with the bias gradient.
__global__ __launch_bounds__(256) void chain_unfold_kernel(const float* __restrict__ gw0h, const float* __restrict__ db,
                                                           const float* __restrict__ cond, int ncond, int nfeat,
                                                           float* __restrict__ gW0, int ldw, float* __restrict__ gb0) {
    const int K = ncond + nfeat;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < CW * (K + 1); e += gridDim.x * 256) {
        const int n = e / (K + 1), c = e % (K + 1);
        if (c == K) gb0[n] += db[n];
        else if (c < ncond) gW0[(size_t)n * ldw + c] += db[n] * cond[c];
        else gW0[(size_t)n * ldw + c] += gw0h[n * W0H_LD + (c - ncond)];
    }
}

extern "C" int hos_mlp_chain_unfold_grad(const float* gw0h, const float* db, const float* cond, int ncond, int nfeat,
                                         float* gW0, int ldw, float* gb0, hos_stream_t stream) {
    if (!gw0h || !db || !cond || !gW0 || !gb0 || ncond <= 0 || nfeat <= 0) return HOS_E_ARG;
    if (nfeat > W0H_LD || ldw < ncond + nfeat) return HOS_E_SHAPE;
    hipLaunchKernelGGL(chain_unfold_kernel, dim3(56), dim3(256), 0, static_cast<hipStream_t>(stream), gw0h, db, cond, ncond, nfeat, gW0, ldw, gb0);
    return hos_launch_status();
}
