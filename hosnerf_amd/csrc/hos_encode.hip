// Sample-point encoders for the background branch.
//
// hos_encode_ipe: conical-frustum moments (H:294-304) -> full-covariance lift (H:318-339) ->
// scene contraction with closed-form Jacobian (H:33-68; SURVEY 7.1: equals functorch.jacrev to
// 1.5e-8) -> projection on the 21-direction icosahedron basis (H:71-74) -> integrated positional
// encoding (H:78-89) -> [IPE(504) | state embedding(64) | 0-pad] rows ready for the layer-0 GEMM.
// In the reference this is ~40 torch launches with [B,S,3,3] / [B,S,12,21] temporaries and a
// vmap(jacrev) autograd pass (13.5 % + 18 % of its forward time).
//
// HBM-bound on the write of X: algorithmic bytes/sample = 4*ldx written + 4 read (tdist).
// Built with -ffp-contract=off: sin(x*2^l + pi/2) must round the sum in fp32 exactly like the
// reference does (for 2^11*x ~ 4096 the fp32 ulp is 4.9e-4 -- a fused or cosf() variant differs
// from the reference by up to 2.4e-4, more than the whole parity budget).
#include "hos_common.h"

namespace {

constexpr int NDIR = 21;
constexpr int NLVL = 12;
constexpr int NIPE = 2 * NDIR * NLVL;   // 504
constexpr int NEMB = 64;
constexpr int SB = 64;                  // samples per workgroup
constexpr float EPS = 1.1920929e-07f;
constexpr float HALF_PI = 1.57079637050628662109375f;   // float32(0.5*pi)

// sin(x) for the encoder's argument range: 2^l * (contracted mean . unit basis vector) (+ pi/2), |x| <= ~4100.  The library's sinf is
// branch-free Payne-Hanek on this target (30 v_mad_u64_u32 per loop iteration of the feature loop, 638 instructions for 4 sines +
// 2 exponentials); here: k = rint(x * 2/pi), three fused Cody-Waite steps with pi/2 = A + B + C (72 bits), the Cephes single-precision
// kernels on [-pi/4, pi/4].  Checked against double precision over the range (oracle/.. tests/test_encoder_sine_cpu.py restates it):
// <= 1.56 ulp, 9.3e-8 absolute.  Arguments beyond 2^15 (never produced by the model; a caller's free basis could) take the library path,
// decided once per workgroup from the lifted means (a test per sine cost as much as the library's reduction saved).
__device__ __forceinline__ float enc_sin(float x) {
    const float kf = __builtin_rintf(x * 0x1.45f306p-1f);
    float r = __builtin_fmaf(kf, -0x1.921fb6p+0f, x);
    r = __builtin_fmaf(kf, 0x1.777a5cp-25f, r);
    r = __builtin_fmaf(kf, 0x1.ee59dap-50f, r);
    const int k = (int)kf;
    const float z = r * r;
    float sp = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    sp = __builtin_fmaf(sp * z, r, r);
    float cp = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    cp = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const float v = (k & 1) ? cp : sp;
    return (k & 2) ? -v : v;
}
// exp(x), x <= 0: 2^(x log2 e) with the rounding error of the product carried into a first-order correction
__device__ __forceinline__ float enc_exp(float x) {
    const float t = x * 0x1.715476p+0f;
    const float e = __builtin_fmaf(x, 0x1.715476p+0f, -t) + x * 0x1.4ae0cp-26f;
    return __builtin_amdgcn_exp2f(t) * __builtin_fmaf(e, 0.69314718f, 1.0f);
}

// (hi, lo) split of one value for the interleaved-planes layout of hos_gemmp.hip: fp16 hi saturates at +-65504
template <typename E> __device__ __forceinline__ float hi_clamp(float x) { return x; }
template <> __device__ __forceinline__ float hi_clamp<_Float16>(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }
template <typename E> __device__ __forceinline__ void split_store(unsigned short* __restrict__ P, size_t o, float x) {
    const E h = (E)hi_clamp<E>(x);
    const E l = (E)(x - (float)h);
    P[o] = __builtin_bit_cast(unsigned short, h);
    P[o + 32] = __builtin_bit_cast(unsigned short, l);
}
// two adjacent columns (c even) of one row: the (hi, hi) and (lo, lo) pairs as one 4-byte store each -- half the store
// instructions of the per-column form (the stores were a quarter of this kernel's time).
// Round 5, measured and dropped (profiles/r05_ab_encoder_output_stage.txt): staging 16 samples' rows as fp32 in LDS and writing
// whole 128-byte plane lines with 16-byte stores (the planes GEMM's epilogue pattern): 420 vs 348 us per 262 144 samples with both
// formats, 277 vs 220 us with one -- the kernel is bound by its 756 sinf / expf evaluations per sample (one format: 2.7 TB/s of
// stores, two formats: 3.5), not by store issue, and the staging's barriers + 48 KB of LDS per workgroup cost occupancy.
template <typename E> __device__ __forceinline__ void split_store2(unsigned short* __restrict__ P, size_t o, float x0, float x1) {
    const E h0 = (E)hi_clamp<E>(x0), h1 = (E)hi_clamp<E>(x1);
    const E l0 = (E)(x0 - (float)h0), l1 = (E)(x1 - (float)h1);
    *reinterpret_cast<uint32_t*>(P + o) = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
    *reinterpret_cast<uint32_t*>(P + o + 32) = (uint32_t)__builtin_bit_cast(unsigned short, l0) | ((uint32_t)__builtin_bit_cast(unsigned short, l1) << 16);
}
__device__ __forceinline__ size_t plane_off(long row, int c, int ld) { return (size_t)row * (2 * ld) + (c >> 5) * 64 + (c & 31); }

// PLANES = false: X fp32 [P][ldx].  PLANES = true: the same rows written directly as interleaved 16-bit planes
// (fp16 planes p16 = first-layer operand, optional bf16 planes pb = weight-gradient operand), so the MLP never
// sees an fp32 copy of the 576-wide encoding (saves the 151 MB write and the hos_split_planes2 pass per level).
template <bool PLANES>
__global__ __launch_bounds__(256) void encode_ipe_kernel(
    const float* __restrict__ tdist, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ radii, const float* __restrict__ basis, const float* __restrict__ embed,
    int B, int S, float* __restrict__ X, int ldx, unsigned short* __restrict__ p16, unsigned short* __restrict__ pb) {
    __shared__ float s_mean[SB][3];
    __shared__ float s_cov[SB][9];
    __shared__ float s_lm[SB][NDIR];
    __shared__ float s_lv[SB][NDIR];
    __shared__ float s_basis[3][NDIR];
    __shared__ float s_embed[NEMB];
    __shared__ int s_big;              // some |2^11 * lifted mean| of this block is beyond enc_sin's range: library sinf for the block

    const int t = threadIdx.x;
    const long P = (long)B * S;
    const long p0 = (long)blockIdx.x * SB;
    if (t < 3 * NDIR) s_basis[t / NDIR][t % NDIR] = basis[t];
    if (t == 255) s_big = 0;
    if (t >= 64 && t < 64 + NEMB) s_embed[t - 64] = embed[t - 64];

    if (t < SB && p0 + t < P) {
        const long p = p0 + t;
        const int ray = (int)(p / S), s = (int)(p % S);
        const float t0 = tdist[(size_t)ray * (S + 1) + s], t1 = tdist[(size_t)ray * (S + 1) + s + 1];
        const float ox = rays_o[ray * 3 + 0], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
        const float d[3] = {rays_d[ray * 3 + 0], rays_d[ray * 3 + 1], rays_d[ray * 3 + 2]};
        const float rad = radii[ray];
        // H:296-302
        const float mu = (t0 + t1) / 2.f, hw = (t1 - t0) / 2.f;
        const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
        const float denom = fmaxf(3.f * mu2 + hw2, EPS);
        const float t_mean = mu + (2.f * mu * hw2) / denom;
        const float t_var = hw2 / 3.f - (4.f / 15.f) * hw4 * (12.f * mu2 - hw2) / (denom * denom);
        float r_var = mu2 / 4.f + (5.f / 12.f) * hw2 - (4.f / 15.f) * hw4 / denom;
        r_var *= rad * rad;
        // H:320-338 (diag=False)
        const float dmag = fmaxf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], 1e-10f);
        float x[3] = {d[0] * t_mean + ox, d[1] * t_mean + oy, d[2] * t_mean + oz};
        float cov[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float outer = d[a] * d[c];
                const float nul = (a == c ? 1.f : 0.f) - d[a] * (d[c] / dmag);
                cov[a][c] = t_var * outer + r_var * nul;
            }
        // H:37-42 contraction and its Jacobian
        const float r2 = fmaxf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2], 1e-32f);
        float J[3][3];
        float z[3];
        if (r2 <= 1.f) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                z[a] = x[a];
#pragma unroll
                for (int c = 0; c < 3; ++c) J[a][c] = (a == c) ? 1.f : 0.f;
            }
        } else {
            const float r = sqrtf(r2);
            const float sc = (2.f * r - 1.f) / r2;
            const float cc = (2.f / (r2 * r) - 2.f / r2) / r;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                z[a] = sc * x[a];
#pragma unroll
                for (int c = 0; c < 3; ++c) J[a][c] = (a == c ? sc : 0.f) + cc * (x[a] * x[c]);
            }
        }
        float JC[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) JC[a][c] = J[a][0] * cov[0][c] + J[a][1] * cov[1][c] + J[a][2] * cov[2][c];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_mean[t][a] = z[a];
#pragma unroll
            for (int c = 0; c < 3; ++c)   // (J cov) J^T
                s_cov[t][a * 3 + c] = JC[a][0] * J[c][0] + JC[a][1] * J[c][1] + JC[a][2] * J[c][2];
        }
    }
    __syncthreads();
    // lift: mean . b_j  and  b_j^T cov b_j   (H:71-74)
    for (int it = t; it < SB * NDIR; it += 256) {
        const int s = it / NDIR, j = it % NDIR;
        const float b0 = s_basis[0][j], b1 = s_basis[1][j], b2 = s_basis[2][j];
        const float lm = s_mean[s][0] * b0 + s_mean[s][1] * b1 + s_mean[s][2] * b2;
        s_lm[s][j] = lm;
        if (!(fabsf(lm) * 2048.f < 32000.f)) s_big = 1;        // (also NaN; same value from every writer)
        const float* c = s_cov[s];
        const float c0 = c[0] * b0 + c[1] * b1 + c[2] * b2;
        const float c1 = c[3] * b0 + c[4] * b1 + c[5] * b2;
        const float c2 = c[6] * b0 + c[7] * b1 + c[8] * b2;
        s_lv[s][j] = b0 * c0 + b1 * c1 + b2 * c2;
    }
    __syncthreads();
    // IPE features (H:78-89, H:104-105): col = level*21 + dir ; second half = +pi/2
    constexpr int HALF = NDIR * NLVL;   // 252
    const bool big = __builtin_amdgcn_readfirstlane(s_big) != 0;        // block-uniform, in a scalar register
    auto feat = [&](int s, int c, float& v0, float& v1) {
        const int lvl = c / NDIR, j = c % NDIR;
        const float sc = (float)(1 << lvl);
        const float sm = s_lm[s][j] * sc;
        const float sv = s_lv[s][j] * (sc * sc);
        const float damp = enc_exp(-0.5f * sv);
        if (big) { v0 = damp * sinf(sm); v1 = damp * sinf(sm + HALF_PI); }
        else     { v0 = damp * enc_sin(sm); v1 = damp * enc_sin(sm + HALF_PI); }
    };
    if constexpr (!PLANES) {
        for (int it = t; it < SB * HALF; it += 256) {
            const int s = it / HALF, c = it % HALF;
            if (p0 + s >= P) break;
            float v0, v1;
            feat(s, c, v0, v1);
            float* row = X + (size_t)(p0 + s) * ldx;
            row[c] = v0;
            row[c + HALF] = v1;
        }
    } else {
        // A thread owns two adjacent columns (c even; c and c + 252 are both even and pairs never straddle a 32-column block) of
        // every second sample of the block: 252 of the 256 threads = 126 column pairs x 2 sample parities.  (level, direction) of
        // its columns, their scales and their offsets inside a plane row are then loop invariants -- the per-item form
        // (it -> sample, column by division) spent ~100 of its ~256 instructions per item on index arithmetic, and the kernel is
        // bound by VALU issue.  Round 5, same box, per 262 144 / 131 072 / 4 194 304 samples, one plane format
        // (profiles/r05_encoder_variants.txt): per-item loop + library sinf / expf 226 / 114 / 3025 us; this loop + library 189 / 89 /
        // 2533; + enc_sin 181 / 81 / 2379; + enc_exp 169 / 76 / 2284 (kept); hardware v_sin / v_exp (NOT to parity) 149 / 58 / 2214.
        if (t < 2 * (HALF / 2)) {
            const int q = t % (HALF / 2), sub = t / (HALF / 2);
            const int c = 2 * q;
            const int j0 = c % NDIR, j1 = (c + 1) % NDIR;
            const float sc0 = (float)(1 << (c / NDIR)), sc1 = (float)(1 << ((c + 1) / NDIR));
            const float sq0 = sc0 * sc0, sq1 = sc1 * sc1;
            const int co0 = (c >> 5) * 64 + (c & 31), co1 = ((c + HALF) >> 5) * 64 + ((c + HALF) & 31);
            const size_t pitch = (size_t)2 * ldx;
            const int n_s = (int)((P - p0) < (long)SB ? (P - p0) : (long)SB);
            auto one = [&](float lm, float lv, float sc, float sq, float& v0, float& v1) {      // == feat()
                const float sm = lm * sc;
                const float sv = lv * sq;
                const float damp = enc_exp(-0.5f * sv);
                if (big) { v0 = damp * sinf(sm); v1 = damp * sinf(sm + HALF_PI); }
                else     { v0 = damp * enc_sin(sm); v1 = damp * enc_sin(sm + HALF_PI); }
            };
            size_t row = (size_t)(p0 + sub) * pitch;
            for (int s = sub; s < n_s; s += 2, row += 2 * pitch) {
                float a0, a1, b0, b1;
                one(s_lm[s][j0], s_lv[s][j0], sc0, sq0, a0, a1);
                one(s_lm[s][j1], s_lv[s][j1], sc1, sq1, b0, b1);
                if (p16 != nullptr) { split_store2<_Float16>(p16, row + co0, a0, b0); split_store2<_Float16>(p16, row + co1, a1, b1); }
                if (pb != nullptr) { split_store2<__bf16>(pb, row + co0, a0, b0); split_store2<__bf16>(pb, row + co1, a1, b1); }
            }
        }
    }
    const int tail = ldx - NIPE;   // embedding + zero pad (even: ldx % 32 == 0, NIPE = 504)
    if constexpr (!PLANES) {
        for (int it = t; it < SB * tail; it += 256) {
            const int s = it / tail, c = it % tail;
            if (p0 + s >= P) break;
            X[(size_t)(p0 + s) * ldx + NIPE + c] = (c < NEMB) ? s_embed[c] : 0.f;
        }
    } else {
        for (int it = t; it < SB * (tail / 2); it += 256) {
            const int s = it / (tail / 2), c = (it % (tail / 2)) * 2;
            if (p0 + s >= P) break;
            const float v0 = (c < NEMB) ? s_embed[c] : 0.f, v1 = (c + 1 < NEMB) ? s_embed[c + 1] : 0.f;
            const size_t o = plane_off(p0 + s, NIPE + c, ldx);
            if (p16 != nullptr) split_store2<_Float16>(p16, o, v0, v1);
            if (pb != nullptr) split_store2<__bf16>(pb, o, v0, v1);
        }
    }
}

// pos_enc(viewdirs, 0, 4, append_identity=True) (H:93-100) broadcast over the S samples of a ray.
__global__ __launch_bounds__(256) void encode_viewdirs_kernel(const float* __restrict__ viewdirs, int B, int S,
                                                              float* __restrict__ Xv, int ldx, int col0) {
    const int width = ldx - col0;   // 27 features + zero pad
    const long total = (long)B * S * width;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long p = it / width;
        const int c = (int)(it % width);
        const int ray = (int)(p / S);
        float v = 0.f;
        if (c < 3) {
            v = viewdirs[ray * 3 + c];
        } else if (c < 27) {
            const int k = (c - 3) % 12, half = (c - 3) / 12;
            const int lvl = k / 3, ax = k % 3;
            const float xb = viewdirs[ray * 3 + ax] * (float)(1 << lvl);
            v = half ? sinf(xb + HALF_PI) : sinf(xb);
        }
        Xv[(size_t)p * ldx + col0 + c] = v;
    }
}

}  // namespace

extern "C" int hos_encode_ipe(const float* tdist, const float* rays_o, const float* rays_d, const float* radii,
                              const float* basis, const float* embed, int B, int S, float* X, int ldx,
                              hos_stream_t stream) {
    if (!tdist || !rays_o || !rays_d || !radii || !basis || !embed || !X || B <= 0 || S <= 0) return HOS_E_ARG;
    if (ldx < NIPE + NEMB) return HOS_E_SHAPE;
    const long P = (long)B * S;
    hipLaunchKernelGGL(encode_ipe_kernel<false>, dim3((unsigned)((P + SB - 1) / SB)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), tdist, rays_o, rays_d, radii, basis, embed, B, S, X, ldx,
                       (unsigned short*)nullptr, (unsigned short*)nullptr);
    return hos_launch_status();
}

extern "C" int hos_encode_ipe_planes(const float* tdist, const float* rays_o, const float* rays_d, const float* radii,
                                     const float* basis, const float* embed, int B, int S, void* p16, void* pb, int ld,
                                     hos_stream_t stream) {
    if (!tdist || !rays_o || !rays_d || !radii || !basis || !embed || (!p16 && !pb) || B <= 0 || S <= 0) return HOS_E_ARG;
    if (ld < NIPE + NEMB || (ld & 31)) return HOS_E_SHAPE;
    const long P = (long)B * S;
    hipLaunchKernelGGL(encode_ipe_kernel<true>, dim3((unsigned)((P + SB - 1) / SB)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), tdist, rays_o, rays_d, radii, basis, embed, B, S, (float*)nullptr, ld,
                       (unsigned short*)p16, (unsigned short*)pb);
    return hos_launch_status();
}

extern "C" int hos_encode_viewdirs(const float* viewdirs, int B, int S, float* Xv, int ldx, int col0,
                                   hos_stream_t stream) {
    if (!viewdirs || !Xv || B <= 0 || S <= 0) return HOS_E_ARG;
    if (ldx - col0 < 27 || col0 < 0) return HOS_E_SHAPE;
    const long total = (long)B * S * (ldx - col0);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(encode_viewdirs_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       viewdirs, B, S, Xv, ldx, col0);
    return hos_launch_status();
}
