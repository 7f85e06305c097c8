// Per-ray compositing and stage-1 regularisers, one wavefront per ray (wave64 scans).
//
//   hos_alpha_weights_{fwd,bwd}   H:235-261  compute_alpha_weights
//   hos_volrender_{fwd,bwd}       H:265-275  volumetric_rendering
//   hos_interlevel_{fwd,bwd}      M1:611-620 -> H:136-138 / H:117-132 / H:109-114
//   hos_distortion_{fwd,bwd}      M1:622-627 -> H:142-149
//   hos_head_grad                 softplus' / sigmoid' of M:316, M:345-346
//
// All of these are HBM-bound streaming kernels (tens of bytes per sample); the in-ray prefix
// products are wave-level scans in registers -- nothing is staged through memory.
#include "hos_common.h"

namespace {

constexpr int CH = 4;            // samples per lane -> up to 256 samples per ray
constexpr float EPS = 1.1920929e-07f;

// Per-lane chunk of `CH` consecutive samples; exclusive prefix sum over the ray.
__device__ __forceinline__ void ray_excl_scan(const float (&v)[CH], float (&out)[CH], int lane) {
    float loc[CH];
    float run = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) { loc[c] = run; run += v[c]; }
    // exclusive base = inclusive scan of the lane totals shifted by one lane.  (NOT incl - run: the
    // opaque-background interval is 1e10 and would cancel the whole prefix.)
    const float incl = wave_incl_scan(run, lane);
    float base = __shfl_up(incl, 1, 64);
    if (lane == 0) base = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) out[c] = base + loc[c];
}
// exclusive suffix sum: out[i] = sum_{j>i} v[j]
__device__ __forceinline__ void ray_excl_rscan(const float (&v)[CH], float (&out)[CH], int lane) {
    float loc[CH];
    float run = 0.f;
#pragma unroll
    for (int c = CH - 1; c >= 0; --c) { loc[c] = run; run += v[c]; }
    const float incl = wave_incl_rscan(run, lane);
    float base = __shfl_down(incl, 1, 64);
    if (lane == 63) base = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) out[c] = base + loc[c];
}

struct RaySamples {
    int per;      // samples per lane
};

__device__ __forceinline__ float dir_norm(const float* dirs, int ray) {
    const float x = dirs[ray * 3], y = dirs[ray * 3 + 1], z = dirs[ray * 3 + 2];
    return sqrtf(x * x + y * y + z * z);
}

// density_delta (with the opaque-background substitution), alpha, trans, weights for the lane's chunk
__device__ __forceinline__ void alpha_chain(const float* density, const float* tdist, float dn, int ray, int S,
                                            int per, int lane, bool opaque, float (&dd)[CH], float (&delta)[CH],
                                            float (&trans)[CH], float (&w)[CH]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int s = lane * per + c;
        dd[c] = 0.f; delta[c] = 0.f;
        if (c < per && s < S) {
            const float t0 = tdist[(size_t)ray * (S + 1) + s], t1 = tdist[(size_t)ray * (S + 1) + s + 1];
            delta[c] = (t1 - t0) * dn;
            dd[c] = density[(size_t)ray * S + s] * delta[c];
            if (opaque && s == S - 1) dd[c] = 1e10f;
        }
    }
    float ex[CH];
    ray_excl_scan(dd, ex, lane);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        trans[c] = expf(-ex[c]);
        w[c] = (1.f - expf(-dd[c])) * trans[c];
    }
}

__global__ __launch_bounds__(256) void alpha_weights_fwd_kernel(const float* __restrict__ density,
                                                                const float* __restrict__ tdist,
                                                                const float* __restrict__ dirs, int B, int S,
                                                                int opaque, float* __restrict__ weights) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= B) return;
    const int per = (S + 63) / 64;
    float dd[CH], delta[CH], trans[CH], w[CH];
    alpha_chain(density, tdist, dir_norm(dirs, ray), ray, S, per, lane, opaque != 0, dd, delta, trans, w);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int s = lane * per + c;
        if (c < per && s < S) weights[(size_t)ray * S + s] = w[c];
    }
}

__global__ __launch_bounds__(256) void alpha_weights_bwd_kernel(const float* __restrict__ g_weights,
                                                                const float* __restrict__ density,
                                                                const float* __restrict__ tdist,
                                                                const float* __restrict__ dirs, int B, int S,
                                                                int opaque, float* __restrict__ g_density) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= B) return;
    const int per = (S + 63) / 64;
    float dd[CH], delta[CH], trans[CH], w[CH], gw[CH], suf[CH];
    alpha_chain(density, tdist, dir_norm(dirs, ray), ray, S, per, lane, opaque != 0, dd, delta, trans, w);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int s = lane * per + c;
        gw[c] = (c < per && s < S) ? g_weights[(size_t)ray * S + s] * w[c] : 0.f;   // g_i * w_i
    }
    ray_excl_rscan(gw, suf, lane);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int s = lane * per + c;
        if (c < per && s < S) {
            // d w_k / d dd_k = T_k - w_k ;  d w_i / d dd_k = -w_i for i > k
            const float g = g_weights[(size_t)ray * S + s];
            float gdd = g * (trans[c] - w[c]) - suf[c];
            if (opaque && s == S - 1) gdd = 0.f;           // last interval is the constant 1e10
            g_density[(size_t)ray * S + s] = gdd * delta[c];
        }
    }
}

__global__ __launch_bounds__(256) void volrender_fwd_kernel(const float* __restrict__ rgbs,
                                                            const float* __restrict__ weights, int B, int S,
                                                            float bg, float* __restrict__ rgb) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= B) return;
    float acc = 0.f, r = 0.f, g = 0.f, b = 0.f;
    for (int s = lane; s < S; s += 64) {
        const float w = weights[(size_t)ray * S + s];
        const float* c = rgbs + ((size_t)ray * S + s) * 3;
        acc += w; r += w * c[0]; g += w * c[1]; b += w * c[2];
    }
    acc = wave_sum(acc); r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
    if (lane == 0) {
        const float bgw = fmaxf(1.f - acc, 0.f) * bg;
        rgb[ray * 3 + 0] = r + bgw; rgb[ray * 3 + 1] = g + bgw; rgb[ray * 3 + 2] = b + bgw;
    }
}

__global__ __launch_bounds__(256) void volrender_bwd_kernel(const float* __restrict__ g_rgb,
                                                            const float* __restrict__ rgbs,
                                                            const float* __restrict__ weights, int B, int S,
                                                            float bg, float* __restrict__ g_rgbs,
                                                            float* __restrict__ g_weights) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= B) return;
    float acc = 0.f;
    for (int s = lane; s < S; s += 64) acc += weights[(size_t)ray * S + s];
    acc = wave_sum(acc);
    const float g0 = g_rgb[ray * 3], g1 = g_rgb[ray * 3 + 1], g2 = g_rgb[ray * 3 + 2];
    const float gbg = (1.f - acc >= 0.f) ? bg * (g0 + g1 + g2) : 0.f;   // clamp(min=0) passes grad at >=
    for (int s = lane; s < S; s += 64) {
        const size_t i = (size_t)ray * S + s;
        const float w = weights[i];
        const float* c = rgbs + i * 3;
        if (g_weights) g_weights[i] = g0 * c[0] + g1 * c[1] + g2 * c[2] - gbg;
        if (g_rgbs) { g_rgbs[i * 3] = g0 * w; g_rgbs[i * 3 + 1] = g1 * w; g_rgbs[i * 3 + 2] = g2 * w; }
    }
}

// ---- interlevel (proposal) loss ---------------------------------------------------------------
constexpr int SCMAX = 64, SPMAX = 128;
struct InterLds {
    float c[SCMAX + 1], w[SCMAX], cp[SPMAX + 1], cy[SPMAX + 1], coef[SCMAX];
    int lo[SCMAX + 1], hi[SCMAX + 1];
};

__device__ __forceinline__ int ub(const float* a, int n, float x) {   // #{a_k <= x}
    int l = 0, r = n;
    while (l < r) { int m = (l + r) >> 1; if (a[m] <= x) l = m + 1; else r = m; }
    return l;
}

// shared prologue: loads, cy = [0, cumsum(wp)], lo/hi per NeRF edge (H:109-114), returns per-lane loss sum
__device__ __forceinline__ float interlevel_common(InterLds& L, const float* c, const float* w, const float* cp,
                                                   const float* wp, int ray, int Sc, int Sp, int lane) {
    for (int i = lane; i <= Sc; i += 64) L.c[i] = c[(size_t)ray * (Sc + 1) + i];
    for (int i = lane; i < Sc; i += 64) L.w[i] = w[(size_t)ray * Sc + i];
    for (int i = lane; i <= Sp; i += 64) L.cp[i] = cp[(size_t)ray * (Sp + 1) + i];
    // exclusive+inclusive cumsum of wp: lane handles 2 consecutive bins (Sp <= 128)
    {
        const int j0 = lane * 2;
        const float a = j0 < Sp ? wp[(size_t)ray * Sp + j0] : 0.f;
        const float b = j0 + 1 < Sp ? wp[(size_t)ray * Sp + j0 + 1] : 0.f;
        const float incl = wave_incl_scan(a + b, lane);
        float base = __shfl_up(incl, 1, 64);
        if (lane == 0) base = 0.f;
        if (lane == 0) L.cy[0] = 0.f;
        if (j0 < Sp) L.cy[j0 + 1] = base + a;
        if (j0 + 1 < Sp) L.cy[j0 + 2] = base + a + b;
    }
    __syncthreads();
    for (int i = lane; i <= Sc; i += 64) {
        const int cnt = ub(L.cp, Sp + 1, L.c[i]);
        L.lo[i] = max(cnt - 1, 0);
        L.hi[i] = min(cnt, Sp);
    }
    __syncthreads();
    float part = 0.f;
    for (int i = lane; i < Sc; i += 64) {
        const float wo = L.cy[L.hi[i + 1]] - L.cy[L.lo[i]];       // H:125
        const float d = fmaxf(L.w[i] - wo, 0.f);
        const float den = L.w[i] + EPS;
        part += d * d / den;                                       // H:138
        L.coef[i] = -2.f * d / den;                                // d loss_i / d w_outer_i
    }
    return part;
}

__global__ __launch_bounds__(256) void interlevel_fwd_kernel(const float* __restrict__ c, const float* __restrict__ w,
                                                             const float* __restrict__ cp, const float* __restrict__ wp,
                                                             int B, int Sc, int Sp, float* __restrict__ loss_ray,
                                                             int32_t* __restrict__ idx_lo, int32_t* __restrict__ idx_hi) {
    __shared__ InterLds lds[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray_raw = blockIdx.x * 4 + wave;
    const bool live = ray_raw < B;
    const int ray = live ? ray_raw : B - 1;
    InterLds& L = lds[wave];
    float part = interlevel_common(L, c, w, cp, wp, ray, Sc, Sp, lane);
    part = wave_sum(part);
    if (live && lane == 0) loss_ray[ray] = part;
    if (live && idx_lo) for (int i = lane; i <= Sc; i += 64) { idx_lo[(size_t)ray * (Sc + 1) + i] = L.lo[i]; idx_hi[(size_t)ray * (Sc + 1) + i] = L.hi[i]; }
}

__global__ __launch_bounds__(256) void interlevel_bwd_kernel(const float* __restrict__ c, const float* __restrict__ w,
                                                             const float* __restrict__ cp, const float* __restrict__ wp,
                                                             int B, int Sc, int Sp, float scale, float* __restrict__ g_wp) {
    __shared__ InterLds lds[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray_raw = blockIdx.x * 4 + wave;
    const bool live = ray_raw < B;
    const int ray = live ? ray_raw : B - 1;
    InterLds& L = lds[wave];
    interlevel_common(L, c, w, cp, wp, ray, Sc, Sp, lane);
    __syncthreads();
    // w_outer_i = sum_{j in [lo_i, hi_{i+1})} wp_j   ->   g_wp[j] = scale * sum_i [lo_i <= j < hi_{i+1}] coef_i
    for (int j = lane; j < Sp; j += 64) {
        float g = 0.f;
        for (int i = 0; i < Sc; ++i) g += (L.lo[i] <= j && j < L.hi[i + 1]) ? L.coef[i] : 0.f;
        if (live) g_wp[(size_t)ray * Sp + j] = g * scale;
    }
}

// ---- distortion loss ---------------------------------------------------------------------------
constexpr int SDMAX = 128;
__global__ __launch_bounds__(256) void distortion_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                                         int B, int S, float scale, float* __restrict__ loss_ray,
                                                         float* __restrict__ g_w) {
    __shared__ float s_u[4][SDMAX], s_w[4][SDMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray_raw = blockIdx.x * 4 + wave;
    const bool live = ray_raw < B;
    const int ray = live ? ray_raw : B - 1;
    for (int i = lane; i < S; i += 64) {
        const float t0 = t[(size_t)ray * (S + 1) + i], t1 = t[(size_t)ray * (S + 1) + i + 1];
        s_u[wave][i] = (t1 + t0) / 2.f;
        s_w[wave][i] = w[(size_t)ray * S + i];
    }
    __syncthreads();
    float part = 0.f;
    for (int i = lane; i < S; i += 64) {
        const float ui = s_u[wave][i], wi = s_w[wave][i];
        float inner = 0.f;
        for (int j = 0; j < S; ++j) inner += s_w[wave][j] * fabsf(ui - s_u[wave][j]);
        const float dt = t[(size_t)ray * (S + 1) + i + 1] - t[(size_t)ray * (S + 1) + i];
        part += wi * inner + wi * wi * dt / 3.f;
        if (g_w && live) g_w[(size_t)ray * S + i] = scale * (2.f * inner + 2.f * wi * dt / 3.f);
    }
    part = wave_sum(part);
    if (loss_ray && live && lane == 0) loss_ray[ray] = part;
}

__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ g_density,
                                                        const float* __restrict__ density,
                                                        const float* __restrict__ g_rgb, const float* __restrict__ rgb,
                                                        int P, float pad, float* __restrict__ dz_density, int ld_dd,
                                                        int col_dd, float* __restrict__ dz_rgb, int ld_dr) {
    const float k = 1.f + 2.f * pad;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < P; m += (long)gridDim.x * blockDim.x) {
        if (dz_density) {
            const float g = g_density ? g_density[m] : 0.f;
            dz_density[(size_t)m * ld_dd + col_dd] = g * (1.f - expf(-density[m]));   // softplus'(z) = 1 - exp(-softplus(z))
        }
        if (dz_rgb) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float s = (rgb[m * 3 + c] + pad) / k;
                const float g = g_rgb ? g_rgb[m * 3 + c] : 0.f;
                dz_rgb[(size_t)m * ld_dr + c] = g * k * s * (1.f - s);
            }
        }
    }
}

inline unsigned ray_blocks(int B) { return (unsigned)hos_cdiv(B, 4); }

}  // namespace

extern "C" int hos_alpha_weights_fwd(const float* density, const float* tdist, const float* dirs, int B, int S,
                                     int opaque_background, float* weights, hos_stream_t stream) {
    if (!density || !tdist || !dirs || !weights || B <= 0 || S <= 0) return HOS_E_ARG;
    if (S > 64 * CH) return HOS_E_SHAPE;
    hipLaunchKernelGGL(alpha_weights_fwd_kernel, dim3(ray_blocks(B)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       density, tdist, dirs, B, S, opaque_background, weights);
    return hos_launch_status();
}

extern "C" int hos_alpha_weights_bwd(const float* g_weights, const float* density, const float* tdist,
                                     const float* dirs, int B, int S, int opaque_background, float* g_density,
                                     hos_stream_t stream) {
    if (!g_weights || !density || !tdist || !dirs || !g_density || B <= 0 || S <= 0) return HOS_E_ARG;
    if (S > 64 * CH) return HOS_E_SHAPE;
    hipLaunchKernelGGL(alpha_weights_bwd_kernel, dim3(ray_blocks(B)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g_weights, density, tdist, dirs, B, S, opaque_background, g_density);
    return hos_launch_status();
}

extern "C" int hos_volrender_fwd(const float* rgbs, const float* weights, int B, int S, float bg, float* rgb,
                                 hos_stream_t stream) {
    if (!rgbs || !weights || !rgb || B <= 0 || S <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(volrender_fwd_kernel, dim3(ray_blocks(B)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rgbs, weights, B, S, bg, rgb);
    return hos_launch_status();
}

extern "C" int hos_volrender_bwd(const float* g_rgb, const float* rgbs, const float* weights, int B, int S,
                                 float bg, float* g_rgbs, float* g_weights, hos_stream_t stream) {
    if (!g_rgb || !rgbs || !weights || B <= 0 || S <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(volrender_bwd_kernel, dim3(ray_blocks(B)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g_rgb, rgbs, weights, B, S, bg, g_rgbs, g_weights);
    return hos_launch_status();
}

extern "C" int hos_interlevel_fwd(const float* c, const float* w, const float* cp, const float* wp, int B,
                                  int Sc, int Sp, float* loss_ray, int32_t* idx_lo, int32_t* idx_hi,
                                  hos_stream_t stream) {
    if (!c || !w || !cp || !wp || !loss_ray || B <= 0) return HOS_E_ARG;
    if (Sc < 1 || Sc > SCMAX || Sp < 1 || Sp > SPMAX) return HOS_E_SHAPE;
    if ((idx_lo == nullptr) != (idx_hi == nullptr)) return HOS_E_ARG;
    hipLaunchKernelGGL(interlevel_fwd_kernel, dim3(ray_blocks(B)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       c, w, cp, wp, B, Sc, Sp, loss_ray, idx_lo, idx_hi);
    return hos_launch_status();
}

extern "C" int hos_interlevel_bwd(const float* c, const float* w, const float* cp, const float* wp, int B,
                                  int Sc, int Sp, float scale, float* g_wp, hos_stream_t stream) {
    if (!c || !w || !cp || !wp || !g_wp || B <= 0) return HOS_E_ARG;
    if (Sc < 1 || Sc > SCMAX || Sp < 1 || Sp > SPMAX) return HOS_E_SHAPE;
    hipLaunchKernelGGL(interlevel_bwd_kernel, dim3(ray_blocks(B)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       c, w, cp, wp, B, Sc, Sp, scale, g_wp);
    return hos_launch_status();
}

extern "C" int hos_distortion_fwd(const float* t, const float* w, int B, int S, float* loss_ray,
                                  hos_stream_t stream) {
    if (!t || !w || !loss_ray || B <= 0) return HOS_E_ARG;
    if (S < 1 || S > SDMAX) return HOS_E_SHAPE;
    hipLaunchKernelGGL(distortion_kernel, dim3(ray_blocks(B)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       t, w, B, S, 0.f, loss_ray, (float*)nullptr);
    return hos_launch_status();
}

extern "C" int hos_distortion_bwd(const float* t, const float* w, int B, int S, float scale, float* g_w,
                                  hos_stream_t stream) {
    if (!t || !w || !g_w || B <= 0) return HOS_E_ARG;
    if (S < 1 || S > SDMAX) return HOS_E_SHAPE;
    hipLaunchKernelGGL(distortion_kernel, dim3(ray_blocks(B)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       t, w, B, S, scale, (float*)nullptr, g_w);
    return hos_launch_status();
}

// columns [c0, c1) of every row of a [P, ld] matrix = 0 (the zero padding of an operand row whose live columns other kernels write)
__global__ __launch_bounds__(256) void zero_cols_kernel(float* __restrict__ a, int ld, int c0, int c1, int P, float* __restrict__ b, int ldb, int b0, int b1) {
    const int wa = a ? c1 - c0 : 0, wb = b ? b1 - b0 : 0, w = wa + wb;
    const long total = (long)P * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / w;
        const int c = (int)(i - m * w);
        if (c < wa) a[m * ld + c0 + c] = 0.f; else b[m * ldb + b0 + (c - wa)] = 0.f;
    }
}

// hos_head_grad that also writes the ZERO PADDING of its two operand rows -- columns (col_dd, ld_dd) of dz_density's rows and
// [3, ld_dr) of dz_rgb's -- so that both may be uninitialised storage (the caller's other kernels fill columns [0, col_dd)).
extern "C" int hos_head_grad_padded(const float* g_density, const float* density, const float* g_rgb, const float* rgb,
                                    int P, float rgb_padding, float* dz_density, int ld_dd, int col_dd,
                                    float* dz_rgb, int ld_dr, hos_stream_t stream) {
    if (P <= 0) return HOS_E_ARG;
    if (dz_density && !density) return HOS_E_ARG;
    if (dz_rgb && !rgb) return HOS_E_ARG;
    int blocks = hos_cdiv(P, 256);
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(head_grad_kernel, dim3(blocks), dim3(256), 0, s, g_density, density, g_rgb, rgb, P, rgb_padding, dz_density, ld_dd, col_dd, dz_rgb, ld_dr);
    const int wa = dz_density ? ld_dd - (col_dd + 1) : 0, wb = dz_rgb ? ld_dr - 3 : 0;
    if (wa + wb > 0) {
        long zb = ((long)P * (wa + wb) + 255) / 256;
        if (zb > 4096) zb = 4096;
        hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)zb), dim3(256), 0, s, wa > 0 ? dz_density : nullptr, ld_dd, col_dd + 1, ld_dd, P,
                           wb > 0 ? dz_rgb : nullptr, ld_dr, 3, ld_dr);
    }
    return hos_launch_status();
}

extern "C" int hos_head_grad(const float* g_density, const float* density, const float* g_rgb, const float* rgb,
                             int P, float rgb_padding, float* dz_density, int ld_dd, int col_dd,
                             float* dz_rgb, int ld_dr, hos_stream_t stream) {
    if (P <= 0) return HOS_E_ARG;
    if (dz_density && !density) return HOS_E_ARG;
    if (dz_rgb && !rgb) return HOS_E_ARG;
    int blocks = hos_cdiv(P, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(head_grad_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g_density, density, g_rgb, rgb, P, rgb_padding, dz_density, ld_dd, col_dd, dz_rgb, ld_dr);
    return hos_launch_status();
}
