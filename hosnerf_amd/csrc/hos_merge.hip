// Stage-3 human (+) background composite and the NeRF-style `_raw2outputs`, one wavefront per ray.
//
//   hos_raw2outputs_{fwd,bwd}     M:73-99 (S3 module-level form) / N2:273-299 (S2, activations applied upstream)
//   hos_merge_composite_{fwd,bwd} M:1524-1596: re-project the human samples onto the background ray (C1),
//                                 fg/bg split + z-sort of 32 background + 128 human samples (C2),
//                                 masked alpha composite of the merged 160 samples (C3)
//
// The reference does this with torch.sort + three advanced-indexing gathers + cumprod on [B,160,*]
// temporaries.  Here a ray's 160 keys live in LDS, the (stable) sort is a rank-by-counting pass
// (160^2/64 compares per lane), and the transmittance is a wave-level product scan.
// HBM-bound: algorithmic bytes/ray = 4*(33 + 32*4 + 128*(4+3+1) + 6) in, 12 B out (+ 4*160 for total_order).
#include "hos_common.h"

namespace {

constexpr int MAXS = 256;        // merged samples per ray
constexpr int CHM = MAXS / 64;   // per-lane chunk

__device__ __forceinline__ float wave_incl_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float n = __shfl_up(v, o, 64);
        if (lane >= o) v *= n;
    }
    return v;
}

struct MergeLds {
    float key[MAXS];     // z (unsorted, by original index)
    float zs[MAXS];      // sorted z
    int order[MAXS];     // sorted position -> original index
};

// alpha/T/w for the lane's chunk of `per` consecutive sorted samples; samples >= S are padding.
struct Chain {
    float alpha[CHM], T[CHM], w[CHM], dist[CHM], e[CHM];   // e = exp(-sigma*dist)
};

__device__ __forceinline__ void composite_chain(const float* zs, int S, float dnorm, float last_dist, int per, int lane,
                                                const float (&sigma)[CHM], const float (&m)[CHM], Chain& c) {
    float f[CHM];
#pragma unroll
    for (int k = 0; k < CHM; ++k) {
        const int j = lane * per + k;
        c.alpha[k] = 0.f; c.dist[k] = 0.f; c.e[k] = 1.f;
        if (k < per && j < S) {
            const float d = (j == S - 1) ? last_dist : (zs[j + 1] - zs[j]);            // M:77-80
            c.dist[k] = d * dnorm;                                                       // M:81
            c.e[k] = expf(-sigma[k] * c.dist[k]);
            c.alpha[k] = (1.f - c.e[k]) * m[k];                                          // M:84-86
        }
        f[k] = (k < per && j < S) ? (1.f - c.alpha[k] + 1e-10f) : 1.f;                   // M:88-90
    }
    // exclusive product scan: within the lane, then across lanes
    float loc[CHM], run = 1.f;
#pragma unroll
    for (int k = 0; k < CHM; ++k) { loc[k] = run; run *= f[k]; }
    const float incl = wave_incl_scan_mul(run, lane);
    float base = __shfl_up(incl, 1, 64);
    if (lane == 0) base = 1.f;
#pragma unroll
    for (int k = 0; k < CHM; ++k) { c.T[k] = base * loc[k]; c.w[k] = c.alpha[k] * c.T[k]; }
}

// d loss / d alpha_k = gw_k T_k - (sum_{i>k} gw_i w_i) / (1 - alpha_k + 1e-10)
__device__ __forceinline__ void composite_chain_bwd(const Chain& c, const float (&gw)[CHM], int per, int lane, float (&galpha)[CHM]) {
    float v[CHM], loc[CHM], run = 0.f;
#pragma unroll
    for (int k = 0; k < CHM; ++k) v[k] = gw[k] * c.w[k];
#pragma unroll
    for (int k = CHM - 1; k >= 0; --k) { loc[k] = run; run += v[k]; }
    const float incl = wave_incl_rscan(run, lane);
    float base = __shfl_down(incl, 1, 64);
    if (lane == 63) base = 0.f;
#pragma unroll
    for (int k = 0; k < CHM; ++k) galpha[k] = gw[k] * c.T[k] - (base + loc[k]) / (1.f - c.alpha[k] + 1e-10f);
}

// ---------------------------------------------------------------------------------------- raw2outputs
__global__ __launch_bounds__(256) void raw2outputs_kernel(
    const float* __restrict__ rgbs, int rgb_ld, const float* __restrict__ sigma, int sig_ld,
    const float* __restrict__ z, const float* __restrict__ dirs, const float* __restrict__ mask,
    const float* __restrict__ bg, float last_dist, int B, int S,
    float* __restrict__ rgb, float* __restrict__ acc, float* __restrict__ weights, float* __restrict__ depth,
    // backward (all NULL in forward)
    const float* __restrict__ g_rgb, const float* __restrict__ g_w, float* __restrict__ g_rgbs, int grgb_ld,
    float* __restrict__ g_sigma, int gsig_ld, float* __restrict__ g_mask) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= B) return;
    const int per = (S + 63) / 64;
    const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    float sg[CHM], m[CHM];
#pragma unroll
    for (int k = 0; k < CHM; ++k) {
        const int j = lane * per + k;
        const bool ok = k < per && j < S;
        sg[k] = ok ? sigma[((size_t)ray * S + j) * sig_ld] : 0.f;
        m[k] = ok ? (mask ? mask[(size_t)ray * S + j] : 1.f) : 0.f;
    }
    Chain c;
    composite_chain(z + (size_t)ray * S, S, dnorm, last_dist, per, lane, sg, m, c);
    if (g_rgb == nullptr) {
        float r = 0.f, g = 0.f, b = 0.f, a = 0.f, dep = 0.f;
#pragma unroll
        for (int k = 0; k < CHM; ++k) {
            const int j = lane * per + k;
            if (k < per && j < S) {
                const float* col = rgbs + ((size_t)ray * S + j) * rgb_ld;
                r += c.w[k] * col[0]; g += c.w[k] * col[1]; b += c.w[k] * col[2];
                a += c.w[k]; dep += c.w[k] * z[(size_t)ray * S + j];
                if (weights) weights[(size_t)ray * S + j] = c.w[k];
            }
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); a = wave_sum(a); dep = wave_sum(dep);
        if (lane == 0) {
            if (bg) { const float t = 1.f - a; r += t * bg[0] / 255.f; g += t * bg[1] / 255.f; b += t * bg[2] / 255.f; }   // M:96-97
            rgb[ray * 3] = r; rgb[ray * 3 + 1] = g; rgb[ray * 3 + 2] = b;
            if (acc) acc[ray] = a;
            if (depth) depth[ray] = dep;
        }
    } else {
        const float g0 = g_rgb[ray * 3], g1 = g_rgb[ray * 3 + 1], g2 = g_rgb[ray * 3 + 2];
        const float gbg = bg ? (g0 * bg[0] + g1 * bg[1] + g2 * bg[2]) / 255.f : 0.f;   // d/dw of (1-acc)*bg/255
        float gw[CHM], ga[CHM];
#pragma unroll
        for (int k = 0; k < CHM; ++k) {
            const int j = lane * per + k;
            gw[k] = 0.f;
            if (k < per && j < S) {
                const float* col = rgbs + ((size_t)ray * S + j) * rgb_ld;
                gw[k] = g0 * col[0] + g1 * col[1] + g2 * col[2] - gbg + (g_w ? g_w[(size_t)ray * S + j] : 0.f);
                if (g_rgbs) { float* o = g_rgbs + ((size_t)ray * S + j) * grgb_ld; o[0] = g0 * c.w[k]; o[1] = g1 * c.w[k]; o[2] = g2 * c.w[k]; }
            }
        }
        composite_chain_bwd(c, gw, per, lane, ga);
#pragma unroll
        for (int k = 0; k < CHM; ++k) {
            const int j = lane * per + k;
            if (k < per && j < S) {
                if (g_sigma) g_sigma[((size_t)ray * S + j) * gsig_ld] = ga[k] * m[k] * c.dist[k] * c.e[k];
                if (g_mask) g_mask[(size_t)ray * S + j] = ga[k] * (1.f - c.e[k]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- merge composite
struct MergeArgs {
    const float* bkg_tdist; const float* bkg_rgb; const float* bkg_density;    // [B,Sb+1], [B,Sb,3], [B,Sb]
    const float* human; const float* pts; const float* mask;                   // [B,Sh,4], [B,Sh,3], [B,Sh]
    const float* rays_o; const float* rays_d; const float* A;                   // [B,3], [B,3], [4,4]
    const int* tiny_d_flag;                                                     // device flag: any |d| < 1e-5 (M:1526)
    int B, Sb, Sh;
    float thre_fg;
    // forward outputs
    float* rgb; int* idx_fg; int* total_order; float* hw_sorted; float* z_h;
    // backward
    const float* g_rgb; const float* g_hw; float* g_bkg_rgb; float* g_bkg_density; float* g_human; float* g_mask;
};

__global__ __launch_bounds__(256) void merge_composite_kernel(const MergeArgs a) {
    __shared__ MergeLds lds[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray_raw = blockIdx.x * 4 + wave;
    const bool live = ray_raw < a.B;
    const int ray = live ? ray_raw : a.B - 1;
    MergeLds& L = lds[wave];
    const int Sb = a.Sb, Sh = a.Sh, St = Sb + Sh;
    const float ox = a.rays_o[ray * 3], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
    const float dx = a.rays_d[ray * 3], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const bool first_comp = a.tiny_d_flag != nullptr && *a.tiny_d_flag != 0;
    // C1: human sample depth along the background ray (M:1524, :1526-1545)
    float msum = 0.f;
    for (int s = lane; s < Sh; s += 64) {
        const float* p = a.pts + ((size_t)ray * Sh + s) * 3;
        const float wx = a.A[0] * p[0] + a.A[1] * p[1] + a.A[2] * p[2] + a.A[3];
        const float wy = a.A[4] * p[0] + a.A[5] * p[1] + a.A[6] * p[2] + a.A[7];
        const float wz = a.A[8] * p[0] + a.A[9] * p[1] + a.A[10] * p[2] + a.A[11];
        float zh;
        if (first_comp) {   // first direction component that is not tiny
            if (fabsf(dx) > 1e-5f) zh = (wx - ox) / (dx + 1e-10f);
            else if (fabsf(dy) > 1e-5f) zh = (wy - oy) / (dy + 1e-10f);
            else zh = (wz - oz) / (dz + 1e-10f);
        } else {
            zh = ((wx - ox) / (dx + 1e-10f) + (wy - oy) / (dy + 1e-10f) + (wz - oz) / (dz + 1e-10f)) / 3.f;
        }
        L.key[Sb + s] = zh;
        if (a.z_h && live) a.z_h[(size_t)ray * Sh + s] = zh;
        msum += a.mask[(size_t)ray * Sh + s];
    }
    for (int s = lane; s < Sb; s += 64) L.key[s] = a.bkg_tdist[(size_t)ray * (Sb + 1) + s];
    msum = wave_sum(msum);
    const bool fg = msum > a.thre_fg;                                                   // M:1547-1551
    __syncthreads();
    const int S = fg ? St : Sb;
    if (fg) {
        // C2: stable sort by rank counting (ties keep original order: background first)
        for (int i = lane; i < St; i += 64) {
            const float ki = L.key[i];
            int r = 0;
            for (int j = 0; j < St; ++j) { const float kj = L.key[j]; r += (kj < ki) || (kj == ki && j < i); }
            L.order[r] = i;
            L.zs[r] = ki;
        }
    } else {
        for (int i = lane; i < Sb; i += 64) { L.order[i] = i; L.zs[i] = L.key[i]; }
    }
    __syncthreads();
    const int per = (S + 63) / 64;
    float sg[CHM], m[CHM], col[CHM][3];
    int src[CHM];
#pragma unroll
    for (int k = 0; k < CHM; ++k) {
        const int j = lane * per + k;
        sg[k] = 0.f; m[k] = 0.f; src[k] = -1; col[k][0] = col[k][1] = col[k][2] = 0.f;
        if (k < per && j < S) {
            const int i = L.order[j];
            src[k] = i;
            if (i < Sb) {
                const float* c3 = a.bkg_rgb + ((size_t)ray * Sb + i) * 3;
                col[k][0] = c3[0]; col[k][1] = c3[1]; col[k][2] = c3[2];
                sg[k] = a.bkg_density[(size_t)ray * Sb + i];
                m[k] = 1.f;                                                              // M:1574
            } else {
                const float* c4 = a.human + ((size_t)ray * Sh + (i - Sb)) * 4;
                col[k][0] = c4[0]; col[k][1] = c4[1]; col[k][2] = c4[2];
                sg[k] = c4[3];
                m[k] = a.mask[(size_t)ray * Sh + (i - Sb)];
            }
        }
    }
    Chain c;
    composite_chain(L.zs, S, dnorm, 1e10f, per, lane, sg, m, c);                         // C3 (M:1586 / :1592)
    if (a.g_rgb == nullptr) {
        float r = 0.f, g = 0.f, b = 0.f;
        int nh = 0;
#pragma unroll
        for (int k = 0; k < CHM; ++k) {
            if (src[k] >= 0) { r += c.w[k] * col[k][0]; g += c.w[k] * col[k][1]; b += c.w[k] * col[k][2]; nh += (src[k] >= Sb); }
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
        if (live && lane == 0) {
            a.rgb[ray * 3] = r; a.rgb[ray * 3 + 1] = g; a.rgb[ray * 3 + 2] = b;
            if (a.idx_fg) a.idx_fg[ray] = fg ? 1 : 0;
        }
        if (live && a.total_order)
            for (int j = lane; j < St; j += 64) a.total_order[(size_t)ray * St + j] = (fg && j < S) ? L.order[j] : -1;
        if (a.hw_sorted) {
            // weights of the human samples in sorted order (M:1588): exclusive count of human entries before me
            const float incl = wave_incl_scan((float)nh, lane);
            int before = (int)__shfl_up(incl, 1, 64);
            if (lane == 0) before = 0;
#pragma unroll
            for (int k = 0; k < CHM; ++k) {
                if (src[k] >= Sb) { if (live && fg) a.hw_sorted[(size_t)ray * Sh + before] = c.w[k]; ++before; }
            }
            if (live && !fg) for (int s = lane; s < Sh; s += 64) a.hw_sorted[(size_t)ray * Sh + s] = 0.f;
        }
    } else {
        const float g0 = a.g_rgb[ray * 3], g1 = a.g_rgb[ray * 3 + 1], g2 = a.g_rgb[ray * 3 + 2];
        float gw[CHM], ga[CHM];
        int nh = 0;
#pragma unroll
        for (int k = 0; k < CHM; ++k) nh += (src[k] >= Sb);
        const float incl = wave_incl_scan((float)nh, lane);
        int before = (int)__shfl_up(incl, 1, 64);
        if (lane == 0) before = 0;
#pragma unroll
        for (int k = 0; k < CHM; ++k) {
            gw[k] = 0.f;
            if (src[k] >= 0) {
                gw[k] = g0 * col[k][0] + g1 * col[k][1] + g2 * col[k][2];
                if (src[k] >= Sb) { if (a.g_hw && fg) gw[k] += a.g_hw[(size_t)ray * Sh + before]; ++before; }
            }
        }
        composite_chain_bwd(c, gw, per, lane, ga);
        if (live) {
#pragma unroll
            for (int k = 0; k < CHM; ++k) {
                const int i = src[k];
                if (i < 0) continue;
                const float gs = ga[k] * m[k] * c.dist[k] * c.e[k];
                if (i < Sb) {
                    if (a.g_bkg_rgb) { float* o = a.g_bkg_rgb + ((size_t)ray * Sb + i) * 3; o[0] = g0 * c.w[k]; o[1] = g1 * c.w[k]; o[2] = g2 * c.w[k]; }
                    if (a.g_bkg_density) a.g_bkg_density[(size_t)ray * Sb + i] = gs;
                } else {
                    float* o = a.g_human + ((size_t)ray * Sh + (i - Sb)) * 4;
                    o[0] = g0 * c.w[k]; o[1] = g1 * c.w[k]; o[2] = g2 * c.w[k]; o[3] = gs;
                    a.g_mask[(size_t)ray * Sh + (i - Sb)] = ga[k] * (1.f - c.e[k]);
                }
            }
            if (!fg) {   // human samples of a background-only ray receive no gradient
                for (int s = lane; s < Sh; s += 64) {
                    float* o = a.g_human + ((size_t)ray * Sh + s) * 4;
                    o[0] = o[1] = o[2] = o[3] = 0.f;
                    a.g_mask[(size_t)ray * Sh + s] = 0.f;
                }
            }
        }
    }
}

}  // namespace

extern "C" int hos_raw2outputs_fwd(const float* rgbs, int rgb_ld, const float* sigma, int sigma_ld, const float* z_vals,
                                   const float* rays_d, const float* mask, const float* bgcolor, float last_dist,
                                   int B, int S, float* rgb, float* acc, float* weights, float* depth,
                                   hos_stream_t stream) {
    if (!rgbs || !sigma || !z_vals || !rays_d || !rgb || B <= 0 || S <= 0) return HOS_E_ARG;
    if (S > MAXS) return HOS_E_SHAPE;
    hipLaunchKernelGGL(raw2outputs_kernel, dim3(hos_cdiv(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rgbs, rgb_ld, sigma, sigma_ld, z_vals, rays_d, mask, bgcolor, last_dist, B, S, rgb, acc, weights,
                       depth, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 0, (float*)nullptr, 0, (float*)nullptr);
    return hos_launch_status();
}

extern "C" int hos_raw2outputs_bwd(const float* g_rgb, const float* g_weights, const float* rgbs, int rgb_ld,
                                   const float* sigma, int sigma_ld, const float* z_vals, const float* rays_d,
                                   const float* mask, const float* bgcolor, float last_dist, int B, int S,
                                   float* g_rgbs, int g_rgb_ld, float* g_sigma, int g_sigma_ld, float* g_mask,
                                   hos_stream_t stream) {
    if (!g_rgb || !rgbs || !sigma || !z_vals || !rays_d || B <= 0 || S <= 0) return HOS_E_ARG;
    if (S > MAXS) return HOS_E_SHAPE;
    hipLaunchKernelGGL(raw2outputs_kernel, dim3(hos_cdiv(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rgbs, rgb_ld, sigma, sigma_ld, z_vals, rays_d, mask, bgcolor, last_dist, B, S, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, g_rgb, g_weights, g_rgbs, g_rgb_ld, g_sigma,
                       g_sigma_ld, g_mask);
    return hos_launch_status();
}

static int merge_launch(MergeArgs& a, hipStream_t s) {
    if (!a.bkg_tdist || !a.bkg_rgb || !a.bkg_density || !a.human || !a.pts || !a.mask || !a.rays_o || !a.rays_d || !a.A)
        return HOS_E_ARG;
    if (a.B <= 0 || a.Sb <= 0 || a.Sh <= 0 || a.Sb + a.Sh > MAXS) return HOS_E_SHAPE;
    hipLaunchKernelGGL(merge_composite_kernel, dim3(hos_cdiv(a.B, 4)), dim3(256), 0, s, a);
    return hos_launch_status();
}

extern "C" int hos_merge_composite_fwd(const float* bkg_tdist, const float* bkg_rgb, const float* bkg_density,
                                       const float* human_rgbsigma, const float* newsmpl_pts, const float* pts_mask,
                                       const float* rays_o_bkg, const float* rays_d_bkg, const float* smpl_to_world,
                                       const int32_t* tiny_d_flag, int B, int Sb, int Sh, float thre_fg,
                                       float* rgb, int32_t* idx_fg, int32_t* total_order, float* human_weights_sorted,
                                       float* z_human, hos_stream_t stream) {
    MergeArgs a{};
    a.bkg_tdist = bkg_tdist; a.bkg_rgb = bkg_rgb; a.bkg_density = bkg_density; a.human = human_rgbsigma;
    a.pts = newsmpl_pts; a.mask = pts_mask; a.rays_o = rays_o_bkg; a.rays_d = rays_d_bkg; a.A = smpl_to_world;
    a.tiny_d_flag = tiny_d_flag; a.B = B; a.Sb = Sb; a.Sh = Sh; a.thre_fg = thre_fg;
    a.rgb = rgb; a.idx_fg = idx_fg; a.total_order = total_order; a.hw_sorted = human_weights_sorted; a.z_h = z_human;
    if (!rgb) return HOS_E_ARG;
    return merge_launch(a, static_cast<hipStream_t>(stream));
}

extern "C" int hos_merge_composite_bwd(const float* g_rgb, const float* g_human_weights_sorted,
                                       const float* bkg_tdist, const float* bkg_rgb, const float* bkg_density,
                                       const float* human_rgbsigma, const float* newsmpl_pts, const float* pts_mask,
                                       const float* rays_o_bkg, const float* rays_d_bkg, const float* smpl_to_world,
                                       const int32_t* tiny_d_flag, int B, int Sb, int Sh, float thre_fg,
                                       float* g_bkg_rgb, float* g_bkg_density, float* g_human_rgbsigma, float* g_pts_mask,
                                       hos_stream_t stream) {
    MergeArgs a{};
    a.bkg_tdist = bkg_tdist; a.bkg_rgb = bkg_rgb; a.bkg_density = bkg_density; a.human = human_rgbsigma;
    a.pts = newsmpl_pts; a.mask = pts_mask; a.rays_o = rays_o_bkg; a.rays_d = rays_d_bkg; a.A = smpl_to_world;
    a.tiny_d_flag = tiny_d_flag; a.B = B; a.Sb = Sb; a.Sh = Sh; a.thre_fg = thre_fg;
    a.g_rgb = g_rgb; a.g_hw = g_human_weights_sorted; a.g_bkg_rgb = g_bkg_rgb; a.g_bkg_density = g_bkg_density;
    a.g_human = g_human_rgbsigma; a.g_mask = g_pts_mask;
    if (!g_rgb || !g_human_rgbsigma || !g_pts_mask) return HOS_E_ARG;
    return merge_launch(a, static_cast<hipStream_t>(stream));
}
