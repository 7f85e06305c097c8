// Human-object branch, per-sample kernels (one thread per sample point, bone transforms in LDS).
//
//   hos_human_sample_warp   N:409-424 (ray samples, stratified jitter) + N:451 (points) +
//                           N:304-355 (_sample_motion_fields: backward LBS through the motion-weight volume)
//   hos_lbs_forward         N:357-399 (_sample_motion_fields_forward)
//   hos_embed_hannw         embedders/hannw_fourier.py:15-71 (+ condition code columns, mlp_offset.py:55)
//   hos_embed_fourier       embedders/fourier.py:11-57 (+ state embedding columns, N:248-249)
//
// The reference runs 52 grid_sample launches + ~80 elementwise launches per chunk for the warp; here
// the 26 bone transforms (26x12 floats) sit in LDS, the 27x32^3 volume (3.5 MB) is L2/MALL resident
// and each point does its 26x8 trilinear taps in registers.  Gather-bound: algorithmic bytes per
// point = 12 B in (o,d,near,far amortised) + 32 B out (z, pts, x_skel, mask) + 26*8*4 B of L2 gathers.
#include "hos_common.h"

namespace {

constexpr int KMAX = 32;   // bones

// F.grid_sample(..., mode='bilinear', padding_mode='zeros', align_corners=True) on one channel of a
// [V,V,V] (z,y,x) volume at normalised (gx,gy,gz); tap order and weights as PyTorch's 3-D kernel.
__device__ __forceinline__ float trilinear_zero(const float* __restrict__ vol, int V, float gx, float gy, float gz) {
    const float ix = ((gx + 1.f) / 2.f) * (float)(V - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(V - 1);
    const float iz = ((gz + 1.f) / 2.f) * (float)(V - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    // NaN / huge coordinates: every tap out of range -> 0
    if (!(fx >= -1.f && fx <= (float)V && fy >= -1.f && fy <= (float)V && fz >= -1.f && fz <= (float)V)) return 0.f;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz;
    const float wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
    float out = 0.f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                if (x >= 0 && x < V && y >= 0 && y < V && z >= 0 && z < V) {
                    const float w = (dx ? wx1 : wx0) * (dy ? wy1 : wy0) * (dz ? wz1 : wz0);
                    out += vol[((size_t)z * V + y) * V + x] * w;
                }
            }
    return out;
}

__global__ __launch_bounds__(256) void human_sample_warp_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ near_,
    const float* __restrict__ far_, const float* __restrict__ t_vals, const float* __restrict__ t_rand,
    const float* __restrict__ R, const float* __restrict__ T, const float* __restrict__ vol, int V,
    const float* __restrict__ bbox_min, const float* __restrict__ bbox_scale, int B, int N, int K,
    float* __restrict__ z_vals, float* __restrict__ pts, float* __restrict__ x_skel, float* __restrict__ mask) {
    __shared__ float sR[KMAX * 9], sT[KMAX * 3], sB[6];
    for (int i = threadIdx.x; i < K * 9; i += blockDim.x) sR[i] = R[i];
    for (int i = threadIdx.x; i < K * 3; i += blockDim.x) sT[i] = T[i];
    if (threadIdx.x < 3) { sB[threadIdx.x] = bbox_min[threadIdx.x]; sB[3 + threadIdx.x] = bbox_scale[threadIdx.x]; }
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)B * N) return;
    const int ray = (int)(p / N), s = (int)(p % N);
    const float nr = near_[ray], fr = far_[ray];
    auto zf = [&](int k) { const float t = t_vals[k]; return nr * (1.f - t) + fr * t; };   // N:411
    float z = zf(s);
    if (t_rand != nullptr) {                                                               // N:416-424
        const float lower = (s == 0) ? z : 0.5f * (z + zf(s - 1));
        const float upper = (s == N - 1) ? z : 0.5f * (zf(s + 1) + z);
        z = lower + (upper - lower) * t_rand[p];
    }
    const float px = rays_o[ray * 3 + 0] + rays_d[ray * 3 + 0] * z;                        // N:451
    const float py = rays_o[ray * 3 + 1] + rays_d[ray * 3 + 1] * z;
    const float pz = rays_o[ray * 3 + 2] + rays_d[ray * 3 + 2] * z;
    float wsum = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    const size_t V3 = (size_t)V * V * V;
    for (int i = 0; i < K; ++i) {
        const float* r = sR + i * 9;
        const float qx = (r[0] * px + r[1] * py + r[2] * pz) + sT[i * 3 + 0];              // N:319
        const float qy = (r[3] * px + r[4] * py + r[5] * pz) + sT[i * 3 + 1];
        const float qz = (r[6] * px + r[7] * py + r[8] * pz) + sT[i * 3 + 2];
        const float gx = (qx - sB[0]) * sB[3] - 1.f, gy = (qy - sB[1]) * sB[4] - 1.f, gz = (qz - sB[2]) * sB[5] - 1.f;
        const float w = trilinear_zero(vol + i * V3, V, gx, gy, gz);                        // N:322-324
        wsum += w;
        ax += w * qx; ay += w * qy; az += w * qz;                                           // N:333-338
    }
    const float den = fmaxf(wsum, 1e-4f);                                                   // N:339
    if (z_vals) z_vals[p] = z;
    if (pts) { pts[p * 3] = px; pts[p * 3 + 1] = py; pts[p * 3 + 2] = pz; }
    x_skel[p * 3] = ax / den; x_skel[p * 3 + 1] = ay / den; x_skel[p * 3 + 2] = az / den;
    mask[p] = wsum;
}

// forward LBS: one K-channel tap at the canonical point; volume given channel-LAST [V,V,V,CL]
__global__ __launch_bounds__(256) void lbs_forward_kernel(const float* __restrict__ cnl, const float* __restrict__ R,
                                                          const float* __restrict__ T, const float* __restrict__ vol_cl,
                                                          int V, int CL, const float* __restrict__ bbox_min,
                                                          const float* __restrict__ bbox_scale, long P, int K,
                                                          float* __restrict__ x_def) {
    __shared__ float sR[KMAX * 9], sT[KMAX * 3], sB[6];
    for (int i = threadIdx.x; i < K * 9; i += blockDim.x) sR[i] = R[i];
    for (int i = threadIdx.x; i < K * 3; i += blockDim.x) sT[i] = T[i];
    if (threadIdx.x < 3) { sB[threadIdx.x] = bbox_min[threadIdx.x]; sB[3 + threadIdx.x] = bbox_scale[threadIdx.x]; }
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float px = cnl[p * 3], py = cnl[p * 3 + 1], pz = cnl[p * 3 + 2];
    const float gx = (px - sB[0]) * sB[3] - 1.f, gy = (py - sB[1]) * sB[4] - 1.f, gz = (pz - sB[2]) * sB[5] - 1.f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(V - 1), iy = ((gy + 1.f) / 2.f) * (float)(V - 1), iz = ((gz + 1.f) / 2.f) * (float)(V - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    float w[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) w[i] = 0.f;
    if (fx >= -1.f && fx <= (float)V && fy >= -1.f && fy <= (float)V && fz >= -1.f && fz <= (float)V) {
        const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
        const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz;
        const float wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
        for (int dz = 0; dz < 2; ++dz)
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                    if (x >= 0 && x < V && y >= 0 && y < V && z >= 0 && z < V) {
                        const float tw = (dx ? wx1 : wx0) * (dy ? wy1 : wy0) * (dz ? wz1 : wz0);
                        const float4* vp = reinterpret_cast<const float4*>(vol_cl + (((size_t)z * V + y) * V + x) * CL);
#pragma unroll
                        for (int q = 0; q < KMAX / 4; ++q) {
                            if (q * 4 < K) {
                                const float4 v = vp[q];
                                w[q * 4 + 0] += v.x * tw; w[q * 4 + 1] += v.y * tw; w[q * 4 + 2] += v.z * tw; w[q * 4 + 3] += v.w * tw;
                            }
                        }
                    }
                }
    }
    float wsum = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        if (i < K) {
            const float* r = sR + i * 9;
            const float qx = (r[0] * px + r[1] * py + r[2] * pz) + sT[i * 3 + 0];
            const float qy = (r[3] * px + r[4] * py + r[5] * pz) + sT[i * 3 + 1];
            const float qz = (r[6] * px + r[7] * py + r[8] * pz) + sT[i * 3 + 2];
            wsum += w[i];
            ax += w[i] * qx; ay += w[i] * qy; az += w[i] * qz;
        }
    }
    const float den = fmaxf(wsum, 1e-4f);
    x_def[p * 3] = ax / den; x_def[p * 3 + 1] = ay / den; x_def[p * 3 + 2] = az / den;
}

// hann-windowed Fourier features of x (no identity): [w_j sin(2^j x), w_j cos(2^j x)]_j, 3 each.
// Row layout of E [P, lde]: cols [0,C) = condition code, [C, C+6F) = features, rest 0.
// Optionally also writes the features alone into PE [P, ldpe] (zero padded) for the skip concat.
__global__ __launch_bounds__(256) void embed_hannw_kernel(const float* __restrict__ x, const float* __restrict__ band_w,
                                                          int F, const float* __restrict__ cond, int C, long P,
                                                          float* __restrict__ E, int lde, float* __restrict__ PE, int ldpe) {
    const int W = max(lde, PE ? C + ldpe : 0);     // iterate a virtual row wide enough for both destinations
    const long total = P * W;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long p = it / W;
        const int c = (int)(it % W);
        float v = 0.f;
        const int f = c - C;
        if (c < C) {
            v = cond[c];
        } else if (f < 6 * F) {
            const int j = f / 6, r = f % 6, ax = r % 3;
            const float a = x[p * 3 + ax] * (float)(1 << j);
            v = band_w[j] * ((r < 3) ? sinf(a) : cosf(a));
        }
        if (c < lde) E[p * lde + c] = v;
        if (PE && f >= 0 && f < ldpe) PE[p * ldpe + f] = v;
    }
}

// [x(3), sin(2^j x), cos(2^j x) (j < F)] | state embedding (NE) | 0-pad; optional second destination E2.
__global__ __launch_bounds__(256) void embed_fourier_kernel(const float* __restrict__ x, int F,
                                                            const float* __restrict__ state, int NE, long P,
                                                            float* __restrict__ E, int lde, float* __restrict__ E2, int lde2) {
    const long total = P * lde;
    const int nf = 3 + 6 * F;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long p = it / lde;
        const int c = (int)(it % lde);
        float v = 0.f;
        if (c < 3) {
            v = x[p * 3 + c];
        } else if (c < nf) {
            const int f = c - 3, j = f / 6, r = f % 6, ax = r % 3;
            const float a = x[p * 3 + ax] * (float)(1 << j);
            v = (r < 3) ? sinf(a) : cosf(a);
        } else if (c < nf + NE) {
            v = state[c - nf];
        }
        E[it] = v;
        if (E2 && c < nf + NE) E2[p * lde2 + c] = v;
    }
}

inline int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int hos_human_sample_warp(const float* rays_o, const float* rays_d, const float* near_, const float* far_,
                                     const float* t_vals, const float* t_rand, const float* R, const float* T,
                                     const float* vol, int V, const float* bbox_min, const float* bbox_scale,
                                     int B, int N, int K, float* z_vals, float* pts, float* x_skel, float* mask,
                                     hos_stream_t stream) {
    if (!rays_o || !rays_d || !near_ || !far_ || !t_vals || !R || !T || !vol || !bbox_min || !bbox_scale || !x_skel || !mask)
        return HOS_E_ARG;
    if (B <= 0 || N <= 0 || K <= 0 || K > KMAX || V < 2) return HOS_E_SHAPE;
    const long P = (long)B * N;
    hipLaunchKernelGGL(human_sample_warp_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), rays_o, rays_d, near_, far_, t_vals, t_rand, R, T, vol, V,
                       bbox_min, bbox_scale, B, N, K, z_vals, pts, x_skel, mask);
    return hos_launch_status();
}

extern "C" int hos_lbs_forward(const float* cnl_pts, const float* R_fwd, const float* T_fwd, const float* vol_cl,
                               int V, int CL, const float* bbox_min, const float* bbox_scale, int64_t P, int K,
                               float* x_deform, hos_stream_t stream) {
    if (!cnl_pts || !R_fwd || !T_fwd || !vol_cl || !bbox_min || !bbox_scale || !x_deform || P <= 0) return HOS_E_ARG;
    if (K <= 0 || K > KMAX || CL < K || (CL & 3) || V < 2) return HOS_E_SHAPE;
    HOS_CHECK_ALIGN16(vol_cl);
    hipLaunchKernelGGL(lbs_forward_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), cnl_pts, R_fwd, T_fwd, vol_cl, V, CL, bbox_min, bbox_scale,
                       (long)P, K, x_deform);
    return hos_launch_status();
}

extern "C" int hos_embed_hannw(const float* x, const float* band_w, int num_freqs, const float* cond, int cond_size,
                               int64_t P, float* E, int lde, float* PE, int ldpe, hos_stream_t stream) {
    if (!x || !band_w || !E || P <= 0 || (cond_size > 0 && !cond)) return HOS_E_ARG;
    if (num_freqs < 1 || num_freqs > 16 || lde < cond_size + 6 * num_freqs || (PE && ldpe < 6 * num_freqs)) return HOS_E_SHAPE;
    hipLaunchKernelGGL(embed_hannw_kernel, dim3(grid_for(P * (lde > cond_size + ldpe ? lde : cond_size + ldpe))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, band_w, num_freqs, cond, cond_size, (long)P, E, lde, PE, ldpe);
    return hos_launch_status();
}

extern "C" int hos_embed_fourier(const float* x, int num_freqs, const float* state, int state_size, int64_t P,
                                 float* E, int lde, float* E2, int lde2, hos_stream_t stream) {
    if (!x || !E || P <= 0 || (state_size > 0 && !state)) return HOS_E_ARG;
    if (num_freqs < 1 || num_freqs > 16 || lde < 3 + 6 * num_freqs + state_size) return HOS_E_SHAPE;
    if (E2 && lde2 < 3 + 6 * num_freqs + state_size) return HOS_E_SHAPE;
    hipLaunchKernelGGL(embed_fourier_kernel, dim3(grid_for(P * lde)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, num_freqs, state, state_size, (long)P, E, lde, E2, lde2);
    return hos_launch_status();
}
