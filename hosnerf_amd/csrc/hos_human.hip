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

#include <cstdlib>
// workgroups of the persistent backward kernels (they keep per-block partial sums in LDS and flush them once).  768 = three 256-thread
// workgroups per CU: these kernels are gather-bound (26 x 8 taps per point out of L2) and one wave per SIMD hides little of that
// latency.  Round 5, step level, three alternations on one box (profiles/r05_persist_grid_sweep.txt): stage 2 8.13-8.18 ms at 256,
// 8.02-8.05 at 384, 7.99-8.03 at 512, 7.99-8.00 at 768; stage 3 31.33-31.36 -> 31.29-31.31; 512-ray step equal.
static inline long persist_grid() {
    static const long g = getenv("HOS_PERSIST_GRID") ? atol(getenv("HOS_PERSIST_GRID")) : 768;
    return g > 0 ? g : 768;
}

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
                                                          float* __restrict__ x_def, const int* __restrict__ p_dev) {
    __shared__ float sR[KMAX * 9], sT[KMAX * 3], sB[6];
    if (p_dev) P = min(P, (long)*p_dev);          // fixed-capacity buffer: only the first *p_dev rows are live
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
                                                          float* __restrict__ E, int lde, float* __restrict__ PE, int ldpe,
                                                          const int* __restrict__ p_dev) {
    if (p_dev) P = min(P, (long)*p_dev);
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

// Tiled forms of the two embedders (the per-element kernels above stay as the fallback for rows that are not 16-byte
// aligned): a workgroup takes 64 rows, evaluates their sin / cos features ONCE into LDS (both destinations read them) and
// writes whole rows with 16-byte stores; index arithmetic is 32-bit and per row, not a 64-bit division per element
// (the per-element kernels ran at 2.2-2.6 TB/s of writes).
constexpr int EM_RB = 64;          // rows per workgroup pass
constexpr int EM_FMAX = 100;       // features per row kept in LDS (3 + 6 x 16 rounded up)

__global__ __launch_bounds__(256) void embed_hannw_tiled_kernel(const float* __restrict__ x, const float* __restrict__ band_w,
                                                                int F, const float* __restrict__ cond, int C, long P,
                                                                float* __restrict__ E, int lde, float* __restrict__ PE, int ldpe,
                                                                const int* __restrict__ p_dev) {
    __shared__ float sF[EM_RB][EM_FMAX + 1];
    if (p_dev) P = min(P, (long)*p_dev);
    const int t = threadIdx.x, nf = 6 * F;
    for (long row0 = (long)blockIdx.x * EM_RB; row0 < P; row0 += (long)gridDim.x * EM_RB) {
        const int rows = (int)min((long)EM_RB, P - row0);
        for (int i = t; i < rows * 3 * F; i += 256) {       // one (frequency, axis) per thread: sin and cos share the range reduction
            const int r = i / (3 * F), e = i - r * (3 * F);
            const int j = e / 3, ax = e - j * 3;
            const float a = x[(row0 + r) * 3 + ax] * (float)(1 << j);
            float sn, cs;
            sincosf(a, &sn, &cs);
            sF[r][j * 6 + ax] = band_w[j] * sn;
            sF[r][j * 6 + 3 + ax] = band_w[j] * cs;
        }
        __syncthreads();
        const int qe = lde >> 2;
        for (int i = t; i < rows * qe; i += 256) {
            const int r = i / qe, c0 = (i - r * qe) * 4;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k, f = c - C;
                v[k] = c < C ? cond[c] : (f < nf ? sF[r][f] : 0.f);
            }
            *reinterpret_cast<float4*>(E + (row0 + r) * lde + c0) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (PE != nullptr) {
            const int qp = ldpe >> 2;
            for (int i = t; i < rows * qp; i += 256) {
                const int r = i / qp, c0 = (i - r * qp) * 4;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (c0 + k < nf) ? sF[r][c0 + k] : 0.f;
                *reinterpret_cast<float4*>(PE + (row0 + r) * ldpe + c0) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void embed_fourier_tiled_kernel(const float* __restrict__ x, int F,
                                                                  const float* __restrict__ state, int NE, long P,
                                                                  float* __restrict__ E, int lde, float* __restrict__ E2, int lde2) {
    __shared__ float sF[EM_RB][EM_FMAX + 1];
    const int t = threadIdx.x, nf = 3 + 6 * F, nv = nf + NE;      // nv = columns that carry a value
    for (long row0 = (long)blockIdx.x * EM_RB; row0 < P; row0 += (long)gridDim.x * EM_RB) {
        const int rows = (int)min((long)EM_RB, P - row0);
        for (int i = t; i < rows * 3 * (F + 1); i += 256) {  // e < 3: x itself; then one (frequency, axis) per thread
            const int r = i / (3 * (F + 1)), e = i - r * (3 * (F + 1));
            if (e < 3) {
                sF[r][e] = x[(row0 + r) * 3 + e];
            } else {
                const int j = (e - 3) / 3, ax = (e - 3) - j * 3;
                const float a = x[(row0 + r) * 3 + ax] * (float)(1 << j);
                float sn, cs;
                sincosf(a, &sn, &cs);
                sF[r][3 + j * 6 + ax] = sn;
                sF[r][3 + j * 6 + 3 + ax] = cs;
            }
        }
        __syncthreads();
        const int qe = lde >> 2;
        for (int i = t; i < rows * qe; i += 256) {
            const int r = i / qe, c0 = (i - r * qe) * 4;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                v[k] = c < nf ? sF[r][c] : (c < nv ? state[c - nf] : 0.f);
            }
            *reinterpret_cast<float4*>(E + (row0 + r) * lde + c0) = make_float4(v[0], v[1], v[2], v[3]);
            if (E2 != nullptr && c0 < nv) {               // second destination: only the nv value columns (the rest of its row is not ours)
                float* d = E2 + (row0 + r) * lde2 + c0;
                if (c0 + 3 < nv) *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (c0 + k < nv) d[k] = v[k];
                }
            }
        }
        __syncthreads();
    }
}

inline bool em_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int em_grid(long P) {
    const long b = (P + EM_RB - 1) / EM_RB;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
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
                               float* x_deform, const int32_t* rows_dev, hos_stream_t stream) {
    if (!cnl_pts || !R_fwd || !T_fwd || !vol_cl || !bbox_min || !bbox_scale || !x_deform || P <= 0) return HOS_E_ARG;
    if (K <= 0 || K > KMAX || CL < K || (CL & 3) || V < 2) return HOS_E_SHAPE;
    HOS_CHECK_ALIGN16(vol_cl);
    hipLaunchKernelGGL(lbs_forward_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), cnl_pts, R_fwd, T_fwd, vol_cl, V, CL, bbox_min, bbox_scale,
                       (long)P, K, x_deform, rows_dev);
    return hos_launch_status();
}

extern "C" int hos_embed_hannw(const float* x, const float* band_w, int num_freqs, const float* cond, int cond_size,
                               int64_t P, float* E, int lde, float* PE, int ldpe, const int32_t* rows_dev, hos_stream_t stream) {
    if (!x || !band_w || !E || P <= 0 || (cond_size > 0 && !cond)) return HOS_E_ARG;
    if (num_freqs < 1 || num_freqs > 16 || lde < cond_size + 6 * num_freqs || (PE && ldpe < 6 * num_freqs)) return HOS_E_SHAPE;
    if (!(lde & 3) && em_al16(E) && (!PE || (!(ldpe & 3) && em_al16(PE))) && 6 * num_freqs <= EM_FMAX) {
        hipLaunchKernelGGL(embed_hannw_tiled_kernel, dim3(em_grid(P)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x, band_w, num_freqs, cond, cond_size, (long)P, E, lde, PE, ldpe, rows_dev);
        return hos_launch_status();
    }
    hipLaunchKernelGGL(embed_hannw_kernel, dim3(grid_for(P * (lde > cond_size + ldpe ? lde : cond_size + ldpe))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, band_w, num_freqs, cond, cond_size, (long)P, E, lde, PE, ldpe, rows_dev);
    return hos_launch_status();
}

extern "C" int hos_embed_fourier(const float* x, int num_freqs, const float* state, int state_size, int64_t P,
                                 float* E, int lde, float* E2, int lde2, hos_stream_t stream) {
    if (!x || !E || P <= 0 || (state_size > 0 && !state)) return HOS_E_ARG;
    if (num_freqs < 1 || num_freqs > 16 || lde < 3 + 6 * num_freqs + state_size) return HOS_E_SHAPE;
    if (E2 && lde2 < 3 + 6 * num_freqs + state_size) return HOS_E_SHAPE;
    if (!(lde & 3) && em_al16(E) && (!E2 || (!(lde2 & 3) && em_al16(E2))) && 3 + 6 * num_freqs <= EM_FMAX) {
        hipLaunchKernelGGL(embed_fourier_tiled_kernel, dim3(em_grid(P)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x, num_freqs, state, state_size, (long)P, E, lde, E2, lde2);
        return hos_launch_status();
    }
    hipLaunchKernelGGL(embed_fourier_kernel, dim3(grid_for(P * lde)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, num_freqs, state, state_size, (long)P, E, lde, E2, lde2);
    return hos_launch_status();
}

// ================================================================================================
// Backward kernels of the human branch (training stages 2/3).
// ================================================================================================
namespace {

// value and spatial gradient (w.r.t. the *normalised* grid coordinate) of the zero-padded trilinear tap;
// optionally scatters g_out * tapweight into g_vol (same geometry as trilinear_zero).
// The scatter is run-aggregated: the lanes of a wave are consecutive samples of a ray, so neighbouring lanes mostly fall
// into the same voxel cell (~3 samples per cell); a segmented scan over runs of equal cell index sums their eight tap
// contributions and only the last lane of a run issues the atomics.  An fp32 atomic costs one request per cache line it
// touches (scripts/probe/atomic_probe.hip), and these are scattered -- each bone taps its own position --, so the
// number of issuing lanes is what matters.  MUST be called by all 64 lanes of the wave (uses shuffles).
__device__ __forceinline__ float trilinear_zero_grad(const float* __restrict__ vol, float* __restrict__ g_vol, int V,
                                                     float gx, float gy, float gz, float g_out, float (&dgrid)[3], bool scatter) {
    const int lane = threadIdx.x & 63;
    const float half = 0.5f * (float)(V - 1);
    const float ix = ((gx + 1.f) / 2.f) * (float)(V - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(V - 1);
    const float iz = ((gz + 1.f) / 2.f) * (float)(V - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    dgrid[0] = dgrid[1] = dgrid[2] = 0.f;
    const bool inside = (fx >= -1.f && fx <= (float)V && fy >= -1.f && fy <= (float)V && fz >= -1.f && fz <= (float)V);
    const int x0 = inside ? (int)fx : 0, y0 = inside ? (int)fy : 0, z0 = inside ? (int)fz : 0;
    const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz;
    const float wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
    float out = 0.f;
    float contrib[8];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                float cw = 0.f;
                if (inside && x >= 0 && x < V && y >= 0 && y < V && z >= 0 && z < V) {
                    const float wx = dx ? wx1 : wx0, wy = dy ? wy1 : wy0, wz = dz ? wz1 : wz0;
                    const size_t idx = ((size_t)z * V + y) * V + x;
                    const float v = vol[idx];
                    out += v * (wx * wy * wz);
                    dgrid[0] += v * (dx ? 1.f : -1.f) * wy * wz;
                    dgrid[1] += v * wx * (dy ? 1.f : -1.f) * wz;
                    dgrid[2] += v * wx * wy * (dz ? 1.f : -1.f);
                    cw = g_out * (wx * wy * wz);
                }
                contrib[dz * 4 + dy * 2 + dx] = cw;
            }
    dgrid[0] *= half; dgrid[1] *= half; dgrid[2] *= half;
    if (g_vol != nullptr) {       // wave-uniform
        const bool act = scatter && inside && g_out != 0.f;
        const int key = act ? ((z0 + 1) * (V + 2) + (y0 + 1)) * (V + 2) + (x0 + 1) : -1 - lane;      // cell index; unique when idle
        const int key_prev = __shfl_up(key, 1, 64);
        int flag = (lane == 0 || key != key_prev) ? 1 : 0;                   // head of a run
        const int key_next = __shfl_down(key, 1, 64);
        const bool tail = (lane == 63) || (key_next != key);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {                             // segmented inclusive scan of the 8 tap sums
            const int f_up = __shfl_up(flag, off, 64);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float v_up = __shfl_up(contrib[t], off, 64);
                if (lane >= off && !flag) contrib[t] += v_up;
            }
            if (lane >= off) flag |= f_up;
        }
#ifdef HOS_EXP_NO_SCATTER      // timing experiment: everything but the atomics (results invalid)
        if (false) {
#else
        if (act && tail) {
#endif
#pragma unroll
            for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                        const float cw = contrib[dz * 4 + dy * 2 + dx];
                        if (x >= 0 && x < V && y >= 0 && y < V && z >= 0 && z < V && cw != 0.f)
                            __hip_atomic_fetch_add(g_vol + ((size_t)z * V + y) * V + x, cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
        }
    }
    return out;
}

// x_skel = sum_i w_i q_i / max(sum w, 1e-4), mask = sum w;  q_i = R_i p + T_i;  w_i = tap(vol_i, g(q_i)).
// Gradients: g_vol (atomics), g_R [K,9], g_T [K,3] (wave reduce -> LDS -> one atomic per block and entry).
__global__ __launch_bounds__(256) void human_sample_warp_bwd_kernel(
    const float* __restrict__ pts, const float* __restrict__ R, const float* __restrict__ T,
    const float* __restrict__ vol, int V, const float* __restrict__ bbox_min, const float* __restrict__ bbox_scale,
    long P, int K, const float* __restrict__ g_xskel, const float* __restrict__ g_mask,
    float* __restrict__ g_vol, float* __restrict__ g_R, float* __restrict__ g_T, float* __restrict__ aux,
    const float* __restrict__ fwd_xskel, const float* __restrict__ fwd_mask) {
    __shared__ float sR[KMAX * 9], sT[KMAX * 3], sB[6], sAcc[KMAX * 12];
    for (int i = threadIdx.x; i < K * 9; i += blockDim.x) sR[i] = R[i];
    for (int i = threadIdx.x; i < K * 3; i += blockDim.x) sT[i] = T[i];
    for (int i = threadIdx.x; i < K * 12; i += blockDim.x) sAcc[i] = 0.f;
    if (threadIdx.x < 3) { sB[threadIdx.x] = bbox_min[threadIdx.x]; sB[3 + threadIdx.x] = bbox_scale[threadIdx.x]; }
    __syncthreads();
    // persistent workgroups (see lbs_forward_bwd_kernel): the R/T gradient sums stay in LDS across 256-point chunks
    const long nchunks = (P + 255) / 256;
    for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const long p = chunk * 256 + threadIdx.x;
    const bool live = p < P;
    const long pp = live ? p : P - 1;
    const int lane = threadIdx.x & 63;
    const float px = pts[pp * 3], py = pts[pp * 3 + 1], pz = pts[pp * 3 + 2];
    const size_t V3 = (size_t)V * V * V;
    // pass 1: wsum and x_skel -- the forward kernel's own outputs when the caller kept them (mask = wsum, x_skel = the
    // quotients below, computed by the same instruction sequence: bit-identical), else recomputed (K x 8 more taps per point)
    float wsum, xs, ys, zs;
    if (fwd_xskel != nullptr) {
        wsum = fwd_mask[pp];
        xs = fwd_xskel[pp * 3]; ys = fwd_xskel[pp * 3 + 1]; zs = fwd_xskel[pp * 3 + 2];
    } else {
        float ax = 0.f, ay = 0.f, az = 0.f;
        wsum = 0.f;
        for (int i = 0; i < K; ++i) {
            const float* r = sR + i * 9;
            const float qx = (r[0] * px + r[1] * py + r[2] * pz) + sT[i * 3 + 0];
            const float qy = (r[3] * px + r[4] * py + r[5] * pz) + sT[i * 3 + 1];
            const float qz = (r[6] * px + r[7] * py + r[8] * pz) + sT[i * 3 + 2];
            const float w = trilinear_zero(vol + i * V3, V, (qx - sB[0]) * sB[3] - 1.f, (qy - sB[1]) * sB[4] - 1.f, (qz - sB[2]) * sB[5] - 1.f);
            wsum += w; ax += w * qx; ay += w * qy; az += w * qz;
        }
        const float d0 = fmaxf(wsum, 1e-4f);
        xs = ax / d0; ys = ay / d0; zs = az / d0;
    }
    const float den = fmaxf(wsum, 1e-4f);
    const float clampg = (wsum >= 1e-4f) ? 1.f : 0.f;        // clamp(min) passes the gradient at >=
    const float gx_ = live ? g_xskel[pp * 3] : 0.f, gy_ = live ? g_xskel[pp * 3 + 1] : 0.f, gz_ = live ? g_xskel[pp * 3 + 2] : 0.f;
    const float gm = live ? g_mask[pp] : 0.f;
    const float gdot_xs = gx_ * xs + gy_ * ys + gz_ * zs;
    if (aux != nullptr && live) { aux[pp * 2] = den; aux[pp * 2 + 1] = clampg * gdot_xs; }   // for sample_warp_vol_scatter_kernel
    // pass 2: per-bone gradients.  d loss / d q_i of every (point, bone) goes through LDS (16 bones at a time) and the
    // R/T gradients  g_R_i = sum_p gq_pi (x) p,  g_T_i = sum_p gq_pi  are reduced with one thread per (bone, component,
    // 64-point chunk) instead of 12 wave-wide butterfly sums per bone.
    __shared__ float sQ[256][49];          // [point][16 bones x 3 components] (+1: bank spread)
    __shared__ float sP[256][3];
    sP[threadIdx.x][0] = px; sP[threadIdx.x][1] = py; sP[threadIdx.x][2] = pz;
    for (int i0 = 0; i0 < K; i0 += 16) {
        for (int ii = 0; ii < 16; ++ii) {
            const int i = i0 + ii;
            float gq[3] = {0.f, 0.f, 0.f};
            if (i < K) {
                const float* r = sR + i * 9;
                const float qx = (r[0] * px + r[1] * py + r[2] * pz) + sT[i * 3 + 0];
                const float qy = (r[3] * px + r[4] * py + r[5] * pz) + sT[i * 3 + 1];
                const float qz = (r[6] * px + r[7] * py + r[8] * pz) + sT[i * 3 + 2];
                // d loss / d w_i
                const float gw = ((gx_ * qx + gy_ * qy + gz_ * qz) - clampg * gdot_xs) / den + gm;
                float dg[3];
                const float w = trilinear_zero_grad(vol + i * V3, g_vol + i * V3, V, (qx - sB[0]) * sB[3] - 1.f,
                                                    (qy - sB[1]) * sB[4] - 1.f, (qz - sB[2]) * sB[5] - 1.f, gw, dg, live);
                // d loss / d q_i = g_xs * w/den + gw * dw/dq   (all zero for dead lanes: their g_xskel / g_mask are 0)
                gq[0] = gx_ * w / den + gw * dg[0] * sB[3];
                gq[1] = gy_ * w / den + gw * dg[1] * sB[4];
                gq[2] = gz_ * w / den + gw * dg[2] * sB[5];
            }
            sQ[threadIdx.x][ii * 3 + 0] = gq[0]; sQ[threadIdx.x][ii * 3 + 1] = gq[1]; sQ[threadIdx.x][ii * 3 + 2] = gq[2];
        }
        __syncthreads();
        {
            const int e = threadIdx.x & 63, chunk = threadIdx.x >> 6;      // e = (bone in group) * 3 + component; 4 chunks of 64 points
            const int bone = i0 + e / 3, comp = e % 3;
            if (e < 48 && bone < K) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (int j = 0; j < 64; ++j) {
                    const int q = chunk * 64 + j;
                    const float v = sQ[q][e];
                    a0 += v * sP[q][0]; a1 += v * sP[q][1]; a2 += v * sP[q][2]; a3 += v;
                }
                atomicAdd(&sAcc[bone * 12 + comp * 3 + 0], a0);
                atomicAdd(&sAcc[bone * 12 + comp * 3 + 1], a1);
                atomicAdd(&sAcc[bone * 12 + comp * 3 + 2], a2);
                atomicAdd(&sAcc[bone * 12 + 9 + comp], a3);
            }
        }
        __syncthreads();
    }
    }   // chunk loop
    __syncthreads();
    for (int i = threadIdx.x; i < K * 12; i += blockDim.x) {
        const int b = i / 12, c = i % 12;
        const float v = sAcc[i];
        if (v != 0.f) {
            if (c < 9) __hip_atomic_fetch_add(g_R + b * 9 + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(g_T + b * 3 + (c - 9), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Volume gradient of the backward warp, one BONE per workgroup row: g_vol[i] += sum_p gw_pi * tap weights of q_pi.
// The per-point scatter inside human_sample_warp_bwd_kernel was 60 % of its time (26 bones x 8 taps of scattered 4-byte
// global atomics per point, 485 of 816 us); a bone's 32^3 gradient volume is 128 KB and fits LDS, so the workgroups of
// row i accumulate bone i's volume with LDS atomics over their share of the points and flush the non-zero cells once.
// gw_pi = ((g_p . q_pi) - c_p) / den_p + gm_p with (den_p, c_p) from `aux` (written by human_sample_warp_bwd_kernel).
__global__ __launch_bounds__(1024) void sample_warp_vol_scatter_kernel(
    const float* __restrict__ pts, const float* __restrict__ R, const float* __restrict__ T, int V,
    const float* __restrict__ bbox_min, const float* __restrict__ bbox_scale, long P,
    const float* __restrict__ g_xskel, const float* __restrict__ g_mask, const float* __restrict__ aux,
    float* __restrict__ g_vol) {
    extern __shared__ float s_gvol[];
    const int i = blockIdx.y;
    const int V3 = V * V * V;
    for (int k = threadIdx.x; k < V3; k += blockDim.x) s_gvol[k] = 0.f;
    float r[9], tt[3], bm[3], bs[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = R[i * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { tt[k] = T[i * 3 + k]; bm[k] = bbox_min[k]; bs[k] = bbox_scale[k]; }
    __syncthreads();
    // Consecutive lanes are consecutive samples of a ray and mostly share a voxel cell: the eight tap contributions are
    // summed over runs of equal cells with a segmented wave scan and only the last lane of a run touches LDS (LDS float
    // atomics retire at well under one lane per clock: 305 us for the 54 M tap updates of a step without this).
    const int lane = threadIdx.x & 63;
    const long span = (long)gridDim.x * blockDim.x;
    const long p_end = ((P + span - 1) / span) * span;                 // whole waves take part in the shuffles
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < p_end; p += span) {
        const bool live = p < P;
        const long pp = live ? p : P - 1;
        const float px = pts[pp * 3], py = pts[pp * 3 + 1], pz = pts[pp * 3 + 2];
        const float qx = (r[0] * px + r[1] * py + r[2] * pz) + tt[0];
        const float qy = (r[3] * px + r[4] * py + r[5] * pz) + tt[1];
        const float qz = (r[6] * px + r[7] * py + r[8] * pz) + tt[2];
        const float gw = ((g_xskel[pp * 3] * qx + g_xskel[pp * 3 + 1] * qy + g_xskel[pp * 3 + 2] * qz) - aux[pp * 2 + 1]) / aux[pp * 2] + g_mask[pp];
        const float gx = (qx - bm[0]) * bs[0] - 1.f, gy = (qy - bm[1]) * bs[1] - 1.f, gz = (qz - bm[2]) * bs[2] - 1.f;
        const float ix = ((gx + 1.f) / 2.f) * (float)(V - 1), iy = ((gy + 1.f) / 2.f) * (float)(V - 1), iz = ((gz + 1.f) / 2.f) * (float)(V - 1);
        const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
        const bool act = live && gw != 0.f && (fx >= -1.f && fx <= (float)V && fy >= -1.f && fy <= (float)V && fz >= -1.f && fz <= (float)V);
        const int x0 = act ? (int)fx : 0, y0 = act ? (int)fy : 0, z0 = act ? (int)fz : 0;
        const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz;
        const float wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
        float contrib[8];
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
            const int dx = t8 & 1, dy = (t8 >> 1) & 1, dz = t8 >> 2;
            contrib[t8] = act ? gw * ((dx ? wx1 : wx0) * (dy ? wy1 : wy0) * (dz ? wz1 : wz0)) : 0.f;
        }
        const int key = act ? ((z0 + 1) * (V + 2) + (y0 + 1)) * (V + 2) + (x0 + 1) : -1 - lane;
        const int key_prev = __shfl_up(key, 1, 64);
        int flag = (lane == 0 || key != key_prev) ? 1 : 0;
        const int key_next = __shfl_down(key, 1, 64);
        const bool tail = (lane == 63) || (key_next != key);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int f_up = __shfl_up(flag, off, 64);
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                const float v_up = __shfl_up(contrib[t8], off, 64);
                if (lane >= off && !flag) contrib[t8] += v_up;
            }
            if (lane >= off) flag |= f_up;
        }
        if (act && tail) {
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                const int x = x0 + (t8 & 1), y = y0 + ((t8 >> 1) & 1), z = z0 + (t8 >> 2);
                if (x >= 0 && x < V && y >= 0 && y < V && z >= 0 && z < V && contrib[t8] != 0.f)
                    atomicAdd(&s_gvol[(z * V + y) * V + x], contrib[t8]);
            }
        }
    }
    __syncthreads();
    float* const out = g_vol + (size_t)i * V3;
    for (int k = threadIdx.x; k < V3; k += blockDim.x) {
        const float v = s_gvol[k];
        if (v != 0.f) __hip_atomic_fetch_add(out + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// forward-LBS backward: x_def = sum_i w_i(c) (R_i c + T_i) / max(sum w, 1e-4)
__global__ __launch_bounds__(256) void lbs_forward_bwd_kernel(
    const float* __restrict__ cnl, const float* __restrict__ R, const float* __restrict__ T,
    const float* __restrict__ vol_cl, int V, int CL, const float* __restrict__ bbox_min,
    const float* __restrict__ bbox_scale, long P, int K, const float* __restrict__ g_xdef,
    float* __restrict__ g_cnl, float* __restrict__ g_vol_cl, float* __restrict__ g_R, float* __restrict__ g_T,
    const int* __restrict__ p_dev) {
    __shared__ float sR[KMAX * 9], sT[KMAX * 3], sB[6], sAcc[KMAX * 12];
    if (p_dev) P = min(P, (long)*p_dev);
    for (int i = threadIdx.x; i < K * 9; i += blockDim.x) sR[i] = R[i];
    for (int i = threadIdx.x; i < K * 3; i += blockDim.x) sT[i] = T[i];
    for (int i = threadIdx.x; i < K * 12; i += blockDim.x) sAcc[i] = 0.f;
    if (threadIdx.x < 3) { sB[threadIdx.x] = bbox_min[threadIdx.x]; sB[3 + threadIdx.x] = bbox_scale[threadIdx.x]; }
    __syncthreads();
    // persistent workgroups: a block walks 256-point chunks and keeps the R/T gradient sums in LDS across them, so the
    // 12 K global atomics at the end (all blocks hit the same 12 K addresses) are issued once per block, not per chunk
    const long nchunks = (P + 255) / 256;
    for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const long p = chunk * 256 + threadIdx.x;
    const bool live = p < P;
    const long pp = live ? p : P - 1;
    const int lane = threadIdx.x & 63;
    const float px = cnl[pp * 3], py = cnl[pp * 3 + 1], pz = cnl[pp * 3 + 2];
    const float gx = (px - sB[0]) * sB[3] - 1.f, gy = (py - sB[1]) * sB[4] - 1.f, gz = (pz - sB[2]) * sB[5] - 1.f;
    const float half = 0.5f * (float)(V - 1);
    const float ix = ((gx + 1.f) / 2.f) * (float)(V - 1), iy = ((gy + 1.f) / 2.f) * (float)(V - 1), iz = ((gz + 1.f) / 2.f) * (float)(V - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const bool inside = (fx >= -1.f && fx <= (float)V && fy >= -1.f && fy <= (float)V && fz >= -1.f && fz <= (float)V);
    const int x0 = inside ? (int)fx : 0, y0 = inside ? (int)fy : 0, z0 = inside ? (int)fz : 0;
    const float wx1 = ix - fx, wy1 = iy - fy, wz1 = iz - fz;
    const float wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy, wz0 = (fz + 1.f) - iz;
    // pass 1: weights, wsum, x_def
    float w[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) w[i] = 0.f;
    if (inside) {
        for (int t = 0; t < 8; ++t) {
            const int dx = t & 1, dy = (t >> 1) & 1, dz = t >> 2;
            const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
            if (x >= 0 && x < V && y >= 0 && y < V && z >= 0 && z < V) {
                const float tw = (dx ? wx1 : wx0) * (dy ? wy1 : wy0) * (dz ? wz1 : wz0);
                const float4* vp = reinterpret_cast<const float4*>(vol_cl + (((size_t)z * V + y) * V + x) * CL);
#pragma unroll
                for (int q = 0; q < KMAX / 4; ++q) {
                    if (q * 4 < K) { const float4 v = vp[q]; w[q * 4] += v.x * tw; w[q * 4 + 1] += v.y * tw; w[q * 4 + 2] += v.z * tw; w[q * 4 + 3] += v.w * tw; }
                }
            }
        }
    }
    float wsum = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        if (i < K) {
            const float* r = sR + i * 9;
            wsum += w[i];
            ax += w[i] * ((r[0] * px + r[1] * py + r[2] * pz) + sT[i * 3 + 0]);
            ay += w[i] * ((r[3] * px + r[4] * py + r[5] * pz) + sT[i * 3 + 1]);
            az += w[i] * ((r[6] * px + r[7] * py + r[8] * pz) + sT[i * 3 + 2]);
        }
    }
    const float den = fmaxf(wsum, 1e-4f);
    const float xd = ax / den, yd = ay / den, zd = az / den;
    const float clampg = (wsum >= 1e-4f) ? 1.f : 0.f;
    const float g0 = live ? g_xdef[pp * 3] : 0.f, g1 = live ? g_xdef[pp * 3 + 1] : 0.f, g2 = live ? g_xdef[pp * 3 + 2] : 0.f;
    const float gdot = g0 * xd + g1 * yd + g2 * zd;
    // pass 2: g_w_i and g wrt c through q_i.  The R/T gradients  g_R_i = sum_p (g_p w_pi/den_p) (x) p,  g_T_i = sum_p g_p
    // w_pi/den_p  are reduced AFTERWARDS with lanes = bones (the per-point factor s_pi = w_pi/den_p goes through LDS):
    // 26 x 12 wave-wide butterfly sums per wave were half of this kernel's instructions.
    __shared__ float sGw[256][33];
    __shared__ float sPG[256][6];
    float gw[KMAX];
    float gc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        gw[i] = 0.f;
        float sfac = 0.f;
        if (i < K) {
            const float* r = sR + i * 9;
            const float qx = (r[0] * px + r[1] * py + r[2] * pz) + sT[i * 3 + 0];
            const float qy = (r[3] * px + r[4] * py + r[5] * pz) + sT[i * 3 + 1];
            const float qz = (r[6] * px + r[7] * py + r[8] * pz) + sT[i * 3 + 2];
            gw[i] = ((g0 * qx + g1 * qy + g2 * qz) - clampg * gdot) / den;
            sfac = w[i] / den;
            const float gq[3] = {g0 * sfac, g1 * sfac, g2 * sfac};
            gc[0] += r[0] * gq[0] + r[3] * gq[1] + r[6] * gq[2];
            gc[1] += r[1] * gq[0] + r[4] * gq[1] + r[7] * gq[2];
            gc[2] += r[2] * gq[0] + r[5] * gq[1] + r[8] * gq[2];
        }
        if (i < 32) sGw[threadIdx.x][i] = sfac;
    }
    sPG[threadIdx.x][0] = px; sPG[threadIdx.x][1] = py; sPG[threadIdx.x][2] = pz;
    sPG[threadIdx.x][3] = g0; sPG[threadIdx.x][4] = g1; sPG[threadIdx.x][5] = g2;       // (0 for dead lanes)
    __syncthreads();
    {
        const int bone = threadIdx.x & 31, chunk = threadIdx.x >> 5;                   // 8 chunks of 32 points
        if (bone < K) {
            float a[12];
#pragma unroll
            for (int c = 0; c < 12; ++c) a[c] = 0.f;
            for (int j = 0; j < 32; ++j) {
                const int q = chunk * 32 + j;
                const float sf = sGw[q][bone];
                const float qx_ = sPG[q][0], qy_ = sPG[q][1], qz_ = sPG[q][2];
                const float e0 = sPG[q][3] * sf, e1 = sPG[q][4] * sf, e2 = sPG[q][5] * sf;
                a[0] += e0 * qx_; a[1] += e0 * qy_; a[2] += e0 * qz_;
                a[3] += e1 * qx_; a[4] += e1 * qy_; a[5] += e1 * qz_;
                a[6] += e2 * qx_; a[7] += e2 * qy_; a[8] += e2 * qz_;
                a[9] += e0; a[10] += e1; a[11] += e2;
            }
#pragma unroll
            for (int c = 0; c < 12; ++c) atomicAdd(&sAcc[bone * 12 + c], a[c]);
        }
    }
    __syncthreads();
    // pass 3: through the taps: d w / d c per point; the volume gradient is scattered AFTERWARDS with lanes = channels:
    // an fp32 atomic costs one request per 128-byte line it touches, not per lane (scripts/probe/atomic_probe.hip:
    // 20 G lane-atomics/s with 64 scattered lines per instruction, 320 G/s with 64 consecutive floats), and the
    // channel-last volume keeps the K <= 32 bone channels of a voxel in one line.  A point-per-lane scatter issued
    // 8 taps x 26 channels x 64 different lines per wave instruction (4.9 ms for 262 144 points).
    __shared__ int sBase[256][8];
    __shared__ float sTw[256][8];
    __shared__ int sKey[256];
    {
        const bool act = inside && live;
        sKey[threadIdx.x] = act ? (((z0 + 1) * (V + 2) + (y0 + 1)) * (V + 2) + (x0 + 1)) : -1;       // voxel cell of the point
#pragma unroll
        for (int i = 0; i < KMAX; ++i) if (i < 32) sGw[threadIdx.x][i] = (act && i < K) ? gw[i] : 0.f;
        float dgx = 0.f, dgy = 0.f, dgz = 0.f;
        for (int t = 0; t < 8; ++t) {
            const int dx = t & 1, dy = (t >> 1) & 1, dz = t >> 2;
            const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
            int tap_base = -1;
            float tap_w = 0.f;
            if (act && x >= 0 && x < V && y >= 0 && y < V && z >= 0 && z < V) {
                const float wx = dx ? wx1 : wx0, wy = dy ? wy1 : wy0, wz = dz ? wz1 : wz0;
                const size_t base = (((size_t)z * V + y) * V + x) * CL;
                float dot = 0.f;    // sum_i gw_i * v_i(tap)
                for (int i = 0; i < K; ++i) dot += gw[i] * vol_cl[base + i];
                tap_base = (int)base;
                tap_w = wx * wy * wz;
                dgx += dot * (dx ? 1.f : -1.f) * wy * wz;
                dgy += dot * wx * (dy ? 1.f : -1.f) * wz;
                dgz += dot * wx * wy * (dz ? 1.f : -1.f);
            }
            sBase[threadIdx.x][t] = tap_base;
            sTw[threadIdx.x][t] = tap_w;
        }
        if (act) { gc[0] += dgx * half * sB[3]; gc[1] += dgy * half * sB[4]; gc[2] += dgz * half * sB[5]; }
    }
    __syncthreads();
    if (g_vol_cl != nullptr) {
        // Each half wave walks 32 CONSECUTIVE points (samples along a ray): neighbours usually fall into the same voxel
        // cell, so their contributions are summed per tap in registers and one line-atomic per tap is issued when the
        // cell changes (about a third of the per-point count on ray samples: 128 samples cross ~40 cells).
        const int c = lane & 31, hw = lane >> 5, w0 = (threadIdx.x >> 6) * 64 + 32 * hw;
        float run[8];
        int rbase[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { run[t] = 0.f; rbase[t] = -1; }
        int rkey = -1;
        auto flush = [&]() {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (rbase[t] >= 0 && c < K) __hip_atomic_fetch_add(g_vol_cl + rbase[t] + c, run[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                run[t] = 0.f;
            }
        };
        for (int j = 0; j < 32; ++j) {
            const int q = w0 + j;
            const int key = sKey[q];
            if (key != rkey) {                                  // uniform over the half wave
                if (rkey >= 0) flush();
                rkey = key;
#pragma unroll
                for (int t = 0; t < 8; ++t) rbase[t] = sBase[q][t];
            }
            if (key >= 0) {
                const float gv = sGw[q][c];
#pragma unroll
                for (int t = 0; t < 8; ++t) run[t] += gv * sTw[q][t];
            }
        }
        if (rkey >= 0) flush();
    }
    if (live && g_cnl) { g_cnl[pp * 3] = gc[0]; g_cnl[pp * 3 + 1] = gc[1]; g_cnl[pp * 3 + 2] = gc[2]; }
    __syncthreads();
    }   // chunk loop
    for (int i = threadIdx.x; i < K * 12; i += blockDim.x) {
        const int b = i / 12, c = i % 12;
        const float v = sAcc[i];
        if (v != 0.f) {
            if (c < 9) __hip_atomic_fetch_add(g_R + b * 9 + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(g_T + b * 3 + (c - 9), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// g_x[p, ax] (+)= [identity] + sum_j w_j 2^j (cos(2^j x) dS_j - sin(2^j x) dC_j), features gathered from up to
// two gradient matrices (first-layer input gradient and skip-concat gradient).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ x, const float* __restrict__ band_w, int F,
                                                        int identity, const float* __restrict__ dA, int lda, int colA,
                                                        const float* __restrict__ dB, int ldb, int colB, long P,
                                                        float* __restrict__ g_x, int accumulate, const int* __restrict__ p_dev,
                                                        const float* __restrict__ res) {
    const long P_full = P;
    if (p_dev) P = min(P, (long)*p_dev);
    const long total = P * 3;
    if (res != nullptr)
        for (long it = total + (long)blockIdx.x * blockDim.x + threadIdx.x; it < P_full * 3; it += (long)gridDim.x * blockDim.x) g_x[it] = res[it];
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long p = it / 3;
        const int ax = (int)(it % 3);
        const float xv = x[it];
        auto feat = [&](int c) {
            float v = dA[p * lda + colA + c];
            if (dB) v += dB[p * ldb + colB + c];
            return v;
        };
        float g = 0.f;
        int base = 0;
        if (identity) { g += feat(ax); base = 3; }
        for (int j = 0; j < F; ++j) {
            const float fr = (float)(1 << j);
            const float a = xv * fr;
            const float wj = band_w ? band_w[j] : 1.f;
            g += wj * fr * (cosf(a) * feat(base + j * 6 + ax) - sinf(a) * feat(base + j * 6 + 3 + ax));
        }
        g_x[it] = res != nullptr ? res[it] + g : (accumulate ? g_x[it] + g : g);
    }
}

// Tiled form: a workgroup stages the feature gradients of 64 rows (dA + dB, 3 + 6F columns each) in LDS with coalesced
// reads -- consecutive lanes read consecutive columns of a row -- and thread (row, axis) then walks ITS features in LDS.
// The per-element kernel above has every lane of a wave reading another row (21 rows x 12 B per load instruction).
__global__ __launch_bounds__(256) void embed_bwd_tiled_kernel(const float* __restrict__ x, const float* __restrict__ band_w, int F,
                                                              int identity, const float* __restrict__ dA, int lda, int colA,
                                                              const float* __restrict__ dB, int ldb, int colB, long P,
                                                              float* __restrict__ g_x, int accumulate, const int* __restrict__ p_dev,
                                                              const float* __restrict__ res) {
    __shared__ float sG[64][101];
    const long P_full = P;
    if (p_dev) P = min(P, (long)*p_dev);
    const int t = threadIdx.x, nf = (identity ? 3 : 0) + 6 * F;
    // `res` (the residual path's cotangent, xyz = x + offset): g_x = res + g, and rows past the device-side row count -- which this
    // kernel otherwise leaves alone -- receive res alone, so that g_x may be uninitialised storage (no clone of res beforehand)
    if (res != nullptr)
        for (long it = P * 3 + (long)blockIdx.x * 256 + t; it < P_full * 3; it += (long)gridDim.x * 256) g_x[it] = res[it];
    for (long row0 = (long)blockIdx.x * 64; row0 < P; row0 += (long)gridDim.x * 64) {
        const int rows = (int)min(64L, P - row0);
        for (int i = t; i < rows * nf; i += 256) {
            const int r = i / nf, c = i - r * nf;
            float v = dA[(row0 + r) * lda + colA + c];
            if (dB) v += dB[(row0 + r) * ldb + colB + c];
            sG[r][c] = v;
        }
        __syncthreads();
        if (t < rows * 3) {
            const int r = t / 3, ax = t - r * 3;
            const long it = (row0 + r) * 3 + ax;
            const float xv = x[it];
            float g = 0.f;
            int base = 0;
            if (identity) { g += sG[r][ax]; base = 3; }
            for (int j = 0; j < F; ++j) {
                const float fr = (float)(1 << j);
                const float a = xv * fr;
                const float wj = band_w ? band_w[j] : 1.f;
                float sn, cs;
                sincosf(a, &sn, &cs);
                g += wj * fr * (cs * sG[r][base + j * 6 + ax] - sn * sG[r][base + j * 6 + 3 + ax]);
            }
            g_x[it] = res != nullptr ? res[it] + g : (accumulate ? g_x[it] + g : g);
        }
        __syncthreads();
    }
}

// out[p, c] = src[p*lds + col0 + c] * (mask_src[p*ldm + mcol0 + c] > 0)   (c < width); used to pull the
// h-part out of the canonical skip-concat gradient, and (mask NULL) for plain strided slices.
__global__ __launch_bounds__(256) void slice_mask_kernel(const float* __restrict__ src, int lds, int col0,
                                                         const float* __restrict__ msk, int ldm, int mcol0, long P,
                                                         int width, float* __restrict__ out, int ldo, const int* __restrict__ p_dev) {
    if (p_dev) P = min(P, (long)*p_dev);
    const long total = P * width;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long p = it / width;
        const int c = (int)(it % width);
        float v = src[p * lds + col0 + c];
        if (msk && !(msk[p * ldm + mcol0 + c] > 0.f)) v = 0.f;
        out[p * ldo + c] = v;
    }
}

// d(pre-activation) of the canonical head: cols 0..2 sigmoid' = s(1-s), col 3 relu'; the whole [P, ldo] row is written
// (16 bytes per thread, columns 4.. zero): the caller passes uninitialised storage, no fill launch
__global__ __launch_bounds__(256) void rgbsigma_grad_kernel(const float* __restrict__ g, const float* __restrict__ y, long P,
                                                            float* __restrict__ out, int ldo) {
    const int q = ldo >> 2;
    const long total = P * q;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long p = it / q;
        const int c4 = (int)(it % q);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 == 0) {
            const float4 yv = *reinterpret_cast<const float4*>(y + p * 4), gv = *reinterpret_cast<const float4*>(g + p * 4);
            o = make_float4(gv.x * yv.x * (1.f - yv.x), gv.y * yv.y * (1.f - yv.y), gv.z * yv.z * (1.f - yv.z), yv.w > 0.f ? gv.w : 0.f);
        }
        *reinterpret_cast<float4*>(out + p * ldo + c4 * 4) = o;
    }
}

// out[p, :] = [src[p*lds + col0 .. + width) | 0 ...]  for the whole [P, ldo] row (ldo % 4 == 0, width <= 4): a narrow gradient
// ([P,3] offsets) widened to the zero-padded operand row of the layer backward, without a fill launch
__global__ __launch_bounds__(256) void slice_pad_kernel(const float* __restrict__ src, int lds, int col0, long P, int width,
                                                        float* __restrict__ out, int ldo, const int* __restrict__ p_dev) {
    if (p_dev) P = min(P, (long)*p_dev);
    const int q = ldo >> 2;
    const long total = P * q;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long p = it / q;
        const int c4 = (int)(it % q);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c4 == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < width) v[c] = src[p * lds + col0 + c];
        }
        *reinterpret_cast<float4*>(out + p * ldo + c4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace

extern "C" int hos_human_sample_warp_bwd(const float* pts, const float* R, const float* T, const float* vol, int V,
                                         const float* bbox_min, const float* bbox_scale, int64_t P, int K,
                                         const float* g_x_skel, const float* g_mask, float* g_vol, float* g_R,
                                         float* g_T, float* scratch, const float* fwd_x_skel, const float* fwd_mask,
                                         hos_stream_t stream) {
    if (!pts || !R || !T || !vol || !bbox_min || !bbox_scale || !g_x_skel || !g_mask || !g_vol || !g_R || !g_T || P <= 0)
        return HOS_E_ARG;
    if (K <= 0 || K > KMAX || V < 2) return HOS_E_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long sw_chunks = (P + 255) / 256;
    const dim3 grid((unsigned)(sw_chunks < persist_grid() ? sw_chunks : persist_grid()));
    // with a scratch [P,2] and a volume that fits LDS the scatter runs as its own bone-per-workgroup pass
    const size_t vol_bytes = (size_t)V * V * V * sizeof(float);
    const bool split = scratch != nullptr && vol_bytes <= 128 * 1024 && P >= 8192;
    hipLaunchKernelGGL(human_sample_warp_bwd_kernel, grid, dim3(256), 0, s, pts, R, T, vol, V, bbox_min, bbox_scale, (long)P, K,
                       g_x_skel, g_mask, split ? (float*)nullptr : g_vol, g_R, g_T, split ? scratch : (float*)nullptr,
                       (fwd_x_skel && fwd_mask) ? fwd_x_skel : (const float*)nullptr, (fwd_x_skel && fwd_mask) ? fwd_mask : (const float*)nullptr);
    if (split) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_warp_vol_scatter_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        const int nb = 256 / K > 0 ? 256 / K : 1;         // one round: at most one workgroup per CU (128 KB of LDS each)
        hipLaunchKernelGGL(sample_warp_vol_scatter_kernel, dim3(nb, K), dim3(1024), vol_bytes, s, pts, R, T, V, bbox_min,
                           bbox_scale, (long)P, g_x_skel, g_mask, scratch, g_vol);
    }
    return hos_launch_status();
}

extern "C" int hos_lbs_forward_bwd(const float* cnl_pts, const float* R_fwd, const float* T_fwd, const float* vol_cl,
                                   int V, int CL, const float* bbox_min, const float* bbox_scale, int64_t P, int K,
                                   const float* g_x_deform, float* g_cnl, float* g_vol_cl, float* g_R, float* g_T,
                                   const int32_t* rows_dev, hos_stream_t stream) {
    if (!cnl_pts || !R_fwd || !T_fwd || !vol_cl || !bbox_min || !bbox_scale || !g_x_deform || !g_R || !g_T || P <= 0)
        return HOS_E_ARG;
    if (K <= 0 || K > KMAX || CL < K || (CL & 3) || V < 2) return HOS_E_SHAPE;
    const long lb_chunks = (P + 255) / 256;
    hipLaunchKernelGGL(lbs_forward_bwd_kernel, dim3((unsigned)(lb_chunks < persist_grid() ? lb_chunks : persist_grid())), dim3(256), 0,
                       static_cast<hipStream_t>(stream), cnl_pts, R_fwd, T_fwd, vol_cl, V, CL, bbox_min, bbox_scale,
                       (long)P, K, g_x_deform, g_cnl, g_vol_cl, g_R, g_T, rows_dev);
    return hos_launch_status();
}

static int embed_bwd_launch(const float* x, const float* band_w, int num_freqs, int identity, const float* dA, int lda,
                            int colA, const float* dB, int ldb, int colB, int64_t P, float* g_x, int accumulate,
                            const int32_t* rows_dev, const float* res, hos_stream_t stream);

extern "C" int hos_embed_bwd(const float* x, const float* band_w, int num_freqs, int identity, const float* dA, int lda,
                             int colA, const float* dB, int ldb, int colB, int64_t P, float* g_x, int accumulate,
                             const int32_t* rows_dev, hos_stream_t stream) {
    return embed_bwd_launch(x, band_w, num_freqs, identity, dA, lda, colA, dB, ldb, colB, P, g_x, accumulate, rows_dev, nullptr, stream);
}

// The same gradient added to a residual cotangent: g_x [P, 3] = res [P, 3] + d(features)/dx (rows past *rows_dev: res alone), so
// g_x may be uninitialised storage -- the backward of xyz = x + MLP(embed(x)) (mlp_offset.py:66-70) without a clone of `res`.
extern "C" int hos_embed_bwd_res(const float* x, const float* band_w, int num_freqs, int identity, const float* dA, int lda,
                                 int colA, const float* dB, int ldb, int colB, int64_t P, const float* res, float* g_x,
                                 const int32_t* rows_dev, hos_stream_t stream) {
    if (!res) return HOS_E_ARG;
    return embed_bwd_launch(x, band_w, num_freqs, identity, dA, lda, colA, dB, ldb, colB, P, g_x, 0, rows_dev, res, stream);
}

static int embed_bwd_launch(const float* x, const float* band_w, int num_freqs, int identity, const float* dA, int lda,
                            int colA, const float* dB, int ldb, int colB, int64_t P, float* g_x, int accumulate,
                            const int32_t* rows_dev, const float* res, hos_stream_t stream) {
    if (!x || !dA || !g_x || P <= 0) return HOS_E_ARG;
    if (num_freqs < 1 || num_freqs > 16) return HOS_E_SHAPE;
    static const bool tiled = !(getenv("HOS_EMBED_BWD_TILED") && atoi(getenv("HOS_EMBED_BWD_TILED")) == 0);
    if (tiled) {
        const long b = (P + 63) / 64;
        hipLaunchKernelGGL(embed_bwd_tiled_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                           band_w, num_freqs, identity, dA, lda, colA, dB, ldb, colB, (long)P, g_x, accumulate, rows_dev, res);
        return hos_launch_status();
    }
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(P * 3)), dim3(256), 0, static_cast<hipStream_t>(stream), x, band_w,
                       num_freqs, identity, dA, lda, colA, dB, ldb, colB, (long)P, g_x, accumulate, rows_dev, res);
    return hos_launch_status();
}

extern "C" int hos_slice_mask(const float* src, int lds, int col0, const float* mask_src, int ldm, int mcol0, int64_t P,
                              int width, float* out, int ldo, const int32_t* rows_dev, hos_stream_t stream) {
    if (!src || !out || P <= 0 || width <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(slice_mask_kernel, dim3(grid_for(P * width)), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds,
                       col0, mask_src, ldm, mcol0, (long)P, width, out, ldo, rows_dev);
    return hos_launch_status();
}

extern "C" int hos_rgbsigma_grad(const float* g_rgbsigma, const float* rgbsigma, int64_t P, float* dz, int ldz,
                                 hos_stream_t stream) {
    if (!g_rgbsigma || !rgbsigma || !dz || P <= 0 || ldz < 4) return HOS_E_ARG;
    if ((ldz & 3) || (((uintptr_t)g_rgbsigma | (uintptr_t)rgbsigma | (uintptr_t)dz) & 15u)) return HOS_E_ALIGN;
    hipLaunchKernelGGL(rgbsigma_grad_kernel, dim3(grid_for(P * (ldz >> 2))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g_rgbsigma, rgbsigma, (long)P, dz, ldz);
    return hos_launch_status();
}

extern "C" int hos_slice_pad(const float* src, int lds, int col0, int64_t P, int width, float* out, int ldo,
                             const int32_t* rows_dev, hos_stream_t stream) {
    if (!src || !out || P <= 0 || width <= 0) return HOS_E_ARG;
    if (width > 4 || ldo < 4) return HOS_E_SHAPE;
    if ((ldo & 3) || ((uintptr_t)out & 15u)) return HOS_E_ALIGN;
    hipLaunchKernelGGL(slice_pad_kernel, dim3(grid_for(P * (ldo >> 2))), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds,
                       col0, (long)P, width, out, ldo, rows_dev);
    return hos_launch_status();
}
