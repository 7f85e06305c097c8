// Training losses of the human-object stages on the device: photometric MSE on the
// rendered rays, the optical-flow term of the previous-frame points and the cycle-consistency term, values and
// gradients, without a host round trip and with fixed shapes (so the step can live in a hipGraph).
//
//   hos_train_losses_fwd   M:1690-1716 `get_loss` (stage 3) / M2:918-944 (stage 2), `flow_func` M:1680-1688,
//                          `img2mae` M:61-71, `img2mse` M:36, `_unpack_imgs` M:41-50 -- the LPIPS term (third-party VGG)
//                          is not part of this library.
//   hos_train_losses_bwd   the gradients torch.autograd derives from those lines.
//   (M = 3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py, M2 = 2nd_State_Conditional_Human-Object/.../model.py)
//
// mse   = (sum_rays |rgb - target|^2 + mse_const) / mse_count          (the patch pixels outside the ray mask hold the
//                                                                         background colour: a constant, M2:41-50)
// flow  = sum_{b,s,c} |uv_c - (x,y)_c - f_c| * w[b,s] * M_b / (S * sum_b M_b + 1e-8) / 2,  M_b = ray_grid[b,4] (* fg_b in
//         stage 3, where the reference first selects the foreground rows, M:1704), uv = pinhole projection of the
//         forward-warped previous-frame point (M:1680-1686)
// cycle = mean_rows( |observe - deform|^2 / 2 )  over the n_cyc rows of the cycle set (n_cyc may live in device memory:
//         the set is data dependent, N:505-536; 0 rows -> 0, like the reference's single-point fallback)
//
// Deterministic: block partials in a workspace, summed in a fixed order by the last block to finish.
#include "hos_common.h"

namespace {

constexpr int LT = 256;          // threads per block
constexpr int LMAXB = 1024;      // max blocks (partials rows)

struct LossArgs {
    const float* rgb; const float* target; long long B;
    float mse_const, mse_count;
    const float* pts_prev; const float* weights; const float* ray_grid; const int32_t* fg; const float* cam; const float* Kin;
    int S;
    const float* observe; const float* deform; long long n_cyc; const int32_t* n_cyc_dev;
    float w_mse, w_flow, w_cycle;
    float* partials; unsigned int* ticket; float* out;
    // backward
    const float* g_total; float* g_rgb; float* g_pts_prev; float* g_weights; float* g_deform;
};

__device__ __forceinline__ float block_sum(float v, float* sh, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < LT / 64; ++w) r += sh[w];
    return r;
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

struct Proj {
    float u, v, iz;        // projected pixel, 1/depth
    float cx, cy, cz;      // K @ cam point
};

__device__ __forceinline__ Proj project(const float* cam, const float* Kin, float px, float py, float pz) {
    // M:1682-1684: (cam @ [p,1])[:3] then K @ . then divide by the last component
    const float x = cam[0] * px + cam[1] * py + cam[2] * pz + cam[3];
    const float y = cam[4] * px + cam[5] * py + cam[6] * pz + cam[7];
    const float z = cam[8] * px + cam[9] * py + cam[10] * pz + cam[11];
    Proj r;
    r.cx = Kin[0] * x + Kin[1] * y + Kin[2] * z;
    r.cy = Kin[3] * x + Kin[4] * y + Kin[5] * z;
    r.cz = Kin[6] * x + Kin[7] * y + Kin[8] * z;
    r.iz = 1.f / r.cz;
    r.u = r.cx / r.cz;
    r.v = r.cy / r.cz;
    return r;
}

__global__ __launch_bounds__(LT) void losses_fwd_kernel(LossArgs a) {
    __shared__ float sh[LT / 64];
    __shared__ bool last;
    const int tid = threadIdx.x;
    const long long gid = (long long)blockIdx.x * LT + tid, gstride = (long long)gridDim.x * LT;
    float s_mse = 0.f, s_flow = 0.f, s_m = 0.f, s_cyc = 0.f;
    for (long long i = gid; i < a.B * 3; i += gstride) {
        const float d = a.rgb[i] - a.target[i];
        s_mse += d * d;
    }
    if (a.pts_prev) {
        const long long P = a.B * a.S;
        for (long long i = gid; i < P; i += gstride) {
            const long long b = i / a.S;
            float M = a.ray_grid[b * 5 + 4];
            if (a.fg && a.fg[b] == 0) M = 0.f;
            if (M != 0.f) {
                const Proj p = project(a.cam, a.Kin, a.pts_prev[3 * i], a.pts_prev[3 * i + 1], a.pts_prev[3 * i + 2]);
                const float eu = (p.u - a.ray_grid[b * 5]) - a.ray_grid[b * 5 + 2];
                const float ev = (p.v - a.ray_grid[b * 5 + 1]) - a.ray_grid[b * 5 + 3];
                const float w = a.weights[i];
                s_flow += fabsf(eu) * w * M + fabsf(ev) * w * M;
            }
        }
        for (long long b = gid; b < a.B; b += gstride) {
            float M = a.ray_grid[b * 5 + 4];
            if (a.fg && a.fg[b] == 0) M = 0.f;
            s_m += M;
        }
    }
    long long nc = a.n_cyc;
    if (a.n_cyc_dev) nc = min(nc, (long long)*a.n_cyc_dev);
    if (a.observe) {
        for (long long i = gid; i < nc; i += gstride) {
            const float dx = a.observe[3 * i] - a.deform[3 * i], dy = a.observe[3 * i + 1] - a.deform[3 * i + 1],
                        dz = a.observe[3 * i + 2] - a.deform[3 * i + 2];
            s_cyc += (dx * dx + dy * dy + dz * dz) / 2.0f;
        }
    }
    const float b_mse = block_sum(s_mse, sh, tid), b_flow = block_sum(s_flow, sh, tid), b_m = block_sum(s_m, sh, tid),
                b_cyc = block_sum(s_cyc, sh, tid);
    if (tid == 0) {
        float* p = a.partials + 4 * blockIdx.x;
        p[0] = b_mse; p[1] = b_flow; p[2] = b_m; p[3] = b_cyc;
        __threadfence();
        const unsigned int t = atomicAdd(a.ticket, 1u);
        last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // fixed-order final sum by the last block: thread t owns partial rows t, t+LT, ...
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = tid; r < (int)gridDim.x; r += LT) {
        const volatile float* p = a.partials + 4 * r;
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] += p[k];
    }
    float tot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tot[k] = block_sum(f[k], sh, tid);
    if (tid == 0) {
        const float mse = (tot[0] + a.mse_const) / a.mse_count;
        float flow = 0.f, inv_den = 0.f;
        if (a.pts_prev) {
            inv_den = 1.f / ((float)a.S * tot[2] + 1e-8f) / 2.f;        // img2mae: / (sum(M) + 1e-8) / x.shape[-1]
            flow = tot[1] * inv_den;
        }
        const float cyc = (a.observe && nc > 0) ? tot[3] / (float)nc : 0.f;
        a.out[0] = a.w_mse * mse + a.w_flow * flow + a.w_cycle * cyc;
        a.out[1] = mse; a.out[2] = flow; a.out[3] = cyc;
        a.out[4] = inv_den;                                              // for the backward pass
        a.out[5] = nc > 0 ? 1.f / (float)nc : 0.f;
        a.out[6] = tot[2];
        a.out[7] = (float)nc;
        *a.ticket = 0u;                                                  // re-armed for the next launch
    }
}

__global__ __launch_bounds__(LT) void losses_bwd_kernel(LossArgs a) {
    const int tid = threadIdx.x;
    const long long gid = (long long)blockIdx.x * LT + tid, gstride = (long long)gridDim.x * LT;
    const float g = a.g_total ? *a.g_total : 1.f;
    if (a.g_rgb) {
        const float c = g * a.w_mse * 2.f / a.mse_count;
        for (long long i = gid; i < a.B * 3; i += gstride) a.g_rgb[i] = c * (a.rgb[i] - a.target[i]);
    }
    if (a.pts_prev && (a.g_pts_prev || a.g_weights)) {
        const float c = g * a.w_flow * a.out[4];
        const long long P = a.B * a.S;
        const float* cam = a.cam;
        const float* K = a.Kin;
        for (long long i = gid; i < P; i += gstride) {
            const long long b = i / a.S;
            float M = a.ray_grid[b * 5 + 4];
            if (a.fg && a.fg[b] == 0) M = 0.f;
            float gx = 0.f, gy = 0.f, gz = 0.f, gw = 0.f;
            if (M != 0.f) {
                const Proj p = project(cam, K, a.pts_prev[3 * i], a.pts_prev[3 * i + 1], a.pts_prev[3 * i + 2]);
                const float eu = (p.u - a.ray_grid[b * 5]) - a.ray_grid[b * 5 + 2];
                const float ev = (p.v - a.ray_grid[b * 5 + 1]) - a.ray_grid[b * 5 + 3];
                const float w = a.weights[i];
                gw = c * M * (fabsf(eu) + fabsf(ev));
                const float gu = c * M * w * sgn(eu), gv = c * M * w * sgn(ev);
                // u = cx / cz, v = cy / cz
                const float gcx = gu * p.iz, gcy = gv * p.iz, gcz = -(gu * p.u + gv * p.v) * p.iz;
                // (cx,cy,cz) = K @ (x,y,z)
                const float qx = K[0] * gcx + K[3] * gcy + K[6] * gcz;
                const float qy = K[1] * gcx + K[4] * gcy + K[7] * gcz;
                const float qz = K[2] * gcx + K[5] * gcy + K[8] * gcz;
                // (x,y,z) = cam[:3,:3] @ p + cam[:3,3]
                gx = cam[0] * qx + cam[4] * qy + cam[8] * qz;
                gy = cam[1] * qx + cam[5] * qy + cam[9] * qz;
                gz = cam[2] * qx + cam[6] * qy + cam[10] * qz;
            }
            if (a.g_pts_prev) { a.g_pts_prev[3 * i] = gx; a.g_pts_prev[3 * i + 1] = gy; a.g_pts_prev[3 * i + 2] = gz; }
            if (a.g_weights) a.g_weights[i] = gw;
        }
    }
    if (a.observe && a.g_deform) {
        long long nc = a.n_cyc;
        if (a.n_cyc_dev) nc = min(nc, (long long)*a.n_cyc_dev);
        const float c = g * a.w_cycle * a.out[5];
        for (long long i = gid; i < a.n_cyc; i += gstride) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                a.g_deform[3 * i + k] = i < nc ? -c * (a.observe[3 * i + k] - a.deform[3 * i + k]) : 0.f;
        }
    }
}

int grid_for(const LossArgs& a) {
    long long n = a.B * 3;
    if (a.pts_prev) n = max(n, a.B * (long long)a.S);
    if (a.observe) n = max(n, a.n_cyc);
    long long g = (n + LT - 1) / LT;
    return (int)max(1LL, min(g, (long long)LMAXB));
}

}  // namespace

extern "C" long long hos_train_losses_workspace_floats(void) { return 4LL * LMAXB + 4; }

extern "C" int hos_train_losses_fwd(const float* rgb, const float* target, long long n_rays, float mse_const, float mse_count,
                                    const float* pts_prev, const float* weights, const float* ray_grid, const int32_t* fg,
                                    const float* cam_prev, const float* intrinsics_prev, int S,
                                    const float* observe, const float* deform, long long n_cyc, const int32_t* n_cyc_dev,
                                    float w_mse, float w_flow, float w_cycle, float* workspace, float* out8,
                                    hos_stream_t stream) {
    if (!rgb || !target || !workspace || !out8 || n_rays <= 0 || mse_count <= 0.f) return HOS_E_ARG;
    if (pts_prev && (!weights || !ray_grid || !cam_prev || !intrinsics_prev || S <= 0)) return HOS_E_ARG;
    if (observe && (!deform || n_cyc < 0)) return HOS_E_ARG;
    LossArgs a{};
    a.rgb = rgb; a.target = target; a.B = n_rays; a.mse_const = mse_const; a.mse_count = mse_count;
    a.pts_prev = pts_prev; a.weights = weights; a.ray_grid = ray_grid; a.fg = fg; a.cam = cam_prev; a.Kin = intrinsics_prev; a.S = S;
    a.observe = observe; a.deform = deform; a.n_cyc = observe ? n_cyc : 0; a.n_cyc_dev = n_cyc_dev;
    a.w_mse = w_mse; a.w_flow = w_flow; a.w_cycle = w_cycle;
    a.partials = workspace; a.ticket = reinterpret_cast<unsigned int*>(workspace + 4 * LMAXB); a.out = out8;
    hipLaunchKernelGGL(losses_fwd_kernel, dim3(grid_for(a)), dim3(LT), 0, static_cast<hipStream_t>(stream), a);
    return hos_launch_status();
}

extern "C" int hos_train_losses_bwd(const float* g_total, const float* out8, const float* rgb, const float* target, long long n_rays,
                                    float mse_count, const float* pts_prev, const float* weights, const float* ray_grid,
                                    const int32_t* fg, const float* cam_prev, const float* intrinsics_prev, int S,
                                    const float* observe, const float* deform, long long n_cyc, const int32_t* n_cyc_dev,
                                    float w_mse, float w_flow, float w_cycle,
                                    float* g_rgb, float* g_pts_prev, float* g_weights, float* g_deform, hos_stream_t stream) {
    if (!out8 || !rgb || !target || n_rays <= 0 || mse_count <= 0.f) return HOS_E_ARG;
    if (pts_prev && (!weights || !ray_grid || !cam_prev || !intrinsics_prev || S <= 0)) return HOS_E_ARG;
    LossArgs a{};
    a.rgb = rgb; a.target = target; a.B = n_rays; a.mse_count = mse_count;
    a.pts_prev = pts_prev; a.weights = weights; a.ray_grid = ray_grid; a.fg = fg; a.cam = cam_prev; a.Kin = intrinsics_prev; a.S = S;
    a.observe = observe; a.deform = deform; a.n_cyc = observe ? n_cyc : 0; a.n_cyc_dev = n_cyc_dev;
    a.w_mse = w_mse; a.w_flow = w_flow; a.w_cycle = w_cycle;
    a.out = const_cast<float*>(out8); a.g_total = g_total;
    a.g_rgb = g_rgb; a.g_pts_prev = g_pts_prev; a.g_weights = g_weights; a.g_deform = g_deform;
    hipLaunchKernelGGL(losses_bwd_kernel, dim3(grid_for(a)), dim3(LT), 0, static_cast<hipStream_t>(stream), a);
    return hos_launch_status();
}

// ================================================================================================================
// Tail of the stage-1 loss (1st_State-Conditional_Scene/src/model/mipnerf360/model.py:491-514): from the rendered colours and the
// per-ray interlevel / distortion terms (hos_interlevel_fwd, hos_distortion_fwd) to the scalar the step differentiates,
//   total = m_data sqrt(mean((rgb - target)^2) + pad^2) + m_inter sum_l sum_rays(inter_l) / (B Sc) + m_dist mean_rays(dist),
// as one single-workgroup launch (fixed summation order) and one backward launch -- the torch form was ~15 element-wise / reduction
// launches forward and as many backward, 2 % of the 1024-ray step.  out = [total, mse, interlevel, distortion].
// ================================================================================================================
namespace {

__device__ __forceinline__ float block_sum_1024(float v, float* s_red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) for (int i = 0; i < 16; ++i) t += s_red[i];
    __syncthreads();
    return t;                         // valid in thread 0
}

__global__ __launch_bounds__(1024) void stage1_loss_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ target, int B,
                                                               const float* __restrict__ inter0, const float* __restrict__ inter1, int Sc,
                                                               const float* __restrict__ dist, float m_data, float m_inter, float m_dist,
                                                               float pad, float* __restrict__ out) {
    __shared__ float s_red[16];
    float a = 0.f, b0 = 0.f, b1 = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < 3 * B; i += 1024) { const float d = rgb[i] - target[i]; a += d * d; }
    for (int i = threadIdx.x; i < B; i += 1024) {
        if (inter0) b0 += inter0[i];
        if (inter1) b1 += inter1[i];
        c += dist[i];
    }
    const float sa = block_sum_1024(a, s_red), s0 = block_sum_1024(b0, s_red), s1 = block_sum_1024(b1, s_red), sc = block_sum_1024(c, s_red);
    if (threadIdx.x == 0) {
        const float mse = sa / (3.f * B);
        const float inter = s0 / ((float)B * Sc) + s1 / ((float)B * Sc);
        const float dm = sc / (float)B;
        out[0] = sqrtf(mse + pad * pad) * m_data + inter * m_inter + dm * m_dist;
        out[1] = mse; out[2] = inter; out[3] = dm;
    }
}

// g_rgb = gout m_data (rgb - target) / (3 B sqrt(mse + pad^2));  g_inter (both levels) = gout m_inter / (B Sc);  g_dist = gout m_dist / B
__global__ __launch_bounds__(256) void stage1_loss_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ target, int B, int Sc,
                                                              const float* __restrict__ fwd_out, const float* __restrict__ gout, float m_data,
                                                              float m_inter, float m_dist, float pad, float* __restrict__ g_rgb,
                                                              float* __restrict__ g_inter, float* __restrict__ g_dist) {
    const float g = gout[0];
    const float k = g * m_data / (3.f * B * sqrtf(fwd_out[1] + pad * pad));
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * B; i += gridDim.x * 256) g_rgb[i] = k * (rgb[i] - target[i]);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256) {
        g_inter[i] = g * m_inter / ((float)B * Sc);
        g_dist[i] = g * m_dist / (float)B;
    }
}

}  // namespace

extern "C" int hos_stage1_loss_fwd(const float* rgb, const float* target, int B, const float* inter0, const float* inter1, int Sc, const float* dist,
                                   float m_data, float m_inter, float m_dist, float charb_padding, float* out4, hos_stream_t stream) {
    if (!rgb || !target || !dist || !out4 || B <= 0 || Sc <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(stage1_loss_fwd_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), rgb, target, B, inter0, inter1, Sc, dist, m_data,
                       m_inter, m_dist, charb_padding, out4);
    return hos_launch_status();
}

extern "C" int hos_stage1_loss_bwd(const float* rgb, const float* target, int B, int Sc, const float* fwd_out4, const float* gout, float m_data,
                                   float m_inter, float m_dist, float charb_padding, float* g_rgb, float* g_inter, float* g_dist, hos_stream_t stream) {
    if (!rgb || !target || !fwd_out4 || !gout || !g_rgb || !g_inter || !g_dist || B <= 0 || Sc <= 0) return HOS_E_ARG;
    int blocks = (3 * B + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(stage1_loss_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), rgb, target, B, Sc, fwd_out4, gout, m_data,
                       m_inter, m_dist, charb_padding, g_rgb, g_inter, g_dist);
    return hos_launch_status();
}
