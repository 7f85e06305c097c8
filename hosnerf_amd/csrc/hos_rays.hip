// Per-frame ray set-up on the device.  The reference recomputes full-image rays in numpy twice per
// training item plus a six-plane AABB test inside a DataLoader worker and ships them pinned
// (S3/core/data/human_nerf/train.py:513-548, core/utils/camera_util.py:154-265 = `C:`); at the renderer's speed that
// becomes the bottleneck.  Here one thread per pixel:
//   hos_camera_rays  C:154-216  origin -R^T T, direction ((K^-1 [i,j,1]) - T) R - o, unit view direction, and the
//                               mip-NeRF pixel radius  |d(row) - d(row+1)| * 2/sqrt(12)  (last row repeats row H-2)
//   hos_rays_aabb    C:219-265  six-plane AABB (bounds grown by 0.01), a ray is valid iff exactly two of the six
//                               plane hits lie inside the box (+-1e-6); near/far = distances of the two hits / |d|
// HBM-bound on the outputs: 40 B per pixel (o, d, viewdir, radius) + 12 B (near, far, mask).
#include "hos_common.h"

namespace {

struct Cam { float Kinv[9]; float R[9]; float T[3]; };

__device__ __forceinline__ void pixel_dir(const Cam& c, float i, float j, const float (&o)[3], float (&d)[3]) {
    float pc[3], q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) pc[a] = (i * c.Kinv[a * 3 + 0] + j * c.Kinv[a * 3 + 1]) + c.Kinv[a * 3 + 2];   // xy1 . Kinv^T
#pragma unroll
    for (int a = 0; a < 3; ++a) q[a] = pc[a] - c.T[a];
#pragma unroll
    for (int b = 0; b < 3; ++b) d[b] = ((q[0] * c.R[0 * 3 + b] + q[1] * c.R[1 * 3 + b]) + q[2] * c.R[2 * 3 + b]) - o[b];   // (pc - T) R - o
}

__global__ __launch_bounds__(256) void camera_rays_kernel(Cam c, int H, int W, float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                          float* __restrict__ viewdirs, float* __restrict__ radii) {
    const long n = (long)H * W;
    float o[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) o[b] = -((c.R[0 * 3 + b] * c.T[0] + c.R[1 * 3 + b] * c.T[1]) + c.R[2 * 3 + b] * c.T[2]);     // -R^T T
    float e[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) e[b] = (c.Kinv[0 * 3 + 1] * c.R[0 * 3 + b] + c.Kinv[1 * 3 + 1] * c.R[1 * 3 + b]) + c.Kinv[2 * 3 + 1] * c.R[2 * 3 + b];
    const float radius = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) * 2.f / sqrtf(12.f);
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const int row = (int)(p / W), col = (int)(p % W);
        float d[3];
        pixel_dir(c, (float)col, (float)row, o, d);
#pragma unroll
        for (int b = 0; b < 3; ++b) { rays_o[p * 3 + b] = o[b]; rays_d[p * 3 + b] = d[b]; }
        if (viewdirs != nullptr) {
            const float inv = 1.f / sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
            for (int b = 0; b < 3; ++b) viewdirs[p * 3 + b] = d[b] * inv;
        }
        // C:212-214: |d(row) - d(row+1)| * 2/sqrt(12).  For a pin-hole camera that row difference is the same vector
        // for every pixel, (K^-1 e_y) R; evaluating it in closed form avoids the cancellation of two O(1) directions
        // that differ by 1/f (an fp32 subtraction would carry 2e-4 relative error at f = 1500).
        if (radii != nullptr) radii[p] = radius;
    }
}

__global__ __launch_bounds__(256) void rays_aabb_kernel(const float* __restrict__ rays_o, float* __restrict__ rays_d, long n,
                                                        float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                                        float* __restrict__ near, float* __restrict__ far, unsigned char* __restrict__ mask) {
    const float lo[3] = {bx0 - 0.01f, by0 - 0.01f, bz0 - 0.01f}, hi[3] = {bx1 + 0.01f, by1 + 0.01f, bz1 + 0.01f};
    const float eps = 1e-6f;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const float o[3] = {rays_o[p * 3], rays_o[p * 3 + 1], rays_o[p * 3 + 2]};
        float d[3] = {rays_d[p * 3], rays_d[p * 3 + 1], rays_d[p * 3 + 2]};
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (fabsf(d[a]) < 1e-5f) { d[a] = 1e-5f; rays_d[p * 3 + a] = 1e-5f; }        // C:238 mutates ray_d in place
        int hits = 0;
        float t[2] = {0.f, 0.f};
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float s = ((side ? hi[a] : lo[a]) - o[a]) / d[a];
                const float x = s * d[0] + o[0], y = s * d[1] + o[1], z = s * d[2] + o[2];
                const bool in = x >= lo[0] - eps && x <= hi[0] + eps && y >= lo[1] - eps && y <= hi[1] + eps &&
                                z >= lo[2] - eps && z <= hi[2] + eps;
                if (in) {
                    if (hits < 2) {
                        // |p - o| / |d| = |s| (the reference measures both norms explicitly, C:259-262)
                        const float px = x - o[0], py = y - o[1], pz = z - o[2];
                        t[hits] = sqrtf((px * px + py * py) + pz * pz) / sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
                    }
                    ++hits;
                }
            }
        const bool ok = hits == 2;
        mask[p] = ok ? 1 : 0;
        near[p] = ok ? fminf(t[0], t[1]) : 0.f;
        far[p] = ok ? fmaxf(t[0], t[1]) : 0.f;
    }
}

inline int grid_n(long n) { long b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int hos_camera_rays(const float* Kinv9, const float* R9, const float* T3, int H, int W, float* rays_o, float* rays_d,
                               float* viewdirs, float* radii, hos_stream_t stream) {
    if (!Kinv9 || !R9 || !T3 || !rays_o || !rays_d || H < 2 || W < 1) return HOS_E_ARG;
    Cam c;
    for (int i = 0; i < 9; ++i) { c.Kinv[i] = Kinv9[i]; c.R[i] = R9[i]; }      // HOST pointers: 21 camera scalars by value
    for (int i = 0; i < 3; ++i) c.T[i] = T3[i];
    hipLaunchKernelGGL(camera_rays_kernel, dim3(grid_n((long)H * W)), dim3(256), 0, static_cast<hipStream_t>(stream), c, H, W,
                       rays_o, rays_d, viewdirs, radii);
    return hos_launch_status();
}

extern "C" int hos_rays_aabb(const float* rays_o, float* rays_d, int64_t n, const float* bounds6, float* near, float* far,
                             unsigned char* mask, hos_stream_t stream) {
    if (!rays_o || !rays_d || !bounds6 || !near || !far || !mask || n <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(rays_aabb_kernel, dim3(grid_n(n)), dim3(256), 0, static_cast<hipStream_t>(stream), rays_o, rays_d, (long)n,
                       bounds6[0], bounds6[1], bounds6[2], bounds6[3], bounds6[4], bounds6[5], near, far, mask);   // HOST pointer: 6 floats
    return hos_launch_status();
}
