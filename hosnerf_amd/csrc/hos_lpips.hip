// The LPIPS term of the stage-2 / stage-3 training loss (weight 1.0, configs/default.yaml:97-101) on the unpacked P x P patches:
//   third_parties/lpips/lpips.py:82-122 (L), pretrained_networks.py:97-135 (P), __init__.py:10-12 (I), called at
//   src/model/mipnerf360/model.py:1664-1678 (M) -- `lpips_func(2 rgb - 1, 2 target - 1)`, eval mode, VGG-16 frozen.
// The thirteen 3 x 3 convolutions run as im2col + the library's own GEMM entry points (hos_linear_fwd with the bias + ReLU
// epilogue; hos_linear_dgrad for the input gradient -- the filters are frozen, there is no weight gradient); this file holds the
// pieces around them, all channel-last [image, y, x, channel] so that a patch [P, P, 3] needs no permute:
//   prep      (2 x - 1 - shift) / scale                                  L:124-131 ScalingLayer + M:1661
//   im2col / col2im for kernel 3, padding 1 (col2im applies the ReLU mask of the layer below)
//   maxpool 2 x 2 forward / backward (first maximum of a window, like torch; times the ReLU mask)
//   head      per pixel: unit-normalise both feature vectors over channels, weighted squared difference, spatial mean   L:92-100
//   unpack    rays -> patch pixels (background colour where a patch pixel has no ray)   M:41-50 `_unpack_imgs`
// A few hundred KB per step: every kernel is latency-bound; one thread per output element, wave-wide channel sums in the head.
#include "hos_common.h"

namespace {

__global__ __launch_bounds__(256) void lpips_prep_kernel(const float* __restrict__ x, long n3, float* __restrict__ out) {
    const float shift[3] = {-0.030f, -0.088f, -0.188f}, scale[3] = {0.458f, 0.448f, 0.450f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n3; i += (long)gridDim.x * 256) {
        const int c = (int)(i % 3);
        out[i] = ((2.f * x[i] - 1.f) - shift[c]) / scale[c];
    }
}

// col[(n, y, x), tap * C + c] = in[n, y + dy, x + dx, c] (zero outside the image), tap = (dy + 1) * 3 + (dx + 1); columns >= 9 C zero
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ in, int NI, int H, int W, int C, float* __restrict__ col, int ld) {
    const long total = (long)NI * H * W * ld;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int k = (int)(e % ld);
        const long row = e / ld;
        float v = 0.f;
        if (k < 9 * C) {
            const int tap = k / C, c = k % C;
            const int x = (int)(row % W), y = (int)((row / W) % H);
            const long n = row / ((long)W * H);
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = in[((n * H + yy) * W + xx) * C + c];
        }
        col[e] = v;
    }
}

// dx[n, y, x, c] = sum_tap dcol[(n, y - dy, x - dx), tap * C + c]  (* [mask[n, y, x, c] > 0]);  accumulate: dx +=
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcol, int ld, int NI, int H, int W, int C,
                                                        const float* __restrict__ mask, float* __restrict__ dx) {
    const long total = (long)NI * H * W * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long pix = e / C;
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const long n = pix / ((long)W * H);
        float s = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y - (tap / 3 - 1), xx = x - (tap % 3 - 1);
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += dcol[((n * H + yy) * W + xx) * ld + tap * C + c];
        }
        if (mask != nullptr && !(mask[e] > 0.f)) s = 0.f;
        dx[e] = s;
    }
}

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ in, int NI, int H, int W, int C, float* __restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)NI * Ho * Wo * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long p = e / C;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho);
        const long n = p / ((long)Wo * Ho);
        const float* b = in + ((n * H + 2 * yo) * W + 2 * xo) * C + c;
        out[e] = fmaxf(fmaxf(b[0], b[C]), fmaxf(b[(long)W * C], b[(long)W * C + C]));
    }
}

// g_in[n, y, x, c] = g_out[window] if (y, x) is the FIRST maximum of its window in row-major order (torch's max_pool2d), else 0;
// times [in > 0]: `in` is a ReLU output and the gradient continues through that ReLU
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ in, int NI, int H, int W, int C,
                                                           float* __restrict__ g_in) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)NI * H * W * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long pix = e / C;
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const long n = pix / ((long)W * H);
        const float* b = in + ((n * H + (y & ~1)) * W + (x & ~1)) * C + c;
        const float v[4] = {b[0], b[C], b[(long)W * C], b[(long)W * C + C]};
        int first = 0;
        float m = v[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) if (v[q] > m) { m = v[q]; first = q; }
        const int me = (y & 1) * 2 + (x & 1);
        float g = 0.f;
        if (me == first && v[me] > 0.f) g = g_out[((n * Ho + y / 2) * Wo + x / 2) * C + c];
        g_in[e] = g;
    }
}

constexpr float LP_EPS = 1e-10f;

// part[i * LP_CHUNKS + j] += coef * sum over the pixels p = j (mod LP_CHUNKS) of pair i of sum_c w_c (f0_c / R0 - f1_c / R1)^2,
// R = sqrt(sum f^2 + eps) + eps (I:10-12); f0 = features of prediction i, f1 = of target i (rows Np * HW further).  One workgroup
// per (pair, chunk), one wave per pixel, fixed summation order (hos_lpips_finish adds the chunks in index order).
constexpr int LP_CHUNKS = 32;
__global__ __launch_bounds__(256) void lpips_head_fwd_kernel(const float* __restrict__ f, const float* __restrict__ w, int Np, int HW, int C,
                                                             float coef, float* __restrict__ part) {
    __shared__ float s_w[4];
    const int i = blockIdx.x, chunk = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int p = chunk + LP_CHUNKS * wave; p < HW; p += LP_CHUNKS * 4) {
        const float* f0 = f + ((size_t)i * HW + p) * C;
        const float* f1 = f + ((size_t)(Np + i) * HW + p) * C;
        float s0 = 0.f, s1 = 0.f;
        for (int c = lane; c < C; c += 64) { s0 += f0[c] * f0[c]; s1 += f1[c] * f1[c]; }
        s0 = wave_sum(s0); s1 = wave_sum(s1);
        const float R0 = sqrtf(s0 + LP_EPS) + LP_EPS, R1 = sqrtf(s1 + LP_EPS) + LP_EPS;
        float d = 0.f;
        for (int c = lane; c < C; c += 64) { const float u = f0[c] / R0 - f1[c] / R1; d += w[c] * (u * u); }
        acc += wave_sum(d);
    }
    if (lane == 0) s_w[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[i * LP_CHUNKS + chunk] += coef * (((s_w[0] + s_w[1]) + s_w[2]) + s_w[3]);
}

// g[(i, p), c] (+)= gscale * coef * d/d f0_c of the pixel's term, times [f0_c > 0] (the tap is a ReLU output):
//   gn_c = 2 w_c (f0_c / R0 - f1_c / R1);  g_c = gn_c / R0 - f0_c (sum_k gn_k f0_k) / (r0 R0^2),  r0 = sqrt(sum f0^2 + eps)
__global__ __launch_bounds__(256) void lpips_head_bwd_kernel(const float* __restrict__ f, const float* __restrict__ w, int Np, int HW, int C,
                                                             float coef, const float* __restrict__ gscale, int accumulate, float* __restrict__ g) {
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long)Np * HW) return;
    const float up = coef * (gscale != nullptr ? gscale[0] : 1.f);
    const float* f0 = f + (size_t)pix * C;
    const float* f1 = f + ((size_t)Np * HW + pix) * C;
    float s0 = 0.f, s1 = 0.f;
    for (int c = lane; c < C; c += 64) { s0 += f0[c] * f0[c]; s1 += f1[c] * f1[c]; }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const float r0 = sqrtf(s0 + LP_EPS), R0 = r0 + LP_EPS, R1 = sqrtf(s1 + LP_EPS) + LP_EPS;
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot += 2.f * w[c] * (f0[c] / R0 - f1[c] / R1) * f0[c];
    dot = wave_sum(dot);
    const float k2 = dot / (r0 * R0 * R0);
    float* go = g + (size_t)pix * C;
    for (int c = lane; c < C; c += 64) {
        const float gn = 2.f * w[c] * (f0[c] / R0 - f1[c] / R1);
        float v = f0[c] > 0.f ? up * (gn / R0 - f0[c] * k2) : 0.f;
        go[c] = accumulate ? go[c] + v : v;
    }
}

__global__ void lpips_finish_kernel(const float* __restrict__ part, int Np, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < Np * LP_CHUNKS; ++i) s += part[i];
        out[0] = s;
    }
}

// img[p] = idx[p] >= 0 ? rgb[idx[p]] : bg   (M:41-50: `_unpack_imgs` fills the patch pixels without a ray with the background colour)
__global__ __launch_bounds__(256) void unpack_patches_fwd_kernel(const float* __restrict__ rgb, const int* __restrict__ idx, const float* __restrict__ bg,
                                                                 float bg_scale, long npix, float* __restrict__ img) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < npix * 3; e += (long)gridDim.x * 256) {
        const long p = e / 3;
        const int c = (int)(e % 3), r = idx[p];
        img[e] = r >= 0 ? rgb[(size_t)r * 3 + c] : bg[c] * bg_scale;
    }
}
// g_rgb[idx[p]] = scale_c * g_img[p]  (every ray owns exactly one patch pixel; rays without one keep the zero the caller wrote)
__global__ __launch_bounds__(256) void unpack_patches_bwd_kernel(const float* __restrict__ g_img, const int* __restrict__ idx, long npix,
                                                                 float s0, float s1, float s2, float* __restrict__ g_rgb) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < npix * 3; e += (long)gridDim.x * 256) {
        const long p = e / 3;
        const int c = (int)(e % 3), r = idx[p];
        if (r >= 0) g_rgb[(size_t)r * 3 + c] = g_img[e] * (c == 0 ? s0 : (c == 1 ? s1 : s2));
    }
}

// y = relu(y + bias) in place (behind hos_linear_fwd_splitk, which has no epilogue)
__global__ __launch_bounds__(256) void bias_relu_kernel(float* __restrict__ y, const float* __restrict__ bias, long M, int N) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < M * N; e += (long)gridDim.x * 256) y[e] = fmaxf(y[e] + bias[e % N], 0.f);
}

inline int blocks_for(long total) { long b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int hos_lpips_prep(const float* x01, int64_t n_pixels, float* out, hos_stream_t stream) {
    if (!x01 || !out || n_pixels <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(lpips_prep_kernel, dim3(blocks_for(n_pixels * 3)), dim3(256), 0, static_cast<hipStream_t>(stream), x01, (long)n_pixels * 3, out);
    return hos_launch_status();
}

extern "C" int hos_im2col3x3(const float* in, int NI, int H, int W, int C, float* col, int ld, hos_stream_t stream) {
    if (!in || !col || NI <= 0 || H <= 0 || W <= 0 || C <= 0) return HOS_E_ARG;
    if (ld < 9 * C) return HOS_E_SHAPE;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(blocks_for((long)NI * H * W * ld)), dim3(256), 0, static_cast<hipStream_t>(stream), in, NI, H, W, C, col, ld);
    return hos_launch_status();
}

extern "C" int hos_col2im3x3(const float* dcol, int ld, int NI, int H, int W, int C, const float* relu_src, float* dx, hos_stream_t stream) {
    if (!dcol || !dx || NI <= 0 || H <= 0 || W <= 0 || C <= 0) return HOS_E_ARG;
    if (ld < 9 * C) return HOS_E_SHAPE;
    hipLaunchKernelGGL(col2im3x3_kernel, dim3(blocks_for((long)NI * H * W * C)), dim3(256), 0, static_cast<hipStream_t>(stream), dcol, ld, NI, H, W, C, relu_src, dx);
    return hos_launch_status();
}

extern "C" int hos_maxpool2x2_fwd(const float* in, int NI, int H, int W, int C, float* out, hos_stream_t stream) {
    if (!in || !out || NI <= 0 || H <= 0 || W <= 0 || C <= 0) return HOS_E_ARG;
    if ((H & 1) || (W & 1)) return HOS_E_SHAPE;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(blocks_for((long)NI * (H / 2) * (W / 2) * C)), dim3(256), 0, static_cast<hipStream_t>(stream), in, NI, H, W, C, out);
    return hos_launch_status();
}

extern "C" int hos_maxpool2x2_bwd(const float* g_out, const float* in, int NI, int H, int W, int C, float* g_in, hos_stream_t stream) {
    if (!g_out || !in || !g_in || NI <= 0 || H <= 0 || W <= 0 || C <= 0) return HOS_E_ARG;
    if ((H & 1) || (W & 1)) return HOS_E_SHAPE;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(blocks_for((long)NI * H * W * C)), dim3(256), 0, static_cast<hipStream_t>(stream), g_out, in, NI, H, W, C, g_in);
    return hos_launch_status();
}

extern "C" int hos_lpips_head_fwd(const float* feats, const float* lin_w, int Np, int HW, int C, float coef, float* part, hos_stream_t stream) {
    if (!feats || !lin_w || !part || Np <= 0 || HW <= 0 || C <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(lpips_head_fwd_kernel, dim3(Np, LP_CHUNKS), dim3(256), 0, static_cast<hipStream_t>(stream), feats, lin_w, Np, HW, C, coef, part);
    return hos_launch_status();
}

extern "C" int hos_lpips_head_bwd(const float* feats, const float* lin_w, int Np, int HW, int C, float coef, const float* gscale,
                                  int accumulate, float* g_feats, hos_stream_t stream) {
    if (!feats || !lin_w || !g_feats || Np <= 0 || HW <= 0 || C <= 0) return HOS_E_ARG;
    const long pixels = (long)Np * HW;
    hipLaunchKernelGGL(lpips_head_bwd_kernel, dim3((unsigned)((pixels + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), feats, lin_w, Np, HW, C,
                       coef, gscale, accumulate, g_feats);
    return hos_launch_status();
}

extern "C" int hos_lpips_finish(const float* part, int Np, float* out, hos_stream_t stream) {
    if (!part || !out || Np <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(lpips_finish_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), part, Np, out);
    return hos_launch_status();
}

extern "C" int hos_unpack_patches_fwd(const float* rgb, const int32_t* idx, const float* bgcolor, float bg_scale, int64_t n_pixels, float* img,
                                      hos_stream_t stream) {
    if (!rgb || !idx || !bgcolor || !img || n_pixels <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(unpack_patches_fwd_kernel, dim3(blocks_for(n_pixels * 3)), dim3(256), 0, static_cast<hipStream_t>(stream), rgb, idx, bgcolor, bg_scale,
                       (long)n_pixels, img);
    return hos_launch_status();
}

extern "C" int hos_unpack_patches_bwd(const float* g_img, const int32_t* idx, int64_t n_pixels, float s0, float s1, float s2, float* g_rgb,
                                      hos_stream_t stream) {
    if (!g_img || !idx || !g_rgb || n_pixels <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(unpack_patches_bwd_kernel, dim3(blocks_for(n_pixels * 3)), dim3(256), 0, static_cast<hipStream_t>(stream), g_img, idx, (long)n_pixels,
                       s0, s1, s2, g_rgb);
    return hos_launch_status();
}

extern "C" int hos_lpips_part_floats(int Np) { return Np > 0 ? Np * LP_CHUNKS : 0; }

extern "C" int hos_bias_relu(float* y, const float* bias, int64_t M, int N, hos_stream_t stream) {
    if (!y || !bias || M <= 0 || N <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(bias_relu_kernel, dim3(blocks_for((long)M * N)), dim3(256), 0, static_cast<hipStream_t>(stream), y, bias, (long)M, N);
    return hos_launch_status();
}
