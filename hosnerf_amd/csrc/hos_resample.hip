// Proposal resampling, one wavefront per ray, everything for a ray resident in LDS.
//
// Fuses the reference chain  max_dilate_weights (H:187-194) -> trim + logits (M:469-482) ->
// softmax / CDF (H:227-229, H:197-204) -> inverse CDF (H:208-224) -> interval edges (H:373-399)
// -> s_to_t (H:169-174), which in torch materialises [B,193,64] and 4x [B,191,S] broadcast
// temporaries (17 % of reference forward time).
//
// Traffic per ray: reads (n+1)+n floats, writes 2*(S+1) floats (+S int32 if bin_idx) -- HBM-bound,
// algorithmic bytes/ray = 4*(2n+1 + 2S+2).
//
// Bit-exactness notes:
//  * the sort of cat[t, t-d, t+d] is a merge of three sorted runs; the sorted multiset is unique,
//    so the result equals torch.sort regardless of tie order;
//  * the CDF is accumulated sequentially (fp32, index order) like torch.cumsum on the CPU
//    so that the knot values -- and with them the bin index of each sample -- track the oracle;
//  * u = u_base + jitter*scale is evaluated as two rounded fp32 ops (file is built with
//    -ffp-contract=off).
#include "hos_common.h"

namespace {

constexpr int NMAX = 64;             // max bins of the previous level
constexpr int EMAX = 3 * NMAX + 1;   // dilated edges
constexpr int SMAX = 64;             // max new samples
constexpr float EPS = 1.1920929e-07f;

struct RayLds {
    float t[NMAX + 1];
    float p[NMAX];
    float lo[NMAX];
    float hi[NMAX];
    float e[EMAX + 3];
    float w[EMAX + 3];
    float cw[EMAX + 3];
    float cen[SMAX];
};

// number of elements of sorted a[0..n) that are < x  /  <= x
__device__ __forceinline__ int count_lt(const float* a, int n, float x) {
    int l = 0, r = n;
    while (l < r) { int m = (l + r) >> 1; if (a[m] < x) l = m + 1; else r = m; }
    return l;
}
__device__ __forceinline__ int count_le(const float* a, int n, float x) {
    int l = 0, r = n;
    while (l < r) { int m = (l + r) >> 1; if (a[m] <= x) l = m + 1; else r = m; }
    return l;
}

__global__ __launch_bounds__(256) void resample_kernel(
    const float* __restrict__ sdist_prev, const float* __restrict__ w_prev, int n, int B, int S,
    float dilation, float anneal, const float* __restrict__ train_frac_dev, float anneal_slope, float pad,
    const float* __restrict__ u_base,
    const float* __restrict__ jitter, float jitter_scale, float s_near, float s_far,
    float* __restrict__ sdist, float* __restrict__ tdist, int32_t* __restrict__ bin_idx) {
    __shared__ RayLds lds[4];
    if (train_frac_dev) {      // M:459-460 bias(train_frac, slope) from device memory: a captured step anneals like an eager one
        const double x = (double)*train_frac_dev;
        anneal = anneal_slope > 0.f ? (float)(((double)anneal_slope * x) / (((double)anneal_slope - 1.0) * x + 1.0)) : 1.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray_raw = blockIdx.x * 4 + wave;
    const bool live = ray_raw < B;
    const int ray = live ? ray_raw : B - 1;
    RayLds& L = lds[wave];

    for (int i = lane; i <= n; i += 64) L.t[i] = sdist_prev[(size_t)ray * (n + 1) + i];
    for (int i = lane; i < n; i += 64) {
        const float wv = w_prev[(size_t)ray * n + i];
        L.p[i] = wv;   // becomes the pdf below
    }
    __syncthreads();

    int nb;   // bins of the histogram we sample from; edges live in L.e[0..nb], weights in L.w[0..nb)
    if (n > 1) {
        for (int j = lane; j < n; j += 64) {
            const float t0 = L.t[j], t1 = L.t[j + 1];
            L.p[j] = L.p[j] / fmaxf(t1 - t0, EPS);       // H:177-178
            L.lo[j] = t0 - dilation;                      // H:154
            L.hi[j] = t1 + dilation;                      // H:155
        }
        __syncthreads();
        // three-way merge by rank (runs t, lo, hi are each sorted); store clipped (H:156-157)
        const int ne = 3 * n + 1;
        for (int i = lane; i <= n; i += 64) {
            const float x = L.t[i];
            const int r = i + count_lt(L.lo, n, x) + count_lt(L.hi, n, x);
            L.e[r] = fminf(fmaxf(x, 0.f), 1.f);
        }
        for (int j = lane; j < n; j += 64) {
            float x = L.lo[j];
            int r = j + count_le(L.t, n + 1, x) + count_lt(L.hi, n, x);
            L.e[r] = fminf(fmaxf(x, 0.f), 1.f);
            x = L.hi[j];
            r = j + count_le(L.t, n + 1, x) + count_le(L.lo, n, x);
            L.e[r] = fminf(fmaxf(x, 0.f), 1.f);
        }
        __syncthreads();
        // dilated pdf: max over the (contiguous) set of bins whose dilated support contains e_i (H:158-165)
        float part = 0.f;
        for (int i = lane; i < ne - 1; i += 64) {
            const float x = L.e[i];
            float m = 0.f;
            for (int j = 0; j < n; ++j) {
                const bool in = (L.lo[j] <= x) && (L.hi[j] > x);
                m = fmaxf(m, in ? L.p[j] : 0.f);
            }
            const float wv = m * (L.e[i + 1] - x);        // H:182-183
            L.w[i] = wv;
            part += wv;
        }
        const float total = fmaxf(wave_sum(part), EPS);   // H:191-192
        __syncthreads();
        // renormalise, and trim first/last edge + bin (M:469-470): shift left by one
        nb = ne - 3;                                       // 3n-2 bins, 3n-1 edges
        float ev[4], wv[4];
        int cnt = 0;
        for (int i = lane; i <= nb; i += 64, ++cnt) { ev[cnt] = L.e[i + 1]; wv[cnt] = (i < nb) ? L.w[i + 1] / total : 0.f; }
        __syncthreads();
        cnt = 0;
        for (int i = lane; i <= nb; i += 64, ++cnt) { L.e[i] = ev[cnt]; if (i < nb) L.w[i] = wv[cnt]; }
    } else {
        nb = 1;
        if (lane < 2) L.e[lane] = L.t[lane];
        if (lane == 0) L.w[0] = L.p[0];
    }
    __syncthreads();

    // logits (M:478-482) -> softmax (H:228)
    float lmax = -INFINITY;
    float lg[4];
    {
        int cnt = 0;
        for (int i = lane; i < nb; i += 64, ++cnt) {
            const float v = (L.e[i + 1] > L.e[i]) ? anneal * logf(L.w[i] + pad) : -INFINITY;
            lg[cnt] = v;
            lmax = fmaxf(lmax, v);
        }
    }
    lmax = wave_max(lmax);
    float esum = 0.f;
    {
        int cnt = 0;
        for (int i = lane; i < nb; i += 64, ++cnt) { lg[cnt] = expf(lg[cnt] - lmax); esum += lg[cnt]; }
    }
    esum = wave_sum(esum);
    {
        int cnt = 0;
        for (int i = lane; i < nb; i += 64, ++cnt) L.w[i] = lg[cnt] / esum;
    }
    __syncthreads();
    // CDF knots cw[0..nb] (H:197-204); sequential fp32 accumulation (see header)
    if (lane == 0) {
        float c = 0.f;
        L.cw[0] = 0.f;
        for (int i = 0; i < nb - 1; ++i) { c += L.w[i]; L.cw[i + 1] = fminf(c, 1.f); }
        L.cw[nb] = 1.f;
    }
    __syncthreads();

    // inverse CDF at u_s (H:208-224, restated as upper_bound + lerp)
    if (lane < S) {
        float u = u_base[lane];
        if (jitter != nullptr) { const float j = jitter[ray] * jitter_scale; u = u + j; }
        const int c = count_le(L.cw, nb + 1, u);
        const int ilo = max(c - 1, 0), ihi = min(c, nb);
        const float x0 = L.cw[ilo], x1 = L.cw[ihi];
        const float f0 = L.e[ilo], f1 = L.e[ihi];
        float off = (u - x0) / (x1 - x0);
        if (off != off) off = 0.f;                        // nan_to_num(., 0)
        off = fminf(fmaxf(off, 0.f), 1.f);
        L.cen[lane] = f0 + off * (f1 - f0);
        if (bin_idx != nullptr && live) bin_idx[(size_t)ray * S + lane] = ilo;
    }
    __syncthreads();
    // interval edges (H:390-395) and ray distances (H:172)
    for (int k = lane; k <= S; k += 64) {
        float s;
        if (k == 0) {
            const float mid0 = (L.cen[1] + L.cen[0]) / 2.f;
            s = fmaxf(2.f * L.cen[0] - mid0, 0.f);
        } else if (k == S) {
            const float midl = (L.cen[S - 1] + L.cen[S - 2]) / 2.f;
            s = fminf(2.f * L.cen[S - 1] - midl, 1.f);
        } else {
            s = (L.cen[k] + L.cen[k - 1]) / 2.f;
        }
        if (live) {
            sdist[(size_t)ray * (S + 1) + k] = s;
            tdist[(size_t)ray * (S + 1) + k] = 1.f / (s * s_far + (1.f - s) * s_near);
        }
    }
}

}  // namespace

extern "C" int hos_resample(const float* sdist_prev, const float* w_prev, int n_prev, int B, int S,
                            float dilation, float anneal, const float* train_frac_dev, float anneal_slope,
                            float resample_padding, const float* u_base, const float* jitter, float jitter_scale,
                            float near_, float far_, float* sdist, float* tdist, int32_t* bin_idx,
                            hos_stream_t stream) {
    if (!sdist_prev || !w_prev || !u_base || !sdist || !tdist || B <= 0) return HOS_E_ARG;
    if (n_prev < 1 || n_prev > NMAX || S < 2 || S > SMAX) return HOS_E_SHAPE;
    const float s_near = (float)(1.0 / (double)near_), s_far = (float)(1.0 / (double)far_);
    hipLaunchKernelGGL(resample_kernel, dim3(hos_cdiv(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       sdist_prev, w_prev, n_prev, B, S, dilation, anneal, train_frac_dev, anneal_slope, resample_padding, u_base, jitter,
                       jitter_scale, s_near, s_far, sdist, tdist, bin_idx);
    return hos_launch_status();
}
