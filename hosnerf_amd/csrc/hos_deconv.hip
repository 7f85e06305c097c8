// ConvTranspose3d(kernel 4, stride 2, padding 1) of the motion-weight volume decoder (U:21-59, `ConvDecoder3D`;
// deconv_vol_decoder.py:17-42) as GEMM + gather, channel-last activations.
//
//   forward   Ycol[M, Cout*64] = X[M, Cin] @ W[Cin, Cout*64]          (hos_linear_dgrad form, W = the reference's
//                                                                      [Cin, Cout, 4,4,4] weight as it lies in memory)
//             out[(2D)^3, Cout] = act( bias + sum over the 8 taps that reach an output voxel )   <- hos_deconv3d_col2im
//   backward  dYcol[i, co*64+k] = dPre[o(i,k), co]                                              <- hos_deconv3d_im2col
//             dX = dYcol @ W^T (hos_linear_fwd form),  dW += X^T @ dYcol (hos_linear_wgrad)
//
// Per dimension an output index o receives input i through tap k with o = 2 i - 1 + k: o odd -> (k=0, i=(o+1)/2),
// (k=2, i=(o-1)/2); o even -> (k=1, i=o/2), (k=3, i=o/2-1).  The decoder is 28 GFLOP per training step and its five
// weight tensors are 253 MB: the GEMMs are weight-streaming (M = 1, 8, 64, 512, 4096 rows), these two kernels move
// < 60 MB.  MIOpen spent 25-45 ms per step on the same layers (batch 1, 1^3..16^3 inputs).
#include "hos_common.h"

namespace {

__device__ __forceinline__ void taps(int o, int D, int (&i)[2], int (&k)[2]) {
    if (o & 1) { k[0] = 0; i[0] = (o + 1) >> 1; k[1] = 2; i[1] = (o - 1) >> 1; }
    else       { k[0] = 1; i[0] = o >> 1;       k[1] = 3; i[1] = (o >> 1) - 1; }
    if (i[0] >= D) i[0] = -1;
    if (i[1] < 0) i[1] = -1;
}

__global__ __launch_bounds__(256) void deconv3d_col2im_kernel(const float* __restrict__ ycol, const float* __restrict__ bias,
                                                              int D, int Cout, float slope, int leaky, float* __restrict__ out) {
    const int O = 2 * D;
    const long total = (long)O * O * O * Cout;
    const int ldy = Cout * 64;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int co = (int)(t % Cout);
        long o = t / Cout;
        const int ox = (int)(o % O); o /= O;
        const int oy = (int)(o % O);
        const int oz = (int)(o / O);
        int iz[2], kz[2], iy[2], ky[2], ix[2], kx[2];
        taps(oz, D, iz, kz); taps(oy, D, iy, ky); taps(ox, D, ix, kx);
        float acc = bias ? bias[co] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (iz[a] < 0 || iy[b] < 0 || ix[c] < 0) continue;
                    const long m = ((long)iz[a] * D + iy[b]) * D + ix[c];
                    acc += ycol[m * ldy + co * 64 + kz[a] * 16 + ky[b] * 4 + kx[c]];
                }
        if (leaky && acc < 0.f) acc *= slope;
        out[t] = acc;
    }
}

__global__ __launch_bounds__(256) void deconv3d_im2col_kernel(const float* __restrict__ dpre, int D, int Cout,
                                                              float* __restrict__ dycol) {
    const int O = 2 * D;
    const long total = (long)D * D * D * Cout * 64;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int k = (int)(t & 63);
        long r = t >> 6;
        const int co = (int)(r % Cout);
        long m = r / Cout;
        const int ix = (int)(m % D); m /= D;
        const int iy = (int)(m % D);
        const int iz = (int)(m / D);
        const int oz = 2 * iz - 1 + (k >> 4), oy = 2 * iy - 1 + ((k >> 2) & 3), ox = 2 * ix - 1 + (k & 3);
        float v = 0.f;
        if (oz >= 0 && oz < O && oy >= 0 && oy < O && ox >= 0 && ox < O)
            v = dpre[(((long)oz * O + oy) * O + ox) * Cout + co];
        dycol[t] = v;
    }
}

// gW[k][n] += sum_m x[m][k] * dy[m][n] for a HANDFUL of rows m (the decoder's first layers see 1, 8 and 64 voxels against
// 134 / 67 / 33 MB of weights): the gradient is an outer-product stream -- read, add, write every weight once.  A thread owns
// four consecutive n of KR weight rows; x comes from LDS, dy is re-read once per row block (L2 resident: M x N x 4 bytes).
// The tiled GEMM spent its time on a 128-row A tile that is 98 % padding here (2.4 TB/s on the 134 MB layer).
// (OA_KR rows per thread: 32 for the wide layers; 8 when the grid would otherwise be under one workgroup per CU -- the compact
// first layer, [1024, 4096], ran on 128 workgroups at 0.9 TB/s)
template <int OA_KR>
__global__ __launch_bounds__(256) void outer_accum_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                          float* __restrict__ gW, int ldw, int M, int K, int N) {
    __shared__ float xs[64][OA_KR];
    const int k0 = blockIdx.y * OA_KR;
    for (int i = threadIdx.x; i < M * OA_KR; i += blockDim.x) {
        const int m = i / OA_KR, k = i % OA_KR;
        xs[m][k] = (k0 + k < K) ? x[(size_t)m * ldx + k0 + k] : 0.f;
    }
    __syncthreads();
    const int n4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (n4 >= N) return;
    float4 acc[OA_KR];
#pragma unroll
    for (int k = 0; k < OA_KR; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < M; ++m) {
        const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)m * lddy + n4);
#pragma unroll
        for (int k = 0; k < OA_KR; ++k) {
            const float xv = xs[m][k];
            acc[k].x += xv * d.x; acc[k].y += xv * d.y; acc[k].z += xv * d.z; acc[k].w += xv * d.w;
        }
    }
#pragma unroll
    for (int k = 0; k < OA_KR; ++k) {
        if (k0 + k < K) {
            float4* p = reinterpret_cast<float4*>(gW + (size_t)(k0 + k) * ldw + n4);
            float4 v = *p;
            v.x += acc[k].x; v.y += acc[k].y; v.z += acc[k].z; v.w += acc[k].w;
            *p = v;
        }
    }
}

// dpre = g * (out > 0 ? 1 : slope) (LeakyReLU backward; leaky == 0: dpre is g itself and nothing is written) and
// db[c] += sum_rows dpre[row][c] in the same pass.  The 256 threads of a block form (256 / CP row lanes) x (CP column lanes),
// CP = the power of two >= min(C, 256): the last layer has 27 channels and 32 768 voxels, the inner ones 256-512 channels and
// 8-4 096 voxels.  A block owns `rows` rows (chosen for ~1000 blocks); the row lanes are reduced through LDS, then one atomic per column and block.
__global__ __launch_bounds__(256) void deconv3d_dpre_kernel(const float* __restrict__ g, const float* __restrict__ out, long R, int C, int CP, int rows,
                                                            float slope, int leaky, float* __restrict__ dpre, float* __restrict__ db) {
    __shared__ float red[256];
    const int rl = threadIdx.x / CP, cl = threadIdx.x % CP, RL = 256 / CP;
    const long r0 = (long)blockIdx.x * rows;
    const long r1 = r0 + rows < R ? r0 + rows : R;
    for (int c0 = 0; c0 < C; c0 += CP) {
        const int c = c0 + cl;
        float s = 0.f;
        if (c < C) {
            for (long r = r0 + rl; r < r1; r += RL) {
                float v = g[r * C + c];
                if (leaky) { if (!(out[r * C + c] > 0.f)) v *= slope; dpre[r * C + c] = v; }
                s += v;
            }
        }
        if (db == nullptr) continue;
        red[threadIdx.x] = s;
        __syncthreads();
        if (rl == 0 && c < C) {
            for (int k = 1; k < RL; ++k) s += red[k * CP + cl];
            __hip_atomic_fetch_add(db + c, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

}  // namespace

// dpre [R, C] = g * (out > 0 ? 1 : slope) (leaky != 0; otherwise dpre is not touched and g is the pre-activation gradient)
// and db [C] += column sums of it (db may be NULL).  R = (2D)^3 output voxels, C = Cout, channel-last.
// Reference: LeakyReLU(0.2) + bias gradient of the ConvTranspose3d blocks, deconv_vol_decoder.py:17-42.
extern "C" int hos_deconv3d_dpre(const float* g, const float* out, long long R, int C, float leaky_slope, int leaky, float* dpre,
                                 float* db, hos_stream_t stream) {
    if (!g || R <= 0 || C <= 0 || (leaky && (!out || !dpre))) return HOS_E_ARG;
    int CP = 1;
    while (CP < C && CP < 256) CP <<= 1;
    const int RL = 256 / CP;
    long rows = (R + 1023) / 1024;
    rows = (rows + RL - 1) / RL * RL;
    const long blocks = (R + rows - 1) / rows;
    hipLaunchKernelGGL(deconv3d_dpre_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g, out, (long)R, C, CP,
                       (int)rows, leaky_slope, leaky, dpre, db);
    return hos_launch_status();
}

// gW [K, ldw] += x[M, :K]^T . dy[M, :N] for M <= 64 rows; N % 4 == 0, 16-byte aligned dy / gW rows.  Exact fp32 (FMA order:
// m ascending).  Reference: the weight gradient of ConvTranspose3d in deconv_vol_decoder.py:34-42 for its first layers.
extern "C" int hos_outer_accum(const float* x, int ldx, const float* dy, int lddy, float* gW, int ldw, int M, int K, int N,
                               hos_stream_t stream) {
    if (!x || !dy || !gW || M <= 0 || K <= 0 || N <= 0) return HOS_E_ARG;
    if (M > 64) return HOS_E_SHAPE;
    if ((N & 3) || (lddy & 3) || (ldw & 3) || (((uintptr_t)dy | (uintptr_t)gW) & 15u)) return HOS_E_ALIGN;
    const unsigned gx = (unsigned)((N / 4 + 255) / 256);
    if (gx * (unsigned)((K + 31) / 32) < 256u)
        hipLaunchKernelGGL(outer_accum_kernel<8>, dim3(gx, (unsigned)((K + 7) / 8)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, dy, lddy, gW, ldw, M, K, N);
    else
        hipLaunchKernelGGL(outer_accum_kernel<32>, dim3(gx, (unsigned)((K + 31) / 32)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, dy, lddy, gW, ldw, M, K, N);
    return hos_launch_status();
}

// y[n] = act(sum_k x[k] W[k][n] + bias[n % bias_mod]) for ONE row x against a long weight stream (the compact first layer of the
// decoder: [1, 1024] x [1024, 4096], 16.8 MB): two launches with a fixed summation order -- partial[slab][n] over slabs of
// GV_SLAB weight rows on K / GV_SLAB x N / 1024 workgroups, then the slabs in ascending order.  The 32-row tile of the exact-fp32
// GEMM ran this shape on N / 128 = 32 workgroups (34 us for 16.8 MB).
constexpr int GV_SLAB = 16;
__global__ __launch_bounds__(256) void gemv_partial_kernel(const float* __restrict__ x, const float* __restrict__ W, int ldw, int K, int N,
                                                           float* __restrict__ partial) {
    const int n4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int k0 = blockIdx.y * GV_SLAB;
    if (n4 >= N) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < GV_SLAB; ++i) {
        const int k = k0 + i;
        if (k < K) {
            const float xv = x[k];
            const float4 w = *reinterpret_cast<const float4*>(W + (size_t)k * ldw + n4);
            acc.x += xv * w.x; acc.y += xv * w.y; acc.z += xv * w.z; acc.w += xv * w.w;
        }
    }
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * N + n4) = acc;
}
__global__ __launch_bounds__(256) void gemv_finish_kernel(const float* __restrict__ partial, int slabs, int N, const float* __restrict__ bias,
                                                          int bias_mod, float slope, int leaky, float* __restrict__ y) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int g = 0; g < slabs; ++g) s += partial[(size_t)g * N + n];
    if (bias) s += bias[n % bias_mod];
    if (leaky && s < 0.f) s *= slope;
    y[n] = s;
}

extern "C" long long hos_gemv_ws_floats(int K, int N) { return (long long)((K + GV_SLAB - 1) / GV_SLAB) * N; }

extern "C" int hos_gemv_rowvec(const float* x, const float* W, int ldw, int K, int N, const float* bias, int bias_mod,
                               float leaky_slope, int leaky, float* ws, float* y, hos_stream_t stream) {
    if (!x || !W || !ws || !y || K <= 0 || N <= 0 || (bias && bias_mod <= 0)) return HOS_E_ARG;
    if ((N & 3) || (ldw & 3) || (((uintptr_t)W | (uintptr_t)ws) & 15u)) return HOS_E_ALIGN;
    const int slabs = (K + GV_SLAB - 1) / GV_SLAB;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gemv_partial_kernel, dim3((unsigned)((N / 4 + 255) / 256), (unsigned)slabs), dim3(256), 0, s, x, W, ldw, K, N, ws);
    hipLaunchKernelGGL(gemv_finish_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, ws, slabs, N, bias, bias_mod, leaky_slope, leaky, y);
    return hos_launch_status();
}

extern "C" int hos_deconv3d_col2im(const float* ycol, const float* bias, int D, int Cout, float leaky_slope, int leaky,
                                   float* out, hos_stream_t stream) {
    if (!ycol || !out || D <= 0 || Cout <= 0) return HOS_E_ARG;
    const long total = 8L * D * D * D * Cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(deconv3d_col2im_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), ycol, bias, D, Cout,
                       leaky_slope, leaky, out);
    return hos_launch_status();
}

// ---- round 5: pieces of the volume decoder sharded over the data-parallel ranks by INPUT channel ------------------------
// A rank multiplies its rows [c0, c1) of a layer's weight: the forward's partial pre-activations are summed over the ranks (after
// the linear col2im), then bias + LeakyReLU are applied here; the backward's input-gradient slices are gathered rank after rank
// and interleaved into channel order here.
namespace {
__global__ __launch_bounds__(256) void bias_lrelu_kernel(float* __restrict__ y, const float* __restrict__ bias, long total, int N, float slope, int leaky) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        float v = y[e] + (bias ? bias[e % N] : 0.f);
        if (leaky && v < 0.f) v *= slope;
        y[e] = v;
    }
}
__global__ __launch_bounds__(256) void shard_interleave_kernel(const float* __restrict__ parts, int world, int M, int cs, float* __restrict__ out) {
    const long total = (long)world * M * cs;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % cs);
        const int m = (int)((e / cs) % M);
        const int r = (int)(e / ((long)cs * M));
        out[(size_t)m * ((size_t)world * cs) + (size_t)r * cs + c] = parts[e];
    }
}
}  // namespace
// y [M, N] = LeakyReLU?(y + bias [N]) in place (bias may be NULL)
extern "C" int hos_bias_lrelu(float* y, const float* bias, long long M, int N, float leaky_slope, int leaky, hos_stream_t stream) {
    if (!y || M <= 0 || N <= 0) return HOS_E_ARG;
    const long total = (long)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bias_lrelu_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), y, bias, total, N, leaky_slope, leaky);
    return hos_launch_status();
}
// out [M, world * cs] with out[m][r * cs + c] = parts[r][m][c] (parts [world, M, cs]: what an all-gather of per-rank [M, cs] slices returns)
extern "C" int hos_shard_interleave(const float* parts, int world, int M, int cs, float* out, hos_stream_t stream) {
    if (!parts || !out || world <= 0 || M <= 0 || cs <= 0) return HOS_E_ARG;
    const long total = (long)world * M * cs;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(shard_interleave_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), parts, world, M, cs, out);
    return hos_launch_status();
}

extern "C" int hos_deconv3d_im2col(const float* dpre, int D, int Cout, float* dycol, hos_stream_t stream) {
    if (!dpre || !dycol || D <= 0 || Cout <= 0) return HOS_E_ARG;
    const long total = (long)D * D * D * Cout * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(deconv3d_im2col_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dpre, D, Cout, dycol);
    return hos_launch_status();
}

// =====================================================================================================================
// Head and tail of the motion-weight volume decoder (deconv_vol_decoder.py:34-42, U:21-59) -- round 4: these were the last
// torch / library launches of the captured step (F.linear -> a rocBLAS GEMM, leaky_relu, log, add, softmax, pad + permute).
//   head:  h = LeakyReLU(W [N, K] . e [K] + b)          `block_mlp.0` on the learned constant embedding (one row)
//   tail:  vol [C, V^3] = softmax_c( z [V^3, C] + log prior [C, V^3] ),  vol_cl [V^3, 32] = its first Kb channels, channel-last
// =====================================================================================================================
namespace {

// one wave per output n: lanes stride the reduction in float4 (K % 4 == 0), wave sum in a fixed order
__global__ __launch_bounds__(256) void rowdot_lrelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, int ldw,
                                                               const float* __restrict__ b, int N, int K, float slope, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 w = *reinterpret_cast<const float4*>(W + (size_t)n * ldw + k);
        const float4 v = *reinterpret_cast<const float4*>(x + k);
        s += w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w;
    }
    s = wave_sum(s);
    if (lane == 0) { s += b ? b[n] : 0.f; y[n] = s < 0.f ? s * slope : s; }
}

// backward of the head.  Block j owns the 32 outputs n = 32 j .. 32 j + 31; thread k (k < K <= 1024, four columns per thread at
// K > 256 via the loop): gW[n][k] += d_n x[k], gx[k] += sum_n d_n W[n][k] (one atomic per block and column), gb[n] += d_n,
// d_n = g[n] * (y[n] > 0 ? 1 : slope).
__global__ __launch_bounds__(256) void rowdot_lrelu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ x,
                                                               const float* __restrict__ W, int ldw, int N, int K, float slope,
                                                               float* __restrict__ gW, int ldgw, float* __restrict__ gb, float* __restrict__ gx) {
    __shared__ float d[32];
    const int n0 = blockIdx.x * 32;
    if (threadIdx.x < 32) {
        const int n = n0 + threadIdx.x;
        float v = 0.f;
        if (n < N) { v = g[n]; if (!(y[n] > 0.f)) v *= slope; if (gb) gb[n] += v; }
        d[threadIdx.x] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        const float xv = x[k];
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int n = n0 + i;
            if (n >= N) break;
            const float dn = d[i];
            gW[(size_t)n * ldgw + k] += dn * xv;
            acc += dn * W[(size_t)n * ldw + k];
        }
        if (gx) __hip_atomic_fetch_add(gx + k, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// softmax over the C <= 32 channels of every voxel: thread = voxel.  z is channel-last (the last deconvolution's output), the
// prior and the result channel-major [C, V3] (what the backward warp samples per bone).
__global__ __launch_bounds__(256) void vol_softmax_fwd_kernel(const float* __restrict__ z, const float* __restrict__ prior, int C, long V3,
                                                              float* __restrict__ vol) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V3) return;
    float t[32];
    float mx = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < 32; ++c)
        if (c < C) { t[c] = z[v * C + c] + logf(prior[(long)c * V3 + v]); mx = fmaxf(mx, t[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c)
        if (c < C) { t[c] = expf(t[c] - mx); s += t[c]; }
#pragma unroll
    for (int c = 0; c < 32; ++c)
        if (c < C) vol[(long)c * V3 + v] = t[c] / s;
}

// gz[v][c] = vol_c (g_c - sum_j g_j vol_j)
__global__ __launch_bounds__(256) void vol_softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ vol, int C, long V3,
                                                              float* __restrict__ gz) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V3) return;
    float p[32], gg[32];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c)
        if (c < C) { p[c] = vol[(long)c * V3 + v]; gg[c] = g[(long)c * V3 + v]; dot += p[c] * gg[c]; }
#pragma unroll
    for (int c = 0; c < 32; ++c)
        if (c < C) gz[v * C + c] = p[c] * (gg[c] - dot);
}

// vol [C, V3] -> channel-last copy of its first Kb channels, zero padded to 32: vol_cl [V3, 32] (the forward warp taps all bones at
// ONE position: 8 x 128-byte lines per point)
__global__ __launch_bounds__(256) void vol_channel_last_kernel(const float* __restrict__ vol, int Kb, long V3, float* __restrict__ cl) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V3) return;
    float t[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) t[c] = c < Kb ? vol[(long)c * V3 + v] : 0.f;
    float4* o = reinterpret_cast<float4*>(cl + v * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
}

// gradient of the pair (vol, vol_cl) w.r.t. vol: g [C, V3] = g_vol (or 0) + channel-major scatter of g_cl's first Kb channels
__global__ __launch_bounds__(256) void vol_pair_bwd_kernel(const float* __restrict__ g_vol, const float* __restrict__ g_cl, int C, int Kb, long V3,
                                                           float* __restrict__ g) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V3) return;
    float t[32];
    if (g_cl != nullptr) {
        const float4* i4 = reinterpret_cast<const float4*>(g_cl + v * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 w = i4[q]; t[4 * q] = w.x; t[4 * q + 1] = w.y; t[4 * q + 2] = w.z; t[4 * q + 3] = w.w; }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c)
        if (c < C) {
            float s = g_vol ? g_vol[(long)c * V3 + v] : 0.f;
            if (g_cl != nullptr && c < Kb) s += t[c];
            g[(long)c * V3 + v] = s;
        }
}

}  // namespace

// y [N] = LeakyReLU(W [N, ldw] . x [K] + b [N]).  K % 4 == 0, 16-byte aligned x / W rows.
// Reference: `block_mlp` of ConvDecoder3D (U:21-30: Linear + LeakyReLU(0.2)) applied to the constant embedding, deconv_vol_decoder.py:36-37.
extern "C" int hos_rowdot_lrelu_fwd(const float* x, const float* W, int ldw, const float* b, int N, int K, float slope, float* y,
                                    hos_stream_t stream) {
    if (!x || !W || !y || N <= 0 || K <= 0) return HOS_E_ARG;
    if ((K & 3) || (ldw & 3) || (((uintptr_t)x | (uintptr_t)W) & 15u)) return HOS_E_ALIGN;
    hipLaunchKernelGGL(rowdot_lrelu_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, W, ldw, b, N, K, slope, y);
    return hos_launch_status();
}

// Backward of hos_rowdot_lrelu_fwd: gW [N, ldgw] += d x^T, gb [N] += d (NULL: skip), gx [K] += W^T d (NULL: skip), d = g * (y > 0 ? 1 : slope).
extern "C" int hos_rowdot_lrelu_bwd(const float* g, const float* y, const float* x, const float* W, int ldw, int N, int K, float slope,
                                    float* gW, int ldgw, float* gb, float* gx, hos_stream_t stream) {
    if (!g || !y || !x || !W || !gW || N <= 0 || K <= 0) return HOS_E_ARG;
    hipLaunchKernelGGL(rowdot_lrelu_bwd_kernel, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, static_cast<hipStream_t>(stream), g, y, x, W, ldw, N, K,
                       slope, gW, ldgw, gb, gx);
    return hos_launch_status();
}

// vol [C, V3] = softmax over c of (z [V3, C] + log prior [C, V3]); C <= 32.  Reference: deconv_vol_decoder.py:38-42
// (`F.softmax(decoded_weights + torch.log(motion_weights_priors), dim=1)`).
extern "C" int hos_volume_softmax_fwd(const float* z, const float* prior, int C, long long V3, float* vol, hos_stream_t stream) {
    if (!z || !prior || !vol || C <= 0 || V3 <= 0) return HOS_E_ARG;
    if (C > 32) return HOS_E_SHAPE;
    hipLaunchKernelGGL(vol_softmax_fwd_kernel, dim3((unsigned)((V3 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), z, prior, C, (long)V3, vol);
    return hos_launch_status();
}

extern "C" int hos_volume_softmax_bwd(const float* g_vol, const float* vol, int C, long long V3, float* gz, hos_stream_t stream) {
    if (!g_vol || !vol || !gz || C <= 0 || V3 <= 0) return HOS_E_ARG;
    if (C > 32) return HOS_E_SHAPE;
    hipLaunchKernelGGL(vol_softmax_bwd_kernel, dim3((unsigned)((V3 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g_vol, vol, C, (long)V3, gz);
    return hos_launch_status();
}

// vol_cl [V3, 32] = channel-last copy of vol[:Kb] (zero padded); Kb <= 32.  Reference: N:357-399 samples all bone channels at one
// position (`F.grid_sample` on the [K, V, V, V] volume); the layout is this library's.
extern "C" int hos_volume_channel_last(const float* vol, int Kb, long long V3, float* vol_cl, hos_stream_t stream) {
    if (!vol || !vol_cl || Kb <= 0 || V3 <= 0) return HOS_E_ARG;
    if (Kb > 32) return HOS_E_SHAPE;
    if ((uintptr_t)vol_cl & 15u) return HOS_E_ALIGN;
    hipLaunchKernelGGL(vol_channel_last_kernel, dim3((unsigned)((V3 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), vol, Kb, (long)V3, vol_cl);
    return hos_launch_status();
}

// g [C, V3] = g_vol [C, V3] (NULL: 0) + the first Kb channels of g_cl [V3, 32] (NULL: 0), channel-major: the gradient w.r.t. the
// volume from its two consumers (backward warp: channel-major; forward warp: channel-last) in one pass.
extern "C" int hos_volume_pair_bwd(const float* g_vol, const float* g_cl, int C, int Kb, long long V3, float* g, hos_stream_t stream) {
    if (!g || (!g_vol && !g_cl) || C <= 0 || Kb <= 0 || V3 <= 0) return HOS_E_ARG;
    if (C > 32 || Kb > C) return HOS_E_SHAPE;
    if (g_cl && ((uintptr_t)g_cl & 15u)) return HOS_E_ALIGN;
    hipLaunchKernelGGL(vol_pair_bwd_kernel, dim3((unsigned)((V3 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g_vol, g_cl, C, Kb, (long)V3, g);
    return hos_launch_status();
}
