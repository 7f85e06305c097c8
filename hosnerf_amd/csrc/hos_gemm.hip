// Dense MLP contractions on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// One tile kernel, three operand-layout instantiations:
//   FWD   C[m,n]  = epi(sum_k A[m,k] W[n,k] + b[n])     A,W k-contiguous  -> LDS-transposed staging
//   DGRAD dX[m,k] = (sum_n dY[m,n] W[n,k]) * relu'(X)   dY k(=n)-contiguous, W row = reduction index
//   WGRAD dW[n,k] += sum_m dY[m,n] X[m,k]               both operands: reduction index = row
//
// Workgroup = 256 threads = 4 wave64.  Tile BM x BN x 32; LDS holds both operand tiles as
// [k][i] (reduction-major) so an MFMA fragment read (lane -> i = lane&31, k = lane>>5) is a
// conflict-free ds_read_b32 of 32 consecutive floats.  Register-staged global->LDS double buffer,
// one barrier per K tile.  Every reduction dimension is a multiple of 32 by construction (the
// host pads activations / weights), so there is no K-tail path.
//
// Numerics: v_mfma_f32_32x32x2_f32 is bitwise an fp32 fmaf chain in k order (MI355X guide), i.e.
// the same class as the reference's fp32 addmm; SURVEY 7.1 shows bf16/TF32 inputs break the 1e-4
// RGB budget, so this path is the parity-safe one.
#include "hos_common.h"

#include "hos_gemm_common.h"
#include <cstdlib>

namespace {

constexpr int BK = 32;
constexpr int NT = 256;


__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- global -> register staging -------------------------------------------------------------
// K-contiguous operand P[i][k]: tile rows i0.., 32 floats of k.  thread -> (k4 = t&7, i = t>>3 + 32r)
template <int BMN>
__device__ __forceinline__ void load_kc(float4 (&v)[BMN / 32], const float* __restrict__ P, int ld,
                                        int i0, int limit, int k0, int t) {
    const int k4 = t & 7, ir = t >> 3;
#pragma unroll
    for (int r = 0; r < BMN / 32; ++r) {
        const int gi = i0 + ir + 32 * r;
        v[r] = gi < limit ? ldg4(P + (size_t)gi * ld + k0 + k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int BMN, int LD>
__device__ __forceinline__ void store_kc(const float4 (&v)[BMN / 32], float* __restrict__ S, int t) {
    const int k4 = t & 7, ir = t >> 3;
#pragma unroll
    for (int r = 0; r < BMN / 32; ++r) {
        const int i = ir + 32 * r;
        S[(k4 * 4 + 0) * LD + i] = v[r].x;
        S[(k4 * 4 + 1) * LD + i] = v[r].y;
        S[(k4 * 4 + 2) * LD + i] = v[r].z;
        S[(k4 * 4 + 3) * LD + i] = v[r].w;
    }
}
// reduction-row operand P[red][i]: 32 rows, BMN contiguous floats.  thread -> (i4 = t % (BMN/4), row = t/(BMN/4) + RP*r)
template <int BMN>
__device__ __forceinline__ void load_rc(float4 (&v)[BMN / 32], const float* __restrict__ P, int ld,
                                        int i0, int limit, int k0, int t, int row_limit) {
    constexpr int C4 = BMN / 4, RP = NT / C4;
    const int i4 = t % C4, rr = t / C4;
    const int gi = i0 + i4 * 4;
#pragma unroll
    for (int r = 0; r < BMN / 32; ++r) {
        const int row = k0 + rr + RP * r;
        v[r] = (gi < limit && row < row_limit) ? ldg4(P + (size_t)row * ld + gi) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int BMN, int LD>
__device__ __forceinline__ void store_rc(const float4 (&v)[BMN / 32], float* __restrict__ S, int t) {
    constexpr int C4 = BMN / 4, RP = NT / C4;
    const int i4 = t % C4, rr = t / C4;
#pragma unroll
    for (int r = 0; r < BMN / 32; ++r)
        *reinterpret_cast<float4*>(S + (rr + RP * r) * LD + i4 * 4) = v[r];
}

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const GemmArgs a) {
    constexpr bool A_KC = (MODE != MODE_WGRAD);
    constexpr bool B_KC = (MODE == MODE_FWD);
    constexpr int LDA = BM + (A_KC ? 2 : 4);
    constexpr int LDB = BN + (B_KC ? 2 : 4);
    constexpr int WN = (BN == 32) ? 1 : ((BM == 32) ? 4 : 2);   // waves along n
    constexpr int WM = 4 / WN;                                  // waves along m
    constexpr int TM = BM / (WM * 32);           // 32x32 MFMA tiles per wave along m
    constexpr int TN = BN / (WN * 32);
    static_assert(TM >= 1 && TN >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BK][LDA]
    float* Bs = smem + 2 * BK * LDA;        // [2][BK][LDB]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tile ids
    // (n fastest) so neighbouring tiles that share the A row panel hit the same L2.
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nb >> 3, r = nb & 7, x = bid & 7, y = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tn_i = bid % a.tiles_n;
    const int tm_i = (bid / a.tiles_n) % a.tiles_m;
    const int split = bid / (a.tiles_n * a.tiles_m);
    const int i0 = tm_i * BM, j0 = tn_i * BN;
    if (MODE == MODE_FWD && a.m_dev && i0 >= *a.m_dev) return;     // fixed-capacity buffer: rows past the device-side count are dead
    const int kt_begin = split * a.kt_per_split;
    const int kt_end = min(a.nk, kt_begin + a.kt_per_split);
    if (kt_begin >= kt_end) return;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

    float4 ra[BM / 32], rb[BN / 32];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);   // WGRAD bias-gradient partial (column sums of dY)
    const bool do_db = (MODE == MODE_WGRAD) && a.db != nullptr && tn_i == 0;

    auto gload = [&](int kt) {
        if constexpr (A_KC) {
            const float* P = a.A0; int ld = a.lda0; int k0 = kt * BK;
            if (MODE == MODE_FWD && kt >= a.kt0) { P = a.A1; ld = a.lda1; k0 = (kt - a.kt0) * BK; }
            load_kc<BM>(ra, P, ld, i0, a.Mload, k0, t);
        } else {
            load_rc<BM>(ra, a.A0, a.lda0, i0, a.Mload, kt * BK, t, a.red_limit);
            if (do_db) {
#pragma unroll
                for (int r = 0; r < BM / 32; ++r) { bsum.x += ra[r].x; bsum.y += ra[r].y; bsum.z += ra[r].z; bsum.w += ra[r].w; }
            }
        }
        if constexpr (B_KC) load_kc<BN>(rb, a.B, a.ldb, j0, a.Nload, kt * BK, t);
        else                load_rc<BN>(rb, a.B, a.ldb, j0, a.Nload, kt * BK, t, a.red_limit);
    };
    auto sstore = [&](int buf) {
        if constexpr (A_KC) store_kc<BM, LDA>(ra, As + buf * BK * LDA, t);
        else                store_rc<BM, LDA>(ra, As + buf * BK * LDA, t);
        if constexpr (B_KC) store_kc<BN, LDB>(rb, Bs + buf * BK * LDB, t);
        else                store_rc<BN, LDB>(rb, Bs + buf * BK * LDB, t);
    };

    gload(kt_begin);
    sstore(0);
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) gload(kt + 1);
        const float* Ab = As + buf * BK * LDA + wm * (TM * 32) + l31;
        const float* Bb = Bs + buf * BK * LDB + wn * (TN * 32) + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kk = ks * 2 + lhi;
            float fa[TM], fb[TN];
#pragma unroll
            for (int x = 0; x < TM; ++x) fa[x] = Ab[kk * LDA + x * 32];
#pragma unroll
            for (int y = 0; y < TN; ++y) fb[y] = Bb[kk * LDB + y * 32];
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int y = 0; y < TN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x], fb[y], acc[x][y], 0, 0, 0);
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue -------------------------------------------------------------------------
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y)
        {
            const int r0 = i0 + wm * (TM * 32) + x * 32, c0 = j0 + wn * (TN * 32) + y * 32;
            if constexpr (MODE == MODE_WGRAD) {
                if (a.kt_per_split >= a.nk) {
                    // a single split owns the whole reduction: plain 16-byte read-add-write instead of fp32 atomics
                    // (the decoder's outer-product gradients are 134 MB of them per step)
                    GemmArgs e = a;
                    e.mask = nullptr; e.accumulate = 1;
                    gemm_epilogue_tile<MODE_DGRAD>(e, acc[x][y], r0, c0, lane);
                } else {
                    gemm_epilogue_tile<MODE_WGRAD>(a, acc[x][y], r0, c0, lane);
                }
            } else if constexpr (MODE == MODE_DGRAD) {
                // split-K DGRAD (few rows, long weight stream): partial tiles are added with atomics into a zeroed C
                if (a.kt_per_split < a.nk) gemm_epilogue_tile<MODE_WGRAD>(a, acc[x][y], r0, c0, lane);
                else gemm_epilogue_tile<MODE_DGRAD>(a, acc[x][y], r0, c0, lane);
            } else {
                // split-K FWD (hos_linear_fwd_splitk*: small output, long reduction): partial tiles with atomics into a zeroed C,
                // or -- deterministic form, a.aux = slab workspace [splits][M][aux_col] -- as plain stores into slab `split`,
                // summed in a fixed order by splitk_reduce_kernel (round 5: LPIPS' deep convolutions; an atomic sum that lands
                // a pre-activation on the other side of 0 flips a ReLU / pooling winner from run to run)
                if (a.kt_per_split < a.nk) {
                    if (a.aux != nullptr) {
                        GemmArgs w = a;
                        w.C = a.aux + (size_t)split * a.M * a.aux_col;
                        w.ldc = a.aux_col; w.bias = nullptr; w.epi = HOS_EPI_NONE; w.mask = nullptr; w.aux = nullptr;
                        gemm_epilogue_tile<MODE_FWD>(w, acc[x][y], r0, c0, lane);
                    } else {
                        gemm_epilogue_tile<MODE_WGRAD>(a, acc[x][y], r0, c0, lane);
                    }
                } else gemm_epilogue_tile<MODE>(a, acc[x][y], r0, c0, lane);
            }
        }

    if constexpr (MODE == MODE_WGRAD) {
        if (do_db) {
            // reduce the per-thread float4 column sums over the NT/(BM/4) row groups, then one atomic per column
            constexpr int C4 = BM / 4, RP = NT / C4;
            float* red = smem;  // reuse (all MFMA reads are behind the loop's final barrier)
            const int i4 = t % C4, rr = t / C4;
            *reinterpret_cast<float4*>(red + rr * BM + i4 * 4) = bsum;
            __syncthreads();
            if (t < BM) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < RP; ++r) s += red[r * BM + t];
                if (i0 + t < a.M)
                    __hip_atomic_fetch_add(a.db + i0 + t, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int BM, int BN, int MODE>
int launch(const GemmArgs& a, int splits, hipStream_t stream) {
    constexpr bool A_KC = (MODE != MODE_WGRAD);
    constexpr bool B_KC = (MODE == MODE_FWD);
    constexpr int LDA = BM + (A_KC ? 2 : 4);
    constexpr int LDB = BN + (B_KC ? 2 : 4);
    constexpr size_t smem = sizeof(float) * 2 * BK * (LDA + LDB);
    static bool attr_set = false;   // idempotent; racing setters write the same value
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nblocks = a.tiles_m * a.tiles_n * splits;
    hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE>), dim3(nblocks), dim3(NT), smem, stream, a);
    return hos_launch_status();
}

// Zero a [rows][cols] window of a row-major fp32 matrix.  A KERNEL, not hipMemset2DAsync: inside a captured hipGraph the memset
// node of ROCm 7.2 was observed to lose its ordering against the neighbouring kernel nodes once in ~10^4 replays (the split-K
// accumulators then started from the previous replay's sums or were cleared under the atomics: gradients of 1e14 .. inf in
// captured training steps, scripts/soak_graph.py); kernel nodes keep stream order.
__global__ __launch_bounds__(256) void zero2d_kernel(float* __restrict__ p, long ld, int rows, int cols) {
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        p[(i / cols) * ld + (i % cols)] = 0.f;
}
inline int zero2d(float* p, long ld, int rows, int cols, hipStream_t s) {
    const long total = (long)rows * cols;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero2d_kernel, dim3(blocks), dim3(256), 0, s, p, ld, rows, cols);
    return hos_launch_status();
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Arithmetic mode: a process DEFAULT (set once at start-up) and a per-THREAD override (-1 = follow the default).  The override
// is what scoped switches use (ops.gemm_mode: "this module's GEMMs in exact fp32"), so two host threads -- e.g. the caller and
// the autograd engine's worker, or two modules pinned to different modes -- never see each other's setting.
int g_default_mode = HOS_GEMM_BF16X3;
thread_local int t_mode = -1;
#define g_gemm_mode (t_mode >= 0 ? t_mode : g_default_mode)
unsigned int* g_range_flag = nullptr;      // caller-owned device word (hos_set_range_flag); NULL: no range reporting

}  // namespace

unsigned int* hos_range_flag_ptr() { return g_range_flag; }
extern "C" int hos_set_range_flag(unsigned int* flag) { g_range_flag = flag; return HOS_OK; }

extern "C" int hos_set_gemm_mode(int mode) {
    if (mode != HOS_GEMM_FP32 && mode != HOS_GEMM_BF16X3) return HOS_E_ARG;
    g_default_mode = mode;
    return HOS_OK;
}
extern "C" int hos_set_thread_gemm_mode(int mode) {
    if (mode != -1 && mode != HOS_GEMM_FP32 && mode != HOS_GEMM_BF16X3) return HOS_E_ARG;
    t_mode = mode;
    return HOS_OK;
}
extern "C" int hos_get_gemm_mode(void) { return g_gemm_mode; }

extern "C" int hos_linear_fwd(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1,
                              const float* W, int ldw, const float* bias, float* C, int ldc,
                              int M, int N, int epilogue, float* aux, int aux_col, float p0, float p1,
                              const int32_t* rows_dev, hos_stream_t stream) {
    if (!A0 || !W || M <= 0 || N <= 0 || K0 <= 0 || K1 < 0) return HOS_E_ARG;
    if (K1 > 0 && !A1) return HOS_E_ARG;
    if ((K0 % BK) || (K1 % BK)) return HOS_E_SHAPE;
    if ((lda0 & 3) || (ldw & 3) || (K1 > 0 && (lda1 & 3))) return HOS_E_ALIGN;
    if (!al16(A0) || !al16(W) || (K1 > 0 && !al16(A1))) return HOS_E_ALIGN;
    const bool aux_only = (epilogue == HOS_EPI_DENSITY);
    if (!aux_only && !C) return HOS_E_ARG;
    if ((epilogue == HOS_EPI_DENSITY || epilogue == HOS_EPI_NERF_HEAD) && !aux) return HOS_E_ARG;
    if (epilogue == HOS_EPI_DENSITY && N != 1) return HOS_E_SHAPE;
    if (epilogue == HOS_EPI_SIGMOID_RELU4 && N != 4) return HOS_E_SHAPE;
    if (epilogue == HOS_EPI_RESIDUAL && !aux) return HOS_E_ARG;
    GemmArgs a{};
    a.A0 = A0; a.lda0 = lda0; a.kt0 = K0 / BK; a.A1 = A1; a.lda1 = lda1;
    a.B = W; a.ldb = ldw; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.Mload = M; a.Nload = N;
    a.nk = (K0 + K1) / BK; a.kt_per_split = a.nk; a.red_limit = 0x7fffffff;
    a.bias = bias; a.aux = aux; a.aux_col = aux_col; a.p0 = p0; a.p1 = p1; a.epi = epilogue;
    a.m_dev = rows_dev;
    a.range_flag = (g_gemm_mode == HOS_GEMM_BF16X3) ? g_range_flag : nullptr;      // exact-fp32 mode has no fp16 operands
    if (epilogue == HOS_EPI_RESIDUAL) { a.mask = aux; a.ldmask = aux_col; }   // residual [M, ld=aux_col]
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N <= 32) {
        a.tiles_m = hos_cdiv(M, 128); a.tiles_n = hos_cdiv(N, 32);
        return launch<128, 32, MODE_FWD>(a, 1, s);
    }
    if (g_gemm_mode == HOS_GEMM_BF16X3) return hos_gemm3_launch(a, MODE_FWD, 1, s);
    a.tiles_m = hos_cdiv(M, 128); a.tiles_n = hos_cdiv(N, 128);
    return launch<128, 128, MODE_FWD>(a, 1, s);
}

// C[M, N] = A[M, K] . W[N, K]^T in exact fp32 MFMA for a SMALL output with a LONG reduction (the input gradient of the volume
// decoder's transposed convolutions: [M <= 4096, N <= 1024] from K = 1 728 .. 32 768, i.e. 8-64 output tiles): the reduction is
// split over ~256 workgroups that add their partial tiles into the zeroed C with fp32 atomics.  No bias / epilogue.
namespace {
struct SplitKPlan { int tiles_m, tiles_n, splits, kt_per_split; bool few; };
// 32-row tiles up to 64 rows and ~256 workgroups (round 3, scripts/bench_decoder.py: [64,512] from K = 16 384 39 -> 22 us,
// [512,256] 71 -> 63 us, [4096,256] from K = 1 728 57 -> 54 us; 128 or 1024 workgroups are 25-50 % slower)
inline SplitKPlan splitk_plan(int M, int N, int K) {
    static const int few_max = getenv("HOS_SPLITK_FEW_M") ? atoi(getenv("HOS_SPLITK_FEW_M")) : 64;
    static const int target = getenv("HOS_SPLITK_TARGET") ? atoi(getenv("HOS_SPLITK_TARGET")) : 256;
    SplitKPlan p;
    const int nk = K / BK;
    p.few = M <= few_max;
    p.tiles_m = hos_cdiv(M, p.few ? 32 : 128); p.tiles_n = hos_cdiv(N, 128);
    int splits = hos_cdiv(target, p.tiles_m * p.tiles_n);
    if (splits > nk / 4) splits = nk / 4 > 0 ? nk / 4 : 1;          // >= 4 K tiles per split
    p.kt_per_split = hos_cdiv(nk, splits);
    p.splits = hos_cdiv(nk, p.kt_per_split);
    return p;
}
// C[m][n] = act( sum_s ws[s][m][n] + bias[n] ), slabs added in index order (bit-reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, int ld,
                                                            const float* __restrict__ bias, int relu, float* __restrict__ C, int ldc) {
    const int groups = ld >> 2;
    const long total = (long)M * groups;
    const size_t slab = (size_t)M * ld;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / groups), n = (int)(i % groups) * 4;
        const float* p = ws + (size_t)m * ld + n;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j0 = 0; j0 < splits; j0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (j0 + u < splits) ? *reinterpret_cast<const float4*>(p + (size_t)(j0 + u) * slab) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        const float vals[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e < N) {
                float v = vals[e] + (bias != nullptr ? bias[n + e] : 0.f);
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)m * ldc + n + e] = v;
            }
        }
    }
}
int splitk_launch(const float* A, int lda, const float* W, int ldw, const float* bias, int relu, float* C, int ldc, int M, int N, int K,
                  float* ws, long long ws_floats, bool want_det, hipStream_t s) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return HOS_E_ARG;
    if (K % BK) return HOS_E_SHAPE;
    if ((lda & 3) || (ldw & 3) || !al16(A) || !al16(W)) return HOS_E_ALIGN;
    GemmArgs a{};
    a.A0 = A; a.lda0 = lda; a.kt0 = K / BK; a.B = W; a.ldb = ldw; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.Mload = M; a.Nload = N;
    a.nk = K / BK; a.red_limit = 0x7fffffff; a.epi = HOS_EPI_NONE;
    const SplitKPlan p = splitk_plan(M, N, K);
    a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.kt_per_split = p.kt_per_split;
    const int ld = (N + 3) & ~3;
    if (want_det) {
        if (p.splits > 1) {
            if (!ws || !al16(ws) || (long long)p.splits * M * ld > ws_floats) return HOS_E_ARG;
            a.aux = ws; a.aux_col = ld;
        } else {
            a.bias = bias; a.epi = relu ? HOS_EPI_RELU : HOS_EPI_NONE;        // one split: the tile kernel's own epilogue
        }
    } else if (p.splits > 1) {
        const int rc = zero2d(C, ldc, M, N, s);
        if (rc != 0) return rc;
    }
    const int rc = p.few ? launch<32, 128, MODE_FWD>(a, p.splits, s) : launch<128, 128, MODE_FWD>(a, p.splits, s);
    if (rc != 0 || !want_det || p.splits <= 1) return rc;
    const long total = (long)M * (ld >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, p.splits, M, N, ld, bias, relu, C, ldc);
    return hos_launch_status();
}
}  // namespace

extern "C" int hos_linear_fwd_splitk(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                                     hos_stream_t stream) {
    return splitk_launch(A, lda, W, ldw, nullptr, 0, C, ldc, M, N, K, nullptr, 0, false, static_cast<hipStream_t>(stream));
}

// The same split, bit-reproducible: the partial tiles go to `ws` (>= hos_linear_fwd_splitk_ws_floats(M, N, K) floats, 16-byte aligned,
// caller-owned scratch) with plain stores and a second launch adds them in slab order, then bias (may be NULL) and ReLU (relu != 0).
extern "C" long long hos_linear_fwd_splitk_ws_floats(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK)) return 0;
    const SplitKPlan p = splitk_plan(M, N, K);
    return p.splits > 1 ? (long long)p.splits * M * ((N + 3) & ~3) : 0;
}
extern "C" int hos_linear_fwd_splitk_det(const float* A, int lda, const float* W, int ldw, const float* bias, int relu, float* C, int ldc,
                                         int M, int N, int K, float* ws, long long ws_floats, hos_stream_t stream) {
    return splitk_launch(A, lda, W, ldw, bias, relu, C, ldc, M, N, K, ws, ws_floats, true, static_cast<hipStream_t>(stream));
}

extern "C" int hos_linear_dgrad(const float* dY, int lddy, const float* W, int ldw, int Npad,
                                const float* Xact, int ldx, float* dX, int lddx, int M, int K,
                                int accumulate, hos_stream_t stream) {
    if (!dY || !W || !dX || M <= 0 || K <= 0 || Npad <= 0) return HOS_E_ARG;
    if (Npad % BK) return HOS_E_SHAPE;
    if ((lddy & 3) || (ldw & 3) || (K & 3)) return HOS_E_ALIGN;
    if (!al16(dY) || !al16(W)) return HOS_E_ALIGN;
    GemmArgs a{};
    a.A0 = dY; a.lda0 = lddy; a.kt0 = Npad / BK;
    a.B = W; a.ldb = ldw; a.C = dX; a.ldc = lddx;
    a.M = M; a.N = K; a.Mload = M; a.Nload = K;
    a.nk = Npad / BK; a.kt_per_split = a.nk; a.red_limit = 0x7fffffff;
    a.mask = Xact; a.ldmask = ldx; a.accumulate = accumulate;
    if (g_gemm_mode == HOS_GEMM_BF16X3 && K > 32) return hos_gemm3_launch(a, MODE_DGRAD, 1, static_cast<hipStream_t>(stream));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // A handful of rows against a long weight stream (the volume decoder: 1..8 voxels x [1024, 32768] weights): 32-row tiles
    // put 3 workgroups on a CU instead of 2 (the 128-row tile streamed the weights at ~0.8 TB/s).  HOS_DGRAD_SPLIT=1 also
    // splits the reduction (atomics into a zeroed output: +1 % on a stage-2 step, but the result is no longer
    // bit-reproducible from call to call, so it is off by default).
    static const bool few_rows_split = getenv("HOS_DGRAD_SPLIT") && atoi(getenv("HOS_DGRAD_SPLIT")) == 1;
    // (up to 64 rows as two 32-row tiles: the decoder's [64, 512] x [512, 16384] layer 46 -> 20 us against one half-empty 128-row tile)
    static const int few_rows_max = getenv("HOS_FEWROW_M") ? atoi(getenv("HOS_FEWROW_M")) : 64;
    if (M <= few_rows_max && !accumulate && Xact == nullptr && a.nk >= 8) {
        a.tiles_m = hos_cdiv(M, 32); a.tiles_n = hos_cdiv(K, 128);
        int splits = 1;
        if (few_rows_split) {
            splits = a.nk / 4 < 4 ? a.nk / 4 : 4;
            if (a.tiles_n * splits < 512 && a.nk / 8 >= 2) splits = a.nk / 8 < 8 ? a.nk / 8 : 8;
        }
        a.kt_per_split = hos_cdiv(a.nk, splits);
        splits = hos_cdiv(a.nk, a.kt_per_split);
        if (splits > 1) {
            const int rc = zero2d(dX, lddx, M, K, s);
            if (rc != 0) return rc;
        }
        return launch<32, 128, MODE_DGRAD>(a, splits, s);
    }
    a.tiles_m = hos_cdiv(M, 128); a.tiles_n = hos_cdiv(K, 128);
    return launch<128, 128, MODE_DGRAD>(a, 1, s);
}

extern "C" int hos_linear_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, int ldw,
                                float* db, int M, int N, int K, int splits, hos_stream_t stream) {
    if (!dY || !X || !dW || M <= 0 || N <= 0 || K <= 0) return HOS_E_ARG;
    if ((lddy & 3) || (ldx & 3) || (K & 3)) return HOS_E_ALIGN;
    if (!al16(dY) || !al16(X)) return HOS_E_ALIGN;
    GemmArgs a{};
    a.A0 = dY; a.lda0 = lddy; a.B = X; a.ldb = ldx; a.C = dW; a.ldc = ldw;
    a.M = N; a.N = K;
    a.Mload = (N + 3) & ~3;          // dY is zero-padded to a multiple of 32 columns by contract
    if (a.Mload > lddy) a.Mload = lddy & ~3;
    a.Nload = K;
    a.nk = hos_cdiv(M, BK);          // rows >= M are zero-filled by the loaders (red_limit)
    a.red_limit = M;
    a.db = db;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool narrow = (N <= 32);
    if (!narrow && g_gemm_mode == HOS_GEMM_BF16X3) return hos_gemm3_launch(a, MODE_WGRAD, splits, s);
    a.tiles_m = hos_cdiv(N, narrow ? 32 : 128);
    a.tiles_n = hos_cdiv(K, 128);
    if (splits <= 0) {
        const int tiles = a.tiles_m * a.tiles_n;
        splits = hos_cdiv(1024, tiles);          // ~4 workgroups per CU
        if (splits > a.nk / 8) splits = a.nk / 8 > 0 ? a.nk / 8 : 1;   // >= 8 K tiles per split
    }
    if (splits > a.nk) splits = a.nk;
    a.kt_per_split = hos_cdiv(a.nk, splits);
    splits = hos_cdiv(a.nk, a.kt_per_split);
    if (narrow) return launch<32, 128, MODE_WGRAD>(a, splits, s);
    return launch<128, 128, MODE_WGRAD>(a, splits, s);
}
