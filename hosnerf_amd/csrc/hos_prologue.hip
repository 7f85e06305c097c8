// Per-frame prologue of the human-object branch on the device: pose refinement and motion bases
// for F frames (the current frame and, for the flow set, the previous one) in two launches forward and two backward,
// instead of ~120 tiny torch launches (7 nn.Linear on one row, Rodrigues, 25 chained 4x4 products, torch.inverse, and
// their autograd nodes).  Everything here is latency-bound bookkeeping on 26 joints; one workgroup per frame.
//
//   hos_pose_refine_{fwd,bwd}    BodyPoseRefiner (pose_decoders/mlp_delta_body_pose.py:14-73, mlp_depth 4, width 256) +
//                                RodriguesModule (U:66-92) + the composition of N:589-605: R_i <- R_i dR_i, T_i <- T_i + dT_i
//                                for the non-root joints.
//   hos_motion_basis_{fwd,bwd}   MotionBasisComputer.forward (U:134-174): kinematic chain over the SMPL tree (U:100-103),
//                                backward bases G_cnl G_dst^-1 and forward bases G_dst G_cnl^-1.  The reference inverts the
//                                4x4 matrices with torch.inverse (LU); here the affine inverse is closed form
//                                ([A|t]^-1 = [A^-1 | -A^-1 t], A^-1 = adj(A)/det(A)) -- the same function, no pivoting,
//                                no library call, no host synchronisation (torch.inverse blocks hipGraph capture).
//   (U = 3rd_Complete_HOSNeRF/core/utils/network_util.py, N = .../core/nets/human_nerf/network.py)
#include "hos_common.h"

namespace {

constexpr int PW = 256;            // pose_decoder.mlp_width
constexpr int PE = 75;             // pose_decoder.embedding_size = 3 * (K - 1)
constexpr int KJ = 26;             // total_bones
constexpr int PT = 1024;           // threads per block
constexpr int SAVED = 5 * PW + 80; // h1 h2 h3 yR yT rvec(+pad)

__constant__ int c_parent[KJ] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21, 23, 22};   // U:100-103

struct PoseW {     // device pointers, order of the C ABI
    const float* w[14];
};
struct PoseG {
    float* g[14];
};
// index: 0 W0 [256,75]  1 b0   2 W2 [256,256]  3 b2   4 W4  5 b4   6 WR0  7 bR0   8 WR2 [75,256]  9 bR2   10 WT0  11 bT0   12 WT2  13 bT2

// out[n] = act(W[n,:] . in + b[n]),  W row-major [N,Kd]; one wave per row, lanes over k (coalesced), 16 waves stride the rows.
// The prologue is pure latency: a wave issues the loads of ALL its rows before it reduces any of them.
template <int KD>
__device__ __forceinline__ void matvec(const float* __restrict__ W, const float* __restrict__ b, const float* in, float* out,
                                       int N, bool relu, int tid) {
    constexpr int NW = PT / 64, RMAX = (PW + NW - 1) / NW, PER = (KD + 63) / 64;
    const int wave = tid >> 6, lane = tid & 63;
    float w[RMAX][PER];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int n = wave + r * NW;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int k = lane + 64 * j;
            w[r][j] = (n < N && k < KD) ? W[(size_t)n * KD + k] : 0.f;
        }
    }
    float x[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) x[j] = (lane + 64 * j < KD) ? in[lane + 64 * j] : 0.f;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int n = wave + r * NW;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < PER; ++j) acc += w[r][j] * x[j];
        acc = wave_sum(acc);
        if (lane == 0 && n < N) {
            acc += b[n];
            out[n] = relu ? fmaxf(acc, 0.f) : acc;
        }
    }
}

__device__ __forceinline__ void rodrigues(const float* r, float* R) {
    // U:76-92: theta = sqrt(1e-5 + |r|^2); axis = r / theta
    const float th = sqrtf(1e-5f + (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
    const float x = r[0] / th, y = r[1] / th, z = r[2] / th;
    const float c = cosf(th), s = sinf(th);
    R[0] = x * x + (1.f - x * x) * c; R[1] = x * y * (1.f - c) - z * s; R[2] = x * z * (1.f - c) + y * s;
    R[3] = x * y * (1.f - c) + z * s; R[4] = y * y + (1.f - y * y) * c; R[5] = y * z * (1.f - c) - x * s;
    R[6] = x * z * (1.f - c) - y * s; R[7] = y * z * (1.f - c) + x * s; R[8] = z * z + (1.f - z * z) * c;
}

// gradient of a scalar w.r.t. r given its gradient g[9] w.r.t. the rotation entries
__device__ __forceinline__ void rodrigues_bwd(const float* r, const float* g, float* gr) {
    const float n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    const float th = sqrtf(1e-5f + n2);
    const float x = r[0] / th, y = r[1] / th, z = r[2] / th;
    const float c = cosf(th), s = sinf(th), oc = 1.f - c;
    const float s01 = g[1] + g[3], s02 = g[2] + g[6], s12 = g[5] + g[7];
    const float gx = g[0] * 2.f * x * oc + s01 * y * oc + s02 * z * oc + (g[7] - g[5]) * s;
    const float gy = g[4] * 2.f * y * oc + s01 * x * oc + s12 * z * oc + (g[2] - g[6]) * s;
    const float gz = g[8] * 2.f * z * oc + s02 * x * oc + s12 * y * oc + (g[3] - g[1]) * s;
    const float gc = g[0] * (1.f - x * x) + g[4] * (1.f - y * y) + g[8] * (1.f - z * z) - s01 * x * y - s02 * x * z - s12 * y * z;
    const float gs = -g[1] * z + g[2] * y + g[3] * z - g[5] * x - g[6] * y + g[7] * x;
    float gth = -gc * s + gs * c;                               // through cos / sin
    gth += -(gx * r[0] + gy * r[1] + gz * r[2]) / (th * th);    // through axis = r / theta
    const float k = gth / th;                                   // d theta / d r = r / theta
    gr[0] = gx / th + k * r[0];
    gr[1] = gy / th + k * r[1];
    gr[2] = gz / th + k * r[2];
}

__device__ __forceinline__ void mm3(const float* A, const float* B, float* C) {          // C = A B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void mtm3(const float* A, const float* B, float* C) {         // C = A^T B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
__device__ __forceinline__ void mmt3(const float* A, const float* B, float* C) {         // C = A B^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
__device__ __forceinline__ void mv3(const float* A, const float* v, float* o) {          // o = A v
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
__device__ __forceinline__ void mtv3(const float* A, const float* v, float* o) {         // o = A^T v
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}
__device__ __forceinline__ void inv3(const float* A, float* I) {
    const float c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const float det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    const float id = 1.f / det;
    I[0] = c00 * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c01 * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c02 * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// ------------------------------------------------------------------------------------------------ P2 forward
__global__ __launch_bounds__(PT) void pose_refine_fwd_kernel(const float* posevec, const float* Rs, const float* Ts, PoseW pw, int K,
                                                            float* Rs_out, float* Ts_out, float* saved) {
    __shared__ float in[PW], h1[PW], h2[PW], h3[PW], yR[PW], yT[PW], rv[80], dT[80];
    const int f = blockIdx.x, tid = threadIdx.x;
    if (tid < PE) in[tid] = posevec[f * PE + tid];
    __syncthreads();
    matvec<PE>(pw.w[0], pw.w[1], in, h1, PW, true, tid);
    __syncthreads();
    matvec<PW>(pw.w[2], pw.w[3], h1, h2, PW, true, tid);
    __syncthreads();
    matvec<PW>(pw.w[4], pw.w[5], h2, h3, PW, true, tid);
    __syncthreads();
    matvec<PW>(pw.w[6], pw.w[7], h3, yR, PW, true, tid);
    matvec<PW>(pw.w[10], pw.w[11], h3, yT, PW, true, tid);
    __syncthreads();
    matvec<PW>(pw.w[8], pw.w[9], yR, rv, PE, false, tid);
    matvec<PW>(pw.w[12], pw.w[13], yT, dT, PE, false, tid);
    __syncthreads();
    float* sv = saved + (size_t)f * SAVED;
    if (tid < PW) {
        sv[tid] = h1[tid]; sv[PW + tid] = h2[tid]; sv[2 * PW + tid] = h3[tid]; sv[3 * PW + tid] = yR[tid]; sv[4 * PW + tid] = yT[tid];
    }
    if (tid < PE) sv[5 * PW + tid] = rv[tid];
    if (tid < K) {
        const float* R = Rs + ((size_t)f * K + tid) * 9;
        const float* T = Ts + ((size_t)f * K + tid) * 3;
        float* Ro = Rs_out + ((size_t)f * K + tid) * 9;
        float* To = Ts_out + ((size_t)f * K + tid) * 3;
        if (tid == 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e) Ro[e] = R[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) To[e] = T[e];
        } else {
            float dR[9], C[9];
            rodrigues(rv + 3 * (tid - 1), dR);
            mm3(R, dR, C);                                                       // N:595-600
#pragma unroll
            for (int e = 0; e < 9; ++e) Ro[e] = C[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) To[e] = T[e] + dT[3 * (tid - 1) + e];    // N:602-603
        }
    }
}

// out[k] = sum_n W[n,k] g[n] (W row-major [N,Kd], Kd <= 256); 1024 threads = 4 row groups x 256 columns, partials through LDS.
// Each thread's loads are independent (unrolled), so they are all in flight together.
__device__ __forceinline__ void matvec_t(const float* __restrict__ W, const float* g, float* out, float* part, int N, int Kd, int tid,
                                         bool accumulate) {
    const int grp = tid >> 8, k = tid & 255;
    float acc = 0.f;
    if (k < Kd) {
#pragma unroll 16
        for (int n = grp; n < N; n += 4) acc += W[(size_t)n * Kd + k] * g[n];
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < Kd) {
        const float v = part[tid] + part[256 + tid] + part[512 + tid] + part[768 + tid];
        out[tid] = accumulate ? out[tid] + v : v;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ P2 backward
// Pass 1 (one workgroup per frame): the chain of per-layer output gradients.  gvec[f] = {g_rv[80], g_dT[80], g_yR', g_yT', g_h3', g_h2',
// g_h1'} (primes: already masked by the ReLU of that layer's output), 5 x 256 + 160 floats.
constexpr int GVEC = 5 * PW + 160;

__global__ __launch_bounds__(PT) void pose_refine_bwd_vec_kernel(const float* gRs_out, const float* gTs_out, const float* Rs,
                                                                const float* saved, PoseW pw, int K, float* gvec) {
    __shared__ float h1[PW], h2[PW], h3[PW], yR[PW], yT[PW], g_rv[80], g_dT[80];
    __shared__ float gA[PW], gB[PW], gC[PW], part[PT];
    const int tid = threadIdx.x, f = blockIdx.x;
    const float* sv = saved + (size_t)f * SAVED;
    float* gv = gvec + (size_t)f * GVEC;
    if (tid < PW) {
        h1[tid] = sv[tid]; h2[tid] = sv[PW + tid]; h3[tid] = sv[2 * PW + tid]; yR[tid] = sv[3 * PW + tid]; yT[tid] = sv[4 * PW + tid];
    }
    if (tid < 80) { g_rv[tid] = 0.f; g_dT[tid] = 0.f; }
    __syncthreads();
    if (tid >= 1 && tid < K) {
        const float* R = Rs + ((size_t)f * K + tid) * 9;
        const float* gRo = gRs_out + ((size_t)f * K + tid) * 9;
        float gdR[9], gr[3];
        mtm3(R, gRo, gdR);                                   // R_out = R dR  ->  g_dR = R^T g_Rout
        rodrigues_bwd(sv + 5 * PW + 3 * (tid - 1), gdR, gr);
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            g_rv[3 * (tid - 1) + e] = gr[e];
            g_dT[3 * (tid - 1) + e] = gTs_out[((size_t)f * K + tid) * 3 + e];
        }
    }
    __syncthreads();
    if (tid < 80) { gv[tid] = g_rv[tid]; gv[80 + tid] = g_dT[tid]; }
    matvec_t(pw.w[8], g_rv, gA, part, PE, PW, tid, false);          // g_yR
    matvec_t(pw.w[12], g_dT, gB, part, PE, PW, tid, false);         // g_yT
    if (tid < PW) {
        gA[tid] = yR[tid] > 0.f ? gA[tid] : 0.f;
        gB[tid] = yT[tid] > 0.f ? gB[tid] : 0.f;
        gv[160 + tid] = gA[tid];
        gv[160 + PW + tid] = gB[tid];
    }
    __syncthreads();
    matvec_t(pw.w[6], gA, gC, part, PW, PW, tid, false);
    matvec_t(pw.w[10], gB, gC, part, PW, PW, tid, true);            // g_h3
    if (tid < PW) { gC[tid] = h3[tid] > 0.f ? gC[tid] : 0.f; gv[160 + 2 * PW + tid] = gC[tid]; }
    __syncthreads();
    matvec_t(pw.w[4], gC, gA, part, PW, PW, tid, false);            // g_h2
    if (tid < PW) { gA[tid] = h2[tid] > 0.f ? gA[tid] : 0.f; gv[160 + 3 * PW + tid] = gA[tid]; }
    __syncthreads();
    matvec_t(pw.w[2], gA, gB, part, PW, PW, tid, false);            // g_h1
    if (tid < PW) gv[160 + 4 * PW + tid] = h1[tid] > 0.f ? gB[tid] : 0.f;
}

// Pass 2 (grid over all weight elements): gW[n,k] += sum_f g_f[n] in_f[k], gb[n] += sum_f g_f[n] for the seven layers.
struct OuterJob { int N, Kd, g_off, in_off, in_is_pose; };     // offsets into gvec / saved rows
__constant__ OuterJob c_jobs[7] = {
    {PW, PE, 160 + 4 * PW, 0, 1},          // block_mlps.0      g_h1' (x) posevec
    {PW, PW, 160 + 3 * PW, 0, 0},          // block_mlps.2      g_h2' (x) h1
    {PW, PW, 160 + 2 * PW, PW, 0},         // block_mlps.4      g_h3' (x) h2
    {PW, PW, 160, 2 * PW, 0},              // block_mlps_dstR.0 g_yR' (x) h3
    {PE, PW, 0, 3 * PW, 0},                // block_mlps_dstR.2 g_rv  (x) yR
    {PW, PW, 160 + PW, 2 * PW, 0},         // block_mlps_dstT.0 g_yT' (x) h3
    {PE, PW, 80, 4 * PW, 0},               // block_mlps_dstT.2 g_dT  (x) yT
};

__global__ __launch_bounds__(256) void pose_refine_bwd_outer_kernel(const float* gvec, const float* saved, const float* posevec, PoseG pg,
                                                                   int F) {
    const int job = blockIdx.y;
    const OuterJob j = c_jobs[job];
    const int e = blockIdx.x * 256 + threadIdx.x;
    float* gW = pg.g[2 * job];
    float* gb = pg.g[2 * job + 1];
    if (e < j.N * j.Kd) {
        const int n = e / j.Kd, k = e % j.Kd;
        float acc = 0.f;
        for (int f = 0; f < F; ++f) {
            const float x = j.in_is_pose ? posevec[f * PE + k] : saved[(size_t)f * SAVED + j.in_off + k];
            acc += gvec[(size_t)f * GVEC + j.g_off + n] * x;
        }
        gW[e] += acc;
    }
    if (e < j.N) {
        float acc = 0.f;
        for (int f = 0; f < F; ++f) acc += gvec[(size_t)f * GVEC + j.g_off + e];
        gb[e] += acc;
    }
}

// ------------------------------------------------------------------------------------------------ P3
struct Chain {
    float A[KJ][9];
    float t[KJ][3];
};

__device__ void chain_forward(const float* Rs, const float* Ts, int K, Chain& c) {        // U:146-160, serial over the tree
    for (int e = 0; e < 9; ++e) c.A[0][e] = Rs[e];
    for (int e = 0; e < 3; ++e) c.t[0][e] = Ts[e];
    for (int i = 1; i < K; ++i) {
        const int p = c_parent[i];
        mm3(c.A[p], Rs + 9 * i, c.A[i]);
        float v[3];
        mv3(c.A[p], Ts + 3 * i, v);
        for (int e = 0; e < 3; ++e) c.t[i][e] = v[e] + c.t[p][e];
    }
}

__global__ __launch_bounds__(64) void motion_basis_fwd_kernel(const float* Rs, const float* Ts, const float* cnl, int K,
                                                             float* R_b, float* T_b, float* R_f, float* T_f) {
    __shared__ Chain c;
    const int f = blockIdx.x, i = threadIdx.x;
    if (i == 0) chain_forward(Rs + (size_t)f * K * 9, Ts + (size_t)f * K * 3, K, c);
    __syncthreads();
    if (i >= K) return;
    const float* G = cnl + 16 * i;                            // canonical transform, row-major 4x4
    const float Ac[9] = {G[0], G[1], G[2], G[4], G[5], G[6], G[8], G[9], G[10]};
    const float tc[3] = {G[3], G[7], G[11]};
    float Ai[9], ti[3], Aci[9], tci[3], R[9], v[3];
    inv3(c.A[i], Ai);
    mv3(Ai, c.t[i], ti);
    inv3(Ac, Aci);
    mv3(Aci, tc, tci);
    const size_t o = (size_t)f * K + i;
    mm3(Ac, Ai, R);                                           // backward basis: G_cnl G_dst^-1   (U:162-166)
    mv3(Ac, ti, v);
    for (int e = 0; e < 9; ++e) R_b[o * 9 + e] = R[e];
    for (int e = 0; e < 3; ++e) T_b[o * 3 + e] = tc[e] - v[e];
    mm3(c.A[i], Aci, R);                                      // forward basis: G_dst G_cnl^-1    (U:168-172)
    mv3(c.A[i], tci, v);
    for (int e = 0; e < 9; ++e) R_f[o * 9 + e] = R[e];
    for (int e = 0; e < 3; ++e) T_f[o * 3 + e] = c.t[i][e] - v[e];
}

__global__ __launch_bounds__(64) void motion_basis_bwd_kernel(const float* gR_b, const float* gT_b, const float* gR_f, const float* gT_f,
                                                             const float* Rs, const float* Ts, const float* cnl, int K,
                                                             float* gRs, float* gTs) {
    __shared__ Chain c;
    __shared__ float gA[KJ][9], gt[KJ][3];
    const int f = blockIdx.x, i = threadIdx.x;
    const float* Rf = Rs + (size_t)f * K * 9;
    const float* Tf = Ts + (size_t)f * K * 3;
    if (i == 0) chain_forward(Rf, Tf, K, c);
    __syncthreads();
    if (i < K) {
        const float* G = cnl + 16 * i;
        const float Ac[9] = {G[0], G[1], G[2], G[4], G[5], G[6], G[8], G[9], G[10]};
        const float tc[3] = {G[3], G[7], G[11]};
        float Ai[9], Aci[9], tci[3];
        inv3(c.A[i], Ai);
        inv3(Ac, Aci);
        mv3(Aci, tc, tci);
        const size_t o = (size_t)f * K + i;
        float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tt[3] = {0, 0, 0};
        if (gR_b || gT_b) {
            // R_b = Ac Ai, T_b = tc - Ac (Ai t)
            float gAi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gti[3] = {0, 0, 0};     // gti: gradient w.r.t. u = Ai t
            if (gR_b) mtm3(Ac, gR_b + o * 9, gAi);
            if (gT_b) {
                float q[3];
                mtv3(Ac, gT_b + o * 3, q);
                for (int e = 0; e < 3; ++e) gti[e] = -q[e];
            }
            // u = Ai t
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) gAi[3 * r + cc] += gti[r] * c.t[i][cc];
            mtv3(Ai, gti, tt);
            // Ai = inv(A): gA = -Ai^T gAi Ai^T
            float m1[9];
            mtm3(Ai, gAi, m1);
            mmt3(m1, Ai, a);
            for (int e = 0; e < 9; ++e) a[e] = -a[e];
        }
        if (gR_f) {
            float m[9];
            mmt3(gR_f + o * 9, Aci, m);                        // R_f = A Aci
            for (int e = 0; e < 9; ++e) a[e] += m[e];
        }
        if (gT_f) {                                            // T_f = t - A tci
            for (int r = 0; r < 3; ++r) {
                for (int cc = 0; cc < 3; ++cc) a[3 * r + cc] -= gT_f[o * 3 + r] * tci[cc];
                tt[r] += gT_f[o * 3 + r];
            }
        }
        for (int e = 0; e < 9; ++e) gA[i][e] = a[e];
        for (int e = 0; e < 3; ++e) gt[i][e] = tt[e];
    }
    __syncthreads();
    if (i != 0) return;
    // chain backward, children before parents (every parent index is below its child's)
    for (int j = K - 1; j >= 1; --j) {
        const int p = c_parent[j];
        float r[9], v[3], m[9];
        mtm3(c.A[p], gA[j], r);                                // A_j = A_p R_j
        mtv3(c.A[p], gt[j], v);                                // t_j = A_p T_j + t_p
        for (int e = 0; e < 9; ++e) gRs[((size_t)f * K + j) * 9 + e] = r[e];
        for (int e = 0; e < 3; ++e) gTs[((size_t)f * K + j) * 3 + e] = v[e];
        mmt3(gA[j], Rf + 9 * j, m);
        for (int rr = 0; rr < 3; ++rr)
            for (int cc = 0; cc < 3; ++cc) gA[p][3 * rr + cc] += m[3 * rr + cc] + gt[j][rr] * Tf[3 * j + cc];
        for (int e = 0; e < 3; ++e) gt[p][e] += gt[j][e];
    }
    for (int e = 0; e < 9; ++e) gRs[(size_t)f * K * 9 + e] = gA[0][e];
    for (int e = 0; e < 3; ++e) gTs[(size_t)f * K * 3 + e] = gt[0][e];
}

}  // namespace

extern "C" long long hos_pose_refine_saved_floats(void) { return SAVED; }
extern "C" long long hos_pose_refine_workspace_floats(void) { return GVEC; }      // per frame

extern "C" int hos_pose_refine_fwd(const float* posevec, const float* Rs, const float* Ts, const float* const* weights14,
                                   int F, int K, int width, float* Rs_out, float* Ts_out, float* saved, hos_stream_t stream) {
    if (!posevec || !Rs || !Ts || !weights14 || !Rs_out || !Ts_out || !saved || F <= 0) return HOS_E_ARG;
    if (K != KJ || width != PW) return HOS_E_SHAPE;
    PoseW pw;
    for (int i = 0; i < 14; ++i) {
        if (!weights14[i]) return HOS_E_ARG;
        pw.w[i] = weights14[i];
    }
    hipLaunchKernelGGL(pose_refine_fwd_kernel, dim3(F), dim3(PT), 0, static_cast<hipStream_t>(stream), posevec, Rs, Ts, pw, K,
                       Rs_out, Ts_out, saved);
    return hos_launch_status();
}

extern "C" int hos_pose_refine_bwd(const float* g_Rs_out, const float* g_Ts_out, const float* posevec, const float* Rs,
                                   const float* saved, const float* const* weights14, float* const* grads14, int F, int K, int width,
                                   float* workspace, hos_stream_t stream) {
    if (!g_Rs_out || !g_Ts_out || !posevec || !Rs || !saved || !weights14 || !grads14 || !workspace || F <= 0) return HOS_E_ARG;
    if (K != KJ || width != PW) return HOS_E_SHAPE;
    PoseW pw;
    PoseG pg;
    for (int i = 0; i < 14; ++i) {
        if (!weights14[i] || !grads14[i]) return HOS_E_ARG;
        pw.w[i] = weights14[i];
        pg.g[i] = grads14[i];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pose_refine_bwd_vec_kernel, dim3(F), dim3(PT), 0, s, g_Rs_out, g_Ts_out, Rs, saved, pw, K, workspace);
    hipLaunchKernelGGL(pose_refine_bwd_outer_kernel, dim3(PW * PW / 256, 7), dim3(256), 0, s, workspace, saved, posevec, pg, F);
    return hos_launch_status();
}

extern "C" int hos_motion_basis_fwd(const float* dst_Rs, const float* dst_Ts, const float* cnl_gtfms, int F, int K,
                                    float* R_bwd, float* T_bwd, float* R_fwd, float* T_fwd, hos_stream_t stream) {
    if (!dst_Rs || !dst_Ts || !cnl_gtfms || !R_bwd || !T_bwd || !R_fwd || !T_fwd || F <= 0) return HOS_E_ARG;
    if (K != KJ) return HOS_E_SHAPE;
    hipLaunchKernelGGL(motion_basis_fwd_kernel, dim3(F), dim3(64), 0, static_cast<hipStream_t>(stream), dst_Rs, dst_Ts, cnl_gtfms, K,
                       R_bwd, T_bwd, R_fwd, T_fwd);
    return hos_launch_status();
}

extern "C" int hos_motion_basis_bwd(const float* g_R_bwd, const float* g_T_bwd, const float* g_R_fwd, const float* g_T_fwd,
                                    const float* dst_Rs, const float* dst_Ts, const float* cnl_gtfms, int F, int K,
                                    float* g_dst_Rs, float* g_dst_Ts, hos_stream_t stream) {
    if (!dst_Rs || !dst_Ts || !cnl_gtfms || !g_dst_Rs || !g_dst_Ts || F <= 0) return HOS_E_ARG;
    if (K != KJ) return HOS_E_SHAPE;
    hipLaunchKernelGGL(motion_basis_bwd_kernel, dim3(F), dim3(64), 0, static_cast<hipStream_t>(stream), g_R_bwd, g_T_bwd, g_R_fwd,
                       g_T_fwd, dst_Rs, dst_Ts, cnl_gtfms, K, g_dst_Rs, g_dst_Ts);
    return hos_launch_status();
}
