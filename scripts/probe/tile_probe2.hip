// Like tile_probe, but at the GEMM's occupancy (1 workgroup of 512 threads per CU, forced by a 128 KB LDS request)
// and with a selectable number of K tiles in flight (DEPTH) -- is the GEMM's operand ingest latency-bound?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int DEPTH, bool SYNC>
__global__ __launch_bounds__(512) void tile_kernel(const float* __restrict__ A, const float* __restrict__ W, int K, int pitch,
                                                   int useA, int useW, int tiles_n, int xcd_remap, float* sink) {
    extern __shared__ float lds[];
    const int t = threadIdx.x;
    int b = blockIdx.x;
    if (xcd_remap) { const int nb = gridDim.x, q = nb >> 3, x = b & 7, y = b >> 3; b = x * q + y; }
    const int tm = b / tiles_n, tn = b % tiles_n;
    const int r = t >> 3, c4 = (t & 7) * 4;
    float acc = 0.f;
    float4 va[DEPTH][4], vw[DEPTH][4];
    const int nk = K / 32;
    auto issue = [&](int kt, int slot) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = r + p * 64;
            if (useA) va[slot][p] = *(const float4*)(A + (size_t)(tm * 256 + row) * pitch + kt * 32 + c4);
            if (useW) vw[slot][p] = *(const float4*)(W + (size_t)(tn * 256 + row) * pitch + kt * 32 + c4);
        }
    };
    auto consume = [&](int slot) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (useA) acc += va[slot][p].x + va[slot][p].y + va[slot][p].z + va[slot][p].w;
            if (useW) acc += vw[slot][p].x + vw[slot][p].y + vw[slot][p].z + vw[slot][p].w;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
    for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kt = kt0 + d;
            if (kt + DEPTH - 1 < nk) issue(kt + DEPTH - 1, (d + DEPTH - 1) % DEPTH);
            consume(d);
            if (SYNC) __syncthreads();
        }
    }
    if (acc == 123.456f) { sink[0] = acc; lds[t] = acc; }
}

template <int DEPTH, bool SYNC>
void bench(const char* tag, const float* A, const float* W, int M, int K, int tiles_n, int lds_bytes, int remap, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel<DEPTH, SYNC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode : {3, 1, 2}) {
        const int useA = mode & 1, useW = (mode >> 1) & 1;
        const int blocks = (M / 256) * tiles_n;
        auto launch = [&]() { hipLaunchKernelGGL((tile_kernel<DEPTH, SYNC>), dim3(blocks), dim3(512), lds_bytes, 0, A, W, K, K, useA, useW, tiles_n, remap, sink); };
        launch(); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double s = ms / 5 * 1e-3;
        const double bytes = (double)blocks * 256 * K * 4 * (useA + useW);
        printf("%-22s depth=%d sync=%d lds=%3dKB remap=%d load=%s%s  %8.1f us %8.2f TB/s %8.2f B/clk/CU\n", tag, DEPTH, (int)SYNC, lds_bytes >> 10, remap,
               useA ? "A" : "", useW ? "W" : "", s * 1e6, bytes / s / 1e12, bytes / s / 2.4e9 / 256);
    }
}

int main() {
    const int M = 32768, N = 1024, K = 1024;
    float *A, *W, *sink;
    hipMalloc(&A, (size_t)M * K * 4); hipMemset(A, 0, (size_t)M * K * 4);
    hipMalloc(&W, (size_t)N * K * 4); hipMemset(W, 0, (size_t)N * K * 4);
    hipMalloc(&sink, 4);
    for (int remap : {0, 1}) {
        bench<1, false>("occ=max", A, W, M, K, 4, 0, remap, sink);
        bench<1, false>("occ=2/CU", A, W, M, K, 4, 72 * 1024, remap, sink);
        bench<1, false>("occ=1/CU", A, W, M, K, 4, 128 * 1024, remap, sink);
        bench<1, true>("occ=1/CU", A, W, M, K, 4, 128 * 1024, remap, sink);
        bench<2, false>("occ=1/CU", A, W, M, K, 4, 128 * 1024, remap, sink);
        bench<2, true>("occ=1/CU", A, W, M, K, 4, 128 * 1024, remap, sink);
        bench<4, false>("occ=1/CU", A, W, M, K, 4, 128 * 1024, remap, sink);
        bench<4, true>("occ=1/CU", A, W, M, K, 4, 128 * 1024, remap, sink);
    }
    return 0;
}
