// global_load_lds (LDS-DMA) ingest probe at GEMM occupancy: 512 threads, 128 KB LDS, double-buffered 64 KB stages,
// one barrier per K tile.  seg = contiguous bytes per row segment (128: fp32 tile rows, 64: 16-bit plane rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// operand bytes per K tile per block: 2 operands x 32 KB.  Each wave issues 8 DMA instrs of 1 KB.
template <int SEG>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ A, const char* __restrict__ W, int nk, size_t pitchB, int useA, int useW,
                                                  int tiles_n, int plain, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int b = blockIdx.x;
    { const int nb = gridDim.x, q = nb >> 3, x = b & 7, y = b >> 3; b = x * q + y; }
    const int tm = b / tiles_n, tn = b % tiles_n;
    constexpr int LPR = SEG / 16;              // lanes per row segment
    constexpr int RPI = 64 / LPR;              // rows per instruction
    constexpr int ROWS = 32768 / SEG;          // rows per 32 KB operand tile (256 for fp32 rows, 512 = 256 rows x 2 planes for 64 B)
    // wave w: operand = w>>2 (0 A, 1 W), instr q in 0..7 covers rows (w&3)*ROWS/4 + q*RPI ...
    const int op = wave >> 2;
    const char* P = op ? W : A;
    const int rowbase = (op ? tn : tm) * ROWS + (wave & 3) * (ROWS / 4);
    const bool on = op ? useW : useA;
    float acc = 0.f;
    float4 v[8];
    for (int kt = 0; kt < nk; ++kt) {
        char* stage = lds + (kt & 1) * 65536 + op * 32768 + (wave & 3) * 8192;
        if (on) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = rowbase + q * RPI + lane / LPR;
                const char* src = P + (size_t)row * pitchB + (size_t)kt * SEG + (lane % LPR) * 16;
                if (plain) v[q] = *(const float4*)src;
                else dma16(src, stage + q * 1024);
            }
            if (plain) {
#pragma unroll
                for (int q = 0; q < 8; ++q) *(float4*)(stage + q * 1024 + lane * 16) = v[q];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *(const float*)(lds + (kt & 1) * 65536 + t * 4);
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int M = 32768, N = 1024, K = 1024;
    char *A, *W; float* sink;
    const size_t szA = (size_t)M * (K + 64) * 4;
    hipMalloc(&A, szA); hipMemset(A, 0, szA);
    hipMalloc(&W, (size_t)N * (K + 64) * 4); hipMemset(W, 0, (size_t)N * (K + 64) * 4);
    hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int seg : {128, 64}) for (int plain : {0, 1}) for (int mode : {3, 1, 2}) {
        const int useA = mode & 1, useW = mode >> 1;
        const int blocks = (M / 256) * 4;
        const size_t pitch = seg == 128 ? (size_t)K * 4 : (size_t)K * 2;     // fp32 rows / 16-bit plane rows (hi and lo = 2x rows)
        auto launch = [&]() {
            if (seg == 128) hipLaunchKernelGGL(dma_kernel<128>, dim3(blocks), dim3(512), 131072, 0, A, W, K / 32, pitch, useA, useW, 4, plain, sink);
            else hipLaunchKernelGGL(dma_kernel<64>, dim3(blocks), dim3(512), 131072, 0, A, W, K / 32, pitch, useA, useW, 4, plain, sink);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double s = ms / 5 * 1e-3, bytes = (double)blocks * (K / 32) * 32768.0 * (useA + useW);
        printf("seg=%3dB %s load=%s%s  %8.1f us %8.2f TB/s %8.2f B/clk/CU\n", seg, plain ? "regs+ds_write" : "global_load_lds", useA ? "A" : "", useW ? "W" : "",
               s * 1e6, bytes / s / 1e12, bytes / s / 2.4e9 / 256);
    }
    return 0;
}
