// GEMM-shaped ingest probe: every block marches 32-float K tiles over a [256 x K] fp32 A panel (private, HBM-streamed)
// and a [256 x K] W panel (one of N/256 panels, shared by all blocks), with selectable row pitch / blocked layout.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probe/tile_probe.hip -o build/tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// layout 0: row-major pitch (floats); layout 1: k-tile-major blocked [k/32][row][32]
__global__ __launch_bounds__(512) void tile_kernel(const float* __restrict__ A, const float* __restrict__ W, int K, int pitch, int layout,
                                                   int rowsA_total, int rowsW_total, int useA, int useW, int tiles_n, float* sink) {
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    const int tm = b / tiles_n, tn = b % tiles_n;
    const int r = t >> 3, c4 = (t & 7) * 4;               // 64 rows x 8 float4 per pass; 4 passes = 256 rows
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 32) {
        float4 va[4], vw[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = r + p * 64;
            size_t offA, offW;
            if (layout == 0) {
                offA = (size_t)(tm * 256 + row) * pitch + k0 + c4;
                offW = (size_t)(tn * 256 + row) * pitch + k0 + c4;
            } else {
                offA = ((size_t)(k0 >> 5) * rowsA_total + tm * 256 + row) * 32 + c4;
                offW = ((size_t)(k0 >> 5) * rowsW_total + tn * 256 + row) * 32 + c4;
            }
            if (useA) va[p] = *(const float4*)(A + offA);
            if (useW) vw[p] = *(const float4*)(W + offW);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (useA) acc += va[p].x + va[p].y + va[p].z + va[p].w;
            if (useW) acc += vw[p].x + vw[p].y + vw[p].z + vw[p].w;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int M = 32768, N = 1024, K = 1024;
    const int maxpitch = K + 64;
    float *A, *W, *sink;
    hipMalloc(&A, (size_t)M * maxpitch * 4); hipMemset(A, 0, (size_t)M * maxpitch * 4);
    hipMalloc(&W, (size_t)N * maxpitch * 4); hipMemset(W, 0, (size_t)N * maxpitch * 4);
    hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("%-60s %8s %10s %10s\n", "case (M=32768 N=1024 K=1024, 512 blocks of 256x256)", "us", "TB/s(L1)", "B/clk/CU");
    for (int tiles_n : {4, 1}) {
        for (int layout : {0, 1}) {
            for (int pitch : {1024, 1056, 1040}) {
                if (layout == 1 && pitch != 1024) continue;
                for (int mode : {3, 1, 2}) {
                    const int useA = mode & 1, useW = (mode >> 1) & 1;
                    const int blocks = (M / 256) * tiles_n;
                    auto launch = [&]() {
                        hipLaunchKernelGGL(tile_kernel, dim3(blocks), dim3(512), 0, 0, A, W, K, pitch, layout, M, N, useA, useW, tiles_n, sink);
                    };
                    launch(); hipDeviceSynchronize();
                    hipEventRecord(a);
                    for (int i = 0; i < 5; ++i) launch();
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    const double s = ms / 5 * 1e-3;
                    const double bytes = (double)blocks * 256 * K * 4 * (useA + useW);
                    char nm[160];
                    snprintf(nm, 160, "tiles_n=%d layout=%s pitch=%d load=%s%s", tiles_n, layout ? "blocked" : "rowmajor", pitch, useA ? "A" : "", useW ? "W" : "");
                    printf("%-60s %8.1f %10.2f %10.2f\n", nm, s * 1e6, bytes / s / 1e12, bytes / s / 2.4e9 / 256);
                }
            }
        }
    }
    return 0;
}
