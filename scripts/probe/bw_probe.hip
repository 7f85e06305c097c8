// Memory-ingest probe for gfx950: how many bytes/clk can one CU pull (a) from HBM, (b) from its XCD's L2,
// as a function of loads in flight.  Build: hipcc --offload-arch=gfx950 -O3 scripts/probe/bw_probe.hip -o build/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int UNROLL>
__global__ __launch_bounds__(512) void stream_kernel(const float4* __restrict__ src, size_t block_stride_vec, size_t vec_per_block,
                                                     int reps, int share, float* sink) {
    // block b reads region (b / share); every thread float4-coalesced, UNROLL loads in flight
    const float4* p = src + (size_t)(blockIdx.x / share) * block_stride_vec;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        for (size_t i = threadIdx.x; i + (size_t)(UNROLL - 1) * blockDim.x < vec_per_block; i += (size_t)UNROLL * blockDim.x) {
            float4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(512) void stream_kernel_cached(const float4* __restrict__ src, size_t block_stride_vec, size_t vec_per_block,
                                                            int reps, int share, float* sink) {
    const float4* p = src + (size_t)(blockIdx.x / share) * block_stride_vec;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        for (size_t i = threadIdx.x; i + (size_t)(UNROLL - 1) * blockDim.x < vec_per_block; i += (size_t)UNROLL * blockDim.x) {
            float4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int UNROLL, bool NT>
double run(const float4* src, int blocks, int threads, size_t bytes_per_block, size_t stride_bytes, int reps, int share, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto launch = [&]() {
        if (NT) hipLaunchKernelGGL(stream_kernel<UNROLL>, dim3(blocks), dim3(threads), 0, 0, src, stride_bytes / 16, bytes_per_block / 16, reps, share, sink);
        else hipLaunchKernelGGL(stream_kernel_cached<UNROLL>, dim3(blocks), dim3(threads), 0, 0, src, stride_bytes / 16, bytes_per_block / 16, reps, share, sink);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5 * 1e-3;
}

int main() {
    const size_t total = (size_t)2 << 30;
    float4* src; hipMalloc(&src, total); hipMemset(src, 0, total);
    float* sink; hipMalloc(&sink, 4);
    const double clk = 2.4e9;
    printf("%-46s %8s %10s %12s\n", "case", "us", "TB/s", "B/clk/CU");
    auto report = [&](const char* name, int blocks, double s, size_t bytes_per_block, int reps) {
        double bytes = (double)blocks * bytes_per_block * reps;
        int cus = blocks < 256 ? blocks : 256;
        printf("%-46s %8.1f %10.3f %12.2f\n", name, s * 1e6, bytes / s / 1e12, bytes / s / clk / cus);
    };
    char nm[128];
    // (a) HBM streaming, private 4 MB per block, 256 / 512 blocks, unroll sweep, thread-count sweep
    for (int threads : {256, 512}) {
        for (int blocks : {1, 8, 64, 256, 512}) {
            size_t bpb = 4u << 20;
            snprintf(nm, 128, "hbm private 4MB  blk=%d thr=%d unroll=4", blocks, threads);
            report(nm, blocks, run<4, false>(src, blocks, threads, bpb, bpb, 1, 1, sink), bpb, 1);
            snprintf(nm, 128, "hbm private 4MB  blk=%d thr=%d unroll=8", blocks, threads);
            report(nm, blocks, run<8, false>(src, blocks, threads, bpb, bpb, 1, 1, sink), bpb, 1);
            snprintf(nm, 128, "hbm private 4MB  blk=%d thr=%d unroll=16", blocks, threads);
            report(nm, blocks, run<16, false>(src, blocks, threads, bpb, bpb, 1, 1, sink), bpb, 1);
        }
    }
    // (b) L2 resident: every block re-reads the same 1 MB region 16 times
    for (int blocks : {1, 8, 64, 256, 512}) {
        size_t bpb = 1u << 20;
        snprintf(nm, 128, "l2 shared 1MB x16 blk=%d thr=512 unroll=8", blocks);
        report(nm, blocks, run<8, false>(src, blocks, 512, bpb, 0, 16, 1, sink), bpb, 16);
        snprintf(nm, 128, "l2 shared 1MB x16 blk=%d thr=512 unroll=16", blocks);
        report(nm, blocks, run<16, false>(src, blocks, 512, bpb, 0, 16, 1, sink), bpb, 16);
    }
    // (c) per-block private 256 KB region re-read 64 times (fits L2: 256 x 256 KB = 64 MB total -> MALL; per XCD 8 MB > 4 MB L2)
    for (int blocks : {1, 64, 256}) {
        size_t bpb = 256u << 10;
        snprintf(nm, 128, "private 256KB x64 blk=%d thr=512 unroll=8", blocks);
        report(nm, blocks, run<8, false>(src, blocks, 512, bpb, bpb, 64, 1, sink), bpb, 64);
    }
    for (int blocks : {1, 64, 256}) {
        size_t bpb = 64u << 10;     // 256 x 64 KB = 16 MB -> 2 MB per XCD, L2 resident
        snprintf(nm, 128, "private 64KB x256 blk=%d thr=512 unroll=8", blocks);
        report(nm, blocks, run<8, false>(src, blocks, 512, bpb, bpb, 256, 1, sink), bpb, 256);
    }
    // (d) GEMM-like sharing: groups of 4 consecutive... blocks b, b+8, b+16, b+24 share (same XCD) -> emulate with share on b/8
    for (int share : {1, 2, 4, 8}) {
        size_t bpb = 1u << 20;
        snprintf(nm, 128, "hbm 1MB shared by %d adjacent blocks blk=256", share);
        report(nm, 256, run<8, false>(src, 256, 512, bpb, bpb, 1, share, sink), bpb, 1);
    }
    return 0;
}
