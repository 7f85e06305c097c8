// fp32 atomic-add throughput into a 4 MB table: agent scope (coherent across XCDs) vs workgroup scope into a per-XCD copy.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int SCOPE>
__global__ __launch_bounds__(256) void atomic_kernel(float* table, int table_floats, int per_thread, int use_xcc, unsigned* xcc_out) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0 && xcc_out) xcc_out[blockIdx.x] = x;
    float* t = table + (use_xcc ? (size_t)x * table_floats : 0);
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int i = 0; i < per_thread; ++i) {
        // clustered addresses: 26 consecutive floats at a pseudo-random 128-byte aligned base (like the volume gradient)
        if ((i % 26) == 0) h = h * 1664525u + 1013904223u;
        const unsigned base = (h >> 8) % (table_floats / 32) * 32;
        if (SCOPE == 0) __hip_atomic_fetch_add(t + base + (i % 26), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(t + base + (i % 26), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// lanes = channels: a wave adds 64 consecutive floats (two 128-byte lines) at one pseudo-random base per instruction
__global__ __launch_bounds__(256) void atomic_coalesced_kernel(float* table, int table_floats, int per_thread) {
    const int lane = threadIdx.x & 63;
    unsigned h = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u;
    for (int i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        const unsigned base = (h >> 8) % (table_floats / 64) * 64;
        __hip_atomic_fetch_add(table + base + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const int TF = 1 << 20;   // 4 MB
    float* table; hipMalloc(&table, (size_t)8 * TF * 4); hipMemset(table, 0, (size_t)8 * TF * 4);
    unsigned* xo; hipMalloc(&xo, 4096 * 4);
    const int blocks = 1024, per_thread = 208;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 0; variant < 3; ++variant) {
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL(atomic_kernel<0>, dim3(blocks), dim3(256), 0, 0, table, TF, per_thread, 0, xo);
            if (variant == 1) hipLaunchKernelGGL(atomic_kernel<0>, dim3(blocks), dim3(256), 0, 0, table, TF, per_thread, 1, xo);
            if (variant == 2) hipLaunchKernelGGL(atomic_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, TF, per_thread, 1, xo);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double n = (double)blocks * 256 * per_thread;
        printf("%-44s %8.1f us  %6.2f G atomics/s\n", variant == 0 ? "agent scope, one table" : variant == 1 ? "agent scope, per-XCD table" : "workgroup scope, per-XCD table",
               ms / 5 * 1e3, n / (ms / 5 * 1e-3) / 1e9);
    }
    {
        auto launch = [&]() { hipLaunchKernelGGL(atomic_coalesced_kernel, dim3(blocks), dim3(256), 0, 0, table, TF, per_thread); };
        launch(); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double n = (double)blocks * 256 * per_thread;
        printf("%-44s %8.1f us  %6.2f G atomics/s\n", "agent scope, lanes = 64 consecutive floats", ms / 5 * 1e3, n / (ms / 5 * 1e-3) / 1e9);
    }
    unsigned h[64]; hipMemcpy(h, xo, 64 * 4, hipMemcpyDeviceToHost);
    printf("XCC_ID of blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", h[i]); printf("\n");
    // correctness of the workgroup-scope variant: total count over the 8 copies
    hipMemset(table, 0, (size_t)8 * TF * 4);
    hipLaunchKernelGGL(atomic_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, TF, per_thread, 1, xo);
    hipDeviceSynchronize();
    float* host = (float*)malloc((size_t)8 * TF * 4); hipMemcpy(host, table, (size_t)8 * TF * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (size_t i = 0; i < (size_t)8 * TF; ++i) sum += host[i];
    printf("workgroup-scope total %.0f expected %.0f\n", sum, (double)blocks * 256 * per_thread);
    return 0;
}
