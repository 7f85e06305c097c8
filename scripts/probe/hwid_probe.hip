// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], ...)
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 8 * 8);
    hipLaunchKernelGGL(k, dim3(8), dim3(512), 0, 0, d);
    unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 8; ++b) {
        printf("wg %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
