// LDS canary: does some OTHER kernel, co-resident on the same CU, write into LDS it does not own?
// Every workgroup fills its dynamic LDS with a pattern and re-checks it `iters` times (with sleeps in between) while other
// kernels run on another stream.  out[0] = number of corrupted words seen, out[1] = first bad word offset, out[2] = value found,
// out[3] = workgroups that ran.   Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o scripts/probe/liblds_canary.so scripts/probe/lds_canary.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void lds_canary_kernel(int words, int iters, unsigned* out) {
    extern __shared__ unsigned lds[];
    const unsigned pat = 0xC0DE0000u ^ (blockIdx.x * 2654435761u);
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pat + i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int s = 0; s < 8; ++s) __builtin_amdgcn_s_sleep(127);
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            const unsigned v = lds[i];
            if (v != pat + i) {
                if (atomicAdd(out, 1u) == 0) { out[1] = (unsigned)i; out[2] = v; out[4] = pat + i; out[5] = blockIdx.x; }
                lds[i] = pat + i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out + 3, 1u);
}

extern "C" int lds_canary_launch(int blocks, int threads, int lds_bytes, int iters, unsigned* out, void* stream) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(threads), lds_bytes, static_cast<hipStream_t>(stream), lds_bytes / 4, iters, out);
    return (int)hipGetLastError();
}
