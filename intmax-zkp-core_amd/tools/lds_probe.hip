#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(384) void spin(float* out, int iters) {
    extern __shared__ float lds[];
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    lds[threadIdx.x] = x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[1];
}
int main() {
    float* out; hipMalloc(&out, 4096 * 4);
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int kb = 40; kb <= 160; kb += 4) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(512), dim3(384), kb * 1024, 0, out, 400000);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("LDS %3d KB per WG: %.2f ms %s\n", kb, ms, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
