// h2d_bench.hip — what does one pinned -> device copy cost as a function of its size, back to back on ONE stream and alternating over
// TWO streams?  (The witness ring of glp_prove_cols: csrc/glprover.hip::stage_words.)  Also: the same while a compute kernel keeps
// the chip busy, and a blocking hipMemcpy for comparison.
// build: hipcc -O2 --offload-arch=gfx950 -o h2d_bench h2d_bench.hip ; run: ./h2d_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            return 1;                                                            \
        }                                                                        \
    } while (0)

__global__ void spin_kernel(unsigned long long* out, long iters) {
    unsigned long long x = threadIdx.x + blockIdx.x;
    for (long i = 0; i < iters; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (x == 42) out[0] = x;
}

int main() {
    const size_t total = (size_t)512 << 20;
    char *host = nullptr, *dev = nullptr;
    CK(hipHostMalloc((void**)&host, total, hipHostMallocDefault));
    CK(hipMalloc((void**)&dev, total));
    memset(host, 1, total);
    hipStream_t s[4], comp;
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking));
    unsigned long long* sink = nullptr;
    CK(hipMalloc((void**)&sink, 64));
    for (int busy = 0; busy < 2; ++busy)
        for (int n_streams : {1, 2, 4})
            for (size_t piece : {(size_t)4 << 20, (size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20}) {
                double best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipDeviceSynchronize());
                    if (busy) hipLaunchKernelGGL(spin_kernel, dim3(256 * 64), dim3(256), 0, comp, sink, 400000L);
                    const auto t0 = std::chrono::steady_clock::now();
                    size_t i = 0;
                    for (size_t off = 0; off < total; off += piece, ++i)
                        CK(hipMemcpyAsync(dev + off, host + off, piece, hipMemcpyHostToDevice, s[i % n_streams]));
                    for (int k = 0; k < n_streams; ++k) CK(hipStreamSynchronize(s[k]));
                    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    if (ms < best) best = ms;
                    CK(hipDeviceSynchronize());
                }
                printf("%s  %d stream(s)  pieces of %4zu MB: %7.2f ms for 512 MB = %5.1f GB/s, %.3f ms per piece\n", busy ? "chip busy" : "chip idle",
                       n_streams, piece >> 20, best, total / best / 1e6, best / (total / piece));
            }
    // latency of ONE copy of each size (enqueue -> done)
    for (size_t piece : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20}) {
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(dev, host, piece, hipMemcpyHostToDevice, s[0]));
            CK(hipStreamSynchronize(s[0]));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        printf("one copy of %3zu MB: %.3f ms (%.1f GB/s)\n", piece >> 20, best, piece / best / 1e6);
    }
    return 0;
}
