// addsub_bench.hip — issue cost of the modular add / sub pair of an NTT butterfly (u, v) -> (u + v, u - v) on gfx950, in the forms
// considered for the NTT kernels: (a) gl::add / gl::sub as compiled (arbitrary u64 representatives in and out),
// (b) a canonical-in / canonical-out form (inputs < p: one conditional correction each, carries on VCC / SGPR pairs).
// Build: hipcc -O3 --offload-arch=gfx950 addsub_bench.hip -o addsub_bench ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../csrc/gl_field.h"

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

// canonical forms: a, b < p  ->  result < p
__device__ __forceinline__ uint64_t cadd(uint64_t a, uint64_t b) {
    const uint64_t s = a + b;                 // < 2p: wraps 2^64 at most once
    const uint64_t t = s + gl::EPS;           // s - p (mod 2^64)
    return (s < a || t < s) ? t : s;          // wrapped, or s >= p
}
__device__ __forceinline__ uint64_t csub(uint64_t a, uint64_t b) {
    const uint64_t d = a - b;
    return a < b ? d - gl::EPS : d;           // + p (mod 2^64)
}

template <int FORM>
__global__ __launch_bounds__(256) void k(uint64_t* data, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
    uint64_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = data[i * n + t];
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 1; s < 16; s <<= 1)  // four butterfly stages over the 16 values, no multiplications
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (!(j & s)) {
                    const uint64_t u = x[j], v = x[j + s];
                    if (FORM == 0) {
                        x[j] = gl::add(u, v);
                        x[j + s] = gl::sub(u, v);
                    } else if (FORM == 1) {
                        x[j] = cadd(u, v);
                        x[j + s] = csub(u, v);
                    } else {
                        gl::bfly(x[j], x[j + s]);
                    }
                }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) data[i * n + t] = gl::canon(x[i]);
}

int main() {
    const int blocks = 256 * 16, threads = 256, iters = 200;
    const size_t n = (size_t)blocks * threads;
    std::vector<uint64_t> h(16 * n);
    uint64_t s = 1;
    for (auto& v : h) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        v = s % gl::P;
    }
    // the first 8 192 threads start from edge values: both corrections of gl::bfly with a low word that carries (u - v and
    // u + v - p with a low word of 0xFFFFFFFF), zero, p - 1, words around 2^32
    const uint64_t edge[] = {0ull, 1ull, 2ull, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, gl::P - 1, gl::P - 2, gl::P - 0xFFFFFFFFull,
                             0xFFFFFFFEFFFFFFFFull, 0x00000001FFFFFFFFull, 0x7FFFFFFF80000000ull, 0xFFFFFFFE00000001ull, 0x8000000000000000ull};
    const size_t n_edge = sizeof edge / sizeof edge[0];
    for (size_t t = 0; t < 8192; ++t)
        for (size_t i = 0; i < 16; ++i) h[i * n + t] = edge[(t * 7 + i * 13 + (t >> 4) * (i + 1)) % n_edge];
    uint64_t* d = nullptr;
    CHECK(hipMalloc((void**)&d, h.size() * 8));
    std::vector<uint64_t> out[3];
    double ms[3];
    for (int form = 0; form < 3; ++form) {
        CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        if (form == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, 2); else if (form == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, 2); else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, d, 2);
        CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipEventRecord(a));
        if (form == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, iters); else if (form == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, iters); else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, d, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, a, b));
        ms[form] = t;
        out[form].resize(h.size());
        CHECK(hipMemcpy(out[form].data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    }
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); ++i) bad += out[0][i] != out[1][i] || out[0][i] != out[2][i];
    const double pairs = (double)n / 64 * iters * 32;  // wave-level butterflies (an add and a sub each)
    const char* names[3] = {"gl::add / gl::sub        ", "canonical in/out, compiled", "gl::bfly (12 instructions)"};
    for (int form = 0; form < 3; ++form)
        printf("%s: %.3f ms, %.1f SIMD-cycles per (add, sub) pair at 2.4 GHz\n", names[form], ms[form],
               ms[form] * 1e-3 * 2.4e9 * 1024 / pairs);
    printf("results differ in %zu of %zu words\n", bad, h.size());
    return bad != 0;
}
