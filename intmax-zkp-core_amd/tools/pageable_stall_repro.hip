// Standalone reproduction (no library code) of the stall DESIGN.md 3.7 records: a "call" copies with hipMemcpyAsync to and from
// pageable std::vectors and frees them when it returns; the FIRST device interaction of the next call (a 32-byte read-back) then
// takes 13 - 25 ms.  The same call with its transfers staged in pinned memory: 0.06 ms.  (MI355X, ROCm 7.2: profiles/r04h_pageable_stall.md)
//     hipcc -O2 --offload-arch=gfx950 -o pageable_stall_repro pageable_stall_repro.hip && ./pageable_stall_repro 0 && ./pageable_stall_repro 1
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint64_t* p, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) p[i] = p[i] * 3 + 1; }
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0: pageable vectors, 1: pinned staging
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t words = (size_t)20 << 17;  // 20 MB
    uint64_t *d, *pin; hipMalloc(&d, words * 8 * 2); hipHostMalloc((void**)&pin, words * 8 * 2, 0);
    hipMemsetAsync(d, 1, words * 16, s); hipStreamSynchronize(s);
    for (int it = 0; it < 12; ++it) {
        const double t0 = now();
        uint64_t small[4];
        hipMemcpyAsync(small, d, 32, hipMemcpyDeviceToHost, s);   // the first device interaction of the "call"
        hipStreamSynchronize(s);
        const double t1 = now();
        {
            std::vector<uint64_t> a(words), b(words), up(words, 7);  // result vectors of the call, an upload source
            if (mode == 0) {
                hipMemcpyAsync(d + words, up.data(), words * 8, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(touch, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, d, words);
                hipMemcpyAsync(a.data(), d, words * 8, hipMemcpyDeviceToHost, s);
                hipMemcpyAsync(b.data(), d + words, words * 8, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
            } else {
                memcpy(pin, up.data(), words * 8);
                hipMemcpyAsync(d + words, pin, words * 8, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(touch, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, d, words);
                hipMemcpyAsync(pin, d, words * 16, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                memcpy(a.data(), pin, words * 8); memcpy(b.data(), pin + words, words * 8);
            }
            // host-only phase with big short-lived allocations (the replay's node tables)
            std::vector<std::vector<uint64_t>> junk;
            for (int j = 0; j < 8; ++j) { junk.emplace_back((size_t)4 << 20, (uint64_t)j); }
            volatile uint64_t sink = 0; for (auto& v : junk) for (size_t q = 0; q < v.size(); q += 512) sink += v[q];
            const double spin = now() + 20.0; while (now() < spin) {}
        }   // a, b, up, junk freed here
        const double t2 = now();
        printf("iteration %2d: first device interaction %.3f ms, rest of the call %.1f ms\n", it, t1 - t0, t2 - t1);
    }
    return 0;
}
