// perm_latency_bench.hip — what does ONE Poseidon permutation cost when nothing runs beside it?  A Fiat-Shamir transcript is a chain
// of dependent permutations (a 2^11-row proof's transcript: ~130 of them, 65 for the opening set alone), so a transcript kept on
// the device runs at the LATENCY of one permutation, not at the chip's 2.9 G/s.  Measured here for the three mappings of this library
// (16 lanes per state, 4 lanes per state, one lane per state) as a chain of `reps` dependent permutations in one wave, next to the
// same chain on one host core.  VERDICT r05 next 2 asks for a device-side challenger: this is what one costs.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../csrc -o perm_latency_bench perm_latency_bench.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#include "poseidon.h"
#include "poseidon_coop.h"
#include "poseidon_coop4.h"

__global__ void chain_coop16(uint64_t* io, int reps) {
    const int r = threadIdx.x % poseidon_coop::GROUP;
    uint64_t s = io[threadIdx.x];
    for (int i = 0; i < reps; ++i) s = poseidon_coop::permute(s, r);
    io[threadIdx.x] = s;
}
__global__ void chain_coop4(uint64_t* io, int reps) {
    const int q = threadIdx.x % poseidon_coop4::GROUP;
    uint64_t s[poseidon_coop4::WPL];
    for (int a = 0; a < poseidon_coop4::WPL; ++a) s[a] = io[threadIdx.x * 3 + a];
    for (int i = 0; i < reps; ++i) poseidon_coop4::permute(s, q);
    for (int a = 0; a < poseidon_coop4::WPL; ++a) io[threadIdx.x * 3 + a] = s[a];
}
__global__ void chain_lane(uint64_t* io, int reps) {
    uint64_t st[12];
    for (int j = 0; j < 12; ++j) st[j] = io[threadIdx.x * 12 + j];
    for (int i = 0; i < reps; ++i) poseidon::permute<poseidon::MDS_INT>(st);
    for (int j = 0; j < 12; ++j) io[threadIdx.x * 12 + j] = st[j];
}

template <typename F>
static double time_us(F&& launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch(8);  // warm
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    launch(reps);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return 1e3 * ms / reps;
}

int main() {
    uint64_t* io = nullptr;
    hipMalloc((void**)&io, 64 * 12 * 8);
    hipMemset(io, 1, 64 * 12 * 8);
    const int reps = 2000;
    printf("one wave, a chain of %d dependent permutations (us per permutation):\n", reps);
    printf("  16 lanes per state (poseidon_coop):  %.2f us\n", time_us([&](int n) { hipLaunchKernelGGL(chain_coop16, dim3(1), dim3(16), 0, 0, io, n); }, reps));
    printf("   4 lanes per state (poseidon_coop4): %.2f us\n", time_us([&](int n) { hipLaunchKernelGGL(chain_coop4, dim3(1), dim3(4), 0, 0, io, n); }, reps));
    printf("   1 lane per state  (poseidon.h):     %.2f us\n", time_us([&](int n) { hipLaunchKernelGGL(chain_lane, dim3(1), dim3(1), 0, 0, io, n); }, reps));
    uint64_t st[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20000; ++i) poseidon::permute<poseidon::MDS_INT>(st);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20000;
    printf("  one host core (the library's host transcript, poseidon.h compiled for the host): %.2f us  [%llu]\n", us, (unsigned long long)st[0]);
    return 0;
}
