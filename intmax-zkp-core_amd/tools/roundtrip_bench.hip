// What does one Fiat-Shamir round trip cost?  A small kernel produces 512 bytes (a Merkle cap); the host needs them before it can
// launch the next kernel.  Variants: (a) hipMemcpyAsync to PAGEABLE memory + stream sync (what merkle_build did through round 5),
// (b) the same to pinned memory, (c) the kernel stores into host-mapped pinned memory and raises a flag there, the host spins on the
// flag (no copy, no sync call), (d) as (c) with the next kernel's 64 bytes of challenges passed as a kernel argument instead of an
// H2D copy.  Prints microseconds per round trip (kernel launch -> host has the data -> next kernel launched and finished).
// build: hipcc -O2 --offload-arch=gfx950 roundtrip_bench.hip -o roundtrip_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Chal { uint64_t v[8]; };

__global__ void produce(uint64_t* out, uint64_t seed, volatile uint64_t* flag, uint64_t seq) {
    if (threadIdx.x < 64) out[threadIdx.x] = seed * 0x9E3779B97F4A7C15ull + threadIdx.x;
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *flag = seq;
    }
}
__global__ void consume_ptr(const uint64_t* chal, uint64_t* sink) { if (threadIdx.x < 8) sink[threadIdx.x] += chal[threadIdx.x]; }
__global__ void consume_arg(Chal c, uint64_t* sink) { if (threadIdx.x < 8) sink[threadIdx.x] += c.v[threadIdx.x]; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    uint64_t *dev_out, *dev_chal, *sink, *pinned, *mapped, *mapped_dev;
    hipMalloc(&dev_out, 512); hipMalloc(&dev_chal, 64); hipMalloc(&sink, 64);
    hipMemset(sink, 0, 64);
    hipHostMalloc(&pinned, 1024, hipHostMallocDefault);
    hipHostMalloc(&mapped, 1024, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void**)&mapped_dev, mapped, 0);
    std::vector<uint64_t> pageable(128);
    const int reps = 2000;
    uint64_t acc = 0;
    for (int variant = 0; variant < 4; ++variant) {
        double t0 = 0;
        for (int r = -200; r < reps; ++r) {
            if (r == 0) { hipStreamSynchronize(s); t0 = now(); }
            Chal c;
            if (variant <= 1) {
                hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, s, dev_out, (uint64_t)r, (volatile uint64_t*)nullptr, 0ull);
                uint64_t* dst = variant == 0 ? pageable.data() : pinned;
                hipMemcpyAsync(dst, dev_out, 512, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                for (int i = 0; i < 8; ++i) c.v[i] = dst[i] ^ dst[63 - i];
                hipMemcpyAsync(dev_chal, c.v, 64, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(consume_ptr, dim3(1), dim3(64), 0, s, dev_chal, sink);
            } else {
                const uint64_t seq = (uint64_t)(r + 1000);
                hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, s, mapped_dev, (uint64_t)r, (volatile uint64_t*)(mapped_dev + 100), seq);
                while (((volatile uint64_t*)mapped)[100] != seq) {}
                for (int i = 0; i < 8; ++i) c.v[i] = mapped[i] ^ mapped[63 - i];
                if (variant == 2) {
                    hipMemcpyAsync(dev_chal, c.v, 64, hipMemcpyHostToDevice, s);
                    hipLaunchKernelGGL(consume_ptr, dim3(1), dim3(64), 0, s, dev_chal, sink);
                } else {
                    hipLaunchKernelGGL(consume_arg, dim3(1), dim3(64), 0, s, c, sink);
                }
            }
            acc += c.v[0];
        }
        hipStreamSynchronize(s);
        const char* names[4] = {"memcpy to pageable + sync, H2D copy", "memcpy to pinned + sync, H2D copy", "mapped store + spin, H2D copy",
                                "mapped store + spin, kernel argument"};
        printf("%-40s %.1f us per round trip\n", names[variant], (now() - t0) / reps * 1e6);
    }
    printf("(%llu)\n", (unsigned long long)acc);
    return 0;
}
