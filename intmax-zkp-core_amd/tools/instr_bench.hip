// instr_bench.hip — gfx950 issue-rate microbenchmark for the integer / fp64 instructions the Goldilocks and Poseidon
// kernels are made of.  Decides design questions with measurements (DESIGN.md "instruction rates"):
//   * is v_mad_u64_u32 quarter rate?  is v_fma_f64 half rate?  what does a 64-bit add cost?
//   * Poseidon permutations per second for the MDS variants.
// Build: hipcc -O3 --offload-arch=gfx950 instr_bench.hip -o instr_bench ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../csrc/poseidon.h"

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

#define DEFINE_ASM_KERNEL(NAME, DECL, BODY, SINK)                                             \
    __global__ __launch_bounds__(256) void NAME(uint64_t* out, uint64_t seed) {               \
        DECL;                                                                                 \
        for (int it = 0; it < ITERS; ++it) {                                                  \
            BODY                                                                              \
        }                                                                                     \
        SINK;                                                                                 \
    }

// --- v_mad_u64_u32: 8 independent chains ---
__global__ __launch_bounds__(256) void k_mad_u64_u32(uint64_t* out, uint64_t seed) {
    uint64_t acc[CHAINS];
    uint32_t a = (uint32_t)seed + threadIdx.x, b = (uint32_t)(seed >> 32) | 1;
    for (int c = 0; c < CHAINS; ++c) acc[c] = seed + c;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mul_lo_u32(uint64_t* out, uint64_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = (uint32_t)(seed >> 32) | 1;
    for (int c = 0; c < CHAINS; ++c) acc[c] = (uint32_t)seed + c + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
    }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mul_hi_u32(uint64_t* out, uint64_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = (uint32_t)(seed >> 32) | 0x80000001u;
    for (int c = 0; c < CHAINS; ++c) acc[c] = ~((uint32_t)seed + c + threadIdx.x);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
    }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mad_u32_u24(uint64_t* out, uint64_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = (uint32_t)(seed >> 40) | 1;
    for (int c = 0; c < CHAINS; ++c) acc[c] = (uint32_t)seed + c + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[c]) : "v"(b));
    }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_add_u32(uint64_t* out, uint64_t seed) {
    uint32_t acc[CHAINS];
    uint32_t b = (uint32_t)(seed >> 40) | 1;
    for (int c = 0; c < CHAINS; ++c) acc[c] = (uint32_t)seed + c + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
    }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_lshl_add_u64(uint64_t* out, uint64_t seed) {
    uint64_t acc[CHAINS];
    uint64_t b = seed | 1;
    for (int c = 0; c < CHAINS; ++c) acc[c] = seed + c + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(acc[c]) : "v"(b));
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_addc_pair(uint64_t* out, uint64_t seed) {
    uint32_t lo[CHAINS], hi[CHAINS];
    uint32_t b = (uint32_t)seed | 1, bh = (uint32_t)(seed >> 32);
    for (int c = 0; c < CHAINS; ++c) {
        lo[c] = c + threadIdx.x;
        hi[c] = c;
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                         : "+v"(lo[c]), "+v"(hi[c]) : "v"(b), "v"(bh) : "vcc");
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= ((uint64_t)hi[c] << 32) | lo[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma_f64(uint64_t* out, uint64_t seed) {
    double acc[CHAINS];
    double a = 1.0 + (double)(seed & 7) * 1e-9, b = 1e-9 * (double)threadIdx.x;
    for (int c = 0; c < CHAINS; ++c) acc[c] = (double)c;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = __builtin_bit_cast(uint64_t, s);
}

__global__ __launch_bounds__(256) void k_fma_f32(uint64_t* out, uint64_t seed) {
    float acc[CHAINS];
    float a = 1.0f + (float)(seed & 7) * 1e-6f, b = 1e-6f * (float)threadIdx.x;
    for (int c = 0; c < CHAINS; ++c) acc[c] = (float)c;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = (uint64_t)__builtin_bit_cast(uint32_t, s);
}


#define SIMPLE32(NAME, ASM)                                                                     \
    __global__ __launch_bounds__(256) void NAME(uint64_t* out, uint64_t seed) {                  \
        uint32_t acc[CHAINS];                                                                    \
        uint32_t b = (uint32_t)(seed >> 40) | 0x01010101u, c = (uint32_t)seed;                   \
        for (int k = 0; k < CHAINS; ++k) acc[k] = (uint32_t)seed + k + threadIdx.x;              \
        for (int it = 0; it < ITERS; ++it) {                                                     \
            _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) asm volatile(ASM : "+v"(acc[k]) : "v"(b), "v"(c)); \
        }                                                                                        \
        uint32_t s = 0;                                                                          \
        for (int k = 0; k < CHAINS; ++k) s ^= acc[k];                                            \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                 \
    }
SIMPLE32(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
SIMPLE32(k_dot2, "v_dot2_u32_u16 %0, %0, %1, %2")
SIMPLE32(k_perm, "v_perm_b32 %0, %0, %1, %2")
SIMPLE32(k_add3, "v_add3_u32 %0, %0, %1, %2")
SIMPLE32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
SIMPLE32(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
SIMPLE32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
SIMPLE32(k_xor, "v_xor_b32 %0, %0, %1")
SIMPLE32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
SIMPLE32(k_bfe, "v_bfe_u32 %0, %0, 3, 20")
SIMPLE32(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
SIMPLE32(k_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
SIMPLE32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
// carry forms with the carry in an SGPR pair (VOP3 encoding) against the VCC forms (VOP2 encoding); timing only
#define SIMPLE32S(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(256) void NAME(uint64_t* out, uint64_t seed) {                  \
        uint32_t acc[CHAINS];                                                                    \
        uint32_t b = (uint32_t)(seed >> 40) | 0x01010101u, c = (uint32_t)seed;                   \
        for (int k = 0; k < CHAINS; ++k) acc[k] = (uint32_t)seed + k + threadIdx.x;              \
        for (int it = 0; it < ITERS; ++it) {                                                     \
            _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) asm volatile(ASM : "+v"(acc[k]) : "v"(b), "v"(c) : "s20", "s21", "vcc"); \
        }                                                                                        \
        uint32_t s = 0;                                                                          \
        for (int k = 0; k < CHAINS; ++k) s ^= acc[k];                                            \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                 \
    }
SIMPLE32S(k_addco_e64, "v_add_co_u32_e64 %0, s[20:21], %0, %1")
SIMPLE32S(k_addc_e64, "v_addc_co_u32_e64 %0, s[20:21], %0, %1, s[20:21]")
SIMPLE32S(k_addc_e32, "v_addc_co_u32_e32 %0, vcc, %0, %1, vcc")
SIMPLE32S(k_subb_e64_const, "v_subb_co_u32_e64 %0, s[20:21], -1, %0, s[20:21]")
SIMPLE32S(k_cndmask_e64, "v_cndmask_b32_e64 %0, 0, -1, s[20:21]")
SIMPLE32S(k_cndmask_e64v, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
SIMPLE32S(k_add_nop, "v_add_u32 %0, %0, %1\n\ts_nop 0")
SIMPLE32S(k_add_nop1, "v_add_u32 %0, %0, %1\n\ts_nop 1")
SIMPLE32S(k_cmp_e64, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_add_u32 %0, %0, %2")
SIMPLE32S(k_cmp_e32, "v_cmp_lt_u32_e32 vcc, %0, %1\n\tv_add_u32 %0, %0, %2")
SIMPLE32S(k_add_e64, "v_add_u32_e64 %0, %0, %1")
SIMPLE32S(k_sub_const, "v_sub_co_u32_e64 %0, s[20:21], 1, %0")

// --- field multiply, 8 independent chains (compiler-scheduled) ---
__global__ __launch_bounds__(256) void k_gl_mul(uint64_t* out, uint64_t seed) {
    uint64_t acc[CHAINS];
    uint64_t b = seed | 1;
    for (int c = 0; c < CHAINS; ++c) acc[c] = seed * (c + 3) + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = gl::mul(acc[c], b);
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_gl_add(uint64_t* out, uint64_t seed) {
    uint64_t acc[CHAINS];
    uint64_t b = seed | 1;
    for (int c = 0; c < CHAINS; ++c) acc[c] = seed * (c + 3) + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = gl::add(acc[c], b);
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MDS>
__global__ __launch_bounds__(256) void k_permute(uint64_t* out, uint64_t seed) {
    uint64_t st[12];
    for (int i = 0; i < 12; ++i) st[i] = seed * (i + 1) + blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < 16; ++it) poseidon::permute<MDS>(st);
    uint64_t s = 0;
    for (int i = 0; i < 12; ++i) s ^= st[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MDS>
__global__ __launch_bounds__(256) void k_mds_only(uint64_t* out, uint64_t seed) {
    uint64_t st[12];
    for (int i = 0; i < 12; ++i) st[i] = seed * (i + 1) + blockIdx.x * 256 + threadIdx.x;
#pragma unroll 1
    for (int it = 0; it < 480; ++it) poseidon::mds_layer<MDS>(st);
    uint64_t s = 0;
    for (int i = 0; i < 12; ++i) s ^= st[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
double time_kernel(K kernel, int blocks, uint64_t* out, int reps = 5) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, (uint64_t)0x9E3779B97F4A7C15ull);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, (uint64_t)0x9E3779B97F4A7C15ull + r);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    const int blocks = cus * 8;  // 8 x 256 threads = 32 waves per CU
    uint64_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8 * 4));
    const double waves = (double)blocks * 4;
    auto report = [&](const char* name, double secs, double instr_per_thread) {
        // cycles per wave-instruction per SIMD at nominal clock: each SIMD runs waves/(cus*4) waves
        double wave_instrs_per_simd = waves / (cus * 4) * instr_per_thread;
        printf("%-18s %8.3f ms  %7.2f cycles/wave-instr/SIMD (nominal clk)  %8.2f G lane-ops/s\n", name, secs * 1e3,
               secs * clk / wave_instrs_per_simd, waves * 64 * instr_per_thread / secs / 1e9);
    };
    const double n = (double)ITERS * CHAINS;
    report("v_add_u32", time_kernel(k_add_u32, blocks, out), n);
    report("v_mad_u32_u24", time_kernel(k_mad_u32_u24, blocks, out), n);
    report("v_mul_lo_u32", time_kernel(k_mul_lo_u32, blocks, out), n);
    report("v_mul_hi_u32", time_kernel(k_mul_hi_u32, blocks, out), n);
    report("v_mad_u64_u32", time_kernel(k_mad_u64_u32, blocks, out), n);
    report("v_dot4_u32_u8", time_kernel(k_dot4, blocks, out), n);
    report("v_dot2_u32_u16", time_kernel(k_dot2, blocks, out), n);
    report("v_perm_b32", time_kernel(k_perm, blocks, out), n);
    report("v_add3_u32", time_kernel(k_add3, blocks, out), n);
    report("v_lshl_add_u32", time_kernel(k_lshl_add_u32, blocks, out), n);
    report("v_mad_u32_u16", time_kernel(k_mad_u32_u16, blocks, out), n);
    report("v_mad_i32_i24", time_kernel(k_mad_i32_i24, blocks, out), n);
    report("v_mul_u32_u24", time_kernel(k_mul_u32_u24, blocks, out), n);
    report("v_cndmask_b32", time_kernel(k_cndmask, blocks, out), n);
    report("v_xor_b32", time_kernel(k_xor, blocks, out), n);
    report("v_alignbit_b32", time_kernel(k_alignbit, blocks, out), n);
    report("v_bfe_u32", time_kernel(k_bfe, blocks, out), n);
    report("v_add_co_u32", time_kernel(k_addco, blocks, out), n);
    report("v_add_co e64 sgpr", time_kernel(k_addco_e64, blocks, out), n);
    report("v_addc_co e64 sgpr", time_kernel(k_addc_e64, blocks, out), n);
    report("v_addc_co e32 vcc", time_kernel(k_addc_e32, blocks, out), n);
    report("v_subb e64 -1", time_kernel(k_subb_e64_const, blocks, out), n);
    report("v_sub_co e64 1,", time_kernel(k_sub_const, blocks, out), n);
    report("v_cndmask e64 0,-1", time_kernel(k_cndmask_e64, blocks, out), n);
    report("v_cndmask e64 v,v", time_kernel(k_cndmask_e64v, blocks, out), n);
    report("v_add_u32 e64", time_kernel(k_add_e64, blocks, out), n);
    report("v_add + s_nop 0", time_kernel(k_add_nop, blocks, out), n);
    report("v_add + s_nop 1", time_kernel(k_add_nop1, blocks, out), n);
    report("v_cmp e64 + v_add", time_kernel(k_cmp_e64, blocks, out), n);
    report("v_cmp e32 + v_add", time_kernel(k_cmp_e32, blocks, out), n);
    report("v_lshl_add_u64", time_kernel(k_lshl_add_u64, blocks, out), n);
    report("add_co+addc pair", time_kernel(k_addc_pair, blocks, out), n);
    report("v_fma_f32", time_kernel(k_fma_f32, blocks, out), n);
    report("v_fma_f64", time_kernel(k_fma_f64, blocks, out), n);
    report("gl::mul", time_kernel(k_gl_mul, blocks, out), n);
    report("gl::add", time_kernel(k_gl_add, blocks, out), n);
    {
        double s = time_kernel(k_mds_only<poseidon::MDS_INT>, blocks, out);
        printf("mds_layer int      %8.3f ms  %.1f cycles/layer/wave/SIMD\n", s * 1e3, s * clk / (waves / (cus * 4) * 480));
        s = time_kernel(k_mds_only<poseidon::MDS_F64>, blocks, out);
        printf("mds_layer f64      %8.3f ms  %.1f cycles/layer/wave/SIMD\n", s * 1e3, s * clk / (waves / (cus * 4) * 480));
    }
    for (int occ = 0; occ < 2; ++occ) {
        int bl = occ == 0 ? blocks : cus * 4;
        double perms = (double)bl * 256 * 16;
        double s = time_kernel(k_permute<poseidon::MDS_INT>, bl, out);
        printf("permute int  (%2d waves/CU) %8.3f ms  %7.3f Gperm/s\n", bl * 4 / cus, s * 1e3, perms / s / 1e9);
        s = time_kernel(k_permute<poseidon::MDS_F64>, bl, out);
        printf("permute f64  (%2d waves/CU) %8.3f ms  %7.3f Gperm/s\n", bl * 4 / cus, s * 1e3, perms / s / 1e9);
    }
    CHECK(hipFree(out));
    return 0;
}
