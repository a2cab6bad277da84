// mfma_mds_bench.hip — the Poseidon MDS layer (12 x 12, entries <= 41) as int8 MFMA products on byte planes, for the
// lane-per-state layout of the hashing kernels, INCLUDING every data movement it needs, bit-exact against the production
// layer (poseidon::mds_layer_add: 288 v_mad_u64_u32 on 32-bit halves), and timed beside it.  A measurement tool only
// (VERDICT r02 item 6): BASELINE.json's north_star describes the path as non-MFMA, so nothing here is linked into
// libglprover.so; the result is recorded in profiles/r03_poseidon_alternatives.md.
//
// Formulation (no cross-lane traffic at all).  v_mfma_i32_16x16x64_i8 contracts, for output (row m, column n),
//     D[m][n] = sum over lane groups g = 0..3 and byte slots j = 0..15 of  A_byte(lane (m, g), j) * B_byte(lane (n, g), j)
// (lane (x, g) = x + 16 g; true for any k <-> (g, j) assignment as long as A and B use the same one, which every AMD layout
// does).  Lane l = (n, g) puts plane-b bytes of ITS OWN twelve state words into B slots 0..11; the constant operand is block
// diagonal, A(lane (m = 4 kq + r4, g), j) = [g == kq] * M[4 og + r4][j]: so D[4 kq + r4][n] = sum_j M[4 og + r4][j] * x_b[j]
// of the state in lane (n, kq), and since a lane holds rows 4 g .. 4 g + 3 of column l % 16 in its four result registers,
// EVERY LANE RECEIVES ITS OWN four output words — 3 output groups x 8 byte planes = 24 MFMAs per layer per wave, each using
// 12 of its 64 K slots and 1/4 of its rows (12 % of the MACs).  Around them, per lane: a 4 x 4 byte transpose of the 24 state
// halves (48 v_perm_b32), bytes made signed by ^ 0x80 (24 v_xor; the bias 128 * rowsum is the MFMA's accumulator input, so the
// results are the unsigned plane sums), and the 8 plane sums of each output word recombined into the two 32-bit-half
// accumulators of the production form (per half: 2 v_lshl_add_u32 + 2 v_mad_u64_u32 = 96 per layer), then the same mds_combine.
//
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form mfma_mds_bench.hip -o mfma_mds_bench ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../csrc/poseidon.h"

#define CHECK(x)                                                                                 \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);        \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int W = 12;

// coefficient of input word j in output word r:  out[r] = sum_i s[(i + r) % 12] * CIRC[i] + 8 s[0] [r == 0]
__host__ __device__ inline int mds_coeff(int r, int j) {
    const int circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return circ[((j - r) % 12 + 12) % 12] + (r == 0 && j == 0 ? 8 : 0);
}

// ---- production layer, ITERS times in a dependent chain (the round constants make every layer different) ----
template <int ITERS>
__global__ __launch_bounds__(256) void k_mad(uint64_t* st, unsigned long long* cyc) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[W];
#pragma unroll
    for (int i = 0; i < W; ++i) s[i] = st[(size_t)i * gridDim.x * blockDim.x + t];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) poseidon::mds_layer_add(s, &poseidon::RC[(it % 30) * W]);
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < W; ++i) st[(size_t)i * gridDim.x * blockDim.x + t] = gl::canon(s[i]);
    if (cyc && (threadIdx.x & 63) == 0) cyc[t >> 6] = t1 - t0;
}

// ---- the MFMA layer ----
struct MfmaConsts {
    v4i a[3];     // the block-diagonal constant operand of output group og
    v4i bias[3];  // accumulator input: 128 * rowsum of the lane's four output rows (the bytes are fed as x ^ 0x80 = x - 128)
};

__device__ __forceinline__ MfmaConsts mfma_consts() {
    const int l = threadIdx.x & 63, m = l & 15, g = l >> 4, kq = m >> 2, r4 = m & 3;
    MfmaConsts c;
#pragma unroll
    for (int og = 0; og < 3; ++og) {
        unsigned w[4] = {0, 0, 0, 0};
        if (g == kq)
            for (int j = 0; j < 12; ++j) w[j >> 2] |= (unsigned)mds_coeff(4 * og + r4, j) << (8 * (j & 3));
        c.a[og] = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        int rs[4];
        for (int i = 0; i < 4; ++i) {  // result register i of every lane = output word 4 og + i of the lane's own state
            rs[i] = 0;
            for (int j = 0; j < 12; ++j) rs[i] += mds_coeff(4 * og + i, j);
        }
        c.bias[og] = v4i{128 * rs[0], 128 * rs[1], 128 * rs[2], 128 * rs[3]};
    }
    return c;
}

// 4 x 4 byte transpose: in = four 32-bit words, out[b] = (w0.b, w1.b, w2.b, w3.b)   (v_perm_b32: byte k of the result =
// byte sel[k] of the pool {src1 = bytes 0..3, src0 = bytes 4..7})
__device__ __forceinline__ void transpose4(const uint32_t (&w)[4], uint32_t (&p)[4]) {
    const uint32_t t0 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400);  // w0.b0 w1.b0 w0.b1 w1.b1
    const uint32_t t1 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602);  // w0.b2 w1.b2 w0.b3 w1.b3
    const uint32_t t2 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400);
    const uint32_t t3 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602);
    p[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100);  // t0.b0 t0.b1 t2.b0 t2.b1
    p[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302);
    p[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100);
    p[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302);
}

__device__ __forceinline__ void mds_layer_mfma(uint64_t (&s)[W], const uint64_t* next_rc, const MfmaConsts& mc) {
    uint32_t plane[8][3];  // plane[b][blk] = bytes (word 4 blk .. 4 blk + 3) of byte plane b, made signed
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        uint32_t lo[4], hi[4], pl[4], ph[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = (uint32_t)s[4 * blk + k];
            hi[k] = (uint32_t)(s[4 * blk + k] >> 32);
        }
        transpose4(lo, pl);
        transpose4(hi, ph);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            plane[b][blk] = pl[b] ^ 0x80808080u;
            plane[4 + b][blk] = ph[b] ^ 0x80808080u;
        }
    }
    // one output group (four words) at a time: 8 products = 32 result registers live, recombined before the next group.
    // The accumulator input of every product is 128 * rowsum (undoing the ^ 0x80), so the results are the UNSIGNED plane sums
    // U_b < 2^17, and a half-word accumulator of the production form is
    //     rc_half + U_0 + 2^8 U_1 + 2^16 (U_2 + 2^8 U_3)  =  mad(t23, 65536, mad(t01, 1, rc_half)),  t = (U_hi << 8) + U_lo
    // (the multiplies written as v_mad_u64_u32 by hand: left to itself the compiler expands them into 64-bit shift / add
    // sequences, ~5 instructions per plane).
    const uint32_t k65536 = __builtin_amdgcn_readfirstlane(65536);
#pragma unroll
    for (int og = 0; og < 3; ++og) {
        v4i d[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const v4i bop = {(int)plane[b][0], (int)plane[b][1], (int)plane[b][2], 0};
            d[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(mc.a[og], bop, mc.bias[og], 0, 0, 0);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int r = 4 * og + r4;
            uint64_t acc[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t t01 = ((uint32_t)d[4 * h + 1][r4] << 8) + (uint32_t)d[4 * h][r4];      // v_lshl_add_u32
                const uint32_t t23 = ((uint32_t)d[4 * h + 3][r4] << 8) + (uint32_t)d[4 * h + 2][r4];
                uint64_t a0 = h == 0 ? (uint64_t)(uint32_t)next_rc[r] : next_rc[r] >> 32, cy;
                asm("v_mad_u64_u32 %0, %1, %2, 1, %0" : "+v"(a0), "=s"(cy) : "v"(t01));
                asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a0), "=s"(cy) : "v"(t23), "s"(k65536));
                acc[h] = a0;
            }
            s[r] = poseidon::mds_combine(acc[0], acc[1]);
        }
    }
}

template <int ITERS>
__global__ __launch_bounds__(256) void k_mfma(uint64_t* st, unsigned long long* cyc) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const MfmaConsts mc = mfma_consts();
    uint64_t s[W];
#pragma unroll
    for (int i = 0; i < W; ++i) s[i] = st[(size_t)i * gridDim.x * blockDim.x + t];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) mds_layer_mfma(s, &poseidon::RC[(it % 30) * W], mc);
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < W; ++i) st[(size_t)i * gridDim.x * blockDim.x + t] = gl::canon(s[i]);
    if (cyc && (threadIdx.x & 63) == 0) cyc[t >> 6] = t1 - t0;
}

// ---- the bare MFMA issue rate (24 independent products per iteration, nothing else) ----
template <int ITERS>
__global__ __launch_bounds__(256) void k_mfma_only(int* out, unsigned long long* cyc) {
    const MfmaConsts mc = mfma_consts();
    v4i acc[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
    v4i bop = {(int)threadIdx.x, (int)blockIdx.x, 7, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int og = 0; og < 3; ++og) acc[og] = __builtin_amdgcn_mfma_i32_16x16x64_i8(mc.a[og], bop, acc[og], 0, 0, 0);
        bop[0] ^= acc[0][0];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2];
    if (cyc && (threadIdx.x & 63) == 0) cyc[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static uint64_t splitmix(uint64_t& x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    constexpr int ITERS = 240;
    const double clk = 2.4e9, simds = 1024;
    printf("{\n");
    // ---- 1. bit-exactness: edge states + random states, one layer and a chain of ITERS layers ----
    {
        const int blocks = 64, threads = 256, n = blocks * threads;
        std::vector<uint64_t> h((size_t)W * n), ref((size_t)W * n), got((size_t)W * n);
        uint64_t seed = 1;
        const uint64_t edge[6] = {0, 1, gl::P - 1, gl::P, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFF00000000ull};
        for (int t = 0; t < n; ++t)
            for (int i = 0; i < W; ++i)
                h[(size_t)i * n + t] = t < 36 ? edge[(t + i * (t / 6)) % 6] : (t < 100 ? splitmix(seed) | 0x8080808080808080ull : splitmix(seed));
        uint64_t* d = nullptr;
        CHECK(hipMalloc((void**)&d, h.size() * 8));
        CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mad<ITERS>, dim3(blocks), dim3(threads), 0, 0, d, nullptr);
        CHECK(hipMemcpy(ref.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma<ITERS>, dim3(blocks), dim3(threads), 0, 0, d, nullptr);
        CHECK(hipMemcpy(got.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h.size(); ++i) bad += ref[i] != got[i];
        // the production chain itself against the host form of the same header (textbook u128 arithmetic)
        size_t bad_host = 0;
        for (int t = 0; t < 200; ++t) {
            uint64_t s[W];
            for (int i = 0; i < W; ++i) s[i] = h[(size_t)i * n + t];
            for (int it = 0; it < ITERS; ++it) poseidon::mds_layer_add(s, &poseidon::RC[(it % 30) * W]);
            for (int i = 0; i < W; ++i) bad_host += gl::canon(s[i]) != ref[(size_t)i * n + t];
        }
        // one layer only (a wrong byte-plane weight could cancel over a chain only by miracle, but check the single step too)
        CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mad<1>, dim3(blocks), dim3(threads), 0, 0, d, nullptr);
        CHECK(hipMemcpy(ref.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(threads), 0, 0, d, nullptr);
        CHECK(hipMemcpy(got.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        size_t bad1 = 0;
        for (size_t i = 0; i < h.size(); ++i) bad1 += ref[i] != got[i];
        printf(" \"states_checked\": %d, \"words_differing_after_%d_layers\": %zu, \"words_differing_after_1_layer\": %zu, "
               "\"production_chain_vs_host_u128\": %zu,\n", n, ITERS, bad, bad1, bad_host);
        CHECK(hipFree(d));
        if (bad || bad1 || bad_host) {
            printf(" \"error\": \"NOT bit-exact\"\n}\n");
            return 1;
        }
    }
    // ---- 2. throughput with the chip filled (what the leaf-hash kernel sees) and one wave per SIMD (latency view) ----
    for (int mode = 0; mode < 2; ++mode) {
        const int threads = 256, blocks = mode == 0 ? 256 * 32 : 256;  // mode 1: one 4-wave workgroup per CU = one wave per SIMD
        const size_t n = (size_t)blocks * threads, waves = n / 64;
        uint64_t* d = nullptr;
        unsigned long long* cyc = nullptr;
        CHECK(hipMalloc((void**)&d, (size_t)W * n * 8));
        CHECK(hipMalloc((void**)&cyc, waves * 8));
        std::vector<uint64_t> h((size_t)W * n);
        uint64_t seed = 7;
        for (auto& x : h) x = splitmix(seed);
        CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        auto wave_cycles = [&]() {
            std::vector<unsigned long long> c(waves);
            CHECK(hipMemcpy(c.data(), cyc, waves * 8, hipMemcpyDeviceToHost));
            double s = 0;
            for (auto v : c) s += (double)v;
            return s / waves / ITERS;
        };
        const double ms_mad = time_ms([&] { hipLaunchKernelGGL(k_mad<ITERS>, dim3(blocks), dim3(threads), 0, 0, d, cyc); }, 5);
        const double wc_mad = wave_cycles();
        const double ms_mfma = time_ms([&] { hipLaunchKernelGGL(k_mfma<ITERS>, dim3(blocks), dim3(threads), 0, 0, d, cyc); }, 5);
        const double wc_mfma = wave_cycles();
        int* o = nullptr;
        CHECK(hipMalloc((void**)&o, n * 4));
        const double ms_only = time_ms([&] { hipLaunchKernelGGL(k_mfma_only<ITERS>, dim3(blocks), dim3(threads), 0, 0, o, cyc); }, 5);
        const double wc_only = wave_cycles();
        auto simd_cycles = [&](double ms) { return ms * 1e-3 * clk * simds / ((double)waves * ITERS); };
        printf(" \"%s\": {\"waves\": %zu, \"layers_per_wave\": %d,\n"
               "   \"mad_form\":  {\"ms\": %.3f, \"simd_cycles_per_layer_at_2.4GHz\": %.0f, \"wave_clock_per_layer\": %.0f},\n"
               "   \"mfma_form\": {\"ms\": %.3f, \"simd_cycles_per_layer_at_2.4GHz\": %.0f, \"wave_clock_per_layer\": %.0f},\n"
               "   \"mfma_only_24_per_layer\": {\"ms\": %.3f, \"simd_cycles_per_layer_at_2.4GHz\": %.0f, \"wave_clock_per_layer\": %.0f},\n"
               "   \"mfma_over_mad_time\": %.3f}%s\n",
               mode == 0 ? "chip_filled" : "one_wave_per_simd", waves, ITERS, ms_mad, simd_cycles(ms_mad), wc_mad, ms_mfma,
               simd_cycles(ms_mfma), wc_mfma, ms_only, simd_cycles(ms_only), wc_only, ms_mfma / ms_mad, mode == 0 ? "," : "");
        CHECK(hipFree(d));
        CHECK(hipFree(cyc));
        CHECK(hipFree(o));
    }
    printf("}\n");
    return 0;
}
