// ntt_canon.h — the per-thread radix rounds of ntt_index.h on canonical values (gl_canon.h): every x[j] < p on entry and on exit.
// Same transforms, same results mod p; (u + v, u - v) is gl::bfly's 12 instructions instead of 16, the products and shift twiddles
// pay 2 (0 .. 2) instructions each to stay canonical.  Device only; used by ntt_pass_ab_kernel<INVERSE, true> and the coset pass.
#pragma once
#include "gl_canon.h"
#include "ntt_index.h"

#if defined(__HIP_DEVICE_COMPILE__)
namespace nttc {

// (u, v) <- (u + v, (u - v) * 2^X), X = exponent of a 192nd root of unity (2^96 = -1: a negative power swaps the roles)
template <int X>
__device__ __forceinline__ void bfly_shift(uint64_t& u, uint64_t& v) {
    constexpr bool NEG = X >= 96;
    constexpr int S = NEG ? X - 96 : X;
    if (NEG) {
        gl::bfly(v, u);  // v = u + v, u = v - u
        const uint64_t sum = v;
        v = glc::mul_2exp_c<S>(u);
        u = sum;
    } else {
        gl::bfly(u, v);
        v = glc::mul_2exp_c<S>(v);
    }
}
template <bool INVERSE, int E>
__device__ __forceinline__ void bfly_w16(uint64_t& u, uint64_t& v) {  // * w_16^E, w_16 = 2^156 (forward) or 2^36 (inverse)
    bfly_shift<((INVERSE ? 36 : 156) * E) % 192>(u, v);
}
template <bool INVERSE, int J>
__device__ __forceinline__ void bfly_w8(uint64_t& u, uint64_t& v) {  // * w_8^J, w_8 = 2^120 (forward) or 2^72 (inverse)
    bfly_shift<((INVERSE ? 72 : 120) * J) % 192>(u, v);
}

// ntt::radix16_dif_ct on canonical values
template <bool INVERSE>
__device__ __forceinline__ void radix16_dif_ct(uint64_t (&x)[16], const uint64_t* w128, uint32_t r3) {
    gl::bfly(x[0], x[8]);
    bfly_w16<INVERSE, 1>(x[1], x[9]);
    bfly_w16<INVERSE, 2>(x[2], x[10]);
    bfly_w16<INVERSE, 3>(x[3], x[11]);
    bfly_w16<INVERSE, 4>(x[4], x[12]);
    bfly_w16<INVERSE, 5>(x[5], x[13]);
    bfly_w16<INVERSE, 6>(x[6], x[14]);
    bfly_w16<INVERSE, 7>(x[7], x[15]);
#pragma unroll
    for (int h = 0; h < 16; h += 8) {
        gl::bfly(x[h + 0], x[h + 4]);
        bfly_w16<INVERSE, 2>(x[h + 1], x[h + 5]);
        bfly_w16<INVERSE, 4>(x[h + 2], x[h + 6]);
        bfly_w16<INVERSE, 6>(x[h + 3], x[h + 7]);
    }
#pragma unroll
    for (int h = 0; h < 16; h += 4) {
        gl::bfly(x[h + 0], x[h + 2]);
        bfly_w16<INVERSE, 4>(x[h + 1], x[h + 3]);
    }
#pragma unroll
    for (int h = 0; h < 16; h += 2) gl::bfly(x[h], x[h + 1]);
#pragma unroll
    for (int q = 1; q < 16; ++q) {
        const uint32_t k1 = ((q & 1) << 3) | ((q & 2) << 1) | ((q & 4) >> 1) | ((q & 8) >> 3);
        x[q] = glc::mulc(x[q], w128[r3 * k1]);
    }
}

// ntt::radix8_dif_const on canonical values
template <bool INVERSE>
__device__ __forceinline__ void radix8_dif_const(uint64_t (&x)[8]) {
    gl::bfly(x[0], x[4]);
    bfly_w8<INVERSE, 1>(x[1], x[5]);
    bfly_w8<INVERSE, 2>(x[2], x[6]);
    bfly_w8<INVERSE, 3>(x[3], x[7]);
    gl::bfly(x[0], x[2]);
    bfly_w8<INVERSE, 2>(x[1], x[3]);
    gl::bfly(x[4], x[6]);
    bfly_w8<INVERSE, 2>(x[5], x[7]);
    gl::bfly(x[0], x[1]);
    gl::bfly(x[2], x[3]);
    gl::bfly(x[4], x[5]);
    gl::bfly(x[6], x[7]);
}

// ntt::radix8_dif_tw on canonical values
__device__ __forceinline__ void radix8_dif_tw(uint64_t (&x)[8], const uint64_t (&tw)[7]) {
#define GLC_BF(a, b, w) gl::bfly(x[a], x[b]), x[b] = glc::mulc(x[b], w)
    GLC_BF(0, 4, tw[0]);
    GLC_BF(1, 5, tw[1]);
    GLC_BF(2, 6, tw[2]);
    GLC_BF(3, 7, tw[3]);
    GLC_BF(0, 2, tw[4]);
    GLC_BF(1, 3, tw[5]);
    GLC_BF(4, 6, tw[4]);
    GLC_BF(5, 7, tw[5]);
    GLC_BF(0, 1, tw[6]);
    GLC_BF(2, 3, tw[6]);
    GLC_BF(4, 5, tw[6]);
    GLC_BF(6, 7, tw[6]);
#undef GLC_BF
}

// ntt::b_round on canonical values
template <bool INVERSE>
__device__ __forceinline__ void b_round(uint64_t (&x)[8], const uint64_t (&tw)[7]) {
#if GL_RADIX8_CT
    radix8_dif_const<INVERSE>(x);
#pragma unroll
    for (int q = 1; q < 8; ++q) x[q] = glc::mulc(x[q], tw[q - 1]);
#else
    radix8_dif_tw(x, tw);
#endif
}

}  // namespace nttc
#endif
