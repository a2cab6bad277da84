// gl_canon.h — canonical-in / canonical-out Goldilocks arithmetic for the NTT butterflies (gfx950 device code only).
//
// gl_field.h carries arbitrary u64 representatives and keeps both wrap corrections in add / sub: (u + v, u - v) compiles to 16
// instructions.  With every value < p the pair is gl::bfly's 12 — IF the products and the shift twiddles that feed the butterflies
// come out canonical at no more than a couple of instructions each.  They can: in the reductions below the two corrected cases
// (carry: + EPS, borrow: - EPS) land below p by construction, and only the uncorrected sum G can exceed p - 1, exactly when
// G = (0xFFFFFFFF, lo >= 1); that case joins the + EPS case (G + EPS wraps to G - p) for two compares.
//   mulc           16 instructions (gl::mul: 14)
//   mul_2exp_c<S>   8 (S <= 32), 12 (32 < S < 64), 10 (S >= 64)      (gl::mul_2exp + gl::canon: 12 / 14 / 13)
// Replaces, on the device, the same plonky2 arithmetic as gl_field.h (field/src/goldilocks_field.rs), for the transforms reached
// through PolynomialBatch::from_values (reference src/rollup/circuits/mod.rs:1247).
#pragma once
#include "gl_field.h"

#if defined(__HIP_DEVICE_COMPILE__)
namespace glc {

// G + (c2 & !bw ? EPS : bw & !c2 ? -EPS : G >= p ? EPS : 0)   (mod 2^64): < p
__device__ __forceinline__ uint64_t tail2(uint64_t G, uint64_t c2, uint64_t bw) {
    uint32_t e0, e1;
    uint64_t ma, ms, g1, g2;
    asm("v_cmp_eq_u32_e64 %[g1], -1, %[gh]\n\t"
        "v_cmp_ne_u32_e64 %[g2], 0, %[gl]\n\t"
        "s_andn2_b64 %[ms], %[bw], %[c2]\n\t"
        "s_andn2_b64 %[ma], %[c2], %[bw]\n\t"
        "s_and_b64 %[g1], %[g1], %[g2]\n\t"
        "s_andn2_b64 %[g1], %[g1], %[ms]\n\t"
        "s_or_b64 %[ma], %[ma], %[g1]\n\t"
        "v_cndmask_b32_e64 %[e1], 0, -1, %[ms]\n\t"
        "v_cndmask_b32_e64 %[e0], 0, -1, %[ma]\n\t"
        "v_cndmask_b32_e64 %[e0], %[e0], 1, %[ms]"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [ma] "=&s"(ma), [ms] "=&s"(ms), [g1] "=&s"(g1), [g2] "=&s"(g2)
        : [c2] "s"(c2), [bw] "s"(bw), [gh] "v"((uint32_t)(G >> 32)), [gl] "v"((uint32_t)G)
        : "scc");
    const uint64_t E = ((uint64_t)e1 << 32) | e0;
    uint64_t R;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(R) : "v"(G), "v"(E));
    return R;
}

// G + (c2 | G >= p ? EPS : 0)   (mod 2^64): < p   (reductions without a borrow)
__device__ __forceinline__ uint64_t tail1(uint64_t G, uint64_t c2) {
    uint32_t r0, r1;
    uint64_t g1, g2, cc;
    // + EPS = + 2^32 - 1 under the mask: low word minus the mask bit, high word plus it unless the low word borrowed
    asm("v_cmp_eq_u32_e64 %[g1], -1, %[gh]\n\t"
        "v_cmp_ne_u32_e64 %[g2], 0, %[gl]\n\t"
        "s_and_b64 %[g1], %[g1], %[g2]\n\t"
        "s_or_b64 %[g1], %[g1], %[c2]\n\t"
        "v_subb_co_u32_e64 %[r0], %[cc], %[gl], 0, %[g1]\n\t"
        "s_andn2_b64 %[g1], %[g1], %[cc]\n\t"
        "v_addc_co_u32_e64 %[r1], %[cc], %[gh], 0, %[g1]"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [g1] "=&s"(g1), [g2] "=&s"(g2), [cc] "=&s"(cc)
        : [c2] "s"(c2), [gh] "v"((uint32_t)(G >> 32)), [gl] "v"((uint32_t)G)
        : "scc");
    return ((uint64_t)r1 << 32) | r0;
}

// a * b mod p, < p  (gl::mul with the canonical tail)
__device__ __forceinline__ uint64_t mulc(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t X = (uint64_t)a0 * b0, U = (uint64_t)a0 * b1, Z = (uint64_t)a1 * b1;
    uint64_t Y, cY;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(Y), "=s"(cY) : "v"(a1), "v"(b0), "v"(U));
    uint32_t h, m, k, t0, t1;
    uint64_t c1, c4, cd, bb, bw;
    asm("v_add_co_u32_e64 %[h], %[c1], %[x1], %[y0]\n\t"
        GL_NOP1
        "v_addc_co_u32_e64 %[m], %[c4], %[y1], %[z0], %[c1]\n\t"
        GL_NOP1
        "v_addc_co_u32_e64 %[k], %[cd], %[z1], 0, %[c4]\n\t"
        "v_subb_co_u32_e64 %[t0], %[bb], %[x0], %[k], %[cY]\n\t"
        GL_NOP1
        "v_subb_co_u32_e64 %[t1], %[bw], %[h], 0, %[bb]"
        : [h] "=&v"(h), [m] "=&v"(m), [k] "=&v"(k), [t0] "=&v"(t0), [t1] "=&v"(t1), [c1] "=&s"(c1), [c4] "=&s"(c4),
          [cd] "=&s"(cd), [bb] "=&s"(bb), [bw] "=&s"(bw)
        : [x0] "v"((uint32_t)X), [x1] "v"((uint32_t)(X >> 32)), [y0] "v"((uint32_t)Y), [y1] "v"((uint32_t)(Y >> 32)),
          [z0] "v"((uint32_t)Z), [z1] "v"((uint32_t)(Z >> 32)), [cY] "s"(cY));
    const uint64_t T = ((uint64_t)t1 << 32) | t0;
    uint64_t G, c2;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(G), "=s"(c2) : "v"(m), "v"(T));
    return tail2(G, c2, bw);
}

// x * 2^S mod p, < p, for any u64 x and a compile-time 0 < S < 96
template <int S>
__device__ __forceinline__ uint64_t mul_2exp_c(uint64_t x) {
    static_assert(S > 0 && S < 96, "shift out of range");
    if (S <= 32) {  // lo64 + h 2^64 = lo64 + h EPS
        const uint64_t lo = x << S;
        const uint32_t h = (uint32_t)(x >> (64 - S));
        uint64_t G, c2;
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(G), "=s"(c2) : "v"(h), "v"(lo));
        return tail1(G, c2);
    }
    if (S < 64) {  // see gl::mul_2exp_mid
        constexpr int s = S > 32 && S < 64 ? S - 32 : 1;
        const uint32_t a0 = (uint32_t)x, a1 = (uint32_t)(x >> 32);
        const uint32_t u0 = a0 << s, m = __builtin_amdgcn_alignbit(a1, a0, 32 - s), z1 = a1 >> (32 - s);
        uint32_t t0, t1;
        uint64_t bb, bw;
        asm("v_sub_co_u32_e64 %[t0], %[bb], 0, %[z1]\n\t"
            GL_NOP1
            "v_subb_co_u32_e64 %[t1], %[bw], %[u0], 0, %[bb]"
            : [t0] "=&v"(t0), [t1] "=&v"(t1), [bb] "=&s"(bb), [bw] "=&s"(bw)
            : [z1] "v"(z1), [u0] "v"(u0));
        const uint64_t T = ((uint64_t)t1 << 32) | t0;
        uint64_t G, c2;
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(G), "=s"(c2) : "v"(m), "v"(T));
        return tail2(G, c2, bw);
    }
    // x 2^S = 2^64 Y, Y = x << (S - 64) = y0 + 2^32 Yhi:  y0 EPS - Yhi = (p - Yhi) + y0 EPS,  Yhi = x >> (96 - S) < 2^63
    const uint64_t yhi = x >> ((96 - S) & 63);
    const uint32_t y0 = (uint32_t)x << ((S - 64) & 31);
    uint32_t t0, t1;
    uint64_t bb, bx;
    asm("v_sub_co_u32_e64 %[t0], %[bb], 1, %[yl]\n\t"
        GL_NOP1
        "v_subb_co_u32_e64 %[t1], %[bx], -1, %[yh], %[bb]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [bb] "=&s"(bb), [bx] "=&s"(bx)
        : [yl] "v"((uint32_t)yhi), [yh] "v"((uint32_t)(yhi >> 32)));
    const uint64_t T = ((uint64_t)t1 << 32) | t0;  // p - Yhi: no borrow out (Yhi < p)
    uint64_t G, c2;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(G), "=s"(c2) : "v"(y0), "v"(T));
    return tail1(G, c2);
}

}  // namespace glc
#endif
