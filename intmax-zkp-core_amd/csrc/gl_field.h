// gl_field.h — Goldilocks field arithmetic for gfx950 kernels (p = 2^64 - 2^32 + 1).
//
// Replaces plonky2 @ f99ed9c field/src/goldilocks_field.rs (GoldilocksField add/sub/mul/reduce128) and
// field/src/extension/quadratic.rs on the device.  Values are carried as arbitrary u64 representatives
// ("non-canonical allowed", exactly like GoldilocksField(u64) — reference src/bin/block_circuit.rs:506 feeds
// from_noncanonical_u64); kernels canonicalise once, on store.
//
// All functions are __host__ __device__ so the same arithmetic can be unit-tested on the CPU build box
// (tests/test_device_arith_host.py compiles csrc/host_selftest.cpp against this header).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ull;
constexpr uint64_t EPS = 0xFFFFFFFFull;  // 2^64 mod p
constexpr uint64_t TWO_ADIC_ROOT = 1753635133440165772ull;  // 7^((p-1)/2^32), order 2^32
constexpr uint64_t COSET_SHIFT = 7;                        // multiplicative generator

GL_HD uint64_t canon(uint64_t x) { return x >= P ? x - P : x; }

// a + b mod p for arbitrary u64 representatives; result is an arbitrary u64 representative.
GL_HD uint64_t add_c(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) {  // wrapped: 2^64 ≡ EPS
        s += EPS;
        if (s < EPS) s += EPS;  // second wrap only when both inputs were >= 2^64 - EPS; cannot wrap again
    }
    return s;
}

GL_HD uint64_t sub_c(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    if (a < b) {  // borrowed: -2^64 ≡ -EPS
        uint64_t e = d - EPS;
        if (d < EPS) e -= EPS;
        d = e;
    }
    return d;
}
#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_ASM_ADDSUB)
// The same two functions as 32-bit carry chains whose wrap corrections ride the carry inputs (-DGL_ASM_ADDSUB): + EPS = low word
// minus the carry bit, high word plus it unless the low word borrowed; six VALU instructions each (as compiled: eight).  A closed
// experiment, as in round 2 with another formulation: bit-identical (261 GPU parity tests), the fused quotient kernel 4 % fewer
// VALU instructions and 128 instead of 115 registers — and 19.3 ms against 18.7, the transforms +0.2 ms (profiles/r06_asm_addsub.txt):
// a dependent chain with three wait states where the compiled form has two independent halves.  Not built by default.
__device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) {
    uint32_t s0, s1;
    uint64_t c, w, k, m;
    asm("v_add_co_u32_e64 %[s0], %[c], %[a0], %[b0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %[s1], %[w], %[a1], %[b1], %[c]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32_e64 %[s0], %[k], %[s0], 0, %[w]\n\t"
        "s_andn2_b64 %[m], %[w], %[k]\n\t"
        "v_addc_co_u32_e64 %[s1], %[w], %[s1], 0, %[m]\n\t"  // carries again only when both inputs were >= 2^64 - EPS
        "s_nop 1\n\t"
        "v_subb_co_u32_e64 %[s0], %[k], %[s0], 0, %[w]\n\t"
        "s_andn2_b64 %[m], %[w], %[k]\n\t"
        "v_addc_co_u32_e64 %[s1], %[w], %[s1], 0, %[m]"
        : [s0] "=&v"(s0), [s1] "=&v"(s1), [c] "=&s"(c), [w] "=&s"(w), [k] "=&s"(k), [m] "=&s"(m)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [b0] "v"((uint32_t)b), [b1] "v"((uint32_t)(b >> 32))
        : "scc");
    return ((uint64_t)s1 << 32) | s0;
}
__device__ __forceinline__ uint64_t sub(uint64_t a, uint64_t b) {
    uint32_t d0, d1;
    uint64_t c, w, k, m;
    asm("v_sub_co_u32_e64 %[d0], %[c], %[a0], %[b0]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32_e64 %[d1], %[w], %[a1], %[b1], %[c]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %[d0], %[k], %[d0], 0, %[w]\n\t"  // - EPS = - 2^32 + 1 where the difference borrowed
        "s_andn2_b64 %[m], %[w], %[k]\n\t"
        "v_subb_co_u32_e64 %[d1], %[w], %[d1], 0, %[m]\n\t"  // borrows again only when the difference was < EPS
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %[d0], %[k], %[d0], 0, %[w]\n\t"
        "s_andn2_b64 %[m], %[w], %[k]\n\t"
        "v_subb_co_u32_e64 %[d1], %[w], %[d1], 0, %[m]"
        : [d0] "=&v"(d0), [d1] "=&v"(d1), [c] "=&s"(c), [w] "=&s"(w), [k] "=&s"(k), [m] "=&s"(m)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [b0] "v"((uint32_t)b), [b1] "v"((uint32_t)(b >> 32))
        : "scc");
    return ((uint64_t)d1 << 32) | d0;
}
#else
GL_HD uint64_t add(uint64_t a, uint64_t b) { return add_c(a, b); }
GL_HD uint64_t sub(uint64_t a, uint64_t b) { return sub_c(a, b); }
#endif

GL_HD uint64_t neg(uint64_t a) { return sub(0, a); }

// x = lo + 2^64 * hi  ->  x mod p (arbitrary representative).  2^64 ≡ 2^32 - 1, 2^96 ≡ -1.
GL_HD uint64_t reduce128(uint64_t lo, uint64_t hi) {
    uint64_t hi_hi = hi >> 32, hi_lo = hi & EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;
    uint64_t t1 = (hi_lo << 32) - hi_lo;  // hi_lo * EPS, fits in 64 bits
    uint64_t t2 = t0 + t1;
    if (t2 < t0) t2 += EPS;
    return t2;
}

// portable form (host, and the cross-check of the device form in tools/mul_bench.hip)
GL_HD uint64_t mul_c(uint64_t a, uint64_t b) {
    unsigned __int128 m = (unsigned __int128)a * b;
    return reduce128((uint64_t)m, (uint64_t)(m >> 64));
}

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 form of the same product.  The compiler's code for the __int128 formulation spends more issue slots on moving
// 32-bit halves into aligned register pairs (64-bit operands must be even-aligned) and on 64-bit compares than on the
// multiplies.  Here the four 32 x 32 products are v_mad_u64_u32 with their natural alignment, the limbs are joined by
// 32-bit adds whose carries live in SGPR pairs, and the two Goldilocks corrections (borrow of "lo - hi_hi", carry of
// "+ hi_lo * (2^32 - 1)") are applied together:
//     a*b = X + 2^32 Y + 2^64 Z,  X = a0 b0, Y = a0 b1 + a1 b0 (carry cY), Z = a1 b1
//     lo = (x0, x1 + y0), hi_lo = y1 + z0 + c1, hi_hi = z1 + cY + c4          (the 128-bit product, exactly; cY is
//                                                                               subtracted as the borrow-in of "lo - ...")
//     r  = lo - hi_hi + hi_lo * EPS + (c2 - b) * EPS                            (2^64 = EPS, 2^96 = -1 mod p)
// gfx940+ needs two wait states between a VALU write of an SGPR and a VALU read of it as a carry / mask, and the
// compiler cannot see inside asm blocks: every such pair below is separated explicitly, and a block that consumes a carry
// of an earlier block does so no sooner than its third issue slot.
#if defined(GL_EXPERIMENT_NO_NOPS)  // timing experiment only: results are wrong without the wait states
#define GL_NOP1
#define GL_NOP0
#else
#define GL_NOP1 "s_nop 1\n\t"
#define GL_NOP0 "s_nop 0\n\t"
#endif
__device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) {
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
        "v_subb_co_u32_e64 %[t0], %[bb], %[x0], %[k], %[cY]\n\t"  // lo - (z1 + c4) - cY: the third term rides the borrow-in
        GL_NOP1
        "v_subb_co_u32_e64 %[t1], %[bw], %[h], 0, %[bb]"
        : [h] "=&v"(h), [m] "=&v"(m), [k] "=&v"(k), [t0] "=&v"(t0), [t1] "=&v"(t1), [c1] "=&s"(c1), [c4] "=&s"(c4),
          [cd] "=&s"(cd), [bb] "=&s"(bb), [bw] "=&s"(bw)
        : [x0] "v"((uint32_t)X), [x1] "v"((uint32_t)(X >> 32)), [y0] "v"((uint32_t)Y), [y1] "v"((uint32_t)(Y >> 32)),
          [z0] "v"((uint32_t)Z), [z1] "v"((uint32_t)(Z >> 32)), [cY] "s"(cY));
    const uint64_t T = ((uint64_t)t1 << 32) | t0;
    uint64_t G, c2;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(G), "=s"(c2) : "v"(m), "v"(T));
#if !defined(GL_MUL_SELECT_TAIL)
    // The two corrections as ONE signed count (round 6): sg = [carry of the multiply-add] - [borrow of lo - hi_hi] in {-1, 0, 1},
    // R = G + sg EPS = (G - sg) + (sg << 32) — an add-with-carry, a subtract-with-borrow, a signed multiply-add and a 32-bit add on the
    // high word, where rounds 2-5 had two scalar and-nots, three selects and a 64-bit add (-DGL_MUL_SELECT_TAIL): the same fourteen
    // VALU instructions, 57.7 instead of 60.4 SIMD-cycles per multiplication, 230 instead of 240 per S-box (tools/mul_bench.hip, which
    // checks both forms on 4.19 M operand pairs incl. every edge pair).  Both forms add sg EPS modulo 2^64: the same value.
    uint32_t sg;
    uint64_t d0, R1;
    asm(GL_NOP1
        "v_addc_co_u32_e64 %[sg], %[d0], 0, 0, %[c2]\n\t"
        "v_subb_co_u32_e64 %[sg], %[d0], %[sg], 0, %[bw]\n\t"
        "v_mad_i64_i32 %[R1], %[d0], %[sg], -1, %[G]"
        : [sg] "=&v"(sg), [d0] "=&s"(d0), [R1] "=&v"(R1)
        : [c2] "s"(c2), [bw] "s"(bw), [G] "v"(G));
    const uint32_t r1 = (uint32_t)(R1 >> 32) + sg;
    return ((uint64_t)r1 << 32) | (uint32_t)R1;
#else
    uint32_t e0, e1;
    uint64_t ma, ms;
    asm("s_andn2_b64 %[ma], %[c2], %[bw]\n\t"
        "s_andn2_b64 %[ms], %[bw], %[c2]\n\t"
        "v_cndmask_b32_e64 %[e0], 0, -1, %[ma]\n\t"
        "v_cndmask_b32_e64 %[e1], 0, -1, %[ms]\n\t"
        "v_cndmask_b32_e64 %[e0], %[e0], 1, %[ms]"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [ma] "=&s"(ma), [ms] "=&s"(ms)
        : [c2] "s"(c2), [bw] "s"(bw)
        : "scc");
    const uint64_t E = ((uint64_t)e1 << 32) | e0;
    uint64_t R;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(R) : "v"(G), "v"(E));  // one 64-bit add (kept opaque: no re-association)
    return R;
#endif
}
#else
GL_HD uint64_t mul(uint64_t a, uint64_t b) { return mul_c(a, b); }
#endif

GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }

// Canonical butterfly (u, v) <- (u + v, u - v) for u, v < p, results < p: ten 32-bit instructions (twelve until round 6), u + v = u - (p - v) and
// x - y = (x - y mod 2^64) - (borrow ? 2^32 - 1 : 0), three interleaved carry chains: 51 SIMD-cycles per pair against 72 for
// gl::add + gl::sub (tools/addsub_bench.hip; every carry / select instruction costs the same ~4.4 cycles as a 64-bit add,
// tools/instr_bench.hip).  Used by the NTT passes that compute on canonical values (ntt_canon.h, gl_canon.h): a first trial with
// gl::canon behind every product and shift (+4 instructions each) came out even; with the canonical tails of gl_canon.h (+0 .. 2)
// the coset pass runs 4.3 % and the fused A+B pass 1.7 % faster.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void bfly(uint64_t& u, uint64_t& v) {
    uint32_t ul = (uint32_t)u, uh = (uint32_t)(u >> 32), vl = (uint32_t)v, vh = (uint32_t)(v >> 32), t0, t1;
    uint64_t s0, s1, s2, s3;
    asm("v_sub_co_u32_e64 %[t0], %[s0], 1, %[vl]\n\t"              // p - v, low word
        "v_sub_co_u32_e64 %[vl], %[s1], %[ul], %[vl]\n\t"           // u - v
        "v_sub_co_u32_e64 %[ul], %[s2], %[ul], %[t0]\n\t"           // u - (p - v)
        "v_subb_co_u32_e64 %[t1], %[s0], -1, %[vh], %[s0]\n\t"
        "v_subb_co_u32_e64 %[vh], %[s1], %[uh], %[vh], %[s1]\n\t"
        "v_subb_co_u32_e64 %[uh], %[s2], %[uh], %[t1], %[s2]\n\t"
        GL_NOP0
        // + p = - 2^32 + 1 where the difference borrowed: low word plus the borrow bit, high word minus it unless the low word
        // carried (round 6: four VALU instructions and two scalar ones; two selects, two subtracts and two subtracts-with-borrow before)
        "v_addc_co_u32_e64 %[vl], %[s0], %[vl], 0, %[s1]\n\t"
        "v_addc_co_u32_e64 %[ul], %[s3], %[ul], 0, %[s2]\n\t"
        "s_andn2_b64 %[s1], %[s1], %[s0]\n\t"
        "s_andn2_b64 %[s2], %[s2], %[s3]\n\t"
        "v_subb_co_u32_e64 %[vh], %[s0], %[vh], 0, %[s1]\n\t"
        "v_subb_co_u32_e64 %[uh], %[s3], %[uh], 0, %[s2]"
        : [ul] "+v"(ul), [uh] "+v"(uh), [vl] "+v"(vl), [vh] "+v"(vh), [t0] "=&v"(t0), [t1] "=&v"(t1), [s0] "=&s"(s0),
          [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)
        :
        : "scc");
    u = ((uint64_t)uh << 32) | ul;
    v = ((uint64_t)vh << 32) | vl;
}
#else
GL_HD void bfly(uint64_t& u, uint64_t& v) {
    const uint64_t a = u, b = v, nb = P - b;
    uint64_t s = a - nb, d = a - b;
    if (a < nb) s -= EPS;
    if (a < b) d -= EPS;
    u = s;
    v = d;
}
#endif

// x + 2^64 * hi where hi < 2^32 (used for small-constant dot products such as the Poseidon MDS layer)
GL_HD uint64_t reduce96(uint64_t lo, uint32_t hi) {
    uint64_t t1 = ((uint64_t)hi << 32) - hi;
    uint64_t t2 = lo + t1;
    if (t2 < lo) t2 += EPS;
    return t2;
}

// x * 2^S by shifts for a compile-time 0 < S < 96 (2 has order 192 in the Goldilocks field: 2^96 = -1, so every 64th root
// of unity is +-2^(3k); the twiddles inside radix-8 / radix-16 butterflies are never general multiplications)
#if defined(__HIP_DEVICE_COMPILE__)
// 32 < S < 64 on the device: x * 2^S = u0 2^32 + m 2^64 + z1 2^96 with u0 = lo(x) << s, m = (x >> (32 - s)) mod 2^32 (the two
// pieces that meet in the 2^64 limb do not overlap: no carry), z1 = hi(x) >> (32 - s), s = S - 32; reduced like gl::mul's product:
// T = (u0 : 0) - z1 (borrow bw), G = T + m (2^32 - 1) (carry c2), + or - EPS.  10 VALU instructions; the portable form below
// compiles to 17 — and 2^48 and its neighbours are 15 of the 27 shift twiddles of a radix-16 + radix-8 tile pass.
template <int S>
__device__ __forceinline__ uint64_t mul_2exp_mid(uint64_t x) {
    constexpr int s = S - 32;
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
    uint32_t e0, e1;
    uint64_t ma, ms;
    asm("s_andn2_b64 %[ma], %[c2], %[bw]\n\t"
        "s_andn2_b64 %[ms], %[bw], %[c2]\n\t"
        "v_cndmask_b32_e64 %[e0], 0, -1, %[ma]\n\t"
        "v_cndmask_b32_e64 %[e1], 0, -1, %[ms]\n\t"
        "v_cndmask_b32_e64 %[e0], %[e0], 1, %[ms]"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [ma] "=&s"(ma), [ms] "=&s"(ms)
        : [c2] "s"(c2), [bw] "s"(bw)
        : "scc");
    const uint64_t E = ((uint64_t)e1 << 32) | e0;
    uint64_t R;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(R) : "v"(G), "v"(E));
    return R;
}
#endif

template <int S>
GL_HD uint64_t mul_2exp(uint64_t x) {
    static_assert(S > 0 && S < 96, "shift out of range");
    if (S <= 32) return reduce96(x << S, (uint32_t)(x >> (64 - S)));
    if (S < 64) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_NO_SHIFT_ASM)
        return mul_2exp_mid<(S > 32 && S < 64) ? S : 33>(x);
#endif
        // x * 2^S = lo + 2^64 hl + 2^96 hh:  lo - hh + hl * EPS;  lo is 0 or >= 2^S > hh, so a borrow leaves t >= 2^64 - 2^32
        const uint64_t lo = x << S, hh = x >> ((96 - S) & 63);
        uint64_t t = lo - hh;
        if (lo < hh) t -= EPS;
        return reduce96(t, (uint32_t)(x >> ((64 - S) & 63)));
    }
    // x * 2^S = 2^64 Y, Y = x << (S - 64) = y0 + 2^32 (Y >> 32):  y0 * EPS - (Y >> 32),  Y >> 32 = x >> (96 - S) < 2^63 < p
    return reduce96(P - (x >> ((96 - S) & 63)), (uint32_t)(x << ((S - 64) & 31)));
}
GL_HD uint64_t mul_2exp24(uint64_t x) { return mul_2exp<24>(x); }
GL_HD uint64_t mul_2exp48(uint64_t x) { return mul_2exp<48>(x); }
GL_HD uint64_t mul_2exp72(uint64_t x) { return mul_2exp<72>(x); }

GL_HD uint64_t pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
        e >>= 1;
    }
    return r;
}

// a^(p - 2) by an addition chain: with e_k = a^(2^k - 1),  p - 2 = 2^64 - 2^32 - 1 = (2^31 - 1) 2^33 + (2^32 - 1), so the inverse is
// e_31^(2^33) * e_32 — 64 squarings and 9 multiplications (square-and-multiply over the bits of p - 2: 64 + 63)
GL_HD uint64_t sqr_n(uint64_t a, int n) {
    for (int i = 0; i < n; ++i) a = sqr(a);
    return a;
}
GL_HD uint64_t inv(uint64_t a) {
    const uint64_t e2 = mul(sqr(a), a);
    const uint64_t e3 = mul(sqr(e2), a);
    const uint64_t e6 = mul(sqr_n(e3, 3), e3);
    const uint64_t e12 = mul(sqr_n(e6, 6), e6);
    const uint64_t e24 = mul(sqr_n(e12, 12), e12);
    const uint64_t e30 = mul(sqr_n(e24, 6), e6);
    const uint64_t e31 = mul(sqr(e30), a);
    const uint64_t e32 = mul(sqr(e31), a);
    return mul(sqr_n(e31, 33), e32);
}

// the same chain with its squaring runs as rolled loops: ~20x less code for callers where the inversion is a small part of
// the work but code size matters (quotient_fused.h)
GL_HD uint64_t inv_compact(uint64_t a) {
    // e_k = a^(2^k - 1):  e_{2k} = e_k^(2^k) e_k,  e_{k+1} = e_k^2 a.   k: 1 2 3 6 12 24 30 31 32
    uint64_t e = a, e6 = 0, e31 = 0;
    const int steps[8] = {1, 1, 3, 6, 12, 6, 1, 1};  // squarings before each multiplication
    uint64_t mulby = a;                              // e_1
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int s = 0; s < 8; ++s) {
        // multiplier: s = 0,1 -> a (k: 1->2->3);  s = 2 -> e3 (3->6);  3 -> e6 (6->12);  4 -> e12 (12->24);  5 -> e6 (24->30);  6,7 -> a
        const uint64_t prev = e;
        uint64_t t = e;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int i = 0; i < steps[s]; ++i) t = sqr(t);
        const uint64_t m = (s == 2 || s == 3 || s == 4) ? prev : (s == 5 ? e6 : a);
        e = mul(t, m);
        if (s == 2) e6 = e;
        if (s == 6) e31 = e;
        (void)mulby;
    }
    // e = e_32, e31 saved:  a^(p-2) = e_31^(2^33) e_32
    uint64_t t = e31;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 0; i < 33; ++i) t = sqr(t);
    return mul(t, e);
}

// ---- quadratic extension F[X]/(X^2 - 7) ----
struct ext2 {
    uint64_t a, b;  // a + b X
};

GL_HD ext2 ext_add(ext2 x, ext2 y) { return {add(x.a, y.a), add(x.b, y.b)}; }
GL_HD ext2 ext_sub(ext2 x, ext2 y) { return {sub(x.a, y.a), sub(x.b, y.b)}; }
GL_HD uint64_t mul7(uint64_t x) {  // 7x = 8x - x
    uint64_t lo = x << 3;
    uint32_t hi = (uint32_t)(x >> 61);
    return sub(reduce96(lo, hi), x);
}
GL_HD ext2 ext_mul(ext2 x, ext2 y) {
    uint64_t c0 = add(mul(x.a, y.a), mul7(mul(x.b, y.b)));
    uint64_t c1 = add(mul(x.a, y.b), mul(x.b, y.a));
    return {c0, c1};
}
GL_HD ext2 ext_mul_base(ext2 x, uint64_t s) { return {mul(x.a, s), mul(x.b, s)}; }
GL_HD ext2 ext_inv(ext2 x) {
    uint64_t norm = sub(mul(x.a, x.a), mul7(mul(x.b, x.b)));
    uint64_t ni = inv(norm);
    return {mul(x.a, ni), mul(neg(x.b), ni)};
}
GL_HD ext2 ext_canon(ext2 x) { return {canon(x.a), canon(x.b)}; }

}  // namespace gl
