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
GL_HD uint64_t add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) {  // wrapped: 2^64 ≡ EPS
        s += EPS;
        if (s < EPS) s += EPS;  // second wrap only when both inputs were >= 2^64 - EPS; cannot wrap again
    }
    return s;
}

GL_HD uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    if (a < b) {  // borrowed: -2^64 ≡ -EPS
        uint64_t e = d - EPS;
        if (d < EPS) e -= EPS;
        d = e;
    }
    return d;
}

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

GL_HD uint64_t mul(uint64_t a, uint64_t b) {
    unsigned __int128 m = (unsigned __int128)a * b;
    return reduce128((uint64_t)m, (uint64_t)(m >> 64));
}

GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }

// x + 2^64 * hi where hi < 2^32 (used for small-constant dot products such as the Poseidon MDS layer)
GL_HD uint64_t reduce96(uint64_t lo, uint32_t hi) {
    uint64_t t1 = ((uint64_t)hi << 32) - hi;
    uint64_t t2 = lo + t1;
    if (t2 < lo) t2 += EPS;
    return t2;
}

GL_HD uint64_t pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
        e >>= 1;
    }
    return r;
}

GL_HD uint64_t inv(uint64_t a) { return pow(a, P - 2); }

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
