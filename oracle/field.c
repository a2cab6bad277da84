/*
 * field.c — CPU ORACLE (test infrastructure): Goldilocks base field and quadratic extension.
 * Follows plonky2 @ f99ed9c field/src/goldilocks_field.rs (reduce128, add/sub with the 2^64 ≡ 2^32-1 identity)
 * and field/src/extension/quadratic.rs (W = 7).  The dependency source is absent from /root/reference
 * (reference Cargo.toml:12); semantics restated from SURVEY.md App. A.1.
 */
#include "gl_oracle.h"

typedef unsigned __int128 u128;

uint64_t glo_canon(uint64_t x) { return x >= GLO_P ? x - GLO_P : x; }

uint64_t glo_add(uint64_t a, uint64_t b) {
    a = glo_canon(a);
    b = glo_canon(b);
    uint64_t s = a + b;
    if (s < a || s >= GLO_P) s -= GLO_P;
    return s;
}

uint64_t glo_sub(uint64_t a, uint64_t b) {
    a = glo_canon(a);
    b = glo_canon(b);
    return a >= b ? a - b : a - b + GLO_P;
}

/* plonky2 goldilocks_field.rs::reduce128: x = lo + 2^64*(hi_hi*2^32 + hi_lo) ≡ lo - hi_hi + EPS*hi_lo */
static uint64_t reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GLO_EPS;
    uint64_t t0, t2; /* the two corrections without branches (they are taken about half of the time: unpredictable) */
    uint64_t brw = __builtin_sub_overflow(lo, hi_hi, &t0);
    t0 -= (0 - brw) & GLO_EPS;
    uint64_t cry = __builtin_add_overflow(t0, hi_lo * GLO_EPS, &t2);
    t2 += (0 - cry) & GLO_EPS;
    return glo_canon(t2);
}

uint64_t glo_mul(uint64_t a, uint64_t b) { return reduce128((u128)a * b); }

uint64_t glo_mul_slow(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % GLO_P); }

/* out[i] = a[i] * b[i] mod p by the 128-bit remainder (vector form for the GPU parity test of the device multiply) */
void glo_mul_vec(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) out[i] = glo_mul_slow(a[i], b[i]);
}

uint64_t glo_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    a = glo_canon(a);
    while (e) {
        if (e & 1) r = glo_mul(r, a);
        a = glo_mul(a, a);
        e >>= 1;
    }
    return r;
}

uint64_t glo_inv(uint64_t a) { return glo_pow(a, GLO_P - 2); }

/* TWO_ADICITY = 32, POWER_OF_TWO_GENERATOR = 7^((p-1)/2^32) = 1753635133440165772 */
uint64_t glo_root_of_unity(unsigned log_n) {
    uint64_t g = 1753635133440165772ULL;
    for (unsigned i = log_n; i < 32; i++) g = glo_mul(g, g);
    return g;
}

void glo_ext_add(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    out[0] = glo_add(a[0], b[0]);
    out[1] = glo_add(a[1], b[1]);
}

void glo_ext_sub(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    out[0] = glo_sub(a[0], b[0]);
    out[1] = glo_sub(a[1], b[1]);
}

/* (a0 + a1 X)(b0 + b1 X) mod X^2 - 7 */
void glo_ext_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    uint64_t c0 = glo_add(glo_mul(a[0], b[0]), glo_mul(7, glo_mul(a[1], b[1])));
    uint64_t c1 = glo_add(glo_mul(a[0], b[1]), glo_mul(a[1], b[0]));
    out[0] = c0;
    out[1] = c1;
}

/* 1/(a0 + a1 X) = (a0 - a1 X) / (a0^2 - 7 a1^2) */
void glo_ext_inv(const uint64_t a[2], uint64_t out[2]) {
    uint64_t norm = glo_sub(glo_mul(a[0], a[0]), glo_mul(7, glo_mul(a[1], a[1])));
    uint64_t ninv = glo_inv(norm);
    out[0] = glo_mul(glo_canon(a[0]), ninv);
    out[1] = glo_mul(glo_sub(0, a[1]), ninv);
}
