/*
 * fast_vec.h — CPU ORACLE, TIMED LEG ONLY (test infrastructure, NOT product code).
 *
 * Packed Goldilocks arithmetic for the vectorised form of the oracle that bench.py's `cpu_baseline` times: the same
 * algorithms as the scalar restatement (field.c / poseidon.c / ntt.c / plonk.c), VL independent field elements per
 * register.  One source, three instantiations chosen by GLV_BACKEND before this header is included:
 *     512  AVX-512F/DQ (8 lanes, unsigned compares into mask registers)
 *     256  AVX2        (4 lanes, unsigned compares by sign-bias)
 *     1    plain C     (1 lane; what a machine without AVX2 runs, and the reference the vector forms are diffed against)
 * This mirrors what plonky2 itself does on x86 (field/src/arch/x86_64/{avx2,avx512}_goldilocks_field.rs: packed
 * Goldilocks with the same 2^64 = 2^32 - 1 folding); the dependency source is absent from /root/reference
 * (Cargo.toml:12), the arithmetic below is written from the field's definition.
 *
 * The scalar path stays the CHECKER: tests/test_oracle_fast.py asserts vector == scalar on the reference's golden
 * digests, on 10^5 random permutation states and on whole proofs.
 *
 * Representation: every value is a u64 congruent to the element mod p (not necessarily < p), like plonky2's
 * GoldilocksField(u64).  v_add / v_sub canonicalise their SECOND operand (so any two representatives may be combined),
 * v_mul takes any representatives; v_canon gives the representative < p.
 */
#ifndef GLO_FAST_VEC_H
#define GLO_FAST_VEC_H

#include <stddef.h>
#include <stdint.h>

#define GLV_P 0xFFFFFFFF00000001ULL
#define GLV_EPS 0xFFFFFFFFULL

#if GLV_BACKEND == 512
#pragma GCC target("avx512f,avx512dq,avx512vl,avx512bw,bmi2")
#include <immintrin.h>
typedef __m512i V;
#define VL 8
#define VI static inline __attribute__((always_inline))
VI V v_set1(uint64_t x) { return _mm512_set1_epi64((long long)x); }
VI V v_load(const uint64_t *p) { return _mm512_loadu_si512((const void *)p); }
VI void v_store(uint64_t *p, V a) { _mm512_storeu_si512((void *)p, a); }
/* lane l = base[l * stride] */
VI V v_gather(const uint64_t *base, size_t stride) {
    const __m512i idx = _mm512_mullo_epi64(_mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0), _mm512_set1_epi64((long long)stride));
    return _mm512_i64gather_epi64(idx, (const void *)base, 8);
}
VI V v_gather_idx(const uint64_t *base, const uint64_t *idx) { return _mm512_i64gather_epi64(v_load(idx), (const void *)base, 8); }
VI void v_scatter_idx(uint64_t *base, const uint64_t *idx, V a) { _mm512_i64scatter_epi64((void *)base, v_load(idx), a, 8); }
VI V v_canon(V a) {
    const V P = v_set1(GLV_P);
    return _mm512_mask_sub_epi64(a, _mm512_cmpge_epu64_mask(a, P), a, P);
}
/* a any representative, b < p */
VI V v_add_c(V a, V b) {
    V s = _mm512_add_epi64(a, b);
    return _mm512_mask_add_epi64(s, _mm512_cmplt_epu64_mask(s, a), s, v_set1(GLV_EPS));
}
VI V v_sub_c(V a, V b) {
    V d = _mm512_sub_epi64(a, b);
    return _mm512_mask_sub_epi64(d, _mm512_cmplt_epu64_mask(a, b), d, v_set1(GLV_EPS));
}
VI V v_shr32(V a) { return _mm512_srli_epi64(a, 32); }
VI V v_shl32(V a) { return _mm512_slli_epi64(a, 32); }
VI V v_mul32(V a, V b) { return _mm512_mul_epu32(a, b); } /* low 32 bits of every lane, 64-bit products */
VI V v_add64(V a, V b) { return _mm512_add_epi64(a, b); }
VI V v_and(V a, V b) { return _mm512_and_si512(a, b); }
/* low 32 bits from lo, high 32 bits = low 32 bits of hi */
VI V v_pack_lo_hi(V lo, V hi) { return _mm512_mask_mov_epi32(lo, 0xAAAA, v_shl32(hi)); }
/* x + (carry out of the previous add ? EPS : 0): s = a + b computed by the caller */
VI V v_fix_carry(V s, V a) { return _mm512_mask_add_epi64(s, _mm512_cmplt_epu64_mask(s, a), s, v_set1(GLV_EPS)); }
VI V v_fix_borrow(V d, V a, V b) { return _mm512_mask_sub_epi64(d, _mm512_cmplt_epu64_mask(a, b), d, v_set1(GLV_EPS)); }
VI V v_carry_inc(V top, V s, V a) { return _mm512_mask_add_epi64(top, _mm512_cmplt_epu64_mask(s, a), top, v_set1(1)); }
VI int v_any_nonzero(V a) { return _mm512_test_epi64_mask(a, a) != 0; }
/* lanes where c == 0 take a, the others b */
VI V v_if_zero(V c, V a, V b) { return _mm512_mask_mov_epi64(b, _mm512_testn_epi64_mask(c, c), a); }
/* r[i] lane l <-> r[l] lane i */
VI void v_transpose(V r[8]) {
    V t[8], u[8];
    for (int i = 0; i < 4; i++) {
        t[2 * i] = _mm512_unpacklo_epi64(r[2 * i], r[2 * i + 1]);     /* even elements of the pair of rows */
        t[2 * i + 1] = _mm512_unpackhi_epi64(r[2 * i], r[2 * i + 1]); /* odd elements */
    }
    for (int o = 0; o < 2; o++) { /* o = 0: even element indices (0,2,4,6), o = 1: odd */
        u[0] = _mm512_shuffle_i64x2(t[o], t[2 + o], 0x88);     /* rows 0-3, elements o+0 and o+4 */
        u[1] = _mm512_shuffle_i64x2(t[o], t[2 + o], 0xDD);     /* rows 0-3, elements o+2 and o+6 */
        u[2] = _mm512_shuffle_i64x2(t[4 + o], t[6 + o], 0x88); /* rows 4-7 */
        u[3] = _mm512_shuffle_i64x2(t[4 + o], t[6 + o], 0xDD);
        r[o + 0] = _mm512_shuffle_i64x2(u[0], u[2], 0x88);
        r[o + 4] = _mm512_shuffle_i64x2(u[0], u[2], 0xDD);
        r[o + 2] = _mm512_shuffle_i64x2(u[1], u[3], 0x88);
        r[o + 6] = _mm512_shuffle_i64x2(u[1], u[3], 0xDD);
    }
}

#elif GLV_BACKEND == 256
#pragma GCC target("avx2,bmi2")
#include <immintrin.h>
typedef __m256i V;
#define VL 4
#define VI static inline __attribute__((always_inline))
VI V v_set1(uint64_t x) { return _mm256_set1_epi64x((long long)x); }
VI V v_load(const uint64_t *p) { return _mm256_loadu_si256((const __m256i *)p); }
VI void v_store(uint64_t *p, V a) { _mm256_storeu_si256((__m256i *)p, a); }
VI V v_gather(const uint64_t *base, size_t stride) {
    return _mm256_set_epi64x((long long)base[3 * stride], (long long)base[2 * stride], (long long)base[stride], (long long)base[0]);
}
VI V v_gather_idx(const uint64_t *base, const uint64_t *idx) {
    return _mm256_set_epi64x((long long)base[idx[3]], (long long)base[idx[2]], (long long)base[idx[1]], (long long)base[idx[0]]);
}
VI void v_scatter_idx(uint64_t *base, const uint64_t *idx, V a) {
    uint64_t t[4];
    _mm256_storeu_si256((__m256i *)t, a);
    for (int l = 0; l < 4; l++) base[idx[l]] = t[l];
}
/* unsigned a < b as an all-ones / all-zeros lane mask */
VI V v_ltu(V a, V b) {
    const V bias = v_set1(0x8000000000000000ULL);
    return _mm256_cmpgt_epi64(_mm256_xor_si256(b, bias), _mm256_xor_si256(a, bias));
}
VI V v_canon(V a) {
    const V P = v_set1(GLV_P);
    V lt = v_ltu(a, P); /* a < p: keep */
    return _mm256_sub_epi64(a, _mm256_andnot_si256(lt, P));
}
VI V v_add_c(V a, V b) {
    V s = _mm256_add_epi64(a, b);
    return _mm256_add_epi64(s, _mm256_and_si256(v_ltu(s, a), v_set1(GLV_EPS)));
}
VI V v_sub_c(V a, V b) {
    V d = _mm256_sub_epi64(a, b);
    return _mm256_sub_epi64(d, _mm256_and_si256(v_ltu(a, b), v_set1(GLV_EPS)));
}
VI V v_shr32(V a) { return _mm256_srli_epi64(a, 32); }
VI V v_shl32(V a) { return _mm256_slli_epi64(a, 32); }
VI V v_mul32(V a, V b) { return _mm256_mul_epu32(a, b); }
VI V v_add64(V a, V b) { return _mm256_add_epi64(a, b); }
VI V v_and(V a, V b) { return _mm256_and_si256(a, b); }
VI V v_pack_lo_hi(V lo, V hi) { return _mm256_blend_epi32(lo, v_shl32(hi), 0xAA); }
VI V v_fix_carry(V s, V a) { return _mm256_add_epi64(s, _mm256_and_si256(v_ltu(s, a), v_set1(GLV_EPS))); }
VI V v_fix_borrow(V d, V a, V b) { return _mm256_sub_epi64(d, _mm256_and_si256(v_ltu(a, b), v_set1(GLV_EPS))); }
VI V v_carry_inc(V top, V s, V a) { return _mm256_sub_epi64(top, v_ltu(s, a)); } /* mask = -1 */
VI int v_any_nonzero(V a) { return !_mm256_testz_si256(a, a); }
VI V v_if_zero(V c, V a, V b) { return _mm256_blendv_epi8(b, a, _mm256_cmpeq_epi64(c, _mm256_setzero_si256())); }
VI void v_transpose(V r[4]) {
    V t0 = _mm256_unpacklo_epi64(r[0], r[1]), t1 = _mm256_unpackhi_epi64(r[0], r[1]);
    V t2 = _mm256_unpacklo_epi64(r[2], r[3]), t3 = _mm256_unpackhi_epi64(r[2], r[3]);
    r[0] = _mm256_permute2x128_si256(t0, t2, 0x20);
    r[2] = _mm256_permute2x128_si256(t0, t2, 0x31);
    r[1] = _mm256_permute2x128_si256(t1, t3, 0x20);
    r[3] = _mm256_permute2x128_si256(t1, t3, 0x31);
}

#elif GLV_BACKEND == 1
typedef uint64_t V;
#define VL 1
#define VI static inline __attribute__((always_inline))
VI V v_set1(uint64_t x) { return x; }
VI V v_load(const uint64_t *p) { return *p; }
VI void v_store(uint64_t *p, V a) { *p = a; }
VI V v_gather(const uint64_t *base, size_t stride) {
    (void)stride;
    return base[0];
}
VI V v_gather_idx(const uint64_t *base, const uint64_t *idx) { return base[idx[0]]; }
VI void v_scatter_idx(uint64_t *base, const uint64_t *idx, V a) { base[idx[0]] = a; }
VI V v_canon(V a) { return a >= GLV_P ? a - GLV_P : a; }
VI V v_add_c(V a, V b) {
    uint64_t s, c = __builtin_add_overflow(a, b, &s);
    return s + ((0 - c) & GLV_EPS);
}
VI V v_sub_c(V a, V b) {
    uint64_t d, c = __builtin_sub_overflow(a, b, &d);
    return d - ((0 - c) & GLV_EPS);
}
VI int v_any_nonzero(V a) { return a != 0; }
VI V v_if_zero(V c, V a, V b) { return c == 0 ? a : b; }
VI void v_transpose(V r[1]) { (void)r; }
#else
#error "GLV_BACKEND must be 512, 256 or 1"
#endif

VI V v_add(V a, V b) { return v_add_c(a, v_canon(b)); }
VI V v_sub(V a, V b) { return v_sub_c(a, v_canon(b)); }
VI V v_neg(V a) { return v_sub_c(v_set1(0), v_canon(a)); }
VI V v_dbl(V a) { return v_add_c(a, v_canon(a)); }

#if GLV_BACKEND == 1
VI V v_mul(V a, V b) { /* plonky2 goldilocks_field.rs::reduce128 */
    unsigned __int128 m = (unsigned __int128)a * b;
    uint64_t lo = (uint64_t)m, hi = (uint64_t)(m >> 64), t0, t2;
    uint64_t brw = __builtin_sub_overflow(lo, hi >> 32, &t0);
    t0 -= (0 - brw) & GLV_EPS;
    uint64_t cry = __builtin_add_overflow(t0, (hi & GLV_EPS) * GLV_EPS, &t2);
    return t2 + ((0 - cry) & GLV_EPS);
}
VI V v_sqr(V a) { return v_mul(a, a); }
/* a * c for a constant c < 2^32 */
VI V v_mul_small(V a, uint64_t c) { return v_mul(a, c); }
#else
/* (hi, lo) = the 128-bit value; x = lo + 2^64 (hi_hi 2^32 + hi_lo) = lo - hi_hi + EPS hi_lo  (2^64 = EPS, 2^96 = -1) */
VI V v_reduce128(V hi, V lo) {
    V hi_hi = v_shr32(hi);
    V t1 = v_mul32(hi, v_set1(GLV_EPS)); /* hi_lo * EPS < 2^64 */
    V t0 = v_fix_borrow(
#if GLV_BACKEND == 512
        _mm512_sub_epi64(lo, hi_hi),
#else
        _mm256_sub_epi64(lo, hi_hi),
#endif
        lo, hi_hi);
    V t2 = v_add64(t0, t1);
    return v_fix_carry(t2, t1);
}
VI V v_mul(V a, V b) { /* schoolbook on 32-bit halves: four 32 x 32 -> 64 products */
    V ah = v_shr32(a), bh = v_shr32(b);
    V ll = v_mul32(a, b), lh = v_mul32(a, bh), hl = v_mul32(ah, b), hh = v_mul32(ah, bh);
    V mid = v_add64(lh, v_shr32(ll));                       /* < 2^64: (2^32-1)^2 + 2^32 - 1 */
    V mid2 = v_add64(hl, v_and(mid, v_set1(GLV_EPS)));
    V lo = v_pack_lo_hi(ll, mid2);
    V hi = v_add64(v_add64(hh, v_shr32(mid)), v_shr32(mid2));
    return v_reduce128(hi, lo);
}
VI V v_sqr(V a) {
    V ah = v_shr32(a);
    V ll = v_mul32(a, a), lh = v_mul32(a, ah), hh = v_mul32(ah, ah);
    V mid = v_add64(lh, v_shr32(ll));
    V mid2 = v_add64(lh, v_and(mid, v_set1(GLV_EPS)));
    V lo = v_pack_lo_hi(ll, mid2);
    V hi = v_add64(v_add64(hh, v_shr32(mid)), v_shr32(mid2));
    return v_reduce128(hi, lo);
}
VI V v_mul_small(V a, uint64_t c) { /* c < 2^32: two products, value < 2^96 */
    V cv = v_set1(c);
    V ll = v_mul32(a, cv), hl = v_mul32(v_shr32(a), cv);
    V mid = v_add64(hl, v_shr32(ll));
    V lo = v_pack_lo_hi(ll, mid);
    V hi = v_shr32(mid); /* < 2^32: hi_hi = 0 */
    V t1 = v_mul32(hi, v_set1(GLV_EPS));
    V t2 = v_add64(lo, t1);
    return v_fix_carry(t2, t1);
}
#endif

#endif
