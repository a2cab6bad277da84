/*
 * poseidon.c — CPU ORACLE (test infrastructure): Poseidon-Goldilocks width 12, x^7, 4+22+4 rounds.
 *
 * Restates plonky2 @ f99ed9c src/hash/{poseidon.rs,poseidon_goldilocks.rs,hashing.rs} (dependency absent from
 * /root/reference, see reference Cargo.toml:12).  The reference uses these natively at
 * src/merkle_tree/tree.rs:70,79,90,123 (two_to_one), src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:170
 * (hash_pad), src/rollup/circuits/mod.rs:859 (hash_no_pad), src/zkdsa/account.rs:165 (two_to_one).
 *
 * Round constants are NOT copied from anywhere: they are re-derived at start-up as the first 360 draws of
 * `ChaCha8Rng::seed_from_u64(0)` mapped into [0,p) by rand 0.8's widening-multiply uniform sampler
 * (SURVEY.md §8c).  PINNED: tests/golden/poseidon_kat.json (36 digests from the reference's own tests).
 *
 * The "naive" round structure is used (full constant layer + full MDS in every partial round); plonky2's
 * fast partial rounds are an algebraically identical re-parameterisation.
 */
#include <string.h>

#include "gl_oracle.h"

typedef unsigned __int128 u128;

/* ---------- ChaCha8Rng::seed_from_u64(0) ---------- */
static uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static uint32_t rotr32(uint32_t x, unsigned n) {
    n &= 31;
    return n ? (x >> n) | (x << (32 - n)) : x;
}

#define QR(a, b, c, d)        \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7);

static void chacha8_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) in[4 + i] = key[i];
    in[12] = (uint32_t)counter;
    in[13] = (uint32_t)(counter >> 32);
    in[14] = 0;
    in[15] = 0;
    uint32_t x[16];
    memcpy(x, in, sizeof x);
    for (int r = 0; r < 4; r++) { /* 8 rounds = 4 double rounds */
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14])
        QR(x[3], x[7], x[11], x[15]) QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12])
        QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}

static uint64_t RC[360];
static int rc_ready = 0;

static void derive_round_constants(void) {
    /* rand_core::SeedableRng::seed_from_u64: PCG32 expansion of the u64 into a 32-byte key */
    uint64_t s = 0;
    uint32_t key[8];
    for (int i = 0; i < 8; i++) {
        s = s * 6364136223846793005ULL + 11634580027462260723ULL;
        uint32_t xorshifted = (uint32_t)(((s >> 18) ^ s) >> 27);
        key[i] = rotr32(xorshifted, (unsigned)(s >> 59));
    }
    uint64_t counter = 0;
    uint32_t blk[16];
    int have = 0, pos = 0, n = 0;
    while (n < 360) {
        if (pos >= have) {
            chacha8_block(key, counter++, blk);
            have = 16;
            pos = 0;
        }
        uint64_t v = (uint64_t)blk[pos] | ((uint64_t)blk[pos + 1] << 32); /* next_u64 = lo word, then hi word */
        pos += 2;
        u128 m = (u128)v * GLO_P; /* UniformInt::sample_single over 0..p : widening multiply */
        uint64_t lo = (uint64_t)m;
        if (lo <= GLO_P - 1) RC[n++] = (uint64_t)(m >> 64); /* zone = p-1 (p has no leading zeros) */
    }
    rc_ready = 1;
}

const uint64_t *glo_poseidon_round_constants(void) {
    if (!rc_ready) derive_round_constants();
    return RC;
}

/* ---------- permutation ---------- */
static const uint64_t MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static uint64_t sbox7(uint64_t x) {
    uint64_t x2 = glo_mul(x, x), x4 = glo_mul(x2, x2), x3 = glo_mul(x, x2);
    return glo_mul(x3, x4);
}

/* textbook form: 128-bit accumulate, reduce with the integer remainder (slow; cross-check only) */
static void mds_layer_slow(uint64_t st[12]) {
    uint64_t out[12];
    for (int r = 0; r < 12; r++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)st[(i + r) % 12] * MDS_CIRC[i];
        acc += (u128)st[r] * MDS_DIAG[r];
        out[r] = (uint64_t)(acc % GLO_P);
    }
    memcpy(st, out, sizeof out);
}

static void permute_with(uint64_t st[12], void (*mds)(uint64_t *)) {
    const uint64_t *rc = glo_poseidon_round_constants();
    for (int i = 0; i < 12; i++) st[i] = glo_canon(st[i]);
    for (int round = 0; round < 30; round++) {
        for (int i = 0; i < 12; i++) {
            uint64_t s = st[i] + rc[round * 12 + i]; /* both canonical */
            if (s < st[i] || s >= GLO_P) s -= GLO_P;
            st[i] = s;
        }
        if (round < 4 || round >= 26) {
            for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        } else {
            st[0] = sbox7(st[0]);
        }
        mds(st);
    }
}

/* The same textbook round structure once more, written for speed (it is what bench.py's cpu_baseline times): branch-free
 * carry corrections, representatives canonicalised only at the end, everything inlined.  glo_poseidon_permute_slow above
 * (integer-remainder MDS, canonical everywhere) stays as the cross-check; both are pinned by the golden digests. */
static inline uint64_t mul_nc(uint64_t a, uint64_t b) { /* a b mod p, any representative */
    u128 m = (u128)a * b;
    uint64_t lo = (uint64_t)m, hi = (uint64_t)(m >> 64), t0, t2;
    uint64_t brw = __builtin_sub_overflow(lo, hi >> 32, &t0);
    t0 -= (0 - brw) & GLO_EPS; /* borrowed: 2^64 = EPS */
    uint64_t cry = __builtin_add_overflow(t0, (hi & GLO_EPS) * GLO_EPS, &t2);
    return t2 + ((0 - cry) & GLO_EPS);
}
static inline uint64_t sbox7_nc(uint64_t x) {
    uint64_t x2 = mul_nc(x, x), x4 = mul_nc(x2, x2), x3 = mul_nc(x, x2);
    return mul_nc(x3, x4);
}
static inline uint64_t add_canon_nc(uint64_t a, uint64_t c) { /* c canonical: one correction suffices */
    uint64_t s, cry = __builtin_add_overflow(a, c, &s);
    return s + ((0 - cry) & GLO_EPS);
}
/* the MDS sums with the 32-bit halves accumulated separately (every partial sum < 2^42), then 2^64 = 2^32 - 1 */
static inline void mds_layer_nc(uint64_t st[12]) {
    uint64_t lo[24], hi[24], out[12];
    for (int i = 0; i < 12; i++) {
        lo[i] = lo[i + 12] = st[i] & GLO_EPS;
        hi[i] = hi[i + 12] = st[i] >> 32;
    }
    for (int r = 0; r < 12; r++) {
        uint64_t alo = 0, ahi = 0;
        for (int i = 0; i < 12; i++) {
            alo += lo[i + r] * MDS_CIRC[i];
            ahi += hi[i + r] * MDS_CIRC[i];
        }
        alo += lo[r] * MDS_DIAG[r];
        ahi += hi[r] * MDS_DIAG[r];
        u128 v = (u128)alo + ((u128)ahi << 32);
        uint64_t vlo = (uint64_t)v, t2;
        uint64_t cry = __builtin_add_overflow(vlo, (uint64_t)(v >> 64) * GLO_EPS, &t2); /* v >> 64 < 2^10 */
        out[r] = t2 + ((0 - cry) & GLO_EPS);
    }
    memcpy(st, out, sizeof out);
}

void glo_poseidon_permute(uint64_t st[12]) {
    const uint64_t *rc = glo_poseidon_round_constants();
    for (int round = 0; round < 30; round++) {
        for (int i = 0; i < 12; i++) st[i] = add_canon_nc(st[i], rc[round * 12 + i]);
        if (round < 4 || round >= 26) {
            for (int i = 0; i < 12; i++) st[i] = sbox7_nc(st[i]);
        } else {
            st[0] = sbox7_nc(st[0]);
        }
        mds_layer_nc(st);
    }
    for (int i = 0; i < 12; i++) st[i] = glo_canon(st[i]);
}
void glo_poseidon_permute_slow(uint64_t st[12]) { permute_with(st, mds_layer_slow); }

/* ---------- sponge (plonky2 hashing.rs::hash_n_to_m_no_pad, overwrite mode, rate 8) ---------- */
void glo_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; i++) st[i] = glo_canon(in[off + i]);
        glo_poseidon_permute(st);
    }
    memcpy(out, st, 4 * sizeof(uint64_t));
}

/* push 1; push 0 until (len+1) % 12 == 0; push 1; then hash_no_pad */
void glo_hash_pad(const uint64_t *in, size_t len, uint64_t out[4]) {
    uint64_t buf[512];
    size_t n = 0;
    if (len > 480) len = 480; /* oracle use only: SMT leaves are <= 12 elements */
    for (size_t i = 0; i < len; i++) buf[n++] = in[i];
    buf[n++] = 1;
    while ((n + 1) % 12 != 0) buf[n++] = 0;
    buf[n++] = 1;
    glo_hash_no_pad(buf, n, out);
}

void glo_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    glo_poseidon_permute(st);
    memcpy(out, st, 4 * sizeof(uint64_t));
}

void glo_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; i++) out[i] = i < len ? glo_canon(in[i]) : 0;
    } else {
        glo_hash_no_pad(in, len, out);
    }
}

/* ---------- HashOut hex codec: 4 x u64 LE -> 32 bytes, whole array reversed, "0x" + hex ---------- */
void glo_hashout_to_hex(const uint64_t h[4], char out[67]) {
    static const char *hx = "0123456789abcdef";
    out[0] = '0';
    out[1] = 'x';
    int p = 2;
    for (int e = 3; e >= 0; e--)
        for (int b = 7; b >= 0; b--) {
            unsigned byte = (unsigned)((h[e] >> (8 * b)) & 0xff);
            out[p++] = hx[byte >> 4];
            out[p++] = hx[byte & 15];
        }
    out[p] = 0;
}

static int hexval(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

int glo_hashout_from_hex(const char *hex, uint64_t h[4]) {
    if (hex[0] != '0' || hex[1] != 'x') return -1;
    hex += 2;
    size_t n = strlen(hex);
    if (n > 64 || (n & 1)) return -1;
    unsigned char bytes[32] = {0}; /* little endian, zero-extended */
    size_t nb = n / 2;
    for (size_t i = 0; i < nb; i++) {
        int a = hexval(hex[2 * i]), b = hexval(hex[2 * i + 1]);
        if (a < 0 || b < 0) return -1;
        bytes[nb - 1 - i] = (unsigned char)(a * 16 + b);
    }
    for (int e = 0; e < 4; e++) {
        uint64_t v = 0;
        for (int b = 7; b >= 0; b--) v = (v << 8) | bytes[8 * e + b];
        h[e] = v;
    }
    return 0;
}
