/*
 * fast_impl.h — CPU ORACLE, TIMED LEG ONLY (test infrastructure, NOT product code).  Included once per backend by
 * fast_avx512.c / fast_avx2.c / fast_scalar.c with GLV_BACKEND and FN(name) defined; see fast_vec.h.
 *
 * The same algorithms as the scalar restatement, VL independent instances per register:
 *   - Poseidon permutation on VL states (structure-of-arrays), the sponge over VL leaves, two_to_one over VL node pairs
 *     [scalar: poseidon.c, merkle.c; plonky2 hash/poseidon.rs, hash/merkle_tree.rs]
 *   - radix-2 decimation-in-frequency NTT, natural order in, bit-reversed order out, in two cache-sized steps
 *     [scalar: ntt.c; plonky2 field/src/fft.rs]
 *   - the vanishing polynomial of the 14 gate types on VL points of the quotient coset
 *     [scalar: plonk.c; plonky2 plonk/vanishing_poly.rs, gates/]
 * Outputs are bit-identical to the scalar path (canonical values); tests/test_oracle_fast.py holds that.
 */
#include <stdlib.h>
#include <string.h>

#include "fast_vec.h"
#include "gl_oracle.h"
#include "fast.h"

/* ------------------------------------------------ Poseidon ------------------------------------------------ */
static const uint32_t FN(MDS_C)[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

VI V FN(sbox7)(V x) {
    V x2 = v_sqr(x), x4 = v_sqr(x2), x3 = v_mul(x, x2);
    return v_mul(x3, x4);
}

#if GLV_BACKEND == 1
VI void FN(mds)(V st[12]) { /* 32-bit halves accumulated separately (sums < 2^42), then 2^64 = 2^32 - 1 */
    uint64_t lo[24], hi[24], out[12];
    for (int i = 0; i < 12; i++) {
        lo[i] = lo[i + 12] = st[i] & GLV_EPS;
        hi[i] = hi[i + 12] = st[i] >> 32;
    }
    for (int r = 0; r < 12; r++) {
        uint64_t alo = 0, ahi = 0;
        for (int i = 0; i < 12; i++) {
            alo += lo[i + r] * FN(MDS_C)[i];
            ahi += hi[i + r] * FN(MDS_C)[i];
        }
        if (r == 0) {
            alo += lo[0] * 8;
            ahi += hi[0] * 8;
        }
        unsigned __int128 v = (unsigned __int128)alo + ((unsigned __int128)ahi << 32);
        uint64_t vlo = (uint64_t)v, t2;
        uint64_t cry = __builtin_add_overflow(vlo, (uint64_t)(v >> 64) * GLV_EPS, &t2);
        out[r] = t2 + ((0 - cry) & GLV_EPS);
    }
    memcpy(st, out, sizeof out);
}
#else
VI void FN(mds)(V st[12]) {
    V hi[12], out[12], cv[12];
    for (int i = 0; i < 12; i++) {
        hi[i] = v_shr32(st[i]);
        cv[i] = v_set1(FN(MDS_C)[i]);
    }
    for (int r = 0; r < 12; r++) {
        V alo = v_mul32(st[r], cv[0]), ahi = v_mul32(hi[r], cv[0]);
        for (int i = 1; i < 12; i++) {
            const int k = (i + r) % 12;
            alo = v_add64(alo, v_mul32(st[k], cv[i]));
            ahi = v_add64(ahi, v_mul32(hi[k], cv[i]));
        }
        if (r == 0) {
            alo = v_add64(alo, v_mul32(st[0], v_set1(8)));
            ahi = v_add64(ahi, v_mul32(hi[0], v_set1(8)));
        }
        /* value = alo + 2^32 ahi, both < 2^42: low word s (+ carry into top), top = ahi >> 32 (< 2^10) */
        V sh = v_shl32(ahi);
        V s = v_add64(alo, sh);
        V top = v_carry_inc(v_shr32(ahi), s, sh);
        V t1 = v_mul32(top, v_set1(GLV_EPS));
        V t2 = v_add64(s, t1);
        out[r] = v_fix_carry(t2, t1);
    }
    for (int r = 0; r < 12; r++) st[r] = out[r];
}
#endif

/* the textbook round structure of poseidon.c::glo_poseidon_permute (full constant layer + full MDS in every round) */
static void FN(permute)(V st[12]) {
    const uint64_t *rc = glo_poseidon_round_constants();
    for (int round = 0; round < 30; round++) {
        for (int i = 0; i < 12; i++) st[i] = v_add_c(st[i], v_set1(rc[round * 12 + i]));
        if (round < 4 || round >= 26) {
            for (int i = 0; i < 12; i++) st[i] = FN(sbox7)(st[i]);
        } else {
            st[0] = FN(sbox7)(st[0]);
        }
        FN(mds)(st);
    }
}

/* n states, state s at states[12 s ..]; canonical output (= glo_poseidon_permute on each) */
void FN(glo_fast_permute_many)(uint64_t *states, size_t n) {
    size_t s = 0;
    for (; s + VL <= n; s += VL) {
        V st[12];
        for (int i = 0; i < 12; i++) st[i] = v_gather(states + 12 * s + i, 12);
        FN(permute)(st);
        for (int i = 0; i < 12; i++) {
            uint64_t t[VL];
            v_store(t, v_canon(st[i]));
            for (int l = 0; l < VL; l++) states[12 * (s + l) + i] = t[l];
        }
    }
    for (; s < n; s++) glo_poseidon_permute(states + 12 * s);
}

/* hash_or_noop of VL consecutive rows (row-major leaves, leaf_len words each) into dig[4 * VL] */
static void FN(hash_rows)(const uint64_t *leaves, size_t leaf_len, uint64_t *dig) {
    if (leaf_len <= 4) {
        for (int l = 0; l < VL; l++)
            for (size_t i = 0; i < 4; i++) dig[4 * l + i] = i < leaf_len ? glo_canon(leaves[l * leaf_len + i]) : 0;
        return;
    }
    V st[12];
    for (int i = 0; i < 12; i++) st[i] = v_set1(0);
    for (size_t off = 0; off < leaf_len; off += 8) {
        size_t k = leaf_len - off < 8 ? leaf_len - off : 8;
        for (size_t i = 0; i < k; i++) st[i] = v_canon(v_gather(leaves + off + i, leaf_len));
        FN(permute)(st);
    }
    for (int i = 0; i < 4; i++) {
        uint64_t t[VL];
        v_store(t, v_canon(st[i]));
        for (int l = 0; l < VL; l++) dig[4 * l + i] = t[l];
    }
}

/* two_to_one of VL consecutive node pairs: in = VL x 8 words, out = VL x 4 words */
static void FN(compress_rows)(const uint64_t *in, uint64_t *out) {
    V st[12];
    for (int i = 0; i < 8; i++) st[i] = v_canon(v_gather(in + i, 8));
    for (int i = 8; i < 12; i++) st[i] = v_set1(0);
    FN(permute)(st);
    for (int i = 0; i < 4; i++) {
        uint64_t t[VL];
        v_store(t, v_canon(st[i]));
        for (int l = 0; l < VL; l++) out[4 * l + i] = t[l];
    }
}

/* merkle.c::glo_merkle_build with VL leaves / node pairs per permutation */
void FN(glo_fast_merkle_levels)(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned top, uint64_t *dig, int threads) {
    (void)glo_poseidon_round_constants();
    const long nb = (long)(n_leaves / VL);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long b = 0; b < nb; b++) FN(hash_rows)(leaves + (size_t)b * VL * leaf_len, leaf_len, dig + (size_t)b * VL * 4);
    for (size_t i = (size_t)nb * VL; i < n_leaves; i++) glo_hash_or_noop(leaves + i * leaf_len, leaf_len, dig + i * 4);
    uint64_t *prev = dig;
    for (unsigned l = 1; l <= top; l++) {
        size_t cnt = n_leaves >> l;
        uint64_t *cur = prev + (n_leaves >> (l - 1)) * 4;
        const long cb = (long)(cnt / VL);
#pragma omp parallel for num_threads(threads) schedule(static) if (cb >= 64)
        for (long b = 0; b < cb; b++) FN(compress_rows)(prev + (size_t)b * VL * 8, cur + (size_t)b * VL * 4);
        for (size_t i = (size_t)cb * VL; i < cnt; i++) glo_two_to_one(prev + i * 8, prev + i * 8 + 4, cur + i * 4);
        prev = cur;
    }
}

/* ------------------------------------------------ NTT ------------------------------------------------ */
/* Decimation in frequency, natural order in, BIT-REVERSED order out, in place: position rev(k) holds sum_i a_i w^(ik),
 * w = root_of_unity(log_n) (inverse: w^-1, no 1/n).  Layer tables: glo_fast_tw(inverse, s)[j] = w_(2^s)^(+-j), j < 2^(s-1). */
#define GLO_FAST_BLK 12u

/* the last log2(VL) layers of VL x VL contiguous elements: the VL-element groups go across registers */
VI void FN(dif_tail)(uint64_t *a, int inverse) {
#if VL > 1
    V r[VL];
    for (int i = 0; i < VL; i++) r[i] = v_load(a + VL * i);
    v_transpose(r);
    for (unsigned h = VL / 2, s = (VL == 8 ? 3 : 2); h >= 1; h >>= 1, s--) {
        const uint64_t *T = glo_fast_tw(inverse, s);
        for (unsigned k = 0; k < VL; k += 2 * h)
            for (unsigned j = 0; j < h; j++) {
                V x = r[k + j], y = r[k + j + h];
                r[k + j] = v_add(x, y);
                V d = v_sub(x, y);
                r[k + j + h] = j ? v_mul(d, v_set1(T[j])) : d;
            }
    }
    v_transpose(r);
    for (int i = 0; i < VL; i++) v_store(a + VL * i, v_canon(r[i]));
#else
    a[0] = v_canon(a[0]);
    (void)inverse;
#endif
}

/* one contiguous block of 2^m elements, VL^2 <= 2^m <= 2^GLO_FAST_MAX_S; canonical output */
static void FN(dif_block)(uint64_t *a, unsigned m, int inverse) {
    const size_t M = (size_t)1 << m;
    for (unsigned s = m; s >= 1; s--) {
        const size_t h = (size_t)1 << (s - 1);
        if (h < VL) break;
        const uint64_t *T = glo_fast_tw(inverse, s);
        for (size_t k = 0; k < M; k += 2 * h)
            for (size_t j = 0; j < h; j += VL) {
                V x = v_load(a + k + j), y = v_load(a + k + j + h);
                v_store(a + k + j, v_add(x, y));
                v_store(a + k + j + h, v_mul(v_sub(x, y), v_load(T + j)));
            }
    }
    for (size_t k = 0; k < M; k += VL * VL) FN(dif_tail)(a + k, inverse);
}

static inline size_t FN(rev)(size_t x, unsigned bits) {
    if (bits == 0) return 0;
    uint64_t v = __builtin_bswap64((uint64_t)x);
    v = ((v & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((v >> 4) & 0x0F0F0F0F0F0F0F0FULL);
    v = ((v & 0x3333333333333333ULL) << 2) | ((v >> 2) & 0x3333333333333333ULL);
    v = ((v & 0x5555555555555555ULL) << 1) | ((v >> 1) & 0x5555555555555555ULL);
    return (size_t)(v >> (64 - bits));
}

/* row[c] *= rho^c, c < len (len a multiple of VL) */
static void FN(scale_geometric)(uint64_t *row, size_t len, uint64_t rho, uint64_t start) {
    uint64_t lane[VL], p = start;
    for (int l = 0; l < VL; l++) {
        lane[l] = p;
        p = glo_mul(p, rho);
    }
    /* p = start rho^VL; step factor rho^VL */
    uint64_t step = 1;
    for (int l = 0; l < VL; l++) step = glo_mul(step, rho);
    V cur = v_load(lane), st = v_set1(step);
    for (size_t c = 0; c < len; c += VL) {
        v_store(row + c, v_mul(v_load(row + c), cur));
        cur = v_mul(cur, st);
    }
}

#define GLO_FAST_WD 16 /* elements of a row handled together in the column step: two cache lines */

void FN(glo_fast_ntt_dif)(uint64_t *a, unsigned log_n, int inverse, int threads) {
    if (log_n < 6 || log_n > 2 * GLO_FAST_MAX_S) { /* tiny: the scalar layers, same result */
        glo_fast_ntt_dif_plain(a, log_n, inverse);
        return;
    }
    if (log_n <= GLO_FAST_MAX_S) {
        FN(dif_block)(a, log_n, inverse);
        return;
    }
    /* N = R x C, element (r, c) at a[r C + c]:  (1) R-point transforms down the columns (stride C), (2) element (r', c) times
     * w_N^(c rev(r')), (3) C-point transforms along the rows.  Position r' C + c' then holds X[rev(r') + R rev(c')] = the
     * bit-reversed order of the whole transform. */
    const unsigned b = log_n - GLO_FAST_BLK > GLO_FAST_MAX_S ? log_n - GLO_FAST_MAX_S : GLO_FAST_BLK, ra = log_n - b;
    const size_t R = (size_t)1 << ra, C = (size_t)1 << b;
    /* (1) */
#pragma omp parallel num_threads(threads) if (threads > 1)
    {
        uint64_t *buf = (uint64_t *)aligned_alloc(64, R * GLO_FAST_WD * 8);
#pragma omp for schedule(static)
        for (long cg = 0; cg < (long)(C / GLO_FAST_WD); cg++) {
            uint64_t *col = a + (size_t)cg * GLO_FAST_WD;
            for (size_t r = 0; r < R; r++) memcpy(buf + r * GLO_FAST_WD, col + r * C, GLO_FAST_WD * 8);
            for (unsigned s = ra; s >= 1; s--) {
                const size_t h = (size_t)1 << (s - 1);
                const uint64_t *T = glo_fast_tw(inverse, s);
                for (size_t k = 0; k < R; k += 2 * h)
                    for (size_t j = 0; j < h; j++) {
                        uint64_t *px = buf + (k + j) * GLO_FAST_WD, *py = px + h * GLO_FAST_WD;
                        const V tw = v_set1(T[j]);
                        for (int q = 0; q < GLO_FAST_WD; q += VL) {
                            V x = v_load(px + q), y = v_load(py + q);
                            v_store(px + q, v_add(x, y));
                            V d = v_sub(x, y);
                            v_store(py + q, j ? v_mul(d, tw) : d);
                        }
                    }
            }
            for (size_t r = 0; r < R; r++) memcpy(col + r * C, buf + r * GLO_FAST_WD, GLO_FAST_WD * 8);
        }
        free(buf);
    }
    /* (2) + (3) */
    const uint64_t wN = inverse ? glo_inv(glo_root_of_unity(log_n)) : glo_root_of_unity(log_n);
    uint64_t *P = (uint64_t *)malloc(R * 8); /* w_N^e, e < R */
    P[0] = 1;
    for (size_t e = 1; e < R; e++) P[e] = glo_mul(P[e - 1], wN);
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (long r = 0; r < (long)R; r++) {
        uint64_t *row = a + (size_t)r * C;
        if (r) FN(scale_geometric)(row, C, P[FN(rev)((size_t)r, ra)], 1);
        FN(dif_block)(row, b, inverse);
    }
    free(P);
}

/* dst[rev(i)] = src[i] * scale for i0 <= i < i1 (scale 1: no multiplication), out of place; canonical if src is */
void FN(glo_fast_bitrev_scale)(const uint64_t *src, uint64_t *dst, unsigned log_n, uint64_t scale, size_t i0, size_t i1) {
    if (scale == 1) {
        for (size_t i = i0; i < i1; i++) dst[FN(rev)(i, log_n)] = src[i];
        return;
    }
    size_t i = i0;
    if (i1 - i0 >= VL) {
        const V sc = v_set1(scale);
        for (; i + VL <= i1; i += VL) {
            uint64_t t[VL];
            v_store(t, v_canon(v_mul(v_load(src + i), sc)));
            for (int l = 0; l < VL; l++) dst[FN(rev)(i + l, log_n)] = t[l];
        }
    }
    for (; i < i1; i++) dst[FN(rev)(i, log_n)] = glo_mul(src[i], scale);
}

/* a[j] *= start * shift^j (canonical out) */
void FN(glo_fast_scale_powers)(uint64_t *a, size_t n, uint64_t shift, uint64_t start) {
    if (n < VL || n % VL) {
        uint64_t s = start;
        for (size_t i = 0; i < n; i++) {
            a[i] = glo_mul(a[i], s);
            s = glo_mul(s, shift);
        }
        return;
    }
    FN(scale_geometric)(a, n, shift, start);
    for (size_t i = 0; i < n; i += VL) v_store(a + i, v_canon(v_load(a + i)));
}

/* ------------------------------------------------ quotient ------------------------------------------------ */
/* plonk.c::gate_eval / vanishing_terms / eval_vanishing on VL points of the quotient coset at once.  On the prover's coset
 * every wire value is a base-field element, so the `ext` arithmetic of plonk.c (whose second limb is zero there) becomes
 * plain field arithmetic; the extension-algebra gates keep their pairs (Y^2 = 7). */
typedef struct {
    V c0, c1;
} FN(alg);
#define ALG FN(alg)

VI V FN(mul7)(V x) { return v_mul_small(x, 7); }
VI ALG FN(amul)(ALG x, ALG y) {
    ALG r = {v_add(v_mul(x.c0, y.c0), FN(mul7)(v_mul(x.c1, y.c1))), v_add(v_mul(x.c0, y.c1), v_mul(x.c1, y.c0))};
    return r;
}
VI ALG FN(aadd)(ALG x, ALG y) {
    ALG r = {v_add(x.c0, y.c0), v_add(x.c1, y.c1)};
    return r;
}
VI ALG FN(asub)(ALG x, ALG y) {
    ALG r = {v_sub(x.c0, y.c0), v_sub(x.c1, y.c1)};
    return r;
}
VI ALG FN(aw)(const V *w, unsigned i) {
    ALG r = {w[i], w[i + 1]};
    return r;
}

static void FN(partial_interpolate)(const uint64_t *domain, const uint64_t *weights, const ALG *values, unsigned from, unsigned to,
                                    ALG x, ALG *eval, ALG *prod) {
    for (unsigned k = from; k < to; k++) {
        ALG term = x;
        term.c0 = v_sub_c(term.c0, v_set1(domain[k]));
        const V wk = v_set1(weights[k]);
        ALG val = {v_mul(values[k].c0, wk), v_mul(values[k].c1, wk)};
        ALG next_eval = FN(aadd)(FN(amul)(*eval, term), FN(amul)(val, *prod));
        *prod = FN(amul)(*prod, term);
        *eval = next_eval;
    }
}

/* PoseidonGate wire layout (gates/poseidon.rs), as plonk.c */
#define FPG_IN(i) (i)
#define FPG_OUT(i) (12 + (i))
#define FPG_SWAP 24
#define FPG_DELTA(i) (25 + (i))
#define FPG_FULL0(r, i) (29 + 12 * ((r)-1) + (i))
#define FPG_PARTIAL(r) (65 + (r))
#define FPG_FULL1(r, i) (87 + 12 * (r) + (i))

static unsigned FN(poseidon_gate_eval)(const V *w, V *c) {
    const uint64_t *rc = glo_poseidon_round_constants();
    const V one = v_set1(1);
    unsigned k = 0;
    V swap = w[FPG_SWAP];
    c[k++] = v_mul(swap, v_sub_c(swap, one));
    for (int i = 0; i < 4; i++) c[k++] = v_sub(v_mul(swap, v_sub(w[FPG_IN(i + 4)], w[FPG_IN(i)])), w[FPG_DELTA(i)]);
    V st[12];
    for (int i = 0; i < 4; i++) {
        st[i] = v_add(w[FPG_IN(i)], w[FPG_DELTA(i)]);
        st[i + 4] = v_sub(w[FPG_IN(i + 4)], w[FPG_DELTA(i)]);
    }
    for (int i = 8; i < 12; i++) st[i] = w[FPG_IN(i)];
    unsigned round = 0;
    for (int r = 0; r < 4; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = v_add_c(st[i], v_set1(rc[round * 12 + i]));
        if (r != 0)
            for (int i = 0; i < 12; i++) {
                c[k++] = v_sub(st[i], w[FPG_FULL0(r, i)]);
                st[i] = w[FPG_FULL0(r, i)];
            }
        for (int i = 0; i < 12; i++) st[i] = FN(sbox7)(st[i]);
        FN(mds)(st);
    }
    for (int r = 0; r < 22; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = v_add_c(st[i], v_set1(rc[round * 12 + i]));
        c[k++] = v_sub(st[0], w[FPG_PARTIAL(r)]);
        st[0] = FN(sbox7)(w[FPG_PARTIAL(r)]);
        FN(mds)(st);
    }
    for (int r = 0; r < 4; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = v_add_c(st[i], v_set1(rc[round * 12 + i]));
        for (int i = 0; i < 12; i++) {
            c[k++] = v_sub(st[i], w[FPG_FULL1(r, i)]);
            st[i] = w[FPG_FULL1(r, i)];
        }
        for (int i = 0; i < 12; i++) st[i] = FN(sbox7)(st[i]);
        FN(mds)(st);
    }
    for (int i = 0; i < 12; i++) c[k++] = v_sub(st[i], w[FPG_OUT(i)]);
    return k; /* 123 */
}

/* interp: the CosetInterpolationGate tables (domain, weights) of this circuit's gate, made once per call */
static unsigned FN(gate_eval)(const glo_gate *g, const V *consts, const V *w, const uint64_t pi_hash[4], const uint64_t *interp, V *c) {
    const V one = v_set1(1), zero = v_set1(0);
    switch (g->id) {
        case GLO_GATE_NOOP: return 0;
        case GLO_GATE_CONSTANT:
            for (unsigned i = 0; i < g->param; i++) c[i] = v_sub(consts[i], w[i]);
            return g->param;
        case GLO_GATE_PUBLIC_INPUT:
            for (unsigned i = 0; i < 4; i++) c[i] = v_sub_c(w[i], v_set1(glo_canon(pi_hash[i])));
            return 4;
        case GLO_GATE_ARITHMETIC:
            for (unsigned i = 0; i < g->param; i++) {
                V m0 = w[4 * i], m1 = w[4 * i + 1], ad = w[4 * i + 2], out = w[4 * i + 3];
                V comp = v_add(v_mul(v_mul(m0, m1), consts[0]), v_mul(ad, consts[1]));
                c[i] = v_sub(out, comp);
            }
            return g->param;
        case GLO_GATE_POSEIDON: return FN(poseidon_gate_eval)(w, c);
        case GLO_GATE_BASE_SUM: {
            V acc = zero;
            for (unsigned i = g->param; i-- > 0;) acc = v_add(v_dbl(acc), w[1 + i]);
            c[0] = v_sub(acc, w[0]);
            for (unsigned i = 0; i < g->param; i++) c[1 + i] = v_mul(w[1 + i], v_sub_c(w[1 + i], one));
            return 1 + g->param;
        }
        case GLO_GATE_ARITHMETIC_EXT:
        case GLO_GATE_MUL_EXT: {
            const unsigned per = g->id == GLO_GATE_ARITHMETIC_EXT ? 8 : 6;
            for (unsigned i = 0; i < g->param; i++) {
                const V *q = w + per * i;
                V p0 = v_add(v_mul(q[0], q[2]), FN(mul7)(v_mul(q[1], q[3])));
                V p1 = v_add(v_mul(q[0], q[3]), v_mul(q[1], q[2]));
                p0 = v_mul(p0, consts[0]);
                p1 = v_mul(p1, consts[0]);
                const V *out = q + 4;
                if (g->id == GLO_GATE_ARITHMETIC_EXT) {
                    p0 = v_add(p0, v_mul(q[4], consts[1]));
                    p1 = v_add(p1, v_mul(q[5], consts[1]));
                    out = q + 6;
                }
                c[2 * i] = v_sub(out[0], p0);
                c[2 * i + 1] = v_sub(out[1], p1);
            }
            return 2 * g->param;
        }
        case GLO_GATE_RANDOM_ACCESS: {
            const unsigned bits = g->param & 0xFF, copies = (g->param >> 8) & 0xFF, extra = (g->param >> 16) & 0xFF;
            const unsigned vec = 1u << bits, routed = (2 + vec) * copies + extra;
            unsigned k = 0;
            for (unsigned cp = 0; cp < copies; cp++) {
                const V *q = w + (2 + vec) * cp;
                const V *b = w + routed + cp * bits;
                for (unsigned i = 0; i < bits; i++) c[k++] = v_mul(b[i], v_sub_c(b[i], one));
                V rec = zero;
                for (unsigned i = bits; i-- > 0;) rec = v_add(v_dbl(rec), b[i]);
                c[k++] = v_sub(rec, q[0]);
                V items[64];
                for (unsigned j = 0; j < vec; j++) items[j] = q[2 + j];
                for (unsigned i = 0, len = vec; i < bits; i++, len >>= 1)
                    for (unsigned j = 0; j < len / 2; j++)
                        items[j] = v_add(items[2 * j], v_mul(b[i], v_sub(items[2 * j + 1], items[2 * j])));
                c[k++] = v_sub(items[0], q[1]);
            }
            for (unsigned i = 0; i < extra; i++) c[k++] = v_sub(consts[i], w[(2 + vec) * copies + i]);
            return k;
        }
        case GLO_GATE_REDUCING:
        case GLO_GATE_REDUCING_EXT: {
            const unsigned nc = g->param, is_ext = g->id == GLO_GATE_REDUCING_EXT;
            const unsigned start_accs = 6 + (is_ext ? 2 * nc : nc);
            ALG alpha = FN(aw)(w, 2), acc = FN(aw)(w, 4);
            for (unsigned i = 0; i < nc; i++) {
                ALG coeff;
                if (is_ext) {
                    coeff = FN(aw)(w, 6 + 2 * i);
                } else {
                    coeff.c0 = w[6 + i];
                    coeff.c1 = zero;
                }
                ALG next = i == nc - 1 ? FN(aw)(w, 0) : FN(aw)(w, start_accs + 2 * i);
                ALG d = FN(asub)(FN(aadd)(FN(amul)(acc, alpha), coeff), next);
                c[2 * i] = d.c0;
                c[2 * i + 1] = d.c1;
                acc = next;
            }
            return 2 * nc;
        }
        case GLO_GATE_EXPONENTIATION: {
            const unsigned nb = g->param;
            V base = w[0], output = w[1 + nb];
            for (unsigned i = 0; i < nb; i++) {
                V prev = i == 0 ? one : v_sqr(w[2 + nb + i - 1]);
                V bit = w[1 + (nb - 1 - i)];
                V factor = v_add(v_mul(bit, base), v_sub(one, bit));
                c[i] = v_sub(v_mul(prev, factor), w[2 + nb + i]);
            }
            c[nb] = v_sub(output, w[2 + nb + nb - 1]);
            return nb + 1;
        }
        case GLO_GATE_POSEIDON_MDS: {
            for (unsigned r = 0; r < 12; r++) {
                ALG acc = {zero, zero};
                for (unsigned i = 0; i < 12; i++) {
                    ALG in = FN(aw)(w, 2 * ((i + r) % 12));
                    acc.c0 = v_add(acc.c0, v_mul_small(in.c0, FN(MDS_C)[i]));
                    acc.c1 = v_add(acc.c1, v_mul_small(in.c1, FN(MDS_C)[i]));
                }
                if (r == 0) {
                    acc.c0 = v_add(acc.c0, v_mul_small(w[0], 8));
                    acc.c1 = v_add(acc.c1, v_mul_small(w[1], 8));
                }
                ALG out = FN(aw)(w, 24 + 2 * r);
                c[2 * r] = v_sub(out.c0, acc.c0);
                c[2 * r + 1] = v_sub(out.c1, acc.c1);
            }
            return 24;
        }
        case GLO_GATE_COSET_INTERPOLATION: {
            const unsigned bits = g->param & 0xFF, degree = (g->param >> 8) & 0xFF, np = 1u << bits;
            const unsigned ni = (np - 2) / (degree - 1);
            const unsigned start_point = 1 + 2 * np, start_value = start_point + 2, start_inter = start_value + 2;
            const uint64_t *domain = interp, *weights = interp + 64;
            V shift = w[0];
            ALG point = FN(aw)(w, start_point), shifted = FN(aw)(w, start_inter + 4 * ni);
            ALG values[64];
            for (unsigned i = 0; i < np; i++) values[i] = FN(aw)(w, 1 + 2 * i);
            unsigned k = 0;
            ALG ss = {v_mul(shifted.c0, shift), v_mul(shifted.c1, shift)};
            ALG d = FN(asub)(point, ss);
            c[k++] = d.c0;
            c[k++] = d.c1;
            ALG eval = {zero, zero}, prod = {one, zero};
            FN(partial_interpolate)(domain, weights, values, 0, degree, shifted, &eval, &prod);
            for (unsigned i = 0; i < ni; i++) {
                ALG ie = FN(aw)(w, start_inter + 2 * i), ip = FN(aw)(w, start_inter + 2 * (ni + i));
                d = FN(asub)(ie, eval);
                c[k++] = d.c0;
                c[k++] = d.c1;
                d = FN(asub)(ip, prod);
                c[k++] = d.c0;
                c[k++] = d.c1;
                unsigned from = 1 + (degree - 1) * (i + 1), to = from + degree - 1;
                if (to > np) to = np;
                eval = ie;
                prod = ip;
                FN(partial_interpolate)(domain, weights, values, from, to, shifted, &eval, &prod);
            }
            d = FN(asub)(FN(aw)(w, start_value), eval);
            c[k++] = d.c0;
            c[k++] = d.c1;
            return k;
        }
        default: return 0;
    }
}

#define FQ_UNUSED_SELECTOR 0xFFFFFFFFull

/* plonk.c::glo_quotient_chunks up to (not including) the inverse coset transform: vals[k][i] = combined constraints of
 * challenge k at the natural coset index i, divided by Z_H.  The VL points of a step are VL consecutive ROWS of the
 * bit-reversed leaves (natural indices rev(p) ..), so the row gathers have a fixed stride. */
void FN(glo_fast_quotient_values)(const glo_circuit *cc, const uint64_t *cs_leaves, const uint64_t *wires_leaves,
                                  const uint64_t *zs_leaves, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                                  const uint64_t pi_hash[4], uint64_t *vals, int threads) {
    const unsigned qb = 3;
    const unsigned lde_bits = cc->degree_bits + qb;
    const size_t n = (size_t)1 << cc->degree_bits, N = (size_t)1 << lde_bits;
    const unsigned nch = cc->num_challenges, npp = cc->num_partial_products, nr = cc->num_routed_wires;
    const unsigned ncs = cc->num_constants + nr, nzs = nch * (1 + npp), nw = cc->num_wires;
    const unsigned salt = (cc->fri.flags & GLO_FLAG_ZERO_KNOWLEDGE) ? GLO_SALT_SIZE : 0; /* leaf rows of blinded oracles are wider */
    const size_t ww = nw + salt, zw = nzs + salt;
    const unsigned chunk = cc->quotient_degree_factor;
    const unsigned n_terms = nch + nch * (npp + 1) + cc->num_gate_constraints;
    const uint64_t wN = glo_root_of_unity(lde_bits);
    /* w_N^i = A[i >> lo_bits] B[i & mask] */
    const unsigned lo_bits = lde_bits / 2, hi_bits = lde_bits - lo_bits;
    uint64_t *tb = (uint64_t *)malloc((((size_t)1 << lo_bits) + ((size_t)1 << hi_bits)) * 8);
    uint64_t *B = tb, *A = tb + ((size_t)1 << lo_bits);
    B[0] = 1;
    for (size_t i = 1; i < ((size_t)1 << lo_bits); i++) B[i] = glo_mul(B[i - 1], wN);
    {
        uint64_t st = glo_mul(B[((size_t)1 << lo_bits) - 1], wN);
        A[0] = 1;
        for (size_t i = 1; i < ((size_t)1 << hi_bits); i++) A[i] = glo_mul(A[i - 1], st);
    }
    uint64_t zh_inv[8], xn8[8], g_n = glo_pow(7, n), w8 = glo_root_of_unity(qb);
    for (unsigned r = 0; r < 8; r++) {
        xn8[r] = glo_mul(g_n, glo_pow(w8, r));
        zh_inv[r] = glo_inv(glo_sub(xn8[r], 1));
    }
    /* l_0(x) = (x^n - 1) / (n (x - 1)) needs 1 / (x - 1) at every point: one inversion per block of rows (Montgomery) */
    uint64_t interp[128];
    memset(interp, 0, sizeof interp);
    for (unsigned g = 0; g < cc->num_gates; g++)
        if (cc->gates[g].id == GLO_GATE_COSET_INTERPOLATION) glo_coset_interpolation_tables(cc->gates[g].param & 0xFF, interp, interp + 64);
    (void)glo_poseidon_round_constants();
    const uint64_t n_field = (uint64_t)n;
    const size_t BLKR = 64; /* rows per inversion block */
#pragma omp parallel num_threads(threads)
    {
        V *terms = (V *)aligned_alloc(64, sizeof(V) * (n_terms + 8));
        V *tmp = (V *)aligned_alloc(64, sizeof(V) * 160);
        V *wires = (V *)aligned_alloc(64, sizeof(V) * (nw + 8));
        V *sig = (V *)aligned_alloc(64, sizeof(V) * (nr + 8));
        uint64_t xs[64], dinv[64], pref[64];
#pragma omp for schedule(dynamic, 4)
        for (long blk = 0; blk < (long)((N + BLKR - 1) / BLKR); blk++) {
            const size_t p0 = (size_t)blk * BLKR, p1 = p0 + BLKR < N ? p0 + BLKR : N;
            uint64_t run = 1;
            for (size_t p = p0; p < p1; p++) {
                const size_t i = FN(rev)(p, lde_bits);
                uint64_t x = glo_mul(7, glo_mul(A[i >> lo_bits], B[i & (((size_t)1 << lo_bits) - 1)]));
                xs[p - p0] = x;
                uint64_t d = glo_mul(glo_sub(x, 1), n_field);
                dinv[p - p0] = d;
                pref[p - p0] = run;
                run = glo_mul(run, d);
            }
            {
                uint64_t inv = glo_inv(run);
                for (size_t p = p1; p-- > p0;) {
                    uint64_t d = dinv[p - p0];
                    dinv[p - p0] = glo_mul(inv, pref[p - p0]);
                    inv = glo_mul(inv, d);
                }
            }
            for (size_t p = p0; p < p1; p += VL) {
                uint64_t idx_nat[VL], idx_next[VL], zhv[VL], l0v[VL];
                for (int l = 0; l < VL; l++) {
                    const size_t i = FN(rev)(p + l, lde_bits);
                    idx_nat[l] = i;
                    idx_next[l] = FN(rev)((i + 8) % N, lde_bits) * zw;
                    zhv[l] = zh_inv[i % 8];
                    l0v[l] = glo_mul(glo_sub(xn8[i % 8], 1), dinv[p + l - p0]);
                }
                const V x = v_load(xs + (p - p0)), l0 = v_load(l0v);
                V consts[16], zs[4], nzsv[4], pps[32];
                for (unsigned k = 0; k < cc->num_constants; k++) consts[k] = v_gather(cs_leaves + p * ncs + k, ncs);
                for (unsigned k = 0; k < nr; k++) sig[k] = v_gather(cs_leaves + p * ncs + cc->num_constants + k, ncs);
                for (unsigned k = 0; k < nw; k++) wires[k] = v_gather(wires_leaves + p * ww + k, ww);
                for (unsigned k = 0; k < nch; k++) {
                    zs[k] = v_gather(zs_leaves + p * zw + k, zw);
                    nzsv[k] = v_gather_idx(zs_leaves + k, idx_next);
                }
                for (unsigned k = 0; k < nch * npp; k++) pps[k] = v_gather(zs_leaves + p * zw + nch + k, zw);
                /* vanishing_terms */
                unsigned t = 0;
                for (unsigned i = 0; i < nch; i++) terms[t++] = v_mul(l0, v_sub_c(zs[i], v_set1(1)));
                for (unsigned i = 0; i < nch; i++) {
                    const V beta = v_set1(betas[i]), gamma = v_set1(glo_canon(gammas[i]));
                    for (unsigned ck = 0; ck * chunk < nr; ck++) {
                        V nump = v_set1(1), denp = v_set1(1);
                        for (unsigned j = ck * chunk; j < (ck + 1) * chunk && j < nr; j++) {
                            V s_id = v_mul(x, v_set1(cc->k_is[j]));
                            nump = v_mul(nump, v_add_c(v_add(wires[j], v_mul(s_id, beta)), gamma));
                            denp = v_mul(denp, v_add_c(v_add(wires[j], v_mul(sig[j], beta)), gamma));
                        }
                        V prev = ck == 0 ? zs[i] : pps[i * npp + ck - 1];
                        V next = ck == npp ? nzsv[i] : pps[i * npp + ck];
                        terms[t++] = v_sub(v_mul(prev, nump), v_mul(next, denp));
                    }
                }
                V *gc = terms + t;
                for (unsigned j = 0; j < cc->num_gate_constraints; j++) gc[j] = v_set1(0);
                for (unsigned g = 0; g < cc->num_gates; g++) {
                    const glo_gate *gt = &cc->gates[g];
                    V s = consts[gt->selector_index];
                    V filter = v_set1(1);
                    for (unsigned r = gt->group_start; r < gt->group_end; r++)
                        if (r != g) filter = v_mul(filter, v_sub(v_set1(r), s));
                    if (cc->num_selectors > 1) filter = v_mul(filter, v_sub(v_set1(FQ_UNUSED_SELECTOR), s));
                    if (!v_any_nonzero(v_canon(filter))) { /* gate not on any of these rows: its terms are zero */
                        continue;
                    }
                    unsigned nc = FN(gate_eval)(gt, consts + cc->num_selectors, wires, pi_hash, interp, tmp);
                    for (unsigned j = 0; j < nc; j++) gc[j] = v_add(gc[j], v_mul(filter, tmp[j]));
                }
                for (unsigned k = 0; k < nch; k++) {
                    const V alpha = v_set1(alphas[k]);
                    V acc = v_set1(0);
                    for (unsigned j = n_terms; j-- > 0;) acc = v_add(v_mul(acc, alpha), terms[j]);
                    v_scatter_idx(vals + (size_t)k * N, idx_nat, v_canon(v_mul(acc, v_load(zhv))));
                }
            }
        }
        free(terms);
        free(tmp);
        free(wires);
        free(sig);
    }
    free(tb);
}

/* ------------------------------------------------ partial products, openings, FRI ------------------------------------------------ */
VI V FN(vinv)(V a) { /* a^(p-2), p - 2 = 2^64 - 2^32 - 1: the usual addition chain on 2^k - 1 blocks; 0 -> 0 */
    V t2 = v_mul(v_sqr(a), a);          /* 2^2 - 1 */
    V t3 = v_mul(v_sqr(t2), a);         /* 2^3 - 1 */
    V t6 = t3;
    for (int i = 0; i < 3; i++) t6 = v_sqr(t6);
    t6 = v_mul(t6, t3);                 /* 2^6 - 1 */
    V t12 = t6;
    for (int i = 0; i < 6; i++) t12 = v_sqr(t12);
    t12 = v_mul(t12, t6);               /* 2^12 - 1 */
    V t24 = t12;
    for (int i = 0; i < 12; i++) t24 = v_sqr(t24);
    t24 = v_mul(t24, t12);              /* 2^24 - 1 */
    V t30 = t24;
    for (int i = 0; i < 6; i++) t30 = v_sqr(t30);
    t30 = v_mul(t30, t6);               /* 2^30 - 1 */
    V t31 = v_mul(v_sqr(t30), a);       /* 2^31 - 1 */
    V t32 = v_mul(v_sqr(t31), a);       /* 2^32 - 1 */
    /* exponent = (2^31 - 1) 2^33 + (2^32 - 1): bits 63..33 ones, bit 32 zero, bits 31..0 ones */
    V r = t31;
    for (int i = 0; i < 33; i++) r = v_sqr(r);
    return v_mul(r, t32);
}

/* plonk.c::glo_partial_products_and_zs, the per-row chunk quotients: q[(ch n_chunks + ck) n + i] */
void FN(glo_fast_chunk_quotients)(const glo_circuit *cc, const uint64_t *wires, const uint64_t *sigmas, const uint64_t *betas,
                                  const uint64_t *gammas, uint64_t *q, int threads) {
    const size_t n = (size_t)1 << cc->degree_bits;
    const unsigned nr = cc->num_routed_wires, nch = cc->num_challenges, chunk = cc->quotient_degree_factor;
    const unsigned n_chunks = (nr + chunk - 1) / chunk, nq = nch * n_chunks;
    const uint64_t w = glo_root_of_unity(cc->degree_bits);
    if (n < VL || nq > 64) { /* tiny circuits: the scalar loop */
        uint64_t x = 1;
        for (size_t i = 0; i < n; i++, x = glo_mul(x, w))
            for (unsigned ch = 0; ch < nch; ch++)
                for (unsigned ck = 0; ck < n_chunks; ck++) {
                    uint64_t num = 1, den = 1;
                    for (unsigned j = ck * chunk; j < (ck + 1) * chunk && j < nr; j++) {
                        uint64_t wv = wires[(size_t)j * n + i];
                        num = glo_mul(num, glo_add(glo_add(wv, glo_mul(betas[ch], glo_mul(cc->k_is[j], x))), gammas[ch]));
                        den = glo_mul(den, glo_add(glo_add(wv, glo_mul(betas[ch], sigmas[(size_t)j * n + i])), gammas[ch]));
                    }
                    q[((size_t)ch * n_chunks + ck) * n + i] = glo_mul(num, glo_inv(den));
                }
        return;
    }
    const size_t SEG = 1024; /* rows per task: x restarts from a power once per task */
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long sg = 0; sg < (long)((n + SEG - 1) / SEG); sg++) {
        const size_t i0 = (size_t)sg * SEG, i1 = i0 + SEG < n ? i0 + SEG : n;
        uint64_t xl[VL], x0 = glo_pow(w, i0), wvl = glo_pow(w, VL);
        for (int l = 0; l < VL; l++) xl[l] = glo_mul(x0, glo_pow(w, (uint64_t)l));
        V x = v_load(xl);
        const V step = v_set1(wvl);
        for (size_t i = i0; i < i1; i += VL, x = v_mul(x, step)) {
            V num[64], den[64], pre[64];
            for (unsigned ch = 0; ch < nch; ch++) {
                const V beta = v_set1(betas[ch]), gamma = v_set1(glo_canon(gammas[ch]));
                const V bx = v_mul(beta, x);
                for (unsigned ck = 0; ck < n_chunks; ck++) {
                    V nm = v_set1(1), dn = v_set1(1);
                    for (unsigned j = ck * chunk; j < (ck + 1) * chunk && j < nr; j++) {
                        V wv = v_load(wires + (size_t)j * n + i);
                        nm = v_mul(nm, v_add_c(v_add(v_mul(bx, v_set1(cc->k_is[j])), wv), gamma));
                        dn = v_mul(dn, v_add_c(v_add(v_mul(beta, v_load(sigmas + (size_t)j * n + i)), wv), gamma));
                    }
                    num[ch * n_chunks + ck] = nm;
                    den[ch * n_chunks + ck] = v_canon(dn);
                }
            }
            /* one inversion for the nq denominators of these rows (a zero denominator gives quotient 0, as inv(0) = 0 does) */
            V run = v_set1(1);
            for (unsigned k = 0; k < nq; k++) {
                pre[k] = run;
                run = v_mul(run, v_if_zero(den[k], v_set1(1), den[k]));
            }
            V inv = FN(vinv)(run);
            for (unsigned k = nq; k-- > 0;) {
                V dk = v_if_zero(den[k], v_set1(1), den[k]);
                V ik = v_mul(inv, pre[k]);
                inv = v_mul(inv, dk);
                V qv = v_if_zero(den[k], v_set1(0), v_mul(num[k], ik));
                v_store(q + (size_t)k * n + i, v_canon(qv));
            }
        }
    }
}

/* sum_j c_j z^j for base-field coefficients and an extension point, from the table zp[2 j], zp[2 j + 1] = z^j split in two
 * planes zp0[j], zp1[j] (plonk.c: glo_poly_eval_ext, Horner there; the field sum is the same) */
void FN(glo_fast_eval_with_powers)(const uint64_t *coeffs, size_t n, const uint64_t *zp0, const uint64_t *zp1, uint64_t out[2]) {
    uint64_t a0 = 0, a1 = 0;
    size_t j = 0;
    if (n >= VL) {
        V acc0 = v_set1(0), acc1 = v_set1(0);
        for (; j + VL <= n; j += VL) {
            V c = v_load(coeffs + j);
            acc0 = v_add(acc0, v_mul(c, v_load(zp0 + j)));
            acc1 = v_add(acc1, v_mul(c, v_load(zp1 + j)));
        }
        uint64_t t0[VL], t1[VL];
        v_store(t0, acc0);
        v_store(t1, acc1);
        for (int l = 0; l < VL; l++) {
            a0 = glo_add(a0, t0[l]);
            a1 = glo_add(a1, t1[l]);
        }
    }
    for (; j < n; j++) {
        a0 = glo_add(a0, glo_mul(coeffs[j], zp0[j]));
        a1 = glo_add(a1, glo_mul(coeffs[j], zp1[j]));
    }
    out[0] = glo_canon(a0);
    out[1] = glo_canon(a1);
}

/* fri.c::fri_final_poly, the composition sum_k alpha^k p_k: comp[2 j], comp[2 j + 1] */
void FN(glo_fast_reduce_polys)(const uint64_t *const *polys, size_t n_polys, const uint64_t *apow /* n_polys x 2 */, size_t n,
                               uint64_t *comp, int threads) {
    const size_t SEG = 512;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long sg = 0; sg < (long)((n + SEG - 1) / SEG); sg++) {
        const size_t j0 = (size_t)sg * SEG, j1 = j0 + SEG < n ? j0 + SEG : n;
        uint64_t acc0[512], acc1[512];
        const size_t len = j1 - j0;
        memset(acc0, 0, sizeof acc0);
        memset(acc1, 0, sizeof acc1);
        for (size_t k = 0; k < n_polys; k++) {
            const uint64_t *p = polys[k] + j0;
            const uint64_t a0 = apow[2 * k], a1 = apow[2 * k + 1];
            size_t j = 0;
            if (len >= VL) {
                const V va0 = v_set1(a0), va1 = v_set1(a1);
                for (; j + VL <= len; j += VL) {
                    V c = v_load(p + j);
                    v_store(acc0 + j, v_add(v_load(acc0 + j), v_mul(c, va0)));
                    v_store(acc1 + j, v_add(v_load(acc1 + j), v_mul(c, va1)));
                }
            }
            for (; j < len; j++) {
                acc0[j] = glo_add(acc0[j], glo_mul(p[j], a0));
                acc1[j] = glo_add(acc1[j], glo_mul(p[j], a1));
            }
        }
        for (size_t j = 0; j < len; j++) {
            comp[2 * (j0 + j)] = glo_canon(acc0[j]);
            comp[2 * (j0 + j) + 1] = glo_canon(acc1[j]);
        }
    }
}

/* smallest i with canon(permute(base, word `slot` = i)[out_index]) >> (64 - pow_bits) == 0 (fri.c: the proof of work) */
uint64_t FN(glo_fast_pow_search)(const uint64_t base[12], unsigned slot, unsigned out_index, unsigned pow_bits, int threads) {
    if (pow_bits == 0) return 0;
    (void)glo_poseidon_round_constants();
    const uint64_t BATCH = (uint64_t)VL * 256 * (uint64_t)(threads < 1 ? 1 : threads);
    for (uint64_t start = 0;; start += BATCH) {
        uint64_t best = UINT64_MAX;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(min : best)
        for (long g = 0; g < (long)(BATCH / VL); g++) {
            uint64_t cand[VL], res[VL];
            V st[12];
            for (int i = 0; i < 12; i++) st[i] = v_set1(glo_canon(base[i]));
            for (int l = 0; l < VL; l++) cand[l] = start + (uint64_t)g * VL + (uint64_t)l;
            st[slot] = v_load(cand);
            FN(permute)(st);
            v_store(res, v_canon(st[out_index]));
            for (int l = 0; l < VL; l++)
                if ((res[l] >> (64 - pow_bits)) == 0 && cand[l] < best) best = cand[l];
        }
        if (best != UINT64_MAX) return best;
    }
}

/* ext coefficients (interleaved, cur_n of them) -> their values on shift * <w>, interleaved, rows in BIT-REVERSED order:
 * fri.c's ext_coset_fft followed by flatten_leaves (whose leaf layout is the interleaved bit-reversed array itself) */
void FN(glo_fast_ext_coset_lde_bitrev)(const uint64_t *coeffs, unsigned log_n, uint64_t shift, uint64_t *out, int threads) {
    const size_t n = (size_t)1 << log_n;
    uint64_t *l0 = (uint64_t *)malloc(n * 8), *l1 = (uint64_t *)malloc(n * 8);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long i = 0; i < (long)n; i++) {
        l0[i] = coeffs[2 * i];
        l1[i] = coeffs[2 * i + 1];
    }
    /* the scaling by shift^j in parallel segments */
    const size_t SEG = n >= 4096 ? 4096 : n;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long sg = 0; sg < (long)(n / SEG); sg++) {
        uint64_t st = glo_pow(shift, (uint64_t)sg * SEG);
        FN(glo_fast_scale_powers)(l0 + (size_t)sg * SEG, SEG, shift, st);
        FN(glo_fast_scale_powers)(l1 + (size_t)sg * SEG, SEG, shift, st);
    }
    FN(glo_fast_ntt_dif)(l0, log_n, 0, threads);
    FN(glo_fast_ntt_dif)(l1, log_n, 0, threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long i = 0; i < (long)n; i++) {
        out[2 * i] = glo_canon(l0[i]);
        out[2 * i + 1] = glo_canon(l1[i]);
    }
    free(l0);
    free(l1);
}
