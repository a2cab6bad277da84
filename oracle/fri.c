/*
 * fri.c — CPU ORACLE (test infrastructure): Fiat–Shamir challenger, polynomial openings, FRI prover and verifier.
 *
 * Restates plonky2 @ f99ed9c src/iop/challenger.rs (Challenger: overwrite-mode duplex sponge, challenges popped from
 * the END of the rate buffer), src/plonk/proof.rs (OpeningSet::new = evaluation of coefficient-form polynomials at an
 * extension point), src/fri/oracle.rs (PolynomialBatch::prove_openings), src/fri/prover.rs (fri_committed_trees,
 * fri_proof_of_work, fri_prover_query_rounds), src/fri/verifier.rs (verify_fri_proof, fri_combine_initial,
 * compute_evaluation) and src/fri/challenges.rs.  The dependency is absent from /root/reference (Cargo.toml:12); the
 * reference reaches all of this only through CircuitData::prove / verify (src/rollup/circuits/mod.rs:1247,1321).
 *
 * PARITY STATUS: "parity unpinned" — the reference's tests hold no transcript/FRI vectors, only prove->verify round
 * trips (e.g. src/rollup/gadgets/proposal_block/mod.rs:605-614).  Checked here by self-consistency: proofs made by
 * glo_fri_prove (and by the GPU path, which must equal it word for word) are accepted by glo_fri_verify, and
 * corrupted proofs are rejected.  Proof of work: plonky2 accepts ANY witness (rayon find_any); this restatement and
 * the GPU path both return the SMALLEST one, which makes proofs deterministic and comparable.
 */
#include <stdlib.h>
#include <string.h>

#include "fast.h"
#include "gl_oracle.h"

/* ------------------------------------------------ challenger ------------------------------------------------ */
void glo_challenger_init(glo_challenger *c) { memset(c, 0, sizeof *c); }

static void duplexing(glo_challenger *c) {
    for (int i = 0; i < c->n_in; i++) c->state[i] = c->in[i];
    c->n_in = 0;
    glo_poseidon_permute(c->state);
    memcpy(c->out, c->state, 8 * sizeof(uint64_t));
    c->n_out = 8;
}

void glo_challenger_observe(glo_challenger *c, const uint64_t *elems, size_t n) {
    for (size_t k = 0; k < n; k++) {
        c->n_out = 0; /* any observation invalidates buffered outputs */
        c->in[c->n_in++] = glo_canon(elems[k]);
        if (c->n_in == 8) duplexing(c);
    }
}

uint64_t glo_challenger_get(glo_challenger *c) {
    if (c->n_in != 0 || c->n_out == 0) duplexing(c);
    return c->out[--c->n_out]; /* Vec::pop */
}

void glo_challenger_get_n(glo_challenger *c, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = glo_challenger_get(c);
}

/* ------------------------------------------------ small helpers ------------------------------------------------ */
static const uint64_t EXT_ZERO[2] = {0, 0};

static void ext_set(uint64_t d[2], const uint64_t s[2]) { d[0] = s[0]; d[1] = s[1]; }
static void ext_mul_base(const uint64_t a[2], uint64_t b, uint64_t out[2]) {
    out[0] = glo_mul(a[0], b);
    out[1] = glo_mul(a[1], b);
}
static void ext_pow(const uint64_t a[2], uint64_t e, uint64_t out[2]) {
    uint64_t r[2] = {1, 0}, b[2] = {a[0], a[1]}, t[2];
    while (e) {
        if (e & 1) {
            glo_ext_mul(r, b, t);
            ext_set(r, t);
        }
        glo_ext_mul(b, b, t);
        ext_set(b, t);
        e >>= 1;
    }
    ext_set(out, r);
}
static int ext_eq(const uint64_t a[2], const uint64_t b[2]) {
    return glo_canon(a[0]) == glo_canon(b[0]) && glo_canon(a[1]) == glo_canon(b[1]);
}

/* OpeningSet::new building block: p(z) for a base-field coefficient vector and an extension point */
void glo_poly_eval_ext(const uint64_t *coeffs, size_t n, const uint64_t z[2], uint64_t out[2]) {
    uint64_t acc[2] = {0, 0}, t[2];
    for (size_t i = n; i-- > 0;) {
        glo_ext_mul(acc, z, t);
        acc[0] = glo_add(t[0], coeffs[i]);
        acc[1] = t[1];
    }
    ext_set(out, acc);
}

static void ext_poly_eval(const uint64_t *coeffs /* n x 2 */, size_t n, const uint64_t z[2], uint64_t out[2]) {
    uint64_t acc[2] = {0, 0}, t[2];
    for (size_t i = n; i-- > 0;) {
        glo_ext_mul(acc, z, t);
        glo_ext_add(t, coeffs + 2 * i, acc);
    }
    ext_set(out, acc);
}

/* extension-field coset FFT = the base-field transform on each limb (roots and shift are base-field elements) */
static void ext_coset_fft(uint64_t *a /* n x 2 interleaved */, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    uint64_t *l0 = (uint64_t *)malloc(n * 8), *l1 = (uint64_t *)malloc(n * 8);
    for (size_t i = 0; i < n; i++) {
        l0[i] = a[2 * i];
        l1[i] = a[2 * i + 1];
    }
    glo_coset_fft(l0, log_n, shift);
    glo_coset_fft(l1, log_n, shift);
    for (size_t i = 0; i < n; i++) {
        a[2 * i] = l0[i];
        a[2 * i + 1] = l1[i];
    }
    free(l0);
    free(l1);
}

/* ------------------------------------------------ proof layout ------------------------------------------------ */
static unsigned total_arity_bits(const glo_fri_params *p, unsigned upto) {
    unsigned s = 0;
    for (unsigned i = 0; i < upto; i++) s += p->arity_bits[i];
    return s;
}

size_t glo_fri_proof_len(const glo_fri_params *p, const size_t *oracle_cols, size_t n_oracles) {
    unsigned lde_bits = p->degree_bits + p->rate_bits;
    size_t cap = (size_t)4 << p->cap_height;
    size_t len = p->n_rounds * cap;
    size_t per_query = 0;
    for (size_t o = 0; o < n_oracles; o++) per_query += oracle_cols[o] + (size_t)(lde_bits - p->cap_height) * 4;
    for (unsigned r = 0; r < p->n_rounds; r++) {
        unsigned tree_bits = lde_bits - total_arity_bits(p, r + 1);
        per_query += ((size_t)2 << p->arity_bits[r]) + (size_t)(tree_bits - p->cap_height) * 4;
    }
    len += per_query * p->num_queries;
    len += (size_t)2 << (p->degree_bits - total_arity_bits(p, p->n_rounds)); /* final poly, ext */
    len += 1;                                                               /* pow witness */
    return len;
}

/* ------------------------------------------------ prover ------------------------------------------------ */
/* alpha.reduce_polys_base + divide_by_linear + shift_poly for every batch, then "multiply by X" (coeffs.insert(0, 0)) */
static uint64_t *fri_final_poly(const glo_fri_instance *inst, const uint64_t *const *oracle_coeffs, unsigned degree_bits,
                                const uint64_t alpha[2], unsigned flags) {
    size_t n = (size_t)1 << degree_bits;
    const size_t sh = (flags & GLO_FLAG_FRI_NO_MUL_X) ? 0 : 1; /* coefficient j of the sum lands at index j + sh */
    uint64_t *final = (uint64_t *)calloc(2 * n, 8); /* n ext coefficients; index 0 stays zero */
    uint64_t *comp = (uint64_t *)malloc(2 * n * 8);
    uint64_t *quot = (uint64_t *)malloc(2 * n * 8);
    for (size_t b = 0; b < inst->n_batches; b++) {
        const glo_fri_batch *bt = &inst->batches[b];
        /* composition = sum_i alpha^i p_i */
        uint64_t t[2];
        uint64_t *apow = (uint64_t *)malloc(2 * bt->n_polys * 8 + 16); /* alpha^k */
        apow[0] = 1;
        apow[1] = 0;
        for (size_t k = 1; k < bt->n_polys; k++) glo_ext_mul(apow + 2 * (k - 1), alpha, apow + 2 * k);
        if (glo_get_fast()) { /* the timed leg: the same sums, VL coefficients per step (fast_impl.h) */
            const uint64_t **ptrs = (const uint64_t **)malloc(bt->n_polys * sizeof(void *));
            for (size_t k = 0; k < bt->n_polys; k++) ptrs[k] = oracle_coeffs[bt->oracle_index[k]] + (size_t)bt->poly_index[k] * n;
            glo_fast_reduce_polys(ptrs, bt->n_polys, apow, n, comp, glo_get_threads());
            free(ptrs);
        } else {
#pragma omp parallel for num_threads(glo_get_threads()) schedule(static)
        for (long jj = 0; jj < (long)n; jj++) { /* coefficient by coefficient, in parallel (exact field sums: order-free) */
            const size_t j = (size_t)jj;
            uint64_t acc[2] = {0, 0}, tt[2];
            for (size_t k = 0; k < bt->n_polys; k++) {
                const uint64_t *p = oracle_coeffs[bt->oracle_index[k]] + (size_t)bt->poly_index[k] * n;
                ext_mul_base(apow + 2 * k, p[j], tt);
                glo_ext_add(acc, tt, acc);
            }
            comp[2 * j] = acc[0];
            comp[2 * j + 1] = acc[1];
        }
    } /* (scalar path) */
        free(apow);
        /* quotient = (comp(X) - comp(z)) / (X - z): Horner scan from the top, drop the remainder */
        uint64_t acc[2] = {0, 0};
        for (size_t j = n; j-- > 0;) {
            glo_ext_mul(acc, bt->point, t);
            glo_ext_add(t, comp + 2 * j, acc);
            ext_set(quot + 2 * j, acc); /* bs[j] */
        }
        /* bs.pop() removes bs[0] (= comp(z)); quotient coefficient j = bs[j+1] */
        /* final = final * alpha^(n_polys) + quotient ; stored already shifted by one (the insert(0, ZERO)) */
        uint64_t shift[2];
        ext_pow(alpha, bt->n_polys, shift);
        for (size_t j = 0; j + 1 < n; j++) {
            glo_ext_mul(final + 2 * (j + sh), shift, t);
            glo_ext_add(t, quot + 2 * (j + 1), final + 2 * (j + sh));
        }
    }
    free(comp);
    free(quot);
    return final;
}

static void flatten_leaves(const uint64_t *values /* n ext */, size_t n, unsigned log_n, unsigned arity_bits,
                           uint64_t *leaves /* (n >> arity_bits) x (2 << arity_bits) */) {
    /* reverse_index_bits_in_place(values); chunk by arity; flatten */
    size_t arity = (size_t)1 << arity_bits;
    for (size_t p = 0; p < n; p++) {
        size_t src = glo_reverse_bits(p, log_n);
        leaves[(p / arity) * 2 * arity + 2 * (p % arity)] = values[2 * src];
        leaves[(p / arity) * 2 * arity + 2 * (p % arity) + 1] = values[2 * src + 1];
    }
}

int glo_fri_prove(const glo_fri_instance *inst, const uint64_t *const *oracle_coeffs, const uint64_t *const *oracle_leaves,
                  const uint64_t *const *oracle_digests, const size_t *oracle_cols, size_t n_oracles,
                  const glo_fri_params *p, glo_challenger *ch, uint64_t *proof_out) {
    const unsigned lde_bits = p->degree_bits + p->rate_bits;
    const size_t N = (size_t)1 << lde_bits, n = (size_t)1 << p->degree_bits;
    const size_t cap_words = (size_t)4 << p->cap_height;
    uint64_t alpha[2];
    glo_challenger_get_n(ch, alpha, 2); /* get_extension_challenge */

    uint64_t *coeffs = (uint64_t *)calloc(2 * N, 8); /* lde_final_poly: final poly zero-padded to N */
    {
        uint64_t *fp = fri_final_poly(inst, oracle_coeffs, p->degree_bits, alpha, p->flags);
        memcpy(coeffs, fp, 2 * n * 8);
        free(fp);
    }
    const int fast = glo_get_fast(); /* the timed leg (fast.h): same bytes, vectorised / threaded pieces */
    uint64_t *values = (uint64_t *)glo_big_alloc(2 * N * 8);
    if (!fast) {
        memcpy(values, coeffs, 2 * N * 8);
        ext_coset_fft(values, lde_bits, 7);
    }

    /* ---- commit phase (fri_committed_trees) ---- */
    uint64_t **tree_leaves = (uint64_t **)calloc(p->n_rounds + 1, sizeof(void *));
    uint64_t **tree_digests = (uint64_t **)calloc(p->n_rounds + 1, sizeof(void *));
    size_t cur_n = N;
    unsigned cur_bits = lde_bits;
    uint64_t shift = 7;
    uint64_t *w = proof_out;
    for (unsigned r = 0; r < p->n_rounds; r++) {
        unsigned ab = p->arity_bits[r];
        size_t arity = (size_t)1 << ab, n_leaves = cur_n >> ab;
        tree_leaves[r] = (uint64_t *)glo_big_alloc(cur_n * 2 * 8);
        if (fast) /* transform straight into bit-reversed order = the flattened leaves */
            glo_fast_ext_coset_lde_bitrev(coeffs, cur_bits, shift, tree_leaves[r], glo_get_threads());
        else
            flatten_leaves(values, cur_n, cur_bits, ab, tree_leaves[r]);
        tree_digests[r] = (uint64_t *)malloc(glo_merkle_levels_len(n_leaves, p->cap_height) * 8);
        glo_merkle_build(tree_leaves[r], n_leaves, 2 * arity, p->cap_height, tree_digests[r], w);
        glo_challenger_observe(ch, w, cap_words);
        w += cap_words;
        uint64_t beta[2], t[2];
        glo_challenger_get_n(ch, beta, 2);
        /* coeffs'[j] = sum_i beta^i coeffs[arity*j + i] */
        if (fast) { /* out of place so that the folds can run in parallel */
            uint64_t *folded = (uint64_t *)malloc(2 * n_leaves * 8);
#pragma omp parallel for num_threads(glo_get_threads()) schedule(static)
            for (long jj = 0; jj < (long)n_leaves; jj++) {
                const size_t j = (size_t)jj;
                uint64_t acc[2] = {0, 0}, tt[2];
                for (size_t i = arity; i-- > 0;) {
                    glo_ext_mul(acc, beta, tt);
                    glo_ext_add(tt, coeffs + 2 * (arity * j + i), acc);
                }
                ext_set(folded + 2 * j, acc);
            }
            memcpy(coeffs, folded, 2 * n_leaves * 8);
            free(folded);
        } else {
        for (size_t j = 0; j < n_leaves; j++) {
            uint64_t acc[2] = {0, 0};
            for (size_t i = arity; i-- > 0;) {
                glo_ext_mul(acc, beta, t);
                glo_ext_add(t, coeffs + 2 * (arity * j + i), acc);
            }
            ext_set(coeffs + 2 * j, acc);
        }
    } /* (scalar path) */
        cur_n = n_leaves;
        cur_bits -= ab;
        shift = glo_pow(shift, arity);
        if (!fast) {
            memcpy(values, coeffs, 2 * cur_n * 8);
            ext_coset_fft(values, cur_bits, shift);
        }
    }
    /* coeffs.truncate(len >> rate_bits); observe */
    size_t n_final = cur_n >> p->rate_bits;
    uint64_t *final_poly = (uint64_t *)malloc(2 * n_final * 8);
    for (size_t i = 0; i < 2 * n_final; i++) final_poly[i] = glo_canon(coeffs[i]);
    glo_challenger_observe(ch, final_poly, 2 * n_final);

    /* ---- proof of work: smallest i whose response has pow_bits leading zeros ---- */
    uint64_t pow_witness = 0;
    if (p->flags & GLO_FLAG_POW_DUPLEX) {
        /* response = the challenge squeezed right after observing i: overwrite the pending inputs and i into the sponge
         * state, permute, pop the last rate word */
        uint64_t base[12], st[12];
        memcpy(base, ch->state, sizeof base);
        for (int k = 0; k < ch->n_in; k++) base[k] = ch->in[k];
        if (fast) pow_witness = glo_fast_pow_search(base, (unsigned)ch->n_in, 7, p->pow_bits, glo_get_threads());
        else {
        for (uint64_t i = 0;; i++) {
            memcpy(st, base, sizeof st);
            st[ch->n_in] = i;
            glo_poseidon_permute(st);
            if (p->pow_bits == 0 || (glo_canon(st[7]) >> (64 - p->pow_bits)) == 0) {
                pow_witness = i;
                break;
            }
        }
    } /* (scalar path) */
        glo_challenger_observe(ch, &pow_witness, 1);
        uint64_t resp = glo_challenger_get(ch);
        if (p->pow_bits && (resp >> (64 - p->pow_bits)) != 0) return -1; /* cannot happen: same computation */
    } else {
        uint64_t cur_hash[5], h[4];
        glo_challenger_get_n(ch, cur_hash, 4);
        if (fast) {
            uint64_t base[12] = {cur_hash[0], cur_hash[1], cur_hash[2], cur_hash[3], 0, 0, 0, 0, 0, 0, 0, 0};
            pow_witness = glo_fast_pow_search(base, 4, 0, p->pow_bits, glo_get_threads());
        } else {
        for (uint64_t i = 0;; i++) {
            cur_hash[4] = i;
            glo_hash_no_pad(cur_hash, 5, h);
            if (p->pow_bits == 0 || (h[0] >> (64 - p->pow_bits)) == 0) {
                pow_witness = i;
                break;
            }
        }
    } /* (scalar path) */
    }

    /* ---- query phase ---- */
    for (unsigned q = 0; q < p->num_queries; q++) {
        size_t x_index = (size_t)(glo_challenger_get(ch) % N);
        for (size_t o = 0; o < n_oracles; o++) {
            memcpy(w, oracle_leaves[o] + x_index * oracle_cols[o], oracle_cols[o] * 8);
            w += oracle_cols[o];
            glo_merkle_prove(oracle_digests[o], N, p->cap_height, x_index, w);
            w += (size_t)(lde_bits - p->cap_height) * 4;
        }
        size_t tn = N;
        unsigned tb = lde_bits;
        for (unsigned r = 0; r < p->n_rounds; r++) {
            unsigned ab = p->arity_bits[r];
            size_t arity = (size_t)1 << ab, n_leaves = tn >> ab;
            size_t leaf = x_index >> ab;
            memcpy(w, tree_leaves[r] + leaf * 2 * arity, 2 * arity * 8);
            w += 2 * arity;
            glo_merkle_prove(tree_digests[r], n_leaves, p->cap_height, leaf, w);
            w += (size_t)(tb - ab - p->cap_height) * 4;
            x_index = leaf;
            tn = n_leaves;
            tb -= ab;
        }
    }
    memcpy(w, final_poly, 2 * n_final * 8);
    w += 2 * n_final;
    *w++ = pow_witness;

    for (unsigned r = 0; r < p->n_rounds; r++) {
        glo_big_free(tree_leaves[r]);
        free(tree_digests[r]);
    }
    free(tree_leaves);
    free(tree_digests);
    free(final_poly);
    free(coeffs);
    glo_big_free(values);
    return 0;
}

/* ------------------------------------------------ verifier ------------------------------------------------ */
/* interpolate {(coset_start * g^i, evals_natural[i])} and evaluate at beta (barycentric form as in plonky2's
 * interpolate(); any exact interpolation gives the same field element) */
static void compute_evaluation(uint64_t x, size_t x_index_within_coset, unsigned arity_bits, const uint64_t *evals_bitrev,
                               const uint64_t beta[2], uint64_t out[2]) {
    size_t arity = (size_t)1 << arity_bits;
    uint64_t g = glo_root_of_unity(arity_bits);
    size_t rev_idx = glo_reverse_bits(x_index_within_coset, arity_bits);
    uint64_t coset_start = glo_mul(x, glo_pow(g, arity - rev_idx));
    uint64_t pts[64];
    uint64_t ev[64][2];
    uint64_t y = 1;
    for (size_t i = 0; i < arity; i++) {
        pts[i] = glo_mul(coset_start, y);
        y = glo_mul(y, g);
        size_t src = glo_reverse_bits(i, arity_bits); /* reverse_index_bits_in_place(evals) */
        ev[i][0] = evals_bitrev[2 * src];
        ev[i][1] = evals_bitrev[2 * src + 1];
    }
    /* Lagrange: sum_i ev_i * prod_{j != i} (beta - x_j) / (x_i - x_j) */
    uint64_t acc[2] = {0, 0};
    for (size_t i = 0; i < arity; i++) {
        uint64_t num[2] = {1, 0}, t[2], d[2];
        uint64_t den = 1;
        for (size_t j = 0; j < arity; j++) {
            if (j == i) continue;
            d[0] = glo_sub(beta[0], pts[j]);
            d[1] = beta[1];
            glo_ext_mul(num, d, t);
            ext_set(num, t);
            den = glo_mul(den, glo_sub(pts[i], pts[j]));
        }
        ext_mul_base(num, glo_inv(den), t);
        glo_ext_mul(t, ev[i], d);
        glo_ext_add(acc, d, acc);
    }
    ext_set(out, acc);
}

int glo_fri_verify(const glo_fri_instance *inst, const uint64_t *const *batch_openings /* per batch: n_polys ext */,
                   const uint64_t *const *oracle_caps, const size_t *oracle_cols, size_t n_oracles,
                   const glo_fri_params *p, glo_challenger *ch, const uint64_t *proof) {
    const unsigned lde_bits = p->degree_bits + p->rate_bits;
    const size_t N = (size_t)1 << lde_bits;
    const size_t cap_words = (size_t)4 << p->cap_height;
    /* ---- challenges (fri/challenges.rs::fri_challenges) ---- */
    uint64_t alpha[2], betas[16][2];
    glo_challenger_get_n(ch, alpha, 2);
    const uint64_t *caps = proof;
    for (unsigned r = 0; r < p->n_rounds; r++) {
        glo_challenger_observe(ch, caps + r * cap_words, cap_words);
        glo_challenger_get_n(ch, betas[r], 2);
    }
    size_t n_final = (size_t)1 << (p->degree_bits - total_arity_bits(p, p->n_rounds));
    size_t len = glo_fri_proof_len(p, oracle_cols, n_oracles);
    const uint64_t *final_poly = proof + len - 1 - 2 * n_final;
    uint64_t pow_witness = proof[len - 1];
    glo_challenger_observe(ch, final_poly, 2 * n_final);
    if (pow_witness >= GLO_P) return -10;
    if (p->flags & GLO_FLAG_POW_DUPLEX) {
        glo_challenger_observe(ch, &pow_witness, 1);
        uint64_t resp = glo_challenger_get(ch);
        if (p->pow_bits && (resp >> (64 - p->pow_bits)) != 0) return -1; /* invalid proof of work */
    } else {
        uint64_t cur_hash[5], h[4];
        glo_challenger_get_n(ch, cur_hash, 4);
        cur_hash[4] = pow_witness;
        glo_hash_no_pad(cur_hash, 5, h);
        if (p->pow_bits && (h[0] >> (64 - p->pow_bits)) != 0) return -1; /* invalid proof of work */
    }

    /* PrecomputedReducedOpenings: sum_i v_i alpha^i per batch */
    uint64_t reduced[8][2];
    for (size_t b = 0; b < inst->n_batches; b++) {
        uint64_t acc[2] = {0, 0}, t[2];
        for (size_t k = inst->batches[b].n_polys; k-- > 0;) {
            glo_ext_mul(acc, alpha, t);
            glo_ext_add(t, batch_openings[b] + 2 * k, acc);
        }
        ext_set(reduced[b], acc);
    }

    const uint64_t *w = proof + p->n_rounds * cap_words;
    uint64_t wN = glo_root_of_unity(lde_bits);
    for (unsigned q = 0; q < p->num_queries; q++) {
        size_t x_index = (size_t)(glo_challenger_get(ch) % N);
        /* initial trees */
        const uint64_t *rows[8];
        for (size_t o = 0; o < n_oracles; o++) {
            rows[o] = w;
            const uint64_t *path = w + oracle_cols[o];
            if (!glo_merkle_verify(rows[o], oracle_cols[o], x_index, path, lde_bits - p->cap_height, oracle_caps[o],
                                   p->cap_height))
                return -2;
            w = path + (size_t)(lde_bits - p->cap_height) * 4;
        }
        uint64_t subgroup_x = glo_mul(7, glo_pow(wN, glo_reverse_bits(x_index, lde_bits)));
        /* fri_combine_initial */
        uint64_t sum[2] = {0, 0}, t[2], u[2];
        for (size_t b = 0; b < inst->n_batches; b++) {
            const glo_fri_batch *bt = &inst->batches[b];
            uint64_t acc[2] = {0, 0};
            for (size_t k = bt->n_polys; k-- > 0;) {
                glo_ext_mul(acc, alpha, t);
                acc[0] = glo_add(t[0], rows[bt->oracle_index[k]][bt->poly_index[k]]);
                acc[1] = t[1];
            }
            uint64_t numer[2], denom[2], dinv[2], shift[2];
            glo_ext_sub(acc, reduced[b], numer);
            denom[0] = glo_sub(subgroup_x, bt->point[0]);
            denom[1] = glo_sub(0, bt->point[1]);
            glo_ext_inv(denom, dinv);
            ext_pow(alpha, bt->n_polys, shift);
            glo_ext_mul(sum, shift, t);
            glo_ext_mul(numer, dinv, u);
            glo_ext_add(t, u, sum);
        }
        uint64_t old_eval[2];
        if (p->flags & GLO_FLAG_FRI_NO_MUL_X)
            ext_set(old_eval, sum);
        else
            ext_mul_base(sum, subgroup_x, old_eval); /* the "multiply by X" of prove_openings */
        unsigned tb = lde_bits;
        for (unsigned r = 0; r < p->n_rounds; r++) {
            unsigned ab = p->arity_bits[r];
            size_t arity = (size_t)1 << ab;
            const uint64_t *evals = w;
            const uint64_t *path = w + 2 * arity;
            size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (!ext_eq(evals + 2 * within, old_eval)) return -3;
            compute_evaluation(subgroup_x, within, ab, evals, betas[r], old_eval);
            if (!glo_merkle_verify(evals, 2 * arity, coset_index, path, tb - ab - p->cap_height, caps + r * cap_words,
                                   p->cap_height))
                return -4;
            w = path + (size_t)(tb - ab - p->cap_height) * 4;
            for (unsigned k = 0; k < ab; k++) subgroup_x = glo_mul(subgroup_x, subgroup_x);
            x_index = coset_index;
            tb -= ab;
        }
        uint64_t fx[2] = {subgroup_x, 0}, fe[2];
        ext_poly_eval(final_poly, n_final, fx, fe);
        if (!ext_eq(fe, old_eval)) return -5;
    }
    (void)EXT_ZERO;
    return 0;
}
