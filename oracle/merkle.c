/*
 * merkle.c — CPU ORACLE (test infrastructure): Poseidon Merkle tree with cap, and PolynomialBatch commit.
 * Restates plonky2 @ f99ed9c src/hash/{merkle_tree.rs,merkle_proofs.rs} and src/fri/oracle.rs
 * (PolynomialBatch::from_values / from_coeffs); dependency absent from /root/reference (Cargo.toml:12).
 * Reached from the reference only through CircuitData::prove / CircuitBuilder::build
 * (e.g. src/rollup/circuits/mod.rs:605,1247).
 *
 * Conventions: leaf digest = hash_or_noop(row); parent = two_to_one(left,right);
 * cap[i] = root of the i-th contiguous block of n_leaves / 2^cap_height leaves; proof = siblings bottom-up.
 * from_values: coeffs = ifft(values); lde = coset_fft(zero-pad(coeffs, 2^rate_bits), shift 7); leaves =
 * transpose(lde) with the ROW index bit-reversed.  "parity unpinned" for the orderings (no reference vectors);
 * the hashing inside is pinned by tests/golden/poseidon_kat.json.
 */
#include <stdlib.h>
#include <string.h>

#include "fast.h"
#include "gl_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
void glo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int glo_get_threads(void) { return g_threads; }

/* timed leg: transforms of at least 2^GLO_FAST_INNER_LOG points run one column at a time with the threads inside the
 * transform, smaller ones one column per thread (default 22; measured on the bench host, profiles/r05_cpu_baseline.md) */
static unsigned inner_log(void) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("GLO_FAST_INNER_LOG");
        v = e ? atoi(e) : 22;
    }
    return (unsigned)v;
}

/* GLO_TIMING=2: wall time of the pieces of a commitment on stderr */
#include <stdio.h>
#include <time.h>
static double sub_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static int sub_on(void) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("GLO_TIMING");
        v = e && atoi(e) >= 2;
    }
    return v;
}
#define SUB(name)                                                                  \
    do {                                                                           \
        if (sub_on()) {                                                            \
            double t_ = sub_now();                                                 \
            fprintf(stderr, "      commit: %-18s %8.3f s\n", name, t_ - t_sub); \
            t_sub = t_;                                                            \
        }                                                                          \
    } while (0)

static unsigned log2_exact(size_t n) {
    unsigned l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

size_t glo_merkle_levels_len(size_t n_leaves, unsigned cap_height) {
    unsigned top = log2_exact(n_leaves) - cap_height;
    size_t total = 0;
    for (unsigned l = 0; l <= top; l++) total += (n_leaves >> l) * 4;
    return total;
}

void glo_merkle_build(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height,
                      uint64_t *digests_out, uint64_t *cap_out) {
    unsigned top = log2_exact(n_leaves) - cap_height;
    uint64_t *dig = digests_out;
    int own = 0;
    if (!dig) {
        dig = (uint64_t *)malloc(glo_merkle_levels_len(n_leaves, cap_height) * sizeof(uint64_t));
        own = 1;
    }
    (void)glo_poseidon_round_constants(); /* derive before going parallel */
    if (glo_get_fast()) { /* the timed leg: the same digests, VL leaves / node pairs per permutation (fast_impl.h) */
        double t_sub = sub_now();
        glo_fast_merkle_levels(leaves, n_leaves, leaf_len, top, dig, g_threads);
        SUB("  hash levels");
        memcpy(cap_out, dig + glo_merkle_levels_len(n_leaves, cap_height) - ((size_t)4 << cap_height), ((size_t)4 << cap_height) * sizeof(uint64_t));
        if (own) free(dig);
        return;
    }
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long i = 0; i < (long)n_leaves; i++) glo_hash_or_noop(leaves + (size_t)i * leaf_len, leaf_len, dig + (size_t)i * 4);
    uint64_t *prev = dig;
    for (unsigned l = 1; l <= top; l++) {
        size_t cnt = n_leaves >> l;
        uint64_t *cur = prev + (n_leaves >> (l - 1)) * 4;
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (long i = 0; i < (long)cnt; i++) glo_two_to_one(prev + (size_t)i * 8, prev + (size_t)i * 8 + 4, cur + (size_t)i * 4);
        prev = cur;
    }
    memcpy(cap_out, prev, ((size_t)4 << cap_height) * sizeof(uint64_t));
    if (own) free(dig);
}

void glo_merkle_prove(const uint64_t *digests, size_t n_leaves, unsigned cap_height, size_t leaf_index,
                      uint64_t *siblings_out) {
    unsigned top = log2_exact(n_leaves) - cap_height;
    const uint64_t *lvl = digests;
    for (unsigned l = 0; l < top; l++) {
        size_t sib = (leaf_index >> l) ^ 1;
        memcpy(siblings_out + (size_t)l * 4, lvl + sib * 4, 4 * sizeof(uint64_t));
        lvl += (n_leaves >> l) * 4;
    }
}

int glo_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t leaf_index, const uint64_t *siblings,
                      size_t n_siblings, const uint64_t *cap, unsigned cap_height) {
    uint64_t cur[4], nxt[4];
    glo_hash_or_noop(leaf, leaf_len, cur);
    size_t idx = leaf_index;
    for (size_t l = 0; l < n_siblings; l++) {
        if (idx & 1)
            glo_two_to_one(siblings + l * 4, cur, nxt);
        else
            glo_two_to_one(cur, siblings + l * 4, nxt);
        memcpy(cur, nxt, sizeof cur);
        idx >>= 1;
    }
    if (idx >= ((size_t)1 << cap_height)) return 0;
    return memcmp(cur, cap + idx * 4, sizeof cur) == 0;
}

void glo_batch_from_coeffs(const uint64_t *coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits,
                           unsigned cap_height, uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out) {
    glo_batch_from_coeffs_salted(coeffs, n_cols, log_n, rate_bits, cap_height, NULL, leaves_out, digests_out, cap_out);
}

void glo_batch_from_coeffs_salted(const uint64_t *coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                                  const uint64_t *salts, uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    const size_t width = n_cols + (salts ? GLO_SALT_SIZE : 0); /* words per leaf */
    unsigned log_N = log_n + rate_bits;
    /* per-column LDE into a column-major scratch, rows already bit-reversed (plonky2: lde_values, then
     * transpose + reverse_index_bits_in_place) */
    double t_sub = sub_now();
    uint64_t *cm = (uint64_t *)glo_big_alloc(width * N * sizeof(uint64_t));
    if (salts) /* the salt columns ride along as columns n_cols .. n_cols + 3 of the scratch, rows bit-reversed like the rest */
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (long i = 0; i < (long)N; i++) {
            const size_t p = glo_reverse_bits((size_t)i, log_N);
            for (size_t s = 0; s < GLO_SALT_SIZE; s++) cm[(n_cols + s) * N + p] = glo_canon(salts[s * N + (size_t)i]);
        }
    if (glo_get_fast()) { /* the timed leg: the decimation-in-frequency transform leaves the rows bit-reversed by itself */
        /* big columns one after the other with all threads inside (one 2^23-point column is 64 MB: the cores share it in
         * their caches), small ones one per thread */
        const int inner = (log_N >= inner_log() && g_threads > 1) ? g_threads : 1;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1) if (inner == 1)
        for (long c = 0; c < (long)n_cols; c++) glo_fast_lde_column(coeffs + (size_t)c * n, n, cm + (size_t)c * N, log_N, 7, inner);
    } else {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
    for (long c = 0; c < (long)n_cols; c++) {
        uint64_t *col = (uint64_t *)calloc(N, sizeof(uint64_t));
        memcpy(col, coeffs + (size_t)c * n, n * sizeof(uint64_t));
        glo_coset_fft(col, log_N, 7);
        uint64_t *dst = cm + (size_t)c * N;
        for (size_t i = 0; i < N; i++) dst[glo_reverse_bits(i, log_N)] = col[i];
        free(col);
    }
    } /* (scalar path) */
    SUB("lde");
    /* blocked transpose to row-major leaves */
    const size_t RB = 64;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long rb = 0; rb < (long)((N + RB - 1) / RB); rb++) {
        size_t r0 = (size_t)rb * RB, r1 = r0 + RB < N ? r0 + RB : N;
        for (size_t c = 0; c < width; c++) {
            const uint64_t *src = cm + c * N;
            for (size_t r = r0; r < r1; r++) leaves_out[r * width + c] = src[r];
        }
    }
    SUB("transpose");
    glo_big_free(cm);
    SUB("free");
    glo_merkle_build(leaves_out, N, width, cap_height, digests_out, cap_out);
    SUB("merkle");
}

void glo_batch_from_values(const uint64_t *cols, size_t n_cols, unsigned log_n, unsigned rate_bits,
                           unsigned cap_height, uint64_t *coeffs_out, uint64_t *leaves_out,
                           uint64_t *digests_out, uint64_t *cap_out) {
    glo_batch_from_values_salted(cols, n_cols, log_n, rate_bits, cap_height, NULL, coeffs_out, leaves_out, digests_out, cap_out);
}

void glo_batch_from_values_salted(const uint64_t *cols, size_t n_cols, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                                  const uint64_t *salts, uint64_t *coeffs_out, uint64_t *leaves_out, uint64_t *digests_out,
                                  uint64_t *cap_out) {
    size_t n = (size_t)1 << log_n;
    double t_sub = sub_now();
    memcpy(coeffs_out, cols, n_cols * n * sizeof(uint64_t));
    SUB("copy values");
    if (glo_get_fast() && log_n >= inner_log() && g_threads > 1) {
        for (size_t c = 0; c < n_cols; c++) glo_fast_fft_mt(coeffs_out + c * n, log_n, 1, g_threads);
    } else {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
    for (long c = 0; c < (long)n_cols; c++) glo_ifft(coeffs_out + (size_t)c * n, log_n);
    } /* (scalar path) */
    SUB("ifft");
    glo_batch_from_coeffs_salted(coeffs_out, n_cols, log_n, rate_bits, cap_height, salts, leaves_out, digests_out, cap_out);
}
