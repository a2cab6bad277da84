/*
 * fast.h — CPU ORACLE, TIMED LEG ONLY (test infrastructure, NOT product code): the vectorised form of the oracle that
 * bench.py's `cpu_baseline` times.  glo_set_fast(1) routes the heavy loops of the restatement (leaf / node hashing, NTT,
 * quotient evaluation, ...) through the packed-Goldilocks code of fast_impl.h; the proof that comes out is the same
 * byte for byte (tests/test_oracle_fast.py).  The scalar path (glo_set_fast(0), the default) stays the checker.
 */
#ifndef GLO_FAST_H
#define GLO_FAST_H
#include <stddef.h>
#include <stdint.h>

#include "gl_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 = scalar restatement (default; the checker), 1 = vectorised (the timed leg) */
void glo_set_fast(int on);
int glo_get_fast(void);
/* force a backend: 512 / 256 / 1, 0 = best the CPU supports (the default); returns the backend now in use */
int glo_fast_set_isa(int isa);
const char *glo_fast_isa_name(void);

/* large buffers: plain malloc / free in the scalar mode, cached huge-page blocks in the timed leg (fast.c) */
void *glo_big_alloc(size_t bytes);
void glo_big_free(void *p);
void glo_big_trim(void);

/* dispatched entry points (same contracts as their scalar counterparts) */
void glo_fast_permute_many(uint64_t *states, size_t n); /* n x 12 words, in place, canonical out */
void glo_fast_merkle_levels(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned top, uint64_t *dig, int threads);

/* NTT pieces: decimation in frequency, natural order in, bit-reversed order out (inverse: root^-1, no 1/n) */
#define GLO_FAST_MAX_S 13u
const uint64_t *glo_fast_tw(int inverse, unsigned s); /* w_(2^s)^(+-j), j < 2^(s-1); s <= GLO_FAST_MAX_S */
void glo_fast_ntt_dif_plain(uint64_t *a, unsigned log_n, int inverse); /* scalar layers, any size */
void glo_fast_ntt_dif(uint64_t *a, unsigned log_n, int inverse, int threads);
void glo_fast_bitrev_scale(const uint64_t *src, uint64_t *dst, unsigned log_n, uint64_t scale, size_t i0, size_t i1);
/* whole transforms of ONE column with the threads inside (a 2^23-point column is 64 MB: all cores share it in L3) */
void glo_fast_scale_powers_mt(uint64_t *a, size_t n, uint64_t shift, int threads);
void glo_fast_fft_mt(uint64_t *a, unsigned log_n, int inverse, int threads); /* natural in, natural out; inverse: with 1/n */
void glo_fast_lde_column(const uint64_t *coeffs, size_t n, uint64_t *dst, unsigned log_N, uint64_t shift, int threads);
void glo_fast_scale_powers(uint64_t *a, size_t n, uint64_t shift, uint64_t start);

/* plonk.c::glo_quotient_chunks before its inverse transforms: vals[k][i], k < num_challenges, i = natural coset index */
void glo_fast_quotient_values(const glo_circuit *cc, const uint64_t *cs_leaves, const uint64_t *wires_leaves,
                              const uint64_t *zs_leaves, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                              const uint64_t pi_hash[4], uint64_t *vals, int threads);

/* plonk.c::glo_partial_products_and_zs: the chunk quotients q[(ch n_chunks + ck) n + i] */
void glo_fast_chunk_quotients(const glo_circuit *cc, const uint64_t *wires, const uint64_t *sigmas, const uint64_t *betas,
                              const uint64_t *gammas, uint64_t *q, int threads);
void glo_fast_eval_with_powers(const uint64_t *coeffs, size_t n, const uint64_t *zp0, const uint64_t *zp1, uint64_t out[2]);
void glo_fast_reduce_polys(const uint64_t *const *polys, size_t n_polys, const uint64_t *apow, size_t n, uint64_t *comp, int threads);
uint64_t glo_fast_pow_search(const uint64_t base[12], unsigned slot, unsigned out_index, unsigned pow_bits, int threads);
void glo_fast_ext_coset_lde_bitrev(const uint64_t *coeffs, unsigned log_n, uint64_t shift, uint64_t *out, int threads);

#define GLO_FAST_DECLARE(SUF)                                                                                                \
    void glo_fast_permute_many_##SUF(uint64_t *states, size_t n);                                                            \
    void glo_fast_merkle_levels_##SUF(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned top, uint64_t *dig, \
                                      int threads);                                                          \
    void glo_fast_ntt_dif_##SUF(uint64_t *a, unsigned log_n, int inverse, int threads);                                      \
    void glo_fast_bitrev_scale_##SUF(const uint64_t *src, uint64_t *dst, unsigned log_n, uint64_t scale, size_t i0, size_t i1); \
    void glo_fast_scale_powers_##SUF(uint64_t *a, size_t n, uint64_t shift, uint64_t start);                                 \
    void glo_fast_quotient_values_##SUF(const glo_circuit *cc, const uint64_t *cs_leaves, const uint64_t *wires_leaves,      \
                                        const uint64_t *zs_leaves, const uint64_t *betas, const uint64_t *gammas,            \
                                        const uint64_t *alphas, const uint64_t pi_hash[4], uint64_t *vals, int threads);     \
    void glo_fast_chunk_quotients_##SUF(const glo_circuit *cc, const uint64_t *wires, const uint64_t *sigmas,                \
                                        const uint64_t *betas, const uint64_t *gammas, uint64_t *q, int threads);            \
    void glo_fast_eval_with_powers_##SUF(const uint64_t *coeffs, size_t n, const uint64_t *zp0, const uint64_t *zp1,         \
                                         uint64_t out[2]);                                                                   \
    void glo_fast_reduce_polys_##SUF(const uint64_t *const *polys, size_t n_polys, const uint64_t *apow, size_t n,           \
                                     uint64_t *comp, int threads);                                                           \
    uint64_t glo_fast_pow_search_##SUF(const uint64_t base[12], unsigned slot, unsigned out_index, unsigned pow_bits,        \
                                       int threads);                                                                         \
    void glo_fast_ext_coset_lde_bitrev_##SUF(const uint64_t *coeffs, unsigned log_n, uint64_t shift, uint64_t *out, int threads);
GLO_FAST_DECLARE(avx512)
GLO_FAST_DECLARE(avx2)
GLO_FAST_DECLARE(scalar)

#ifdef __cplusplus
}
#endif
#endif
