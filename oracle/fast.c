/*
 * fast.c — CPU ORACLE, TIMED LEG ONLY (test infrastructure, NOT product code): run-time choice between the three
 * instantiations of fast_impl.h (AVX-512, AVX2, plain C) and the switch between the scalar restatement (the checker) and
 * the vectorised form (what bench.py's cpu_baseline times).
 */
#include "fast.h"

static int g_fast = 0;
static int g_isa = -1; /* 512 / 256 / 1 */

void glo_big_trim(void);
void glo_set_fast(int on) {
    g_fast = on != 0;
    if (!g_fast) glo_big_trim();
}
int glo_get_fast(void) { return g_fast; }

static int best_isa(void) {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
        __builtin_cpu_supports("avx512bw"))
        return 512;
    if (__builtin_cpu_supports("avx2")) return 256;
    return 1;
}

int glo_fast_set_isa(int isa) {
    int best = best_isa();
    if (isa == 0 || isa > best) isa = best;
    if (isa != 512 && isa != 256) isa = isa >= 256 ? 256 : 1;
    g_isa = isa;
    return g_isa;
}
static int isa(void) {
    if (g_isa < 0) glo_fast_set_isa(0);
    return g_isa;
}
const char *glo_fast_isa_name(void) { return isa() == 512 ? "avx512" : isa() == 256 ? "avx2" : "scalar"; }

#define DISPATCH(name, ...)                      \
    do {                                         \
        switch (isa()) {                         \
            case 512: name##_avx512(__VA_ARGS__); break; \
            case 256: name##_avx2(__VA_ARGS__); break;   \
            default: name##_scalar(__VA_ARGS__); break;  \
        }                                        \
    } while (0)

void glo_fast_permute_many(uint64_t *states, size_t n) { DISPATCH(glo_fast_permute_many, states, n); }
void glo_fast_merkle_levels(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned top, uint64_t *dig, int threads) {
    DISPATCH(glo_fast_merkle_levels, leaves, n_leaves, leaf_len, top, dig, threads);
}

/* ---- NTT layer tables, shared by the three instantiations ---- */
#include <stdlib.h>

#include "gl_oracle.h"

static uint64_t *g_tw[2][GLO_FAST_MAX_S + 1];
static int g_tw_ready = 0;

static void build_tw(void) {
    for (int inv = 0; inv < 2; inv++)
        for (unsigned s = 1; s <= GLO_FAST_MAX_S; s++) {
            size_t h = (size_t)1 << (s - 1);
            uint64_t w = glo_root_of_unity(s);
            if (inv) w = glo_inv(w);
            uint64_t *t = (uint64_t *)aligned_alloc(64, (h < 8 ? 8 : h) * sizeof(uint64_t));
            t[0] = 1;
            for (size_t j = 1; j < h; j++) t[j] = glo_mul(t[j - 1], w);
            g_tw[inv][s] = t;
        }
}

const uint64_t *glo_fast_tw(int inverse, unsigned s) {
    if (!__atomic_load_n(&g_tw_ready, __ATOMIC_ACQUIRE)) {
#pragma omp critical(glo_fast_tw_build)
        {
            if (!g_tw_ready) {
                build_tw();
                __atomic_store_n(&g_tw_ready, 1, __ATOMIC_RELEASE);
            }
        }
    }
    return g_tw[inverse != 0][s];
}

void glo_fast_ntt_dif_plain(uint64_t *a, unsigned log_n, int inverse) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) a[i] = glo_canon(a[i]);
    for (unsigned s = log_n; s >= 1; s--) {
        const size_t h = (size_t)1 << (s - 1);
        uint64_t w = glo_root_of_unity(s);
        if (inverse) w = glo_inv(w);
        for (size_t k = 0; k < n; k += 2 * h) {
            uint64_t t = 1;
            for (size_t j = 0; j < h; j++) {
                uint64_t x = a[k + j], y = a[k + j + h];
                a[k + j] = glo_add(x, y);
                a[k + j + h] = glo_mul(glo_sub(x, y), t);
                t = glo_mul(t, w);
            }
        }
    }
}

void glo_fast_ntt_dif(uint64_t *a, unsigned log_n, int inverse, int threads) { DISPATCH(glo_fast_ntt_dif, a, log_n, inverse, threads); }
void glo_fast_bitrev_scale(const uint64_t *src, uint64_t *dst, unsigned log_n, uint64_t scale, size_t i0, size_t i1) {
    DISPATCH(glo_fast_bitrev_scale, src, dst, log_n, scale, i0, i1);
}
void glo_fast_scale_powers(uint64_t *a, size_t n, uint64_t shift, uint64_t start) { DISPATCH(glo_fast_scale_powers, a, n, shift, start); }
void glo_fast_quotient_values(const glo_circuit *cc, const uint64_t *cs_leaves, const uint64_t *wires_leaves,
                              const uint64_t *zs_leaves, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                              const uint64_t pi_hash[4], uint64_t *vals, int threads) {
    DISPATCH(glo_fast_quotient_values, cc, cs_leaves, wires_leaves, zs_leaves, betas, gammas, alphas, pi_hash, vals, threads);
}
void glo_fast_chunk_quotients(const glo_circuit *cc, const uint64_t *wires, const uint64_t *sigmas, const uint64_t *betas,
                              const uint64_t *gammas, uint64_t *q, int threads) {
    DISPATCH(glo_fast_chunk_quotients, cc, wires, sigmas, betas, gammas, q, threads);
}
void glo_fast_eval_with_powers(const uint64_t *coeffs, size_t n, const uint64_t *zp0, const uint64_t *zp1, uint64_t out[2]) {
    DISPATCH(glo_fast_eval_with_powers, coeffs, n, zp0, zp1, out);
}
void glo_fast_reduce_polys(const uint64_t *const *polys, size_t n_polys, const uint64_t *apow, size_t n, uint64_t *comp, int threads) {
    DISPATCH(glo_fast_reduce_polys, polys, n_polys, apow, n, comp, threads);
}
uint64_t glo_fast_pow_search(const uint64_t base[12], unsigned slot, unsigned out_index, unsigned pow_bits, int threads) {
    switch (isa()) {
        case 512: return glo_fast_pow_search_avx512(base, slot, out_index, pow_bits, threads);
        case 256: return glo_fast_pow_search_avx2(base, slot, out_index, pow_bits, threads);
        default: return glo_fast_pow_search_scalar(base, slot, out_index, pow_bits, threads);
    }
}
void glo_fast_ext_coset_lde_bitrev(const uint64_t *coeffs, unsigned log_n, uint64_t shift, uint64_t *out, int threads) {
    DISPATCH(glo_fast_ext_coset_lde_bitrev, coeffs, log_n, shift, out, threads);
}

#include <string.h>

void glo_fast_scale_powers_mt(uint64_t *a, size_t n, uint64_t shift, int threads) {
    const size_t SEG = 4096;
    if (threads <= 1 || n < 4 * SEG) {
        glo_fast_scale_powers(a, n, shift, 1);
        return;
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long sg = 0; sg < (long)(n / SEG); sg++) glo_fast_scale_powers(a + (size_t)sg * SEG, SEG, shift, glo_pow(shift, (uint64_t)sg * SEG));
}

void glo_fast_fft_mt(uint64_t *a, unsigned log_n, int inverse, int threads) {
    const size_t n = (size_t)1 << log_n;
    uint64_t *t = (uint64_t *)malloc(n * sizeof(uint64_t));
    const uint64_t scale = inverse ? glo_inv((uint64_t)n) : 1;
    if (threads <= 1 || log_n < 16) {
        memcpy(t, a, n * sizeof(uint64_t));
        glo_fast_ntt_dif(t, log_n, inverse, 1);
        glo_fast_bitrev_scale(t, a, log_n, scale, 0, n);
    } else {
        const size_t SEG = 4096;
#pragma omp parallel for num_threads(threads) schedule(static)
        for (long sg = 0; sg < (long)(n / SEG); sg++) memcpy(t + (size_t)sg * SEG, a + (size_t)sg * SEG, SEG * sizeof(uint64_t));
        glo_fast_ntt_dif(t, log_n, inverse, threads);
#pragma omp parallel for num_threads(threads) schedule(static)
        for (long sg = 0; sg < (long)(n / SEG); sg++) glo_fast_bitrev_scale(t, a, log_n, scale, (size_t)sg * SEG, (size_t)(sg + 1) * SEG);
    }
    free(t);
}

/* dst (2^log_N words) = the values of the polynomial on shift * <w_N>, ROWS IN BIT-REVERSED ORDER */
void glo_fast_lde_column(const uint64_t *coeffs, size_t n, uint64_t *dst, unsigned log_N, uint64_t shift, int threads) {
    const size_t N = (size_t)1 << log_N, SEG = 4096;
    if (threads <= 1 || n < 4 * SEG) {
        memcpy(dst, coeffs, n * sizeof(uint64_t));
        glo_fast_scale_powers(dst, n, shift, 1);
        memset(dst + n, 0, (N - n) * sizeof(uint64_t));
        glo_fast_ntt_dif(dst, log_N, 0, threads);
        return;
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long sg = 0; sg < (long)(N / SEG); sg++) {
        uint64_t *d = dst + (size_t)sg * SEG;
        if ((size_t)sg * SEG < n) {
            memcpy(d, coeffs + (size_t)sg * SEG, SEG * sizeof(uint64_t));
            glo_fast_scale_powers(d, SEG, shift, glo_pow(shift, (uint64_t)sg * SEG));
        } else {
            memset(d, 0, SEG * sizeof(uint64_t));
        }
    }
    glo_fast_ntt_dif(dst, log_N, 0, threads);
}

/* ---- big buffers of the timed leg: kept across calls, huge pages asked for ----
 * A 2^20-row proof allocates ~30 GB in a dozen blocks; handing them back to the kernel and faulting fresh zero pages in for the
 * next proof costs seconds (munmap of the 9 GB leaf array alone: 0.9 s on the bench host).  A prover that serves a stream of
 * proofs keeps its buffers, so the timed leg does: blocks >= 32 MB come from a small cache (best fit within 25 %), new ones
 * are 2 MB-aligned with MADV_HUGEPAGE.  glo_set_fast(0) -> glo_big_trim() gives everything cached back. */
#include <pthread.h>
#include <sys/mman.h>

#define BIG_MIN ((size_t)32 << 20)
#define BIG_SLOTS 64
static struct {
    void *p;
    size_t size;
    int in_use;
} g_big[BIG_SLOTS];
static pthread_mutex_t g_big_mu = PTHREAD_MUTEX_INITIALIZER;

void *glo_big_alloc(size_t bytes) {
    if (!g_fast || bytes < BIG_MIN) return malloc(bytes);
    pthread_mutex_lock(&g_big_mu);
    int best = -1, empty = -1;
    for (int i = 0; i < BIG_SLOTS; i++) {
        if (!g_big[i].p) {
            if (empty < 0) empty = i;
        } else if (!g_big[i].in_use && g_big[i].size >= bytes && g_big[i].size <= bytes + bytes / 4 &&
                   (best < 0 || g_big[i].size < g_big[best].size))
            best = i;
    }
    void *p = NULL;
    if (best >= 0) {
        g_big[best].in_use = 1;
        p = g_big[best].p;
    } else if (empty >= 0) {
        size_t sz = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (posix_memalign(&p, (size_t)2 << 20, sz) == 0) {
            madvise(p, sz, MADV_HUGEPAGE);
            g_big[empty].p = p;
            g_big[empty].size = sz;
            g_big[empty].in_use = 1;
        } else
            p = NULL;
    }
    pthread_mutex_unlock(&g_big_mu);
    return p ? p : malloc(bytes); /* table full or no memory for an aligned block: plain malloc */
}

void glo_big_free(void *p) {
    if (!p) return;
    pthread_mutex_lock(&g_big_mu);
    for (int i = 0; i < BIG_SLOTS; i++)
        if (g_big[i].p == p) {
            g_big[i].in_use = 0;
            pthread_mutex_unlock(&g_big_mu);
            return;
        }
    pthread_mutex_unlock(&g_big_mu);
    free(p);
}

void glo_big_trim(void) {
    pthread_mutex_lock(&g_big_mu);
    for (int i = 0; i < BIG_SLOTS; i++)
        if (g_big[i].p && !g_big[i].in_use) {
            free(g_big[i].p);
            g_big[i].p = NULL;
            g_big[i].size = 0;
        }
    pthread_mutex_unlock(&g_big_mu);
}
