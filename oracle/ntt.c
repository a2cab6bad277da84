/*
 * ntt.c — CPU ORACLE (test infrastructure): radix-2 NTT over Goldilocks, natural order in and out.
 * Restates plonky2 @ f99ed9c field/src/fft.rs (fft / ifft) and field/src/polynomial/mod.rs
 * (coset_fft, coset_ifft, lde); the dependency source is absent from /root/reference (Cargo.toml:12).
 * Convention: fft(c)[i] = sum_j c_j w^(ij), w = primitive_root_of_unity(log_n);
 * ifft(v)[j] = n^-1 sum_i v_i w^(-ij); coset_fft(c, s) = fft(c_j * s^j).
 * "parity unpinned": the reference has no golden vector for orderings; pinned here by glo_dft_naive.
 */
#include <stdlib.h>
#include <string.h>

#include "fast.h"
#include "gl_oracle.h"

size_t glo_reverse_bits(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

void glo_dft_naive(const uint64_t *coeffs, uint64_t *values, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    uint64_t w = glo_root_of_unity(log_n);
    uint64_t wi = 1; /* w^i */
    for (size_t i = 0; i < n; i++) {
        uint64_t acc = 0, x = 1;
        for (size_t j = 0; j < n; j++) {
            acc = glo_add(acc, glo_mul(coeffs[j], x));
            x = glo_mul(x, wi);
        }
        values[i] = acc;
        wi = glo_mul(wi, w);
    }
}

/* iterative DIT: bit-reverse the input, then log_n butterfly layers */
static void fft_root(uint64_t *a, unsigned log_n, uint64_t root) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = glo_reverse_bits(i, log_n);
        if (i < j) {
            uint64_t t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    }
    for (size_t i = 0; i < n; i++) a[i] = glo_canon(a[i]);
    uint64_t *tw = (uint64_t *)malloc((n / 2 + 1) * sizeof(uint64_t));
    tw[0] = 1;
    for (size_t i = 1; i < n / 2; i++) tw[i] = glo_mul(tw[i - 1], root);
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, half = m >> 1, step = n / m;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                uint64_t u = a[k + j], v = glo_mul(a[k + j + half], tw[j * step]);
                a[k + j] = glo_add(u, v);
                a[k + j + half] = glo_sub(u, v);
            }
    }
    free(tw);
}

/* the timed leg (fast.h): decimation in frequency into bit-reversed order, then the permutation back (scaled by 1/n for
 * the inverse) — the same values as the layers above */
static void fft_fast(uint64_t *a, unsigned log_n, int inverse) { glo_fast_fft_mt(a, log_n, inverse, 1); }

void glo_fft(uint64_t *a, unsigned log_n) {
    if (glo_get_fast()) {
        fft_fast(a, log_n, 0);
        return;
    }
    fft_root(a, log_n, glo_root_of_unity(log_n));
}

void glo_ifft(uint64_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    if (glo_get_fast()) {
        fft_fast(a, log_n, 1);
        return;
    }
    fft_root(a, log_n, glo_inv(glo_root_of_unity(log_n)));
    uint64_t ninv = glo_inv((uint64_t)n);
    for (size_t i = 0; i < n; i++) a[i] = glo_mul(a[i], ninv);
}

void glo_coset_fft(uint64_t *a, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    if (glo_get_fast()) {
        glo_fast_scale_powers(a, n, shift, 1);
        fft_fast(a, log_n, 0);
        return;
    }
    uint64_t s = 1;
    for (size_t i = 0; i < n; i++) {
        a[i] = glo_mul(a[i], s);
        s = glo_mul(s, shift);
    }
    glo_fft(a, log_n);
}

void glo_coset_ifft(uint64_t *a, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    glo_ifft(a, log_n);
    if (glo_get_fast()) {
        glo_fast_scale_powers(a, n, glo_inv(shift), 1);
        return;
    }
    uint64_t sinv = glo_inv(shift), s = 1;
    for (size_t i = 0; i < n; i++) {
        a[i] = glo_mul(a[i], s);
        s = glo_mul(s, sinv);
    }
}

uint64_t glo_poly_eval(const uint64_t *coeffs, size_t n, uint64_t x) {
    uint64_t acc = 0;
    for (size_t i = n; i-- > 0;) acc = glo_add(glo_mul(acc, x), coeffs[i]);
    return acc;
}
