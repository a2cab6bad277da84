// ntt_index.h — index algebra of the multi-pass, LDS-tiled Goldilocks NTT (host + device).
//
// Replaces plonky2 @ f99ed9c field/src/fft.rs (fft_classic / ifft) and polynomial/mod.rs (coset_fft, lde), reached
// from the reference through PolynomialBatch::from_values inside CircuitData::prove
// (reference src/rollup/circuits/mod.rs:1247).
//
// A size-2^k transform is cut into passes; a pass transforms the R index bits [s_lo, s_lo+R) of every element:
//     position p = (A << (s_lo+R)) | (r << s_lo) | B
//   DIF pass (natural -> bit-reversed):  local 2^R-point DIF over r, then slot r' *= w_M^(B * rev_R(r')), M = 2^(s_lo+R)
//   DIT pass (bit-reversed -> natural):  slot r' *= w_M^(B * rev_R(r')), then local 2^R-point DIT over r
// DIF passes run from the high bits down, DIT passes from the low bits up.  A workgroup owns a tile of 2^LOGE
// elements = 2^R values of r  x  2^(LOGE-R) neighbours that are contiguous in memory (>= 128-byte runs), so every
// pass is a coalesced in-place read-modify-write of the array.  (Derivation checked against the naive DFT in
// tests/test_ntt_model.py.)
#pragma once
#include "gl_field.h"

namespace ntt {

struct TileGeom {
    int log_e;   // tile holds 2^log_e elements
    int s_lo;    // lowest transformed bit
    int r_bits;  // R
    // derived
    int cb;      // non-r bits inside the tile = log_e - R
    int rshift;  // bit offset of the r field inside the local index e
};

GL_HD TileGeom make_geom(int log_e, int s_lo, int r_bits) {
    TileGeom g;
    g.log_e = log_e;
    g.s_lo = s_lo;
    g.r_bits = r_bits;
    g.cb = log_e - r_bits;
    g.rshift = g.cb < s_lo ? g.cb : s_lo;
    return g;
}

// local element e of tile T  ->  position inside the column
GL_HD size_t tile_pos(const TileGeom& g, size_t tile, uint32_t e) {
    if (g.s_lo >= g.cb) {
        uint32_t b_lo = e & ((1u << g.cb) - 1);
        uint32_t r = e >> g.cb;
        int hb = g.s_lo - g.cb;  // tile bits that select the B run
        size_t a = tile >> hb;
        size_t b_hi = tile & (((size_t)1 << hb) - 1);
        return (a << (g.s_lo + g.r_bits)) | ((size_t)r << g.s_lo) | (b_hi << g.cb) | b_lo;
    }
    return (tile << g.log_e) | e;
}

GL_HD uint32_t tile_r(const TileGeom& g, uint32_t e) { return (e >> g.rshift) & ((1u << g.r_bits) - 1); }

GL_HD uint32_t rev_bits(uint32_t x, int bits) {
    if (bits == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x) >> (32 - bits);
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

// butterfly j (0 <= j < 2^(log_e-1)) of local stage t  ->  local index of its "bit t = 0" element
GL_HD uint32_t butterfly_lo(const TileGeom& g, uint32_t j, int t) {
    int bit = g.rshift + t;
    uint32_t low = j & ((1u << bit) - 1);
    return ((j >> bit) << (bit + 1)) | low;
}

// index into the local table LT[i] = w_{2^R}^i (i < 2^(R-1)) for the butterfly whose low element has r-field r0
GL_HD uint32_t local_twiddle_index(const TileGeom& g, uint32_t r0, int t) {
    return (r0 & ((1u << t) - 1)) << (g.r_bits - 1 - t);
}

// exponent (in units of the master root w_{2^K}) of the inter-pass twiddle of slot r' at low-bits value B
GL_HD uint64_t pass_twiddle_exp(const TileGeom& g, uint32_t r_slot, size_t b_val, int K, bool inverse) {
    uint64_t e = (uint64_t)b_val * rev_bits(r_slot, g.r_bits);  // < 2^(s_lo+R)
    e <<= (K - (g.s_lo + g.r_bits));
    if (inverse) e = (((uint64_t)1 << K) - e) & (((uint64_t)1 << K) - 1);
    return e;
}

constexpr int TW_LO_BITS = 12;  // w^e = W_lo[e & 4095] * W_hi[e >> 12]

// ---- register-blocked rounds: Q consecutive butterfly stages (local bits [t, t+Q)) on 2^Q elements held in registers ----
// group g (0 <= g < 2^(log_e - Q)) -> local index of its element 0; element j sits at base + (j << (rshift + t))
GL_HD uint32_t round_base(const TileGeom& g, uint32_t group, int t, int q) {
    const int bit = g.rshift + t;
    const uint32_t low = group & ((1u << bit) - 1);
    return ((group >> bit) << (bit + q)) | low;
}

// DIF (DIT = false): stages t+Q-1 .. t, (u, v) -> (u + v, (u - v) w);  DIT: stages t .. t+Q-1, (u, v) -> (u + w v, u - w v).
// Twiddle of the butterfly whose lower element has r-field r0 at stage st: ltw[(r0 mod 2^st) << (R-1-st)], ltw[i] = w_{2^R}^i.
// T0 = the round sits on the lowest bits (t == 0): the twiddle index is then a compile-time function of j, and the
// butterflies whose twiddle is w^0 = 1 (7 of the 12 in a radix-8 round) skip the multiplication.
template <int Q, bool DIT, bool T0 = false>
GL_HD void radix_round(uint64_t (&x)[1 << Q], const uint64_t* ltw, const TileGeom& g, uint32_t base, int t) {
    const uint32_t low = T0 ? 0u : (tile_r(g, base) & ((1u << t) - 1));  // r-field bits below the round
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int s = DIT ? k : Q - 1 - k;  // stage inside the round
        const int st = t + s;
        const int half = 1 << s;
#pragma unroll
        for (int j = 0; j < (1 << Q); ++j) {
            if (j & half) continue;
            const uint32_t r0_low = low | ((uint32_t)(j & (half - 1)) << t);
            uint64_t u = x[j], v = x[j + half];
            if (T0 && (j & (half - 1)) == 0) {  // twiddle 1
                x[j] = gl::add(u, v);
                x[j + half] = gl::sub(u, v);
                continue;
            }
            const uint64_t w = ltw[r0_low << (g.r_bits - 1 - st)];
            if (DIT) {
                v = gl::mul(v, w);
                x[j] = gl::add(u, v);
                x[j + half] = gl::sub(u, v);
            } else {
                x[j] = gl::add(u, v);
                x[j + half] = gl::mul(gl::sub(u, v), w);
            }
        }
    }
}

// ---- specialised pass "A": R = 7 bits at s_lo >= 4, tile = 128 values of r x 16 contiguous neighbours, 128 threads x 16 ----
// Thread t loads r = (q << 3) | (t >> 4), q = 0..15 (coalesced 128-byte runs), runs the four upper stages as one radix-16
// round in registers (Cooley-Tukey form: shift twiddles inside, one w_128 twiddle per element after it), exchanges through
// LDS once, and finishes with the three lowest stages as a radix-8 round whose twiddles are the 8th roots of unity
// +-2^24, +-2^48, +-2^72 (shifts).  LDS rows of 128 elements are padded to 144.
constexpr int A_R_BITS = 7;
constexpr int A_THREADS = 128;
#ifndef GL_A_WAVES
#define GL_A_WAVES 4
#endif
constexpr int A_WAVES_PER_SIMD = GL_A_WAVES;  // register budget of the kernel: 512 / this
constexpr int A_ROW = 144;  // elements per padded LDS row (8 values of r x 16 neighbours + 16 pad)
GL_HD uint32_t a_lds_index(uint32_t r, uint32_t b_lo) { return (r >> 3) * A_ROW + ((r & 7) << 4) + b_lo; }
// first position of a tile: tile index = (A << (s_lo - 4)) | b_hi
GL_HD size_t a_tile_base(size_t tile, int s_lo) {
    const int hb = s_lo - 4;
    return ((tile >> hb) << (s_lo + A_R_BITS)) | ((tile & (((size_t)1 << hb) - 1)) << 4);
}

// ---- specialised pass "B": the lowest R = 6 bits (s_lo = 0): a tile is 32 contiguous 64-point transforms, 128 threads x 16 ----
// Thread t loads r = (j << 3) | (t & 7) of blocks (t >> 3) and (t >> 3) + 16 (64-byte segments), runs stages 5..3 as a radix-8
// round in registers (general twiddles w_64^i, seven per thread, fixed for the whole kernel), exchanges through LDS (rows of
// 8 elements padded to 9: conflict-free both ways), finishes with the shift-twiddle radix-8 round and returns through the
// same LDS rows for coalesced stores.
constexpr int B_R_BITS = 6;
constexpr int B_PITCH = 9;  // 8-byte words per LDS row (8 consecutive values of r + 1 pad)
GL_HD uint32_t b_lds_index(uint32_t blk, uint32_t r) { return (blk * 8 + (r >> 3)) * B_PITCH + (r & 7); }

// DIF stages 5..3 on x[j] = element with r = (j << 3) | r3;  tw[0..3] = w_64^(r3 + 8 j), tw[4..5] = w_64^(2 (r3 + 8 j)), tw[6] = w_64^(4 r3)
GL_HD void radix8_dif_tw(uint64_t (&x)[8], const uint64_t (&tw)[7]) {
    uint64_t u, v;
#define GL_BF(a, b, w) u = x[a], v = x[b], x[a] = gl::add(u, v), x[b] = gl::mul(gl::sub(u, v), w)
    GL_BF(0, 4, tw[0]);
    GL_BF(1, 5, tw[1]);
    GL_BF(2, 6, tw[2]);
    GL_BF(3, 7, tw[3]);
    GL_BF(0, 2, tw[4]);
    GL_BF(1, 3, tw[5]);
    GL_BF(4, 6, tw[4]);
    GL_BF(5, 7, tw[5]);
    GL_BF(0, 1, tw[6]);
    GL_BF(2, 3, tw[6]);
    GL_BF(4, 5, tw[6]);
    GL_BF(6, 7, tw[6]);
#undef GL_BF
}

// (u - v) * w_16^e, e = 1..7, for the forward root w_16 = 2^156 = -2^60 or the inverse root 2^36
template <bool INVERSE, int E>
GL_HD uint64_t diff_times_w16(uint64_t u, uint64_t v) {
    static_assert(E >= 1 && E <= 7, "w_16^0 = 1 needs no multiplication");
    constexpr int X = ((INVERSE ? 36 : 156) * E) % 192;  // w_16^E = 2^X
    constexpr bool NEG = X >= 96;                         // 2^96 = -1
    constexpr int S = NEG ? X - 96 : X;
    return gl::mul_2exp<S>(NEG ? gl::sub(v, u) : gl::sub(u, v));
}

// Radix-16 DIF in Cooley-Tukey form on x[q] = element with r = (q << 3) | r3: the 16-point transform over q has only powers of
// w_16 as twiddles (shifts); the factor that depends on the low bits, w_128^(r3 * k1) for output k1 = rev4(q'), is applied
// once per element afterwards.  w128[i] = w_128^i, i < 128.  15 general multiplications and 17 shifts per 16 elements
// (the plain radix-2 form of the same four stages needs 32 general multiplications).
template <bool INVERSE>
GL_HD void radix16_dif_ct(uint64_t (&x)[16], const uint64_t* w128, uint32_t r3) {
    uint64_t u, v;
#define GL_BF1(a, b) u = x[a], v = x[b], x[a] = gl::add(u, v), x[b] = gl::sub(u, v)
#define GL_BFW(a, b, e) u = x[a], v = x[b], x[a] = gl::add(u, v), x[b] = diff_times_w16<INVERSE, e>(u, v)
    GL_BF1(0, 8);
    GL_BFW(1, 9, 1);
    GL_BFW(2, 10, 2);
    GL_BFW(3, 11, 3);
    GL_BFW(4, 12, 4);
    GL_BFW(5, 13, 5);
    GL_BFW(6, 14, 6);
    GL_BFW(7, 15, 7);
#pragma unroll
    for (int h = 0; h < 16; h += 8) {
        GL_BF1(h + 0, h + 4);
        GL_BFW(h + 1, h + 5, 2);
        GL_BFW(h + 2, h + 6, 4);
        GL_BFW(h + 3, h + 7, 6);
    }
#pragma unroll
    for (int h = 0; h < 16; h += 4) {
        GL_BF1(h + 0, h + 2);
        GL_BFW(h + 1, h + 3, 4);
    }
#pragma unroll
    for (int h = 0; h < 16; h += 2) GL_BF1(h, h + 1);
#undef GL_BF1
#undef GL_BFW
#pragma unroll
    for (int q = 1; q < 16; ++q) {
        const uint32_t k1 = ((q & 1) << 3) | ((q & 2) << 1) | ((q & 4) >> 1) | ((q & 8) >> 3);
        x[q] = gl::mul(x[q], w128[r3 * k1]);
    }
}

// (u - v) * w_8^j for the forward root w_8 = 2^120 = -2^24 or the inverse root 2^72
template <bool INVERSE, int J>
GL_HD uint64_t diff_times_w8(uint64_t u, uint64_t v) {
    static_assert(J >= 1 && J <= 3, "w_8^0 = 1 needs no multiplication");
    if (!INVERSE) {
        if (J == 1) return gl::mul_2exp24(gl::sub(v, u));  // -2^24
        if (J == 2) return gl::mul_2exp48(gl::sub(u, v));  //  2^48
        return gl::mul_2exp72(gl::sub(v, u));              // -2^72
    }
    if (J == 1) return gl::mul_2exp72(gl::sub(u, v));      //  2^72
    if (J == 2) return gl::mul_2exp48(gl::sub(v, u));      //  2^144 = -2^48
    return gl::mul_2exp24(gl::sub(u, v));                  //  2^216 = 2^24
}

// DIF stages 2..0 on x[j] = element with r = (hi << 3) | j: every twiddle is a power of w_8
template <bool INVERSE>
GL_HD void radix8_dif_const(uint64_t (&x)[8]) {
    uint64_t u, v;
#define GL_BF(a, b, expr) u = x[a], v = x[b], x[a] = gl::add(u, v), x[b] = (expr)
    GL_BF(0, 4, gl::sub(u, v));
    GL_BF(1, 5, (diff_times_w8<INVERSE, 1>(u, v)));
    GL_BF(2, 6, (diff_times_w8<INVERSE, 2>(u, v)));
    GL_BF(3, 7, (diff_times_w8<INVERSE, 3>(u, v)));
    GL_BF(0, 2, gl::sub(u, v));
    GL_BF(1, 3, (diff_times_w8<INVERSE, 2>(u, v)));
    GL_BF(4, 6, gl::sub(u, v));
    GL_BF(5, 7, (diff_times_w8<INVERSE, 2>(u, v)));
    GL_BF(0, 1, gl::sub(u, v));
    GL_BF(2, 3, gl::sub(u, v));
    GL_BF(4, 5, gl::sub(u, v));
    GL_BF(6, 7, gl::sub(u, v));
#undef GL_BF
}

// Pass B's first radix-8 round (DIF stages 5..3 of a 64-point block, x[j] = element with r = (j << 3) | r3) in Cooley-Tukey form,
// like radix16_dif_ct: the 8-point transform over j has only powers of w_8 as twiddles (shifts), and the factor that depends on the
// low bits, w_64^(r3 * k1) for output k1 = rev3(q), is applied once per element afterwards: SEVEN general products and five shift
// twiddles per eight elements where radix8_dif_tw (twiddles merged into the butterflies) needs twelve general products.
// GL_RADIX8_CT=0 (GL_EXTRA_FLAGS) builds the merged form for A/B runs.
#ifndef GL_RADIX8_CT
#define GL_RADIX8_CT 1
#endif
constexpr int B_W64_ENTRIES = GL_RADIX8_CT ? 64 : 32;  // w_64^i: the merged form reads i < 32, this one i = r3 * k1 <= 49
// the seven per-thread twiddles of the round, from the table w64[i] = w_64^i (inverse roots for an inverse transform)
GL_HD void b_round_twiddles(const uint64_t* w64, uint32_t r3, uint64_t (&tw)[7]) {
#if GL_RADIX8_CT
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        const uint32_t k1 = ((q & 1) << 2) | (q & 2) | ((q & 4) >> 2);
        tw[q - 1] = w64[r3 * k1];
    }
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) tw[j] = w64[r3 + 8 * j];
    tw[4] = w64[2 * r3];
    tw[5] = w64[2 * (r3 + 8)];
    tw[6] = w64[4 * r3];
#endif
}
template <bool INVERSE>
GL_HD void b_round(uint64_t (&x)[8], const uint64_t (&tw)[7]) {
#if GL_RADIX8_CT
    radix8_dif_const<INVERSE>(x);
#pragma unroll
    for (int q = 1; q < 8; ++q) x[q] = gl::mul(x[q], tw[q - 1]);
#else
    radix8_dif_tw(x, tw);
#endif
}

}  // namespace ntt
