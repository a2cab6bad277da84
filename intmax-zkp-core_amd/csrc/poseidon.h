// poseidon.h — Poseidon-Goldilocks permutation (width 12, x^7, 4 + 22 + 4 rounds), one state per lane.
//
// Replaces plonky2 @ f99ed9c src/hash/poseidon.rs::Poseidon::poseidon and src/hash/hashing.rs
// (hash_n_to_m_no_pad / compress) — the hasher of PoseidonGoldilocksConfig that the reference selects at
// src/zkdsa/circuits/mod.rs:362-368, src/rollup/circuits/mod.rs:1284-1311 and calls natively at
// src/merkle_tree/tree.rs:70,79,90,123.
//
// gfx950 notes: the 12-word state lives in 24 VGPRs of one lane (no cross-lane traffic); round constants come from
// constant memory through scalar loads (the round index is wave-uniform); rounds are kept as rolled loops so the
// kernel body stays inside the instruction cache.  The MDS layer has two interchangeable forms:
//   MDS_INT : 32-bit halves x small constants accumulated in u64 (v_mad_u64_u32), round constants folded into the
//             accumulator start values (production path)
//   MDS_F64 : reference round structure with the dot products in fp64 FMAs — exact, every partial sum is an integer
//             < 2^42 < 2^53 (kept as an independent cross-check; measured no faster)
//   MDS_MFMA: OPT-IN, never the default (BASELINE.json's north_star describes this path as non-MFMA): the six full 12 x 12 layers of
//             a permutation as int8 matrix-core products on byte planes, everything else as MDS_INT (mds_layer_add_mfma below;
//             measured in tools/mfma_mds_bench.hip: 0.75x the cycles of the 288-multiply layer).  Lane-per-state kernels with
//             whole waves only.
// All are bit-identical to the oracle.
#pragma once
#include "gl_field.h"
#include "poseidon_constants.h"

// device pass: constant memory (scalar loads); host pass (hipcc host side, or g++): an ordinary table, so the same
// permutation serves the host-side Fiat-Shamir challenger
#if defined(__HIP_DEVICE_COMPILE__)
#define GL_TABLE static __device__ __constant__ const
#else
#define GL_TABLE static const
#endif

namespace poseidon {

constexpr int WIDTH = 12;
constexpr int RATE = 8;
constexpr int HALF_FULL = 4;
constexpr int PARTIAL = 22;
constexpr int ROUNDS = 30;

// 30 x 12 round constants followed by one all-zero row (the "next round" of the last round)
GL_TABLE uint64_t RC[(ROUNDS + 1) * WIDTH] = {GL_POSEIDON_RC_LIST, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
// Partial rounds with the constants of elements 1..11 pushed forward (tools/gen_poseidon_constants.py): only element 0 meets
// the S-box there, so MDS(x + c) = MDS(x) + MDS(c) moves the other eleven constants into the next round's, leaving one
// scalar per partial round 5..25 and a modified vector for the first full round after them.  Same permutation.
GL_TABLE uint64_t RC_PARTIAL_S0[PARTIAL - 1] = {GL_POSEIDON_RC_PARTIAL_S0_LIST};
GL_TABLE uint64_t RC_AFTER_PARTIAL[WIDTH] = {GL_POSEIDON_RC_AFTER_PARTIAL_LIST};
// Round 0 of a permutation whose words 8..11 enter as ZERO (two_to_one of the tree levels: left || right || 0000): their S-box
// outputs are the constants RC[8..11]^7, and so is their share of round 0's MDS layer; RC_CAP0[r] = RC[12 + r] + that share
// (tools/gen_poseidon_constants.py).  permute_cap0 below: eight S-boxes and 192 multiply-adds in round 0 instead of twelve and 288.
GL_TABLE uint64_t RC_CAP0[WIDTH] = {GL_POSEIDON_RC_CAP0_LIST};
// Three partial rounds as ONE linear map (tools/gen_poseidon_constants.py::grouped_partial_rounds): with
// u = (x_1, s_1 .. s_11), x_j the S-box outputs, the S-box input of the 2nd round is P3_IN2 . u, of the 3rd P3_IN3 . (u, x_2),
// and the state after the 3rd round is P3_OUT (u, x_2, x_3).  Coefficients < 2^21: every term is still one v_mad_u64_u32
// on a 32-bit half, sums < 2^57.  One 12 x 14 matrix-vector product and two 13-term dot products replace three 12 x 12
// MDS layers.
GL_TABLE uint32_t P3_IN2[WIDTH + 2] = {GL_POSEIDON_P3_IN2_LIST};
GL_TABLE uint32_t P3_IN3[WIDTH + 2] = {GL_POSEIDON_P3_IN3_LIST};
GL_TABLE uint32_t P3_OUT[WIDTH][WIDTH + 2] = {GL_POSEIDON_P3_OUT_LIST};
// The MDS layer of full round 3 fused with partial round 4 (tools/gen_poseidon_constants.py::fused_mds_partial1): with x the
// twelve S-box outputs of round 3 and x_1' the S-box output of round 4, the state after round 4's MDS is
// P1_OUT (x, x_1') + P1_K (the scalar of round 5 included).  One 12 x 13 map and one 12-term dot product instead of two 12 x 12
// layers; the seven groups of three then cover rounds 5..25 and no single partial round is left over.
GL_TABLE uint32_t P1_OUT[WIDTH][WIDTH + 1] = {GL_POSEIDON_P1_OUT_LIST};
GL_TABLE uint64_t P1_K[WIDTH] = {GL_POSEIDON_P1_K_LIST};

// MDS = circulant(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,...,0):
//   out[r] = sum_i s[(i+r) mod 12] * CIRC[i] + 8*s[0] (r == 0)
#define GL_MDS_CIRC(i) \
    ((i) == 0 ? 17 : (i) == 1 ? 15 : (i) == 2 ? 41 : (i) == 3 ? 16 : (i) == 4 ? 2 : (i) == 5 ? 28 : \
     (i) == 6 ? 13 : (i) == 7 ? 13 : (i) == 8 ? 39 : (i) == 9 ? 18 : (i) == 10 ? 34 : 20)

// the same constants as an opaque table: read through scalar loads they stay in SGPRs and every product is one
// v_mad_u64_u32; as literals the compiler turns x2 and x16 into 64-bit shift-adds that need zero-extended register pairs
GL_TABLE uint32_t MDS_CIRC_TAB[WIDTH] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

enum { MDS_INT = 0, MDS_F64 = 1, MDS_MFMA = 2 };

GL_HD uint64_t sbox7(uint64_t x) {
    uint64_t x2 = gl::sqr(x);
    uint64_t x4 = gl::sqr(x2);
    uint64_t x3 = gl::mul(x, x2);
    return gl::mul(x3, x4);
}

// sum given as lo-half accumulator and hi-half accumulator, each < 2^63 (the layers: < 2^42, the grouped rounds: < 2^57):  value = alo + 2^32 * ahi
GL_HD uint64_t mds_combine_c(uint64_t alo, uint64_t ahi) {
    uint64_t shifted = ahi << 32;
    uint32_t top = (uint32_t)(ahi >> 32);
    uint64_t s = alo + shifted;
    top += (s < alo) ? 1u : 0u;
    return gl::reduce96(s, top);
}
#if defined(__HIP_DEVICE_COMPILE__)
// same value (any u64 representative of it) on the device, carries in SGPR pairs as in gl::mul
__device__ __forceinline__ uint64_t mds_combine(uint64_t alo, uint64_t ahi) {
    // alo + 2^32 (q0 + 2^32 q1) = (alo + q1 EPS) + 2^32 q0  (mod p):  the first sum is ONE multiply-add that cannot carry
    // (q1 < 2^31, alo < 2^63), the second touches the high word only; where it carries, + EPS = low word minus the carry bit, high
    // word plus it unless the low word borrowed.  Four VALU instructions and a scalar one (round 6; six before: the middle and top
    // limbs by an add and an add-with-carry, top * EPS by the multiply-add, the correction by a select and a 64-bit add).
    uint64_t W, cd, c, k, m;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(W), "=s"(cd) : "v"((uint32_t)(ahi >> 32)), "v"(alo));
    uint32_t r0, r1;
    asm("v_add_co_u32_e64 %[r1], %[c], %[w1], %[q0]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32_e64 %[r0], %[k], %[w0], 0, %[c]\n\t"
        "s_andn2_b64 %[m], %[c], %[k]\n\t"
        "v_addc_co_u32_e64 %[r1], %[k], %[r1], 0, %[m]"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [c] "=&s"(c), [k] "=&s"(k), [m] "=&s"(m)
        : [q0] "v"((uint32_t)ahi), [w0] "v"((uint32_t)W), [w1] "v"((uint32_t)(W >> 32))
        : "scc");
    return ((uint64_t)r1 << 32) | r0;
}
#else
GL_HD uint64_t mds_combine(uint64_t alo, uint64_t ahi) { return mds_combine_c(alo, ahi); }
#endif

GL_HD void mds_layer_int(uint64_t (&s)[WIDTH]) {
    uint32_t lo[WIDTH], hi[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < WIDTH; ++r) {
        uint64_t alo = 0, ahi = 0;
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) {
            alo += (uint64_t)lo[(i + r) % WIDTH] * (uint32_t)GL_MDS_CIRC(i);
            ahi += (uint64_t)hi[(i + r) % WIDTH] * (uint32_t)GL_MDS_CIRC(i);
        }
        if (r == 0) {
            alo += (uint64_t)lo[0] * 8u;
            ahi += (uint64_t)hi[0] * 8u;
        }
        s[r] = mds_combine(alo, ahi);
    }
}

GL_HD double u32_to_f64(uint32_t x) {
    // 2^52 + x has x in its low mantissa bits; subtracting 2^52 is exact
    uint64_t bits = 0x4330000000000000ull | x;
    return __builtin_bit_cast(double, bits) - 4503599627370496.0;
}
GL_HD uint64_t f64_to_u64_biased(double acc_plus_2p52) {
    return __builtin_bit_cast(uint64_t, acc_plus_2p52) & 0x000FFFFFFFFFFFFFull;
}

GL_HD void mds_layer_f64(uint64_t (&s)[WIDTH]) {
    double lo[WIDTH], hi[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) {
        lo[i] = u32_to_f64((uint32_t)s[i]);
        hi[i] = u32_to_f64((uint32_t)(s[i] >> 32));
    }
#pragma unroll
    for (int r = 0; r < WIDTH; ++r) {
        double alo = 4503599627370496.0, ahi = 4503599627370496.0;  // 2^52 bias: result bits = integer
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) {
            alo = __builtin_fma((double)GL_MDS_CIRC(i), lo[(i + r) % WIDTH], alo);
            ahi = __builtin_fma((double)GL_MDS_CIRC(i), hi[(i + r) % WIDTH], ahi);
        }
        if (r == 0) {
            alo = __builtin_fma(8.0, lo[0], alo);
            ahi = __builtin_fma(8.0, hi[0], ahi);
        }
        s[r] = mds_combine(f64_to_u64_biased(alo), f64_to_u64_biased(ahi));
    }
}

template <int MDS>
GL_HD void mds_layer(uint64_t (&s)[WIDTH]) {
    if (MDS == MDS_F64)
        mds_layer_f64(s);
    else
        mds_layer_int(s);
}

// ---- reference round structure (constant layer, S-box, MDS — exactly as written in poseidon.rs) ----
template <int MDS>
GL_HD void full_round(uint64_t (&s)[WIDTH], int round) {
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) s[i] = sbox7(gl::add(s[i], RC[round * WIDTH + i]));
    mds_layer<MDS>(s);
}

template <int MDS>
GL_HD void partial_round(uint64_t (&s)[WIDTH], int round) {
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) s[i] = gl::add(s[i], RC[round * WIDTH + i]);
    s[0] = sbox7(s[0]);
    mds_layer<MDS>(s);
}

template <int MDS>
GL_HD void permute_reference(uint64_t (&s)[WIDTH]) {
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; ++r) full_round<MDS>(s, r);
#pragma unroll 1
    for (int r = HALF_FULL; r < HALF_FULL + PARTIAL; ++r) partial_round<MDS>(s, r);
#pragma unroll 1
    for (int r = HALF_FULL + PARTIAL; r < ROUNDS; ++r) full_round<MDS>(s, r);
}

// ---- production form: the constant layer of round r+1 is folded into the MDS layer of round r ----
// MDS output r = sum_i c_i * s_(i+r) is accumulated in two u64 accumulators (32-bit halves); starting them at the
// two halves of the next round's constant instead of 0 adds that constant for free (sums stay < 2^43), which removes
// 348 of the 360 modular additions of the permutation.
// ONLY0: only output 0 receives a constant (next_rc[0]); the other accumulators start at zero
// M[r][0], the first column of the MDS matrix: the coefficient of the newest S-box output in every row of the grouped maps
constexpr uint32_t mds_col0(int r) { return (uint32_t)GL_MDS_CIRC((WIDTH - r) % WIDTH) + (r == 0 ? 8u : 0u); }
// the first term of row r of a layer whose inputs NIN.. do not contribute
constexpr int mds_first_term(int r, int nin) {
    for (int i = 0; i < WIDTH; ++i)
        if ((i + r) % WIDTH < nin) return i;
    return 0;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_NO_MAD_START)
#define GL_MAD_START 1
// x * C + k in ONE instruction, C an inline constant (every MDS entry is < 64), k a 64-bit constant in an SGPR pair: the start of an
// accumulator chain.  A VALU instruction of gfx9 reads at most one scalar operand, so with the coefficient in an SGPR (how the
// chains' other terms get theirs) the start value has to pass through a VGPR pair first — a v_mov / v_lshl_add_u64 per
// accumulator, 24 per 12 x 12 layer and ~30 per group of three partial rounds, in kernels whose only limit is VALU issue
// (round 6: 13.1 k -> 12.7 k VALU instructions per permutation, profiles/r06_mad_start.md).  Only for entries whose value is a
// compile-time constant: the first column of the matrix (the newest S-box output) and the plain MDS rows.
// `after`: any value the chain is to start after — an operand the instruction does not read.  Without it the scheduler runs all
// 24 chains of a layer side by side, coefficient by coefficient (48 accumulator registers live: 112 VGPRs, four waves per SIMD
// instead of six); with each row waiting for the result two rows back the allocation is the old one.
// HI: which of a row's two chains (low halves / high halves).  The unused carry-out goes to vcc in one chain and to an allocated
// SGPR pair in the other: the two chains are issued alternately, and the compiler puts a wait state between two adjacent asm
// statements that write the same register (it cannot know they are not sub-dword writes: gfx940's dst_sel forwarding hazard) —
// with one carry register that was a s_nop per multiply-add, 4 400 per permutation, and a wave alone on its SIMD pays for them.
template <uint32_t C, bool HI>
__device__ __forceinline__ uint64_t mad_start(uint32_t x, uint64_t k, uint32_t after) {
    static_assert(C <= 64, "inline integer constants are 0..64");
    uint64_t acc;
    if (HI) {
        uint64_t carry;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4 ; after %5" : "=v"(acc), "=s"(carry) : "v"(x), "n"(C), "s"(k), "v"(after));
    } else {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3 ; after %4" : "=v"(acc) : "v"(x), "n"(C), "s"(k), "v"(after) : "vcc");
    }
    return acc;
}
// acc += x * c as the same instruction, opaque: a chain that begins with mad_start has to continue in asm, or the compiler
// re-associates "start + t1 + t2 + .." into a chain from zero plus one 64-bit add of the start — the instruction this is about
template <bool HI>
__device__ __forceinline__ void mad_acc(uint64_t& acc, uint32_t x, uint32_t c) {
    if (HI) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(c) : "vcc");
    } else {
        uint64_t carry;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(x), "s"(c));
    }
}
#else
#define GL_MAD_START 0
template <uint32_t C, bool HI>
GL_HD uint64_t mad_start(uint32_t x, uint64_t k, uint32_t) { return (uint64_t)x * C + k; }
template <bool HI>
GL_HD void mad_acc(uint64_t& acc, uint32_t x, uint32_t c) { acc += (uint64_t)x * c; }
#endif

// a + rc for a round constant (rc < p, in SGPRs) and any u64 a: the sum wraps 2^64 at most once and the corrected value cannot wrap
// again (a + rc - 2^64 < p), so the correction is the riding one of mds_combine — four VALU instructions (gl::add as compiled: seven)
#if GL_MAD_START
__device__ __forceinline__ uint64_t add_rc(uint64_t a, uint64_t rc) {
    uint64_t s, w, k, m;
    uint32_t r0, r1;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(s) : "v"(a), "s"(rc));
    asm("v_cmp_lt_u64_e64 %[w], %[s], %[rc]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32_e64 %[r0], %[k], %[s0], 0, %[w]\n\t"
        "s_andn2_b64 %[m], %[w], %[k]\n\t"
        "v_addc_co_u32_e64 %[r1], %[k], %[s1], 0, %[m]"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [w] "=&s"(w), [k] "=&s"(k), [m] "=&s"(m)
        : [s] "v"(s), [s0] "v"((uint32_t)s), [s1] "v"((uint32_t)(s >> 32)), [rc] "s"(rc)
        : "scc");
    return ((uint64_t)r1 << 32) | r0;
}
#else
GL_HD uint64_t add_rc(uint64_t a, uint64_t rc) { return gl::add(a, rc); }
#endif

// row R of mds_layer_add<ONLY0, NIN> with its chains started by mad_start (the hashing kernels; START below)
template <int R, bool ONLY0, int NIN>
GL_HD uint64_t mds_row_started(const uint32_t (&lo)[WIDTH], const uint32_t (&hi)[WIDTH], const uint32_t (&cc)[WIDTH], const uint64_t* next_rc,
                               uint32_t after) {
    constexpr int I0 = mds_first_term(R, NIN);
    constexpr uint32_t C0 = (uint32_t)GL_MDS_CIRC(I0) + (R == 0 && I0 == 0 ? 8u : 0u);
    const uint64_t k = (!ONLY0 || R == 0) ? next_rc[R] : 0;
    uint64_t alo = mad_start<C0, false>(lo[(I0 + R) % WIDTH], (uint32_t)k, after), ahi = mad_start<C0, true>(hi[(I0 + R) % WIDTH], k >> 32, after);
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) {
        if (i == I0 || (i + R) % WIDTH >= NIN) continue;
        mad_acc<false>(alo, lo[(i + R) % WIDTH], cc[i]);
        mad_acc<true>(ahi, hi[(i + R) % WIDTH], cc[i]);
    }
    static_assert(R != 0 || I0 == 0, "row 0 starts with its own word: the diagonal 8 is part of the start coefficient");
    return mds_combine(alo, ahi);
}

// row R of a grouped map (partial_rounds3 / mds_partial1_fused) with its chains started by mad_start: the start term is the NEWEST
// input, word NT — its coefficient is M[R][0], an inline constant — so the chain begins when that S-box output exists; NT other terms
template <int R, int NT, typename Row>
GL_HD uint64_t grouped_row_started(const uint32_t* lo, const uint32_t* hi, Row row, uint64_t k, uint32_t after) {
    uint64_t alo = mad_start<mds_col0(R), false>(lo[NT], (uint32_t)k, after), ahi = mad_start<mds_col0(R), true>(hi[NT], k >> 32, after);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        mad_acc<false>(alo, lo[i], row[i]);
        mad_acc<true>(ahi, hi[i], row[i]);
    }
    return mds_combine(alo, ahi);
}

// zoff (here and in the grouped-round functions below): an offset added to every constant-table index.  It is the literal 0
// everywhere except in the fused quotient kernel, which passes a run-time zero made opaque once per task so that the table
// reads stay inside the task that needs them instead of being hoisted into the prologue of all eight waves (quotient_fused.h).
// START: the accumulator chains begin with mad_start (constant + first term in one instruction)
template <bool ONLY0 = false, int NIN = WIDTH, bool START = false>  // NIN < WIDTH: the words NIN.. do not contribute (folded into next_rc by the caller)
GL_HD void mds_layer_add(uint64_t (&s)[WIDTH], const uint64_t* next_rc, int zoff = 0) {
    uint32_t lo[WIDTH], hi[WIDTH], cc[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
        cc[i] = MDS_CIRC_TAB[i + zoff];
    }
    if (START && GL_MAD_START) {
        s[0] = mds_row_started<0, ONLY0, NIN>(lo, hi, cc, next_rc, lo[0]);
        s[1] = mds_row_started<1, ONLY0, NIN>(lo, hi, cc, next_rc, lo[0]);
        s[2] = mds_row_started<2, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[0]);
        s[3] = mds_row_started<3, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[1]);
        s[4] = mds_row_started<4, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[2]);
        s[5] = mds_row_started<5, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[3]);
        s[6] = mds_row_started<6, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[4]);
        s[7] = mds_row_started<7, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[5]);
        s[8] = mds_row_started<8, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[6]);
        s[9] = mds_row_started<9, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[7]);
        s[10] = mds_row_started<10, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[8]);
        s[11] = mds_row_started<11, ONLY0, NIN>(lo, hi, cc, next_rc, (uint32_t)s[9]);
        return;
    }
#pragma unroll
    for (int r = 0; r < WIDTH; ++r) {
        uint64_t alo = 0, ahi = 0;
        if (!ONLY0 || r == 0) {
            alo = (uint32_t)next_rc[r];
            ahi = next_rc[r] >> 32;
        }
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) {
            if ((i + r) % WIDTH >= NIN) continue;
            alo += (uint64_t)lo[(i + r) % WIDTH] * cc[i];
            ahi += (uint64_t)hi[(i + r) % WIDTH] * cc[i];
        }
        if (r == 0) {
            alo += (uint64_t)lo[0] * 8u;
            ahi += (uint64_t)hi[0] * 8u;
        }
        s[r] = mds_combine(alo, ahi);
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- MDS_MFMA: the same layer on the matrix cores (opt-in variant; formulation and measurements: tools/mfma_mds_bench.hip) ----
// v_mfma_i32_16x16x64_i8 computes D[m][n] = sum over lane groups g and byte slots j of A(lane (m, g), j) * B(lane (n, g), j).
// Every lane puts byte plane b of ITS OWN twelve state words into B slots 0..11; the constant operand is block diagonal
// (A(lane (4 kq + r4, g), j) = [g == kq] * M[4 og + r4][j]), so the four result registers of a lane are output words
// 4 og .. 4 og + 3 of its own state: no cross-lane traffic.  3 output groups x 8 byte planes = 24 products per layer.  Bytes are
// fed as x ^ 0x80 (signed), the accumulator input 128 * rowsum restores the unsigned plane sums U_b < 2^17, and a 32-bit-half
// accumulator of mds_layer_add is  rc_half + U_0 + 2^8 U_1 + 2^16 (U_2 + 2^8 U_3).
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
struct MfmaMds {
    mfma_v4i a[3], bias[3];
};
__device__ __forceinline__ int mds_coeff(int r, int j) {  // coefficient of input word j in output word r
    int d = j - r;
    d = d < 0 ? d + WIDTH : d;
    return (int)GL_MDS_CIRC(d) + (r == 0 && j == 0 ? 8 : 0);
}
__device__ __forceinline__ MfmaMds mfma_mds_consts() {
    const int l = threadIdx.x & 63, m = l & 15, g = l >> 4, kq = m >> 2, r4 = m & 3;
    MfmaMds c;
#pragma unroll
    for (int og = 0; og < 3; ++og) {
        unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < WIDTH; ++j) w[j >> 2] |= (g == kq ? (unsigned)mds_coeff(4 * og + r4, j) : 0u) << (8 * (j & 3));
        c.a[og] = mfma_v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        int rs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rs[i] = 0;
#pragma unroll
            for (int j = 0; j < WIDTH; ++j) rs[i] += mds_coeff(4 * og + i, j);
        }
        c.bias[og] = mfma_v4i{128 * rs[0], 128 * rs[1], 128 * rs[2], 128 * rs[3]};
    }
    return c;
}
// (w0.b, w1.b, w2.b, w3.b) for b = 0..3: a 4 x 4 byte transpose in eight v_perm_b32
__device__ __forceinline__ void byte_transpose4(const uint32_t (&w)[4], uint32_t (&p)[4]) {
    const uint32_t t0 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400), t1 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602);
    const uint32_t t2 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400), t3 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602);
    p[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100);
    p[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302);
    p[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100);
    p[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302);
}
// = mds_layer_add<false>(s, next_rc), bit for bit.  Every lane of the wave must be active.
__device__ __forceinline__ void mds_layer_add_mfma(uint64_t (&s)[WIDTH], const uint64_t* next_rc, const MfmaMds& mc) {
    uint32_t plane[8][3];
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        uint32_t lo[4], hi[4], pl[4], ph[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = (uint32_t)s[4 * blk + k];
            hi[k] = (uint32_t)(s[4 * blk + k] >> 32);
        }
        byte_transpose4(lo, pl);
        byte_transpose4(hi, ph);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            plane[b][blk] = pl[b] ^ 0x80808080u;
            plane[4 + b][blk] = ph[b] ^ 0x80808080u;
        }
    }
    const uint32_t k65536 = __builtin_amdgcn_readfirstlane(65536);
#pragma unroll
    for (int og = 0; og < 3; ++og) {
        uint64_t acc[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // the four planes of one 32-bit half at a time: 16 result registers live
            mfma_v4i d[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const mfma_v4i bop = {(int)plane[4 * h + b][0], (int)plane[4 * h + b][1], (int)plane[4 * h + b][2], 0};
                d[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(mc.a[og], bop, mc.bias[og], 0, 0, 0);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = 4 * og + r4;
                const uint32_t t01 = ((uint32_t)d[1][r4] << 8) + (uint32_t)d[0][r4];
                const uint32_t t23 = ((uint32_t)d[3][r4] << 8) + (uint32_t)d[2][r4];
                uint64_t a0 = h == 0 ? (uint64_t)(uint32_t)next_rc[r] : next_rc[r] >> 32, cy;
                asm("v_mad_u64_u32 %0, %1, %2, 1, %0" : "+v"(a0), "=s"(cy) : "v"(t01));
                asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a0), "=s"(cy) : "v"(t23), "s"(k65536));
                acc[h][r4] = a0;
            }
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) s[4 * og + r4] = mds_combine(acc[0][r4], acc[1][r4]);
    }
}
#endif

// the last MDS layer of a permutation, restricted to the output words [R0, R1) that the caller will read (the others are
// set to zero): a sponge in overwrite mode discards the rate part of every permutation but the last, a digest is words 0..3
template <int R0, int R1>
GL_HD void mds_layer_rows(uint64_t (&s)[WIDTH], int zoff = 0) {
    uint32_t lo[WIDTH], hi[WIDTH], cc[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
        cc[i] = MDS_CIRC_TAB[i + zoff];
    }
#pragma unroll
    for (int r = 0; r < WIDTH; ++r) {
        if (r < R0 || r >= R1) {
            s[r] = 0;
            continue;
        }
        uint64_t alo = 0, ahi = 0;
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) {
            alo += (uint64_t)lo[(i + r) % WIDTH] * cc[i];
            ahi += (uint64_t)hi[(i + r) % WIDTH] * cc[i];
        }
        if (r == 0) {
            alo += (uint64_t)lo[0] * 8u;
            ahi += (uint64_t)hi[0] * 8u;
        }
        s[r] = mds_combine(alo, ahi);
    }
}

// which words of the result the caller reads (the rest of the last MDS layer is skipped)
enum { OUT_ALL = 0, OUT_DIGEST = 1 /* 0..3 */, OUT_CAPACITY = 2 /* 8..11 */, OUT_TAIL = 3 /* 4..11 */, OUT_WORD0 = 4, OUT_WORD7 = 5 };

// Three consecutive partial rounds.  In: s[0] = S-box input of the first of them (constant included), s[1..11] the rest of
// the state.  k2, k3: the scalar constants of the second and third round; next_rc: what follows the group (ONLY0: the next
// round's scalar for element 0; otherwise a full vector).
// sbox_in: applied to every S-box input of the partial rounds before the S-box — identity in the permutation; the
// PoseidonGate evaluator (plonk_kernels.h) compares the input with its wire there and continues from the wire
struct SboxInIdentity {
    GL_HD uint64_t operator()(uint64_t x) const { return x; }
};
// only0 may be a run-time (wave-uniform) flag: the fused quotient kernel runs all seven groups through ONE copy of this code
template <bool START = false, typename F = SboxInIdentity>  // START: chains begun by mad_start (the hashing kernels)
GL_HD void partial_rounds3_rt(uint64_t (&s)[WIDTH], uint64_t k2, uint64_t k3, const uint64_t* next_rc, bool ONLY0, F&& sbox_in = F(),
                              int zoff = 0) {
    uint32_t lo[WIDTH + 2], hi[WIDTH + 2];
    s[0] = sbox7(sbox_in(s[0]));
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
    if (START && GL_MAD_START) {
        // P3_IN2 = MDS row 0 (25 first), P3_IN3[12] = 25 (the coefficient of x_2): both chains start on an inline constant
        uint64_t alo = mad_start<mds_col0(0), false>(lo[0], (uint32_t)k2, lo[0]), ahi = mad_start<mds_col0(0), true>(hi[0], k2 >> 32, lo[0]);
#pragma unroll
        for (int i = 1; i < WIDTH; ++i) {
            mad_acc<false>(alo, lo[i], P3_IN2[i + zoff]);
            mad_acc<true>(ahi, hi[i], P3_IN2[i + zoff]);
        }
        const uint64_t x2 = sbox7(sbox_in(mds_combine(alo, ahi)));
        lo[WIDTH] = (uint32_t)x2;
        hi[WIDTH] = (uint32_t)(x2 >> 32);
        alo = mad_start<mds_col0(0), false>(lo[WIDTH], (uint32_t)k3, lo[WIDTH]);
        ahi = mad_start<mds_col0(0), true>(hi[WIDTH], k3 >> 32, lo[WIDTH]);
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) {
            mad_acc<false>(alo, lo[i], P3_IN3[i + zoff]);
            mad_acc<true>(ahi, hi[i], P3_IN3[i + zoff]);
        }
        const uint64_t x3 = sbox7(sbox_in(mds_combine(alo, ahi)));
        lo[WIDTH + 1] = (uint32_t)x3;
        hi[WIDTH + 1] = (uint32_t)(x3 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const uint32_t __attribute__((address_space(4))) * crow_t;
#define GL_P3_ROW(R, AFTER)                                                                                             \
    {                                                                                                                   \
        crow_t row = (crow_t)(uintptr_t)P3_OUT[R];                                                                      \
        asm volatile("" : "+s"(row));                                                                                   \
        s[R] = grouped_row_started<R, WIDTH + 1>(lo, hi, row, (!ONLY0 || R == 0) ? next_rc[R] : 0, (uint32_t)(AFTER)); \
    }
        GL_P3_ROW(0, lo[WIDTH + 1])
        GL_P3_ROW(1, lo[WIDTH + 1])
        GL_P3_ROW(2, s[0])
        GL_P3_ROW(3, s[1])
        GL_P3_ROW(4, s[2])
        GL_P3_ROW(5, s[3])
        GL_P3_ROW(6, s[4])
        GL_P3_ROW(7, s[5])
        GL_P3_ROW(8, s[6])
        GL_P3_ROW(9, s[7])
        GL_P3_ROW(10, s[8])
        GL_P3_ROW(11, s[9])
#undef GL_P3_ROW
#endif
        return;
    }
    {
        uint64_t alo = (uint32_t)k2, ahi = k2 >> 32;
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) {
            alo += (uint64_t)lo[i] * P3_IN2[i + zoff];
            ahi += (uint64_t)hi[i] * P3_IN2[i + zoff];
        }
        const uint64_t x2 = sbox7(sbox_in(mds_combine(alo, ahi)));
        lo[WIDTH] = (uint32_t)x2;
        hi[WIDTH] = (uint32_t)(x2 >> 32);
    }
    {
        uint64_t alo = (uint32_t)k3, ahi = k3 >> 32;
#pragma unroll
        for (int i = 0; i < WIDTH + 1; ++i) {
            alo += (uint64_t)lo[i] * P3_IN3[i + zoff];
            ahi += (uint64_t)hi[i] * P3_IN3[i + zoff];
        }
        const uint64_t x3 = sbox7(sbox_in(mds_combine(alo, ahi)));
        lo[WIDTH + 1] = (uint32_t)x3;
        hi[WIDTH + 1] = (uint32_t)(x3 >> 32);
    }
#pragma unroll
    for (int r = 0; r < WIDTH; ++r) {
        uint64_t alo = 0, ahi = 0;
        if (!ONLY0 || r == 0) {
            alo = (uint32_t)next_rc[r];
            ahi = next_rc[r] >> 32;
        }
        // 196 matrix constants do not fit the scalar register file: each row is (re)loaded by scalar loads right where it is
        // used — the laundered pointer keeps the compiler from hoisting all rows out of the loop and spilling them
#if defined(__HIP_DEVICE_COMPILE__)
        // constant address space, so the row arrives by scalar loads (a laundered generic pointer would turn into flat loads)
        typedef const uint32_t __attribute__((address_space(4))) * crow_t;
        crow_t row = (crow_t)(uintptr_t)P3_OUT[r];
        asm volatile("" : "+s"(row));
#else
        const uint32_t* row = P3_OUT[r];
#endif
#pragma unroll
        for (int i = 0; i < WIDTH + 2; ++i) {
            alo += (uint64_t)lo[i] * row[i];
            ahi += (uint64_t)hi[i] * row[i];
        }
        s[r] = mds_combine(alo, ahi);
    }
}

template <bool ONLY0, typename F = SboxInIdentity, bool START = false>
GL_HD void partial_rounds3(uint64_t (&s)[WIDTH], uint64_t k2, uint64_t k3, const uint64_t* next_rc, F&& sbox_in = F(), int zoff = 0) {
    partial_rounds3_rt<START>(s, k2, k3, next_rc, ONLY0, sbox_in, zoff);
}

// In: the twelve S-box outputs of full round 3.  Out: the state that enters partial round 5 (element 0 = its S-box input).
template <typename F = SboxInIdentity, bool Z = false, bool START = false>  // Z: add zoff to the table indices (fused quotient kernel only)
GL_HD void mds_partial1_fused(uint64_t (&s)[WIDTH], F&& sbox_in = F(), int zoff = 0) {
    uint32_t lo[WIDTH + 1], hi[WIDTH + 1];
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
    if (START && GL_MAD_START) {
        const uint64_t c = RC[Z ? HALF_FULL * WIDTH + zoff : HALF_FULL * WIDTH];
        uint64_t alo = mad_start<mds_col0(0), false>(lo[0], (uint32_t)c, lo[0]), ahi = mad_start<mds_col0(0), true>(hi[0], c >> 32, lo[0]);
#pragma unroll
        for (int i = 1; i < WIDTH; ++i) {
            mad_acc<false>(alo, lo[i], MDS_CIRC_TAB[Z ? i + zoff : i]);
            mad_acc<true>(ahi, hi[i], MDS_CIRC_TAB[Z ? i + zoff : i]);
        }
        const uint64_t x1 = sbox7(sbox_in(mds_combine(alo, ahi)));
        lo[WIDTH] = (uint32_t)x1;
        hi[WIDTH] = (uint32_t)(x1 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const uint32_t __attribute__((address_space(4))) * crow_t;
#define GL_P1_ROW(R, AFTER)                                                                                   \
    {                                                                                                         \
        crow_t row = (crow_t)(uintptr_t)P1_OUT[R];                                                            \
        asm volatile("" : "+s"(row));                                                                         \
        s[R] = grouped_row_started<R, WIDTH>(lo, hi, row, P1_K[Z ? R + zoff : R], (uint32_t)(AFTER));         \
    }
        GL_P1_ROW(0, lo[WIDTH])
        GL_P1_ROW(1, lo[WIDTH])
        GL_P1_ROW(2, s[0])
        GL_P1_ROW(3, s[1])
        GL_P1_ROW(4, s[2])
        GL_P1_ROW(5, s[3])
        GL_P1_ROW(6, s[4])
        GL_P1_ROW(7, s[5])
        GL_P1_ROW(8, s[6])
        GL_P1_ROW(9, s[7])
        GL_P1_ROW(10, s[8])
        GL_P1_ROW(11, s[9])
#undef GL_P1_ROW
#endif
        return;
    }
    {
        const uint64_t c = RC[Z ? HALF_FULL * WIDTH + zoff : HALF_FULL * WIDTH];  // constant of element 0 in round 4
        uint64_t alo = (uint32_t)c, ahi = c >> 32;
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) {
            alo += (uint64_t)lo[i] * MDS_CIRC_TAB[Z ? i + zoff : i];
            ahi += (uint64_t)hi[i] * MDS_CIRC_TAB[Z ? i + zoff : i];
        }
        alo += (uint64_t)lo[0] * 8u;
        ahi += (uint64_t)hi[0] * 8u;
        const uint64_t x1 = sbox7(sbox_in(mds_combine(alo, ahi)));
        lo[WIDTH] = (uint32_t)x1;
        hi[WIDTH] = (uint32_t)(x1 >> 32);
    }
#pragma unroll
    for (int r = 0; r < WIDTH; ++r) {
        uint64_t alo = (uint32_t)P1_K[Z ? r + zoff : r], ahi = P1_K[Z ? r + zoff : r] >> 32;
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const uint32_t __attribute__((address_space(4))) * crow_t;  // scalar loads, one row at a time (see partial_rounds3)
        crow_t row = (crow_t)(uintptr_t)P1_OUT[r];
        asm volatile("" : "+s"(row));
#else
        const uint32_t* row = P1_OUT[r];
#endif
#pragma unroll
        for (int i = 0; i < WIDTH + 1; ++i) {
            alo += (uint64_t)lo[i] * row[i];
            ahi += (uint64_t)hi[i] * row[i];
        }
        s[r] = mds_combine(alo, ahi);
    }
}

// state in / state out: arbitrary u64 representatives (callers canonicalise on store).  `out` (wave-uniform): OUT_*.
// first_round = 1 (wave-uniform): s is already the state that enters the S-boxes of round 1, constants included — the
// proof-of-work grind (fri_kernels.h) builds it from a per-proof precomputation, its candidates differing in one word only.
// THRU (round 6): the form for launches that fill the chip (wide leaf sets and tree levels; merkle_kernels.h *_thru_kernel).  Its
// accumulator chains begin with mad_start (above) and the six ONLY0 groups run two per loop iteration — the new state of one group
// is the input of the next in place, where the rolled loop copies it back into the loop-carried registers (ten v_mov_b64 per
// group): 13.1 k -> 12.3 k VALU instructions per permutation, +11 KB of code (the leaf kernel: 56 KB of the 64 KB instruction cache).
// The price is instruction-level parallelism (rows wait for each other, wait states between adjacent asm statements): a wave alone
// on its SIMD runs it 3-4 % slower, so launches that cannot fill the chip keep the other form (-DGL_P3_UNROLL=1: no unrolling).
#ifndef GL_P3_UNROLL
#define GL_P3_UNROLL 2
#endif
template <bool MFMA = false, bool THRU = false>
GL_HD void permute_folded(uint64_t (&s)[WIDTH], int out = OUT_ALL, int first_round = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
    MfmaMds mc;
    if (MFMA) mc = mfma_mds_consts();  // a function of the lane index only: hoisted out of the caller's loops
#endif
    // A per-permutation opaque zero on every constant-table index (round 3).  The tables are loop-invariant in the caller's
    // sponge loop (17 permutations per wires leaf), so the compiler read ALL of them — RC rows, P1_K, P3_IN2 / IN3,
    // RC_AFTER_PARTIAL, ~110 SGPRs — once per leaf and, the SGPR file being full, kept them in VGPR LANES: 196 SGPR spills, 181
    // v_readlane in the round bodies, 93 VGPRs.  Behind the laundered offset every table word is read by a scalar load where
    // it is used (the scalar cache holds the tables): 50 spills, 79 VGPRs (six waves per SIMD instead of five), leaf hash at
    // C = 135 50.5 -> 49.2 ms = 2.82 -> 2.90 G perm/s, proof 120.5 -> 117.5 ms.  -DGL_NO_PERM_LAUNDER restores the old form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_NO_PERM_LAUNDER)
    int z = 0;
    asm volatile("" : "+s"(z));
    constexpr bool ZL = true;
#else
    constexpr int z = 0;
    constexpr bool ZL = false;
#endif
    if (first_round == 0) {
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) s[i] = THRU ? add_rc(s[i], RC[i + z]) : gl::add(s[i], RC[i + z]);
    }
    static_assert(PARTIAL == 22, "one fused partial round and 7 groups of three");
    // The two halves share ONE copy of the full-round code (three rounds with their MDS layer, then the S-boxes of round 3
    // resp. 29): the kernel stays inside the 64 KB instruction cache that two CUs share.
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int r0 = half ? HALF_FULL + PARTIAL : 0;
#pragma unroll 1
        for (int r = half ? r0 : first_round; r < r0 + HALF_FULL - 1; ++r) {
#pragma unroll
            for (int i = 0; i < WIDTH; ++i) s[i] = sbox7(s[i]);
#if defined(__HIP_DEVICE_COMPILE__)
            if (MFMA)
                mds_layer_add_mfma(s, &RC[(r + 1) * WIDTH], mc);
            else
#endif
                mds_layer_add<false, WIDTH, THRU>(s, &RC[(r + 1) * WIDTH + z], z);
        }
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) s[i] = sbox7(s[i]);
        if (half == 0) {
            mds_partial1_fused<SboxInIdentity, ZL, THRU>(s, SboxInIdentity(), z);  // the MDS layer of round 3 fused with partial round 4
#pragma clang loop unroll_count(THRU ? GL_P3_UNROLL : 1)
            for (int g = 0; g < 6; ++g)  // partial rounds 5..22, three at a time; RC_PARTIAL_S0[j] is the scalar of round 5 + j
                partial_rounds3<true, SboxInIdentity, THRU>(s, RC_PARTIAL_S0[3 * g + 1 + z], RC_PARTIAL_S0[3 * g + 2 + z],
                                                            &RC_PARTIAL_S0[3 * g + 3 + z], SboxInIdentity(), z);
            // rounds 23..25, followed by the pushed-forward constants of round 26
            partial_rounds3<false, SboxInIdentity, THRU>(s, RC_PARTIAL_S0[19 + z], RC_PARTIAL_S0[20 + z], &RC_AFTER_PARTIAL[z], SboxInIdentity(), z);
        } else if (out == OUT_DIGEST) {  // round 29: no constants follow, and only the words that are read
            mds_layer_rows<0, 4>(s, z);
        } else if (out == OUT_CAPACITY) {
            mds_layer_rows<8, 12>(s, z);
        } else if (out == OUT_TAIL) {
            mds_layer_rows<4, 12>(s, z);
        } else if (out == OUT_WORD0) {
            mds_layer_rows<0, 1>(s, z);
        } else if (out == OUT_WORD7) {
            mds_layer_rows<7, 8>(s, z);
        } else {
            mds_layer_rows<0, 12>(s, z);
        }
    }
}

// the permutation of a state whose words 8..11 are zero (RC_CAP0 above); s[8..11] are not read.  Same result as permute_folded.
template <bool THRU = false>
GL_HD void permute_cap0(uint64_t (&s)[WIDTH], int out = OUT_ALL) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_NO_PERM_LAUNDER)
    int z = 0;
    asm volatile("" : "+s"(z));
#else
    constexpr int z = 0;
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = sbox7(THRU ? add_rc(s[i], RC[i + z]) : gl::add(s[i], RC[i + z]));
#pragma unroll
    for (int i = 8; i < WIDTH; ++i) s[i] = 0;
    mds_layer_add<false, 8, THRU>(s, &RC_CAP0[z], z);
    permute_folded<false, THRU>(s, out, 1);
}

// MDS selects the code path: MDS_INT = production (folded constants), MDS_F64 = reference structure with fp64 MDS
template <int MDS, bool THRU = false>
GL_HD void permute(uint64_t (&s)[WIDTH], int out = OUT_ALL) {
    if (MDS == MDS_F64)
        permute_reference<MDS_F64>(s);
    else if (MDS == MDS_MFMA)
        permute_folded<true>(s, out);
    else
        permute_folded<false, THRU>(s, out);
}

}  // namespace poseidon
