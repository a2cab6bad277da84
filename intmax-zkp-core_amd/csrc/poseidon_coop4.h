// poseidon_coop4.h — Poseidon permutation on FOUR lanes: lane q of a quad holds the state words 3q, 3q + 1, 3q + 2.
//
// Same function as poseidon.h / poseidon_coop.h (plonky2 @ f99ed9c src/hash/poseidon.rs::Poseidon::poseidon), a third
// mapping for the sizes between them:
//   lane-per-state (poseidon.h)        13.6 k dependent instructions per permutation, 64 permutations per wave
//   16 lanes per state (poseidon_coop) ~3 k, 4 per wave   — leaf sets up to 2^13, tree levels up to 2^13 parents
//   4 lanes per state (this file)      ~4.8 k, 16 per wave — leaf sets of 2^13 .. 2^15 leaves (circuits of 2^10 .. 2^12 rows):
//       one lane per leaf leaves half the SIMDs without a wave there and every wave alone on its SIMD (17 sponge blocks
//       x 42 us = 0.72 ms for the wires commitment of a 2^12-row circuit), sixteen lanes per leaf oversubscribe the chip.
// Full rounds: the MDS layer is circulant, so with three CONTIGUOUS words per lane the twelve words a lane needs are its own
// three plus the three of each neighbour at quad distance 1, 2, 3 — nine DPP quad rotations of 64-bit values — and in the
// lane's own numbering (word m' of lane q = state word 3q + m' mod 12) row a' uses coefficient c[(m' - a') mod 12] in EVERY
// lane: the coefficients stay scalar operands.  Partial rounds: the production grouping (poseidon.h: round 3's MDS layer fused
// with partial round 4, then seven groups of three rounds as one 12 x 14 map each) with every lane holding all twelve words
// in the GLOBAL numbering (twelve DPP quad broadcasts): the two inner S-box inputs are computed by all four lanes alike
// (scalar coefficients), each lane then evaluates its own three rows, whose coefficients it keeps in registers for the whole
// permutation.  Constants folded into the accumulators as everywhere else; bit-identical results.
#pragma once
#include <hip/hip_runtime.h>

#include "poseidon.h"

namespace poseidon_coop4 {

constexpr int GROUP = 4;  // lanes per state
constexpr int WPL = 3;    // words per lane

// the value lane (q + B) mod 4 of the quad holds
template <int B>
__device__ __forceinline__ uint32_t quad_rot(uint32_t v) {
    constexpr int ctrl = ((0 + B) & 3) | (((1 + B) & 3) << 2) | (((2 + B) & 3) << 4) | (((3 + B) & 3) << 6);  // quad_perm
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, false);
}
// the value lane K of the quad holds
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, false);  // quad_perm:[K,K,K,K]
}

// s[a] = state word 3q + a.  next_rc[a]: constant of word 3q + a in the round that follows (0 after the last round).
// diag8: 8 in lane 0 (the MDS diagonal entry of word 0), 0 elsewhere.
__device__ __forceinline__ void mds_layer(uint64_t (&s)[WPL], uint32_t diag8, const uint64_t (&next_rc)[WPL]) {
    uint32_t lo[poseidon::WIDTH], hi[poseidon::WIDTH];  // index m': state word (3q + m') mod 12
#pragma unroll
    for (int e = 0; e < WPL; ++e) {
        lo[e] = (uint32_t)s[e];
        hi[e] = (uint32_t)(s[e] >> 32);
        lo[3 + e] = quad_rot<1>(lo[e]);
        hi[3 + e] = quad_rot<1>(hi[e]);
        lo[6 + e] = quad_rot<2>(lo[e]);
        hi[6 + e] = quad_rot<2>(hi[e]);
        lo[9 + e] = quad_rot<3>(lo[e]);
        hi[9 + e] = quad_rot<3>(hi[e]);
    }
#pragma unroll
    for (int a = 0; a < WPL; ++a) {
        uint64_t alo = (uint32_t)next_rc[a], ahi = next_rc[a] >> 32;
#pragma unroll
        for (int m = 0; m < poseidon::WIDTH; ++m) {
            const uint32_t c = poseidon::MDS_CIRC_TAB[(m - a + poseidon::WIDTH) % poseidon::WIDTH];
            alo += (uint64_t)lo[m] * c;
            ahi += (uint64_t)hi[m] * c;
        }
        if (a == 0) {
            alo += (uint64_t)lo[0] * diag8;
            ahi += (uint64_t)hi[0] * diag8;
        }
        s[a] = poseidon::mds_combine(alo, ahi);
    }
}

// all twelve state words in the global numbering, in every lane of the quad
__device__ __forceinline__ void gather_global(const uint64_t (&s)[WPL], uint32_t* lo, uint32_t* hi) {
#pragma unroll
    for (int e = 0; e < WPL; ++e) {
        const uint32_t l = (uint32_t)s[e], h = (uint32_t)(s[e] >> 32);
        lo[0 + e] = quad_bcast<0>(l);
        hi[0 + e] = quad_bcast<0>(h);
        lo[3 + e] = quad_bcast<1>(l);
        hi[3 + e] = quad_bcast<1>(h);
        lo[6 + e] = quad_bcast<2>(l);
        hi[6 + e] = quad_bcast<2>(h);
        lo[9 + e] = quad_bcast<3>(l);
        hi[9 + e] = quad_bcast<3>(h);
    }
}

// poseidon::mds_partial1_fused on four lanes.  In: this lane's S-box outputs of full round 3.  Out: its words of the state that
// enters partial round 5.  row1[a][.] = P1_OUT[3q + a][.], k1[a] = P1_K[3q + a].
__device__ __forceinline__ void mds_partial1_fused(uint64_t (&s)[WPL], const uint32_t (&row1)[WPL][poseidon::WIDTH + 1],
                                                   const uint64_t (&k1)[WPL]) {
    uint32_t lo[poseidon::WIDTH + 1], hi[poseidon::WIDTH + 1];
    gather_global(s, lo, hi);
    {
        const uint64_t c = poseidon::RC[poseidon::HALF_FULL * poseidon::WIDTH];  // constant of element 0 in round 4
        uint64_t alo = (uint32_t)c, ahi = c >> 32;
#pragma unroll
        for (int i = 0; i < poseidon::WIDTH; ++i) {
            alo += (uint64_t)lo[i] * poseidon::MDS_CIRC_TAB[i];
            ahi += (uint64_t)hi[i] * poseidon::MDS_CIRC_TAB[i];
        }
        alo += (uint64_t)lo[0] * 8u;
        ahi += (uint64_t)hi[0] * 8u;
        const uint64_t x1 = poseidon::sbox7(poseidon::mds_combine(alo, ahi));
        lo[poseidon::WIDTH] = (uint32_t)x1;
        hi[poseidon::WIDTH] = (uint32_t)(x1 >> 32);
    }
#pragma unroll
    for (int a = 0; a < WPL; ++a) {
        uint64_t alo = (uint32_t)k1[a], ahi = k1[a] >> 32;
#pragma unroll
        for (int i = 0; i < poseidon::WIDTH + 1; ++i) {
            alo += (uint64_t)lo[i] * row1[a][i];
            ahi += (uint64_t)hi[i] * row1[a][i];
        }
        s[a] = poseidon::mds_combine(alo, ahi);
    }
}

// poseidon::partial_rounds3 on four lanes.  In: s with state word 0 (lane 0, slot 0) = the S-box input of the first of the three
// rounds.  row3[a][.] = P3_OUT[3q + a][.]; next_rc[a]: what follows the group for word 3q + a (zero where nothing does).
__device__ __forceinline__ void partial_rounds3(uint64_t (&s)[WPL], int q, uint64_t k2, uint64_t k3,
                                                const uint32_t (&row3)[WPL][poseidon::WIDTH + 2], const uint64_t (&next_rc)[WPL]) {
    {
        const uint64_t t = poseidon::sbox7(s[0]);
        s[0] = q == 0 ? t : s[0];
    }
    uint32_t lo[poseidon::WIDTH + 2], hi[poseidon::WIDTH + 2];
    gather_global(s, lo, hi);
    {
        uint64_t alo = (uint32_t)k2, ahi = k2 >> 32;
#pragma unroll
        for (int i = 0; i < poseidon::WIDTH; ++i) {
            alo += (uint64_t)lo[i] * poseidon::P3_IN2[i];
            ahi += (uint64_t)hi[i] * poseidon::P3_IN2[i];
        }
        const uint64_t x2 = poseidon::sbox7(poseidon::mds_combine(alo, ahi));
        lo[poseidon::WIDTH] = (uint32_t)x2;
        hi[poseidon::WIDTH] = (uint32_t)(x2 >> 32);
    }
    {
        uint64_t alo = (uint32_t)k3, ahi = k3 >> 32;
#pragma unroll
        for (int i = 0; i < poseidon::WIDTH + 1; ++i) {
            alo += (uint64_t)lo[i] * poseidon::P3_IN3[i];
            ahi += (uint64_t)hi[i] * poseidon::P3_IN3[i];
        }
        const uint64_t x3 = poseidon::sbox7(poseidon::mds_combine(alo, ahi));
        lo[poseidon::WIDTH + 1] = (uint32_t)x3;
        hi[poseidon::WIDTH + 1] = (uint32_t)(x3 >> 32);
    }
#pragma unroll
    for (int a = 0; a < WPL; ++a) {
        uint64_t alo = (uint32_t)next_rc[a], ahi = next_rc[a] >> 32;
#pragma unroll
        for (int i = 0; i < poseidon::WIDTH + 2; ++i) {
            alo += (uint64_t)lo[i] * row3[a][i];
            ahi += (uint64_t)hi[i] * row3[a][i];
        }
        s[a] = poseidon::mds_combine(alo, ahi);
    }
}

// state in / state out: arbitrary u64 representatives; q = lane % 4
__device__ __forceinline__ void permute(uint64_t (&s)[WPL], int q) {
    const uint32_t diag8 = q == 0 ? 8u : 0u;
    const uint64_t* rc = poseidon::RC + 3 * q;  // this lane's three constants of a round: rc[round * 12 + a]
    // this lane's rows of the two grouped-round matrices: per-lane (vector) loads, issued first so that their latency hides
    // behind the four full rounds
    uint32_t row3[WPL][poseidon::WIDTH + 2], row1[WPL][poseidon::WIDTH + 1];
    uint64_t k1[WPL];
#pragma unroll
    for (int a = 0; a < WPL; ++a) {
#pragma unroll
        for (int i = 0; i < poseidon::WIDTH + 2; ++i) row3[a][i] = poseidon::P3_OUT[3 * q + a][i];
#pragma unroll
        for (int i = 0; i < poseidon::WIDTH + 1; ++i) row1[a][i] = poseidon::P1_OUT[3 * q + a][i];
        k1[a] = poseidon::P1_K[3 * q + a];
    }
#pragma unroll
    for (int a = 0; a < WPL; ++a) s[a] = gl::add(s[a], rc[a]);
#pragma unroll 1
    for (int round = 0; round < poseidon::HALF_FULL - 1; ++round) {
#pragma unroll
        for (int a = 0; a < WPL; ++a) s[a] = poseidon::sbox7(s[a]);
        uint64_t nrc[WPL];
#pragma unroll
        for (int a = 0; a < WPL; ++a) nrc[a] = rc[(round + 1) * poseidon::WIDTH + a];
        mds_layer(s, diag8, nrc);
    }
#pragma unroll
    for (int a = 0; a < WPL; ++a) s[a] = poseidon::sbox7(s[a]);  // full round 3 ...
    mds_partial1_fused(s, row1, k1);                               // ... its MDS layer fused with partial round 4
    // partial rounds 5..25 in seven groups of three (pushed constants: after a group only word 0 receives one; the last group
    // is followed by the full vector of round 26)
#pragma unroll 1
    for (int g = 0; g < 7; ++g) {
        uint64_t nrc[WPL] = {0, 0, 0};
        if (g == 6) {
#pragma unroll
            for (int a = 0; a < WPL; ++a) nrc[a] = poseidon::RC_AFTER_PARTIAL[3 * q + a];
        } else if (q == 0) {
            nrc[0] = poseidon::RC_PARTIAL_S0[3 * g + 3];
        }
        partial_rounds3(s, q, poseidon::RC_PARTIAL_S0[3 * g + 1], poseidon::RC_PARTIAL_S0[3 * g + 2], row3, nrc);
    }
#pragma unroll 1
    for (int round = poseidon::HALF_FULL + poseidon::PARTIAL; round < poseidon::ROUNDS; ++round) {
#pragma unroll
        for (int a = 0; a < WPL; ++a) s[a] = poseidon::sbox7(s[a]);
        uint64_t nrc[WPL];  // RC has an all-zero row after the last round
#pragma unroll
        for (int a = 0; a < WPL; ++a) nrc[a] = rc[(round + 1) * poseidon::WIDTH + a];
        mds_layer(s, diag8, nrc);
    }
}

}  // namespace poseidon_coop4
