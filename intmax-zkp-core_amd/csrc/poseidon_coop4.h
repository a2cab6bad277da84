// poseidon_coop4.h — Poseidon permutation on FOUR lanes: lane q of a quad holds the state words 3q, 3q + 1, 3q + 2.
//
// Same function as poseidon.h / poseidon_coop.h (plonky2 @ f99ed9c src/hash/poseidon.rs::Poseidon::poseidon), a third
// mapping for the sizes between them:
//   lane-per-state (poseidon.h)        13.6 k dependent instructions per permutation, 64 permutations per wave
//   16 lanes per state (poseidon_coop) ~3 k, 4 per wave   — leaf sets up to 2^13, tree levels up to 2^15 parents
//   4 lanes per state (this file)      ~5.7 k, 16 per wave — leaf sets of 2^13 .. 2^15 leaves (circuits of 2^10 .. 2^12 rows):
//       one lane per leaf leaves half the SIMDs without a wave there and every wave alone on its SIMD (17 sponge blocks
//       x 42 us = 0.72 ms for the wires commitment of a 2^12-row circuit), sixteen lanes per leaf oversubscribe the chip.
// The MDS layer is circulant, so with three CONTIGUOUS words per lane the twelve words a lane needs are its own three plus
// the three of each neighbour at quad distance 1, 2, 3 — nine DPP quad rotations of 64-bit values — and in the lane's own
// numbering (word m' of lane q = state word 3q + m' mod 12) row a' uses coefficient c[(m' - a') mod 12] in EVERY lane:
// the coefficients stay scalar operands.  Reference round structure (constant layer, S-box, MDS), the next round's
// constants folded into the accumulators as everywhere else; bit-identical results.
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

// state in / state out: arbitrary u64 representatives; q = lane % 4
__device__ __forceinline__ void permute(uint64_t (&s)[WPL], int q) {
    const uint32_t diag8 = q == 0 ? 8u : 0u;
    const uint64_t* rc = poseidon::RC + 3 * q;  // this lane's three constants of a round: rc[round * 12 + a]
#pragma unroll
    for (int a = 0; a < WPL; ++a) s[a] = gl::add(s[a], rc[a]);
#pragma unroll 1
    for (int round = 0; round < poseidon::ROUNDS; ++round) {
        const bool full = round < poseidon::HALF_FULL || round >= poseidon::HALF_FULL + poseidon::PARTIAL;  // wave-uniform
        if (full) {
#pragma unroll
            for (int a = 0; a < WPL; ++a) s[a] = poseidon::sbox7(s[a]);
        } else {
            const uint64_t t = poseidon::sbox7(s[0]);  // only state word 0 (lane 0, slot 0) meets the S-box
            s[0] = q == 0 ? t : s[0];
        }
        uint64_t nrc[WPL];  // RC has an all-zero row after the last round
#pragma unroll
        for (int a = 0; a < WPL; ++a) nrc[a] = rc[(round + 1) * poseidon::WIDTH + a];
        mds_layer(s, diag8, nrc);
    }
}

}  // namespace poseidon_coop4
