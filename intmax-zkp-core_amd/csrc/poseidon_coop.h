// poseidon_coop.h — cooperative Poseidon permutation: ONE state element per lane, 16-lane groups (12 active lanes),
// MDS layer through wavefront shuffles (ds_bpermute, no LDS memory).
//
// Same function as poseidon.h (plonky2 @ f99ed9c src/hash/poseidon.rs::Poseidon::poseidon; reference call sites
// src/merkle_tree/tree.rs:70,79,90,123 and every Merkle cap inside CircuitData::prove), different mapping:
//   lane-per-state (poseidon.h): 64 permutations per wave, ~94 k issue cycles, ~55 us of dependent latency — the right
//       shape for the wide levels (millions of leaves), throughput-bound;
//   cooperative (this file): 4 permutations per wave, each lane does 1/12 of every layer (its own S-box, its own MDS
//       row from 11 shuffled neighbours) — ~4x lower latency per permutation, ~3.6x lower throughput.  Used where a
//       tree level has so few nodes that the lane-per-state kernel is latency-bound: the top <= 256-parent levels of
//       every Merkle tree and the leaf sponges of tiny circuits (zkdsa-sized traces).
// Round constants are folded into the MDS accumulators exactly as in poseidon.h; results are bit-identical.
#pragma once
#include <hip/hip_runtime.h>

#include "poseidon.h"

namespace poseidon_coop {

constexpr int GROUP = 16;  // lanes per state: 12 hold state words, 4 idle (keeps shuffle groups power-of-two aligned)

// MDS row of lane r: sum_k c_k s_{(r+k) mod 12} + 8 s_0 [r == 0] + next round constant, by 11 neighbour shuffles
__device__ __forceinline__ uint64_t mds_row(uint64_t s, int r, int rr, uint64_t next_rc) {
#ifndef GL_COOP_CHAINS
#define GL_COOP_CHAINS 2
#endif
    // GL_COOP_CHAINS interleaved partial sums per half: the cooperative form is used where LATENCY is the cost — one wave, a chain of
    // dependent permutations (tools/perm_latency_bench.hip: 12.13 us per permutation with one chain of twelve dependent multiply-adds
    // per half, 11.37 with two, 12.49 with three, 11.54 with four)
    uint64_t alo[GL_COOP_CHAINS], ahi[GL_COOP_CHAINS];
#pragma unroll
    for (int j = 0; j < GL_COOP_CHAINS; ++j) alo[j] = ahi[j] = 0;
    alo[0] = (uint32_t)next_rc;
    ahi[0] = next_rc >> 32;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        int src = rr + k;
        src = src >= 12 ? src - 12 : src;
        const uint64_t v = k == 0 ? s : __shfl(s, src, GROUP);
        alo[k % GL_COOP_CHAINS] += (uint64_t)(uint32_t)v * (uint32_t)GL_MDS_CIRC(k);
        ahi[k % GL_COOP_CHAINS] += (uint64_t)(uint32_t)(v >> 32) * (uint32_t)GL_MDS_CIRC(k);
    }
    if (r == 0) {
        alo[0] += (uint64_t)(uint32_t)s * 8u;
        ahi[0] += (uint64_t)(uint32_t)(s >> 32) * 8u;
    }
#pragma unroll
    for (int j = 1; j < GL_COOP_CHAINS; ++j) {
        alo[0] += alo[j];
        ahi[0] += ahi[j];
    }
    return poseidon::mds_combine(alo[0], ahi[0]);
}

// three partial rounds at once (poseidon.h: P3_IN2 / P3_IN3 / P3_OUT): ONE round of shuffles gathers u = (x_1, s_1..s_11)
// into every lane; each lane then computes the two inner S-box inputs redundantly (no further shuffles) and its own row
// of the 12 x 14 output map.  `row` = the lane's 14 coefficients of P3_OUT, loaded once per permutation.
__device__ __forceinline__ uint64_t partial_rounds3(uint64_t s, int r, const uint32_t (&row)[poseidon::WIDTH + 2], uint64_t k2,
                                                    uint64_t k3, uint64_t next_rc) {
    const uint64_t sb = poseidon::sbox7(s);
    s = r == 0 ? sb : s;
    uint32_t lo[poseidon::WIDTH + 2], hi[poseidon::WIDTH + 2];
#pragma unroll
    for (int k = 0; k < poseidon::WIDTH; ++k) {
        const uint64_t v = __shfl(s, k, GROUP);
        lo[k] = (uint32_t)v;
        hi[k] = (uint32_t)(v >> 32);
    }
    {
        uint64_t alo = (uint32_t)k2, ahi = k2 >> 32;
#pragma unroll
        for (int k = 0; k < poseidon::WIDTH; ++k) {
            alo += (uint64_t)lo[k] * poseidon::P3_IN2[k];
            ahi += (uint64_t)hi[k] * poseidon::P3_IN2[k];
        }
        const uint64_t x2 = poseidon::sbox7(poseidon::mds_combine(alo, ahi));
        lo[poseidon::WIDTH] = (uint32_t)x2;
        hi[poseidon::WIDTH] = (uint32_t)(x2 >> 32);
    }
    {
        uint64_t alo = (uint32_t)k3, ahi = k3 >> 32;
#pragma unroll
        for (int k = 0; k < poseidon::WIDTH + 1; ++k) {
            alo += (uint64_t)lo[k] * poseidon::P3_IN3[k];
            ahi += (uint64_t)hi[k] * poseidon::P3_IN3[k];
        }
        const uint64_t x3 = poseidon::sbox7(poseidon::mds_combine(alo, ahi));
        lo[poseidon::WIDTH + 1] = (uint32_t)x3;
        hi[poseidon::WIDTH + 1] = (uint32_t)(x3 >> 32);
    }
    uint64_t alo = (uint32_t)next_rc, ahi = next_rc >> 32;
#pragma unroll
    for (int k = 0; k < poseidon::WIDTH + 2; ++k) {
        alo += (uint64_t)lo[k] * row[k];
        ahi += (uint64_t)hi[k] * row[k];
    }
    return poseidon::mds_combine(alo, ahi);
}

// MDS layer of full round 3 fused with partial round 4 (poseidon.h: P1_OUT / P1_K).  In: this lane's S-box output of round 3.
// ONE round of shuffles gathers all twelve; every lane computes the S-box input of round 4 and its S-box redundantly (no
// further shuffles) and then its own row of the 12 x 13 map.
__device__ __forceinline__ uint64_t mds_partial1_fused(uint64_t s, int rr) {
    uint32_t lo[poseidon::WIDTH + 1], hi[poseidon::WIDTH + 1];
#pragma unroll
    for (int k = 0; k < poseidon::WIDTH; ++k) {
        const uint64_t v = __shfl(s, k, GROUP);
        lo[k] = (uint32_t)v;
        hi[k] = (uint32_t)(v >> 32);
    }
    {
        const uint64_t c = poseidon::RC[poseidon::HALF_FULL * poseidon::WIDTH];
        uint64_t alo = (uint32_t)c, ahi = c >> 32;
#pragma unroll
        for (int k = 0; k < poseidon::WIDTH; ++k) {
            alo += (uint64_t)lo[k] * (uint32_t)GL_MDS_CIRC(k);
            ahi += (uint64_t)hi[k] * (uint32_t)GL_MDS_CIRC(k);
        }
        alo += (uint64_t)lo[0] * 8u;
        ahi += (uint64_t)hi[0] * 8u;
        const uint64_t x1 = poseidon::sbox7(poseidon::mds_combine(alo, ahi));
        lo[poseidon::WIDTH] = (uint32_t)x1;
        hi[poseidon::WIDTH] = (uint32_t)(x1 >> 32);
    }
    const uint64_t kc = poseidon::P1_K[rr];
    uint64_t alo = (uint32_t)kc, ahi = kc >> 32;
#pragma unroll
    for (int k = 0; k < poseidon::WIDTH + 1; ++k) {
        const uint32_t c = poseidon::P1_OUT[rr][k];
        alo += (uint64_t)lo[k] * c;
        ahi += (uint64_t)hi[k] * c;
    }
    return poseidon::mds_combine(alo, ahi);
}

// s = state word r of this lane's group (r = lane % 16; lanes with r >= 12 carry a dummy word and idle along)
__device__ __forceinline__ uint64_t permute(uint64_t s, int r) {
    const int rr = r < 12 ? r : 0;  // idle lanes mirror lane 0's indices (their result is discarded)
    // this lane's row of the grouped partial-round matrix: per-lane (vector) loads, issued first so that their latency hides
    // behind the four full rounds (the cooperative form is used where latency, not throughput, is the cost)
    uint32_t row[poseidon::WIDTH + 2];
#pragma unroll
    for (int k = 0; k < poseidon::WIDTH + 2; ++k) row[k] = poseidon::P3_OUT[rr][k];
    s = gl::add(s, poseidon::RC[rr]);
#pragma unroll 1
    for (int round = 0; round < poseidon::HALF_FULL - 1; ++round) {
        s = poseidon::sbox7(s);
        s = mds_row(s, r, rr, poseidon::RC[(round + 1) * 12 + rr]);
    }
    s = poseidon::sbox7(s);          // full round 3 ...
    s = mds_partial1_fused(s, rr);   // ... its MDS layer fused with partial round 4
    // partial rounds 5..25 in seven groups of three (pushed constants: only lane 0 adds one; the last group is followed by
    // the full vector of round 26)
#pragma unroll 1
    for (int g = 0; g < 7; ++g)
        s = partial_rounds3(s, r, row, poseidon::RC_PARTIAL_S0[3 * g + 1], poseidon::RC_PARTIAL_S0[3 * g + 2],
                            g == 6 ? poseidon::RC_AFTER_PARTIAL[rr] : (r == 0 ? poseidon::RC_PARTIAL_S0[g == 6 ? 0 : 3 * g + 3] : 0));
#pragma unroll 1
    for (int round = poseidon::HALF_FULL + poseidon::PARTIAL; round < poseidon::ROUNDS; ++round) {
        s = poseidon::sbox7(s);
        s = mds_row(s, r, rr, poseidon::RC[(round + 1) * 12 + rr]);
    }
    return s;
}

}  // namespace poseidon_coop
