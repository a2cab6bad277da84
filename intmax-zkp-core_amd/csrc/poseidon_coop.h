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
    uint64_t alo = (uint32_t)next_rc, ahi = next_rc >> 32;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        int src = rr + k;
        src = src >= 12 ? src - 12 : src;
        const uint64_t v = k == 0 ? s : __shfl(s, src, GROUP);
        alo += (uint64_t)(uint32_t)v * (uint32_t)GL_MDS_CIRC(k);
        ahi += (uint64_t)(uint32_t)(v >> 32) * (uint32_t)GL_MDS_CIRC(k);
    }
    if (r == 0) {
        alo += (uint64_t)(uint32_t)s * 8u;
        ahi += (uint64_t)(uint32_t)(s >> 32) * 8u;
    }
    return poseidon::mds_combine(alo, ahi);
}

// s = state word r of this lane's group (r = lane % 16; lanes with r >= 12 carry a dummy word and idle along)
__device__ __forceinline__ uint64_t permute(uint64_t s, int r) {
    const int rr = r < 12 ? r : 0;  // idle lanes mirror lane 0's indices (their result is discarded)
    s = gl::add(s, poseidon::RC[rr]);
#pragma unroll 1
    for (int round = 0; round < poseidon::ROUNDS; ++round) {
        const bool full = round < poseidon::HALF_FULL || round >= poseidon::HALF_FULL + poseidon::PARTIAL;
        const uint64_t sb = poseidon::sbox7(s);
        s = (full || r == 0) ? sb : s;
        s = mds_row(s, r, rr, poseidon::RC[(round + 1) * 12 + rr]);
    }
    return s;
}

}  // namespace poseidon_coop
