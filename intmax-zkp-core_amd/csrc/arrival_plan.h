// arrival_plan.h — how a host witness of nw columns is cut into arrival chunks, and which sponge launches follow them.
// Host-only logic (no HIP): included by glprover.hip and by host_selftest.cpp, so that the CPU suite checks its invariants for every
// column count (tests/test_device_arith_host.py::test_arrival_plan_invariants).
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

namespace glpi {

constexpr size_t SPONGE_RATE = 8;  // = poseidon::RATE (static_assert where both are visible)

// Chunk boundaries: multiples of eight columns (the rate of the leaf sponge, which follows the chunks: merkle_leaf_chunk).  The
// first chunks are small — nothing can start before chunk 0 is in HBM, and the GPU needs ~7 ms per 16 columns of a 2^20-row
// witness (inverse NTT, LDE, one permutation per eight columns) where the bus delivers them in ~2.5 — later ones grow, so that
// the launches stay few and wide.  A remainder below eight columns (135 = 16 x 8 + 7) rides with the last chunk.
// widths: the chunk widths (multiples of 8, >= 8; the last one repeats).
inline void plan_arrival_chunks(size_t nw, const std::vector<size_t>& widths, std::vector<size_t>* bounds) {
    bounds->assign(1, 0);
    size_t at = 0;
    for (size_t k = 0; at < nw; ++k) {
        size_t take = widths.empty() ? nw : widths[k < widths.size() ? k : widths.size() - 1];
        if (take < SPONGE_RATE) take = SPONGE_RATE;
        take -= take % SPONGE_RATE;
        if (at + take + SPONGE_RATE > nw) take = nw - at;  // what would be left is less than a block: it rides along
        at += take;
        bounds->push_back(at);
    }
}

// The sponge launch (first word, words) that may follow chunk k of `bounds` when `hashed` words of the `width`-word leaves have been
// absorbed, or (hashed, 0) for none: whole blocks of what has arrived, and at least one whole block stays for the LAST launch — the
// permutation in front of a partial last block must deliver words tail..11, which are not carried between launches.  The last chunk
// launches nothing here: the caller's final launch absorbs [hashed, width) after the salt columns (if any) are in place.
inline std::pair<size_t, size_t> sponge_launch_after_chunk(const std::vector<size_t>& bounds, size_t k, size_t width, size_t hashed) {
    if (k + 2 >= bounds.size()) return {hashed, 0};  // the last chunk
    size_t e = bounds[k + 1] / SPONGE_RATE * SPONGE_RATE;
    while (e > hashed && width - e < SPONGE_RATE) e -= SPONGE_RATE;
    return {hashed, e > hashed ? e - hashed : 0};
}

}  // namespace glpi
