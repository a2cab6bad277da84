// merkle_kernels.h — Poseidon sponge over leaves and 2-to-1 compression levels down to the Merkle cap.
//
// Replaces plonky2 @ f99ed9c src/hash/merkle_tree.rs::MerkleTree::new (fill_digests_buf / fill_subtree) and
// PoseidonHash::{hash_or_noop, two_to_one}; reached from the reference through every CircuitData::prove /
// CircuitBuilder::build (e.g. reference src/rollup/circuits/mod.rs:605,1247).
//
// One lane = one leaf (or one tree node): the 12-word sponge state stays in that lane's VGPRs for the whole
// absorb loop, so the only memory traffic is the compulsory read of the leaf elements and the 32-byte digest write.
// Leaves are addressed as elem(i, c) = leaves[i * row_stride + c * col_stride]; the LDE produced by this library is
// column-major (row_stride = 1), which makes every absorb load a fully coalesced 512-byte wave read.
// Digest buffer: level 0 = leaf digests (n_leaves x 4 u64), level l follows level l-1, last level = the cap.
// A FOREST of T equal trees (the batched prover: one tree per proof) is stored level by level across the trees — level l is
// [tree][n_leaves >> l][4] — so the level kernels see one array of T * n nodes whose sibling pairs never straddle two trees;
// tree t's leaves start at leaves + t * tree_stride (leaf kernels) and its cap is entries [t * n_cap, (t + 1) * n_cap) of the
// last level.
#pragma once
#include <hip/hip_runtime.h>

#include "poseidon.h"
#include "gl_canon.h"
#include "poseidon_coop.h"
#include "poseidon_coop4.h"

namespace merkle {

constexpr int THREADS = 256;

template <int MDS, bool THRU = false>
__device__ __forceinline__ void leaf_hash_body(const uint64_t* __restrict__ leaves, size_t n_leaves, int leaf_len, size_t row_stride,
                                               size_t col_stride, uint64_t* __restrict__ digests, int log_per_tree,
                                               size_t tree_stride) {
    size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;  // leaf of the forest: tree i >> log_per_tree
    if (i >= n_leaves) return;
    const uint64_t* row = leaves + (i >> log_per_tree) * tree_stride + (i & (((size_t)1 << log_per_tree) - 1)) * row_stride;
    uint64_t st[poseidon::WIDTH];
#pragma unroll
    for (int j = 0; j < poseidon::WIDTH; ++j) st[j] = 0;
    if (leaf_len <= 4) {  // hash_or_noop: short leaves are packed, not hashed
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < leaf_len) st[j] = row[(size_t)j * col_stride];
    } else {
        // overwrite-mode sponge: the next chunk replaces the rate words, so a permutation that is followed by a full chunk only
        // has to deliver the capacity words 8..11, the one before a partial last chunk words 4..11, the last one the digest
        const int tail = leaf_len % poseidon::RATE;
        for (int c0 = 0; c0 < leaf_len; c0 += poseidon::RATE) {
#pragma unroll
            for (int j = 0; j < poseidon::RATE; ++j)
                if (c0 + j < leaf_len) st[j] = row[(size_t)(c0 + j) * col_stride];
            const int left = leaf_len - c0 - poseidon::RATE;  // elements after this chunk
            poseidon::permute<MDS, THRU>(st, left <= 0                ? poseidon::OUT_DIGEST
                                       : left >= poseidon::RATE ? poseidon::OUT_CAPACITY
                                       : tail >= 4              ? poseidon::OUT_TAIL  // the last chunk keeps words tail..7
                                                                : poseidon::OUT_ALL);
        }
    }
    ulonglong2* out = reinterpret_cast<ulonglong2*>(digests + i * 4);
    out[0] = make_ulonglong2(gl::canon(st[0]), gl::canon(st[1]));
    out[1] = make_ulonglong2(gl::canon(st[2]), gl::canon(st[3]));
}

template <int MDS>
__global__ __launch_bounds__(THREADS) void leaf_hash_kernel(const uint64_t* __restrict__ leaves, size_t n_leaves,
                                                            int leaf_len, size_t row_stride, size_t col_stride,
                                                            uint64_t* __restrict__ digests, int log_per_tree, size_t tree_stride) {
    leaf_hash_body<MDS>(leaves, n_leaves, leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
}
// The same for leaf sets that fill the chip (>= THRU_MIN_NODES leaves; poseidon.h permute_folded<.., THRU>).  Six waves per SIMD is
// stated: left alone the compiler schedules this form for instruction-level parallelism and takes 108-112 registers (four waves).
constexpr size_t THRU_MIN_NODES = (size_t)1 << 19;  // 8 192 waves: eight per SIMD
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void leaf_hash_thru_kernel(
    const uint64_t* __restrict__ leaves, size_t n_leaves, int leaf_len, size_t row_stride, size_t col_stride,
    uint64_t* __restrict__ digests, int log_per_tree, size_t tree_stride) {
    leaf_hash_body<poseidon::MDS_INT, true>(leaves, n_leaves, leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
}
// the opt-in matrix-core variant with a register budget of its own (four waves per SIMD: left alone the compiler takes 132
// registers for it — three waves — by issuing the products early)
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void leaf_hash_mfma_kernel(
    const uint64_t* __restrict__ leaves, size_t n_leaves, int leaf_len, size_t row_stride, size_t col_stride,
    uint64_t* __restrict__ digests, int log_per_tree, size_t tree_stride) {
    leaf_hash_body<poseidon::MDS_MFMA>(leaves, n_leaves, leaf_len, row_stride, col_stride, digests, log_per_tree, tree_stride);
}

// The leaf sponge in column CHUNKS: one launch absorbs columns [c0, c0 + cnt) of every leaf.  For a witness that is still
// arriving from the host (glp_prove_cols / glp_prove: internal.h, arrival_*): the columns that are already transformed are hashed
// while the rest is on the bus, instead of the whole hashing waiting for the last column.  The overwrite-mode sponge only carries its
// four capacity words across a full 8-word block, so the state between two launches is exactly what the leaf's digest slot
// holds: digests[i][0..4] = words 8..11 of the state (raw representatives: the next launch continues the very computation of
// leaf_hash_body, bit for bit), and after the last launch the digest.  The driver cuts the chunks so that c0 is a multiple of
// eight and every launch but the last ends on a full block that is FOLLOWED by at least one more full block's worth of words
// or by nothing (a partial last block is absorbed in the same launch as the full block before it: the permutation in front
// of it must deliver words tail..11, which are not carried).
template <int MDS, bool THRU>
__device__ __forceinline__ void leaf_hash_chunk_body(const uint64_t* __restrict__ leaves, size_t n_leaves, int leaf_len, int c0, int cnt,
                                                     size_t row_stride, size_t col_stride, uint64_t* __restrict__ digests) {
    const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n_leaves) return;
    const uint64_t* row = leaves + i * row_stride;
    ulonglong2* slot = reinterpret_cast<ulonglong2*>(digests + i * 4);
    uint64_t st[poseidon::WIDTH];
#pragma unroll
    for (int j = 0; j < poseidon::WIDTH; ++j) st[j] = 0;
    if (c0 > 0) {
        const ulonglong2 a = slot[0], b = slot[1];
        st[8] = a.x; st[9] = a.y; st[10] = b.x; st[11] = b.y;
    }
    const int tail = leaf_len % poseidon::RATE, end = c0 + cnt;
    for (int cb = c0; cb < end; cb += poseidon::RATE) {
#pragma unroll
        for (int j = 0; j < poseidon::RATE; ++j)
            if (cb + j < leaf_len) st[j] = row[(size_t)(cb + j) * col_stride];
        const int left = leaf_len - cb - poseidon::RATE;  // elements after this block
        poseidon::permute<MDS, THRU>(st, left <= 0                ? poseidon::OUT_DIGEST
                                   : left >= poseidon::RATE ? poseidon::OUT_CAPACITY
                                   : tail >= 4              ? poseidon::OUT_TAIL
                                                            : poseidon::OUT_ALL);
    }
    if (end >= leaf_len) {
        slot[0] = make_ulonglong2(gl::canon(st[0]), gl::canon(st[1]));
        slot[1] = make_ulonglong2(gl::canon(st[2]), gl::canon(st[3]));
    } else {
        slot[0] = make_ulonglong2(st[8], st[9]);
        slot[1] = make_ulonglong2(st[10], st[11]);
    }
}
template <int MDS>
__global__ __launch_bounds__(THREADS) void leaf_hash_chunk_kernel(const uint64_t* __restrict__ leaves, size_t n_leaves, int leaf_len,
                                                                  int c0, int cnt, size_t row_stride, size_t col_stride,
                                                                  uint64_t* __restrict__ digests) {
    leaf_hash_chunk_body<MDS, false>(leaves, n_leaves, leaf_len, c0, cnt, row_stride, col_stride, digests);
}
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void leaf_hash_chunk_thru_kernel(
    const uint64_t* __restrict__ leaves, size_t n_leaves, int leaf_len, int c0, int cnt, size_t row_stride, size_t col_stride,
    uint64_t* __restrict__ digests) {
    leaf_hash_chunk_body<poseidon::MDS_INT, true>(leaves, n_leaves, leaf_len, c0, cnt, row_stride, col_stride, digests);
}

template <int MDS, bool THRU = false>
__device__ __forceinline__ void compress_node(const uint64_t* __restrict__ children, uint64_t* __restrict__ parent) {
    const ulonglong2* in = reinterpret_cast<const ulonglong2*>(children);
    uint64_t st[poseidon::WIDTH];
    ulonglong2 a = in[0], b = in[1], c = in[2], d = in[3];
    st[0] = a.x; st[1] = a.y; st[2] = b.x; st[3] = b.y;
    st[4] = c.x; st[5] = c.y; st[6] = d.x; st[7] = d.y;
    st[8] = st[9] = st[10] = st[11] = 0;
    if (MDS == poseidon::MDS_INT)
        poseidon::permute_cap0<THRU>(st, poseidon::OUT_DIGEST);  // round 0 knows that the capacity words are zero
    else
        poseidon::permute<MDS>(st, poseidon::OUT_DIGEST);
    ulonglong2* out = reinterpret_cast<ulonglong2*>(parent);
    out[0] = make_ulonglong2(gl::canon(st[0]), gl::canon(st[1]));
    out[1] = make_ulonglong2(gl::canon(st[2]), gl::canon(st[3]));
}

// one tree level: parent[j] = two_to_one(child[2j], child[2j+1])
template <int MDS>
__global__ __launch_bounds__(THREADS) void level_kernel(const uint64_t* __restrict__ children, size_t n_parents,
                                                        uint64_t* __restrict__ parents) {
    size_t j = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (j >= n_parents) return;
    compress_node<MDS>(children + j * 8, parents + j * 4);
}

// the same for levels of >= THRU_MIN_NODES parents
__global__ __launch_bounds__(THREADS) void level_thru_kernel(const uint64_t* __restrict__ children, size_t n_parents,
                                                             uint64_t* __restrict__ parents) {
    size_t j = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (j >= n_parents) return;
    compress_node<poseidon::MDS_INT, true>(children + j * 8, parents + j * 4);
}

// one tree level with the cooperative permutation, 16 lanes per parent: for levels too narrow to fill the chip with one
// lane per node (256 waves of lane-per-node work at 2^14 parents leave 3/4 of the SIMDs idle for ~55 us; the same level
// as 4096 cooperative waves takes one cooperative latency, ~15 us)
__global__ __launch_bounds__(THREADS) void level_coop_kernel(const uint64_t* __restrict__ children, size_t n_parents,
                                                             uint64_t* __restrict__ parents) {
    const size_t j = ((size_t)blockIdx.x * THREADS + threadIdx.x) / poseidon_coop::GROUP;
    const int r = threadIdx.x % poseidon_coop::GROUP;
    if (j >= n_parents) return;  // whole 16-lane groups leave together
    uint64_t s = r < 8 ? children[j * 8 + r] : 0;
    s = poseidon_coop::permute(s, r);
    if (r < 4) parents[j * 4 + r] = gl::canon(s);
}

// the same level with four lanes per parent (poseidon_coop4.h): for the levels where sixteen lanes per parent are already
// several waves per SIMD (2^13 .. 2^15 parents: 2048 .. 8192 sixteen-lane waves, but 512 .. 2048 four-lane ones)
__global__ __launch_bounds__(THREADS) void level_coop4_kernel(const uint64_t* __restrict__ children, size_t n_parents,
                                                              uint64_t* __restrict__ parents) {
    const size_t j = ((size_t)blockIdx.x * THREADS + threadIdx.x) / poseidon_coop4::GROUP;
    const int q = threadIdx.x % poseidon_coop4::GROUP;
    if (j >= n_parents) return;  // whole quads leave together
    uint64_t s[poseidon_coop4::WPL];
#pragma unroll
    for (int a = 0; a < poseidon_coop4::WPL; ++a) s[a] = 3 * q + a < 8 ? children[j * 8 + 3 * q + a] : 0;  // left || right || 0000
    poseidon_coop4::permute(s, q);
#pragma unroll
    for (int a = 0; a < poseidon_coop4::WPL; ++a)
        if (3 * q + a < 4) parents[j * 4 + 3 * q + a] = gl::canon(s[a]);
}

// the narrow top of the tree in ONE workgroup (those levels are latency-bound: nothing to gain from more workgroups):
// starts from a level of `n_children` digests and walks up until `n_stop` digests remain.  Levels with more than
// COOP_MAX_PARENTS parents use one lane per node (one permutation latency, ~55 us, per level); narrower levels switch to
// the cooperative permutation (16 lanes per node, 64 nodes per pass of the 1024-thread workgroup, ~4x lower latency).
constexpr int TOP_THREADS = 1024;
constexpr int COOP_MAX_PARENTS = 256;
// one workgroup per tree of the forest (gridDim.x trees): level arrays hold gridDim.x * n nodes, tree t's at offset t * n
template <int MDS>
__global__ __launch_bounds__(TOP_THREADS) void top_levels_kernel(uint64_t* __restrict__ level0, size_t n_children, size_t n_stop) {
    const size_t n_trees = gridDim.x, tree = blockIdx.x;
    uint64_t* level = level0;  // first node of the current level (all trees)
    size_t n = n_children;
    const int grp = threadIdx.x / poseidon_coop::GROUP, r = threadIdx.x % poseidon_coop::GROUP;
    constexpr int N_GROUPS = TOP_THREADS / poseidon_coop::GROUP;
    while (n > n_stop) {
        const size_t np = n >> 1;
        uint64_t* cur = level + tree * n * 4;
        uint64_t* nxt = level + n_trees * n * 4 + tree * np * 4;
        if (np > COOP_MAX_PARENTS) {
            for (size_t j = threadIdx.x; j < np; j += TOP_THREADS) compress_node<MDS>(cur + j * 8, nxt + j * 4);
        } else {
            for (size_t j = grp; j < np; j += N_GROUPS) {  // uniform per 16-lane group
                uint64_t s = r < 8 ? cur[j * 8 + r] : 0;   // two_to_one: left || right || 0000
                s = poseidon_coop::permute(s, r);
                if (r < 4) nxt[j * 4 + r] = gl::canon(s);
            }
        }
        __threadfence_block();
        __syncthreads();
        level += n_trees * n * 4;
        n = np;
    }
}

// The levels of one CAP SUBTREE in one workgroup (grid: n_cap x n_trees): from a level with n_children nodes per tree
// (n_children / n_cap per subtree) down to the subtree's root = its cap entry.  The 2^cap_height subtrees of a tree never exchange a
// node, so nothing but __syncthreads stands between two levels ("one persistent workgroup cluster per cap
// subtree").  Pays only for the narrow end (profiles/r05_subtree_levels.md: from 32 children per subtree down it replaces four
// launches and the one-workgroup top, -0.4 ms per 2^20-row proof; from 256 children up it LOSES: sixteen workgroups are sixteen
// CUs, and a pass of 64 nodes with four waves per SIMD takes ~10 us where the whole chip hashes a level in one ~15 us launch).
template <int MDS>
__global__ __launch_bounds__(TOP_THREADS) void subtree_levels_kernel(uint64_t* __restrict__ level0, size_t n_children, size_t n_cap,
                                                                     unsigned first_sub = 0, uint64_t* cap_host = nullptr) {
    const size_t n_trees = gridDim.y, tree = blockIdx.y, sub = blockIdx.x + first_sub;  // (first_sub: ONE subtree per launch, glprover.hip)
    // cap_host (optional): host-mapped pinned memory that takes a copy of the cap entry this workgroup ends on — the host, spinning on
    // a sentinel there, has the Merkle cap without a copy kernel and without a stream synchronisation (glprover.hip::merkle_build_t)
    uint64_t* level = level0;
    size_t n = n_children;
    const int grp = threadIdx.x / poseidon_coop::GROUP, r = threadIdx.x % poseidon_coop::GROUP;
    constexpr int N_GROUPS = TOP_THREADS / poseidon_coop::GROUP;
    while (n > n_cap) {
        const size_t np = n >> 1, ns = n / n_cap, nps = ns >> 1;  // per tree: parents; per subtree: children, parents
        const uint64_t* cur = level + (tree * n + sub * ns) * 4;
        uint64_t* nxt = level + n_trees * n * 4 + (tree * np + sub * nps) * 4;
        for (size_t j = grp; j < nps; j += N_GROUPS) {  // uniform per 16-lane group
            uint64_t s = r < 8 ? cur[j * 8 + r] : 0;     // two_to_one: left || right || 0000
            s = poseidon_coop::permute(s, r);
            if (r < 4) {
                const uint64_t v = gl::canon(s);
                nxt[j * 4 + r] = v;
                if (cap_host && np == n_cap)  // the subtree's root = cap entry (tree, sub): nps == 1, j == 0
                    __hip_atomic_store(cap_host + (tree * n_cap + sub) * 4 + r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __threadfence_block();
        __syncthreads();
        level += n_trees * n * 4;
        n = np;
    }
}

// leaf sponge with the cooperative permutation: 16 lanes per leaf.  For tiny leaf sets (a few thousand leaves at most)
// where one-lane-per-leaf would leave the chip idle and pay 55 us per absorbed chunk.
__global__ __launch_bounds__(THREADS) void leaf_hash_coop_kernel(const uint64_t* __restrict__ leaves, size_t n_leaves,
                                                                 int leaf_len, size_t row_stride, size_t col_stride,
                                                                 uint64_t* __restrict__ digests, int log_per_tree, size_t tree_stride) {
    const size_t i = ((size_t)blockIdx.x * THREADS + threadIdx.x) / poseidon_coop::GROUP;
    const int r = threadIdx.x % poseidon_coop::GROUP;
    if (i >= n_leaves) return;  // whole 16-lane groups leave together
    const uint64_t* row = leaves + (i >> log_per_tree) * tree_stride + (i & (((size_t)1 << log_per_tree) - 1)) * row_stride;
    uint64_t s = 0;
    if (leaf_len <= 4) {  // hash_or_noop
        if (r < leaf_len) s = row[(size_t)r * col_stride];
    } else {
        for (int c0 = 0; c0 < leaf_len; c0 += poseidon::RATE) {
            if (r < poseidon::RATE && c0 + r < leaf_len) s = row[(size_t)(c0 + r) * col_stride];  // overwrite mode
            s = poseidon_coop::permute(s, r);
        }
    }
    if (r < 4) digests[i * 4 + r] = gl::canon(s);
}

// leaf sponge on four lanes per leaf (poseidon_coop4.h): lane q of a quad holds state words 3q .. 3q + 2.  For leaf sets
// between the two other forms: too many for sixteen lanes per leaf, too few to give every SIMD more than one lane-per-leaf wave.
__global__ __launch_bounds__(THREADS) void leaf_hash_coop4_kernel(const uint64_t* __restrict__ leaves, size_t n_leaves,
                                                                  int leaf_len, size_t row_stride, size_t col_stride,
                                                                  uint64_t* __restrict__ digests, int log_per_tree, size_t tree_stride) {
    const size_t i = ((size_t)blockIdx.x * THREADS + threadIdx.x) / poseidon_coop4::GROUP;
    const int q = threadIdx.x % poseidon_coop4::GROUP;
    if (i >= n_leaves) return;  // whole quads leave together
    const uint64_t* row = leaves + (i >> log_per_tree) * tree_stride + (i & (((size_t)1 << log_per_tree) - 1)) * row_stride;
    uint64_t s[poseidon_coop4::WPL] = {0, 0, 0};
    if (leaf_len <= 4) {  // hash_or_noop
#pragma unroll
        for (int a = 0; a < poseidon_coop4::WPL; ++a)
            if (3 * q + a < leaf_len) s[a] = row[(size_t)(3 * q + a) * col_stride];
    } else {
        for (int c0 = 0; c0 < leaf_len; c0 += poseidon::RATE) {
#pragma unroll
            for (int a = 0; a < poseidon_coop4::WPL; ++a) {  // overwrite mode: the rate words 0..7 take the next chunk
                const int w = 3 * q + a;
                if (w < poseidon::RATE && c0 + w < leaf_len) s[a] = row[(size_t)(c0 + w) * col_stride];
            }
            poseidon_coop4::permute(s, q);
        }
    }
#pragma unroll
    for (int a = 0; a < poseidon_coop4::WPL; ++a)
        if (3 * q + a < 4) digests[i * 4 + 3 * q + a] = gl::canon(s[a]);
}

// batched raw permutation (tests / microbench): states[i][12] in place
template <int MDS>
__global__ __launch_bounds__(THREADS) void permute_kernel(uint64_t* __restrict__ states, size_t n, int reps) {
    size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    uint64_t st[poseidon::WIDTH];
#pragma unroll
    for (int j = 0; j < poseidon::WIDTH; ++j) st[j] = states[i * 12 + j];
    for (int r = 0; r < reps; ++r) poseidon::permute<MDS>(st);
#pragma unroll
    for (int j = 0; j < poseidon::WIDTH; ++j) states[i * 12 + j] = gl::canon(st[j]);
}

// element-wise field arithmetic with the device routines (glp_field_op_dev)
__global__ __launch_bounds__(THREADS) void field_op_kernel(int op, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                                           size_t n, uint64_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = a[i], y = b[i];
    uint64_t r = 0;
    switch (op) {
        case 0: r = gl::add(x, y); break;
        case 1: r = gl::sub(x, y); break;
        case 2: r = gl::mul(x, y); break;
        case 3:
            switch (y) {
#define GLP_SHL_CASE(S) case S: r = gl::mul_2exp<S>(x); break;
                GLP_SHL_CASE(1) GLP_SHL_CASE(3) GLP_SHL_CASE(12) GLP_SHL_CASE(24) GLP_SHL_CASE(31) GLP_SHL_CASE(32)
                GLP_SHL_CASE(33) GLP_SHL_CASE(36) GLP_SHL_CASE(48) GLP_SHL_CASE(60) GLP_SHL_CASE(63) GLP_SHL_CASE(64)
                GLP_SHL_CASE(65) GLP_SHL_CASE(72) GLP_SHL_CASE(84) GLP_SHL_CASE(95)
#undef GLP_SHL_CASE
            }
            break;
        case 4: r = poseidon::mds_combine(x, y); break;
        // the canonical forms of the NTT passes (gl::bfly, gl_canon.h): stored RAW, a result >= p must show
        case 5:
        case 6: {
            uint64_t u = x, v = y;
            gl::bfly(u, v);
            out[i] = op == 5 ? u : v;
            return;
        }
#if defined(__HIP_DEVICE_COMPILE__)  // (namespace glc exists in the device pass only)
        case 7: out[i] = glc::mulc(x, y); return;
        default:
            switch (y) {
#define GLP_SHLC_CASE(S) case S: out[i] = glc::mul_2exp_c<S>(x); return;
                GLP_SHLC_CASE(1) GLP_SHLC_CASE(3) GLP_SHLC_CASE(12) GLP_SHLC_CASE(24) GLP_SHLC_CASE(31) GLP_SHLC_CASE(32)
                GLP_SHLC_CASE(33) GLP_SHLC_CASE(36) GLP_SHLC_CASE(48) GLP_SHLC_CASE(60) GLP_SHLC_CASE(63) GLP_SHLC_CASE(64)
                GLP_SHLC_CASE(65) GLP_SHLC_CASE(72) GLP_SHLC_CASE(84) GLP_SHLC_CASE(95)
#undef GLP_SHLC_CASE
            }
            break;
#endif
    }
    out[i] = gl::canon(r);
}

}  // namespace merkle
