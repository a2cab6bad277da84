// tree_kernels.h — the reference's NATIVE tree hashing as batched gfx950 kernels.
//
// These are the Poseidon calls the reference makes on the host, outside any circuit, to produce witnesses:
//   PoseidonHash::{two_to_one, hash_pad}                reference src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:161-184
//   SMT path climbs (update / insert / remove walks)    reference src/sparse_merkle_tree/tree.rs:208-225,318-345
//   get_merkle_root / get_merkle_proof_with_zero        reference src/merkle_tree/tree.rs:49-128
//   get_block_hash                                      reference src/transaction/block_header.rs:157-174
// One ITEM (one hash, one path, one header) per lane — or per 16-lane group when the batch is too small to fill the
// chip (the cooperative permutation of poseidon_coop.h: ~4x lower latency, ~3.6x lower throughput).  Each algorithm is
// written once against the two hashers below.
#pragma once
#include <hip/hip_runtime.h>

#include "poseidon.h"
#include "poseidon_coop.h"
#include "smt_replay.h"

namespace trees {

constexpr int THREADS = 256;

// ---- one Poseidon state per lane ----
struct LaneHasher {
    static constexpr int W = 1;  // lanes per item
    uint64_t st[poseidon::WIDTH];
    __device__ __forceinline__ LaneHasher() {
#pragma unroll
        for (int j = 0; j < poseidon::WIDTH; ++j) st[j] = 0;
    }
    // overwrite-mode absorb: rate words [0, count) <- f(j)
    template <class F>
    __device__ __forceinline__ void load(int count, F f) {
#pragma unroll
        for (int j = 0; j < poseidon::RATE; ++j)
            if (j < count) st[j] = f(j);
    }
    __device__ __forceinline__ void permute() { poseidon::permute<poseidon::MDS_INT>(st); }
    // state <- two_to_one input with the current digest on the right (bit = 1) or left (bit = 0) of `sib`, then permute
    template <class F>
    __device__ __forceinline__ void chain(int bit, F sib) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t h = st[j], s = sib(j);
            st[j] = bit ? s : h;
            st[4 + j] = bit ? h : s;
        }
#pragma unroll
        for (int j = 8; j < poseidon::WIDTH; ++j) st[j] = 0;
        permute();
    }
    __device__ __forceinline__ void digest(uint64_t (&h)[4]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = st[j];
    }
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int j = 0; j < poseidon::WIDTH; ++j) st[j] = 0;
    }
    __device__ __forceinline__ void store(uint64_t* out) const {
        ulonglong2* o = reinterpret_cast<ulonglong2*>(out);
        o[0] = make_ulonglong2(gl::canon(st[0]), gl::canon(st[1]));
        o[1] = make_ulonglong2(gl::canon(st[2]), gl::canon(st[3]));
    }
};

// ---- one Poseidon state per 16-lane group: lane r holds state word r ----
struct CoopHasher {
    static constexpr int W = poseidon_coop::GROUP;
    uint64_t s = 0;
    int r;
    __device__ __forceinline__ CoopHasher() : r(threadIdx.x % poseidon_coop::GROUP) {}
    template <class F>
    __device__ __forceinline__ void load(int count, F f) {
        if (r < poseidon::RATE && r < count) s = f(r);
    }
    __device__ __forceinline__ void permute() { s = poseidon_coop::permute(s, r); }
    template <class F>
    __device__ __forceinline__ void chain(int bit, F sib) {
        const uint64_t h = __shfl(s, r & 3, poseidon_coop::GROUP);  // digest word (r mod 4), every lane
        const int part = r >> 2;                                   // 0: left digest, 1: right digest, 2+: capacity
        uint64_t v = 0;
        if (part == (bit ? 1 : 0)) v = h;
        if (part == (bit ? 0 : 1)) v = sib(r & 3);
        s = v;
        permute();
    }
    __device__ __forceinline__ void digest(uint64_t (&h)[4]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = __shfl(s, j, poseidon_coop::GROUP);
    }
    __device__ __forceinline__ void reset() { s = 0; }
    __device__ __forceinline__ void store(uint64_t* out) const {
        if (r < 4) out[r] = gl::canon(s);
    }
};

// item index of this lane (whole groups leave together when out of range)
template <class H>
__device__ __forceinline__ size_t item_index() {
    return ((size_t)blockIdx.x * THREADS + threadIdx.x) / H::W;
}

enum { MODE_NO_PAD = 0, MODE_PAD = 1, MODE_OR_NOOP = 2 };

// out[i] = hash_no_pad / hash_pad / hash_or_noop of in[i * item_stride .. + len)
// hash_pad (plonky2 hash_n_to_m_with_pad as called by the reference, goldilocks_poseidon/mod.rs:170): the input, a 1, zeros
// until one short of a multiple of 12, a final 1.
template <class H>
__global__ __launch_bounds__(THREADS) void hash_batch_kernel(const uint64_t* __restrict__ in, size_t n, int len,
                                                             size_t item_stride, int mode, uint64_t* __restrict__ out) {
    const size_t i = item_index<H>();
    if (i >= n) return;
    const uint64_t* x = in + i * item_stride;
    H h;
    int total = len;
    if (mode == MODE_PAD) total = (len + 2 + 11) / 12 * 12;
    if (mode == MODE_OR_NOOP && len <= 4) {
        h.load(len, [&](int j) { return x[j]; });
    } else {
        for (int c0 = 0; c0 < total; c0 += poseidon::RATE) {
            h.load(total - c0, [&](int j) -> uint64_t {
                const int p = c0 + j;
                if (p < len) return x[p];
                return (p == len || p == total - 1) ? 1 : 0;  // only reached in MODE_PAD
            });
            h.permute();
        }
    }
    h.store(out + i * 4);
}

// out[i] = two_to_one(left[i], right[i])
template <class H>
__global__ __launch_bounds__(THREADS) void two_to_one_batch_kernel(const uint64_t* __restrict__ left,
                                                                   const uint64_t* __restrict__ right, size_t n,
                                                                   uint64_t* __restrict__ out) {
    const size_t i = item_index<H>();
    if (i >= n) return;
    H h;
    h.load(8, [&](int j) { return j < 4 ? left[i * 4 + j] : right[i * 4 + j - 4]; });
    h.permute();
    h.store(out + i * 4);
}

// SMT Node::Leaf hash: hash_pad(key ++ value ++ [1]) = sponge over [k0..k3 v0..v3 | 1 1 0 1]
template <class H>
__global__ __launch_bounds__(THREADS) void smt_leaf_hash_kernel(const uint64_t* __restrict__ keys,
                                                                const uint64_t* __restrict__ values, size_t n,
                                                                int keys_bit_reversed, uint64_t* __restrict__ out) {
    const size_t i = item_index<H>();
    if (i >= n) return;
    H h;
    h.load(8, [&](int j) -> uint64_t {
        if (j >= 4) return values[i * 4 + j - 4];
        const uint64_t k = keys[i * 4 + j];
        return keys_bit_reversed ? __brevll(k) : k;
    });
    h.permute();
    h.load(4, [&](int j) -> uint64_t { return j == 2 ? 0 : 1; });
    h.permute();
    h.store(out + i * 4);
}

// get_merkle_root: climb `depth` siblings (leaf level first), index bit l = 1 puts the running hash on the right
template <class H>
__global__ __launch_bounds__(THREADS) void merkle_root_batch_kernel(const uint64_t* __restrict__ index,
                                                                    const uint64_t* __restrict__ value,
                                                                    const uint64_t* __restrict__ siblings, size_t n,
                                                                    int depth, uint64_t* __restrict__ out) {
    const size_t i = item_index<H>();
    if (i >= n) return;
    H h;
    h.load(4, [&](int j) { return value[i * 4 + j]; });
    const uint64_t idx = index[i];
    const uint64_t* sib = siblings + i * (size_t)depth * 4;
    for (int l = 0; l < depth; ++l) {
        const int bit = l < 64 ? (int)((idx >> l) & 1) : 0;
        h.chain(bit, [&](int j) { return sib[l * 4 + j]; });
    }
    h.store(out + i * 4);
}

// SMT path climb: siblings[0] sits just under the root; key bit `lvl` (LE limbs, LSB first) = 1 puts the running hash on
// the right.  n_siblings == nullptr: every item has max_siblings.
template <class H>
__global__ __launch_bounds__(THREADS) void smt_root_batch_kernel(const uint64_t* __restrict__ keys,
                                                                 const uint64_t* __restrict__ start,
                                                                 const uint64_t* __restrict__ siblings,
                                                                 const uint32_t* __restrict__ n_siblings, size_t n,
                                                                 int max_siblings, uint64_t* __restrict__ out) {
    const size_t i = item_index<H>();
    if (i >= n) return;
    H h;
    h.load(4, [&](int j) { return start[i * 4 + j]; });
    int ns = n_siblings ? (int)n_siblings[i] : max_siblings;
    ns = ns > max_siblings ? max_siblings : ns;
    const uint64_t* sib = siblings + i * (size_t)max_siblings * 4;
    const uint64_t* key = keys + i * 4;
    for (int lvl = ns - 1; lvl >= 0; --lvl) {
        const int bit = (int)((key[lvl >> 6] >> (lvl & 63)) & 1);
        h.chain(bit, [&](int j) { return sib[lvl * 4 + j]; });
    }
    h.store(out + i * 4);
}

// get_block_hash over headers of 29 words: block_number, then prev_block_hash, block_headers_digest,
// transactions_digest, deposit_digest, proposed_world_state_digest, approved_world_state_digest, latest_account_digest
template <class H>
__global__ __launch_bounds__(THREADS) void block_hash_batch_kernel(const uint64_t* __restrict__ headers, size_t n,
                                                                   uint64_t* __restrict__ out) {
    const size_t i = item_index<H>();
    if (i >= n) return;
    const uint64_t* hd = headers + i * 29;
    const uint64_t *block_headers = hd + 5, *transactions = hd + 9, *deposit = hd + 13, *proposed = hd + 17,
                   *approved = hd + 21, *latest_account = hd + 25;
    H h;
    uint64_t b[4], c[4];
    h.load(8, [&](int j) { return j < 4 ? deposit[j] : transactions[j - 4]; });
    h.permute();
    h.digest(b);  // b = two_to_one(deposit, transactions)
    h.reset();
    h.load(8, [&](int j) -> uint64_t { return j == 0 ? (hd[0] & 0xFFFFFFFFu) : j < 4 ? 0 : latest_account[j - 4]; });
    h.permute();                                      // a = two_to_one([number, 0, 0, 0], latest_account)
    h.chain(0, [&](int j) { return b[j]; });          // c = two_to_one(a, b)
    h.digest(c);
    h.reset();
    h.load(8, [&](int j) { return j < 4 ? proposed[j] : approved[j - 4]; });
    h.permute();                                      // d = two_to_one(proposed, approved)
    h.chain(1, [&](int j) { return c[j]; });          // e = two_to_one(c, d)
    h.chain(1, [&](int j) { return block_headers[j]; });  // two_to_one(block_headers_digest, e)
    h.store(out + i * 4);
}

// ---- get_merkle_proof_with_zero: the part above / beside the dense subtree ----
// One wave.  Lanes 0-15 walk the zero-hash chain Z_0 = zero, Z_{l+1} = two_to_one(Z_l, Z_l); lanes 16-31 fold the dense
// subtree's root with Z_l for l >= log_m — both in lock step, so the whole thing costs n_sib cooperative permutations.
// digests: level 0 = the m = 2^log_m padded leaves, level l follows level l-1 (merkle_build layout, cap height 0).
__global__ __launch_bounds__(64) void merkle_proof_finish_kernel(const uint64_t* __restrict__ digests, int log_m, size_t index,
                                                                 int n_sib, uint64_t z0, uint64_t z1, uint64_t z2, uint64_t z3,
                                                                 uint64_t* __restrict__ siblings_out,
                                                                 uint64_t* __restrict__ root_out) {
    const int lane = threadIdx.x, r = lane % poseidon_coop::GROUP, grp = lane / poseidon_coop::GROUP;
    const uint64_t zero[4] = {z0, z1, z2, z3};
    // dense-part siblings
    {
        size_t off = 0, cnt = (size_t)1 << log_m;
        for (int l = 0; l < log_m && l < n_sib; ++l) {
            if (lane < 4) siblings_out[l * 4 + lane] = digests[off + ((index >> l) ^ 1) * 4 + lane];
            off += cnt * 4;
            cnt >>= 1;
        }
    }
    size_t root_off = 0;
    for (int l = 0; l < log_m; ++l) root_off += ((size_t)1 << (log_m - l)) * 4;
    uint64_t s = 0;  // group 0: Z_l in words 0..3; group 1: running root in words 0..3
    if (r < 4) s = grp == 0 ? zero[r] : digests[root_off + r];
    for (int l = 0; l < n_sib; ++l) {
        const uint64_t zl = __shfl(s, r & 3);  // Z_l word (r mod 4), read from group 0 by every lane
        if (l >= log_m && lane < 4) siblings_out[l * 4 + lane] = gl::canon(zl);
        const uint64_t own = __shfl(s, r & 3, poseidon_coop::GROUP);
        uint64_t v = 0;
        if (grp == 0) {
            v = r < 8 ? zl : 0;  // (Z_l, Z_l, 0)
        } else {
            if (r < 4) v = own;  // (root, Z_l, 0)
            else if (r < 8) v = zl;
        }
        const uint64_t next = poseidon_coop::permute(v, r);
        if (grp == 0 || (grp == 1 && l >= log_m)) s = next;
    }
    const uint64_t root_w = __shfl(s, poseidon_coop::GROUP + (lane & 3));
    if (lane < 4) root_out[lane] = gl::canon(root_w);
}

// padded dense leaves: out[i] = i < n_leaves ? leaves[i] : zero
__global__ __launch_bounds__(THREADS) void pad_leaves_kernel(const uint64_t* __restrict__ leaves, size_t n_leaves, size_t m,
                                                             uint64_t z0, uint64_t z1, uint64_t z2, uint64_t z3,
                                                             uint64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= m) return;
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + i * 4);
    if (i < n_leaves) {
        const ulonglong2* s = reinterpret_cast<const ulonglong2*>(leaves + i * 4);
        o[0] = s[0];
        o[1] = s[1];
    } else {
        o[0] = make_ulonglong2(z0, z1);
        o[1] = make_ulonglong2(z2, z3);
    }
}

// ---- bulk root of a key SET (the reference's compressed SMT is canonical: tree.rs:255-350,383-470) ----
// Input: n >= 2 entries sorted by key bit string (bit 0 first); rk = keys with every limb bit-reversed, so that the
// order is the lexicographic order of (rk[0], rk[1], rk[2], rk[3]) and a common prefix length is a count of leading
// zeros.  The branching nodes of the key trie are the n-1 internal nodes of a binary radix tree (Karras 2012); the
// reference's tree additionally has a unary node two_to_one(h, 0) / two_to_one(0, h) at every depth between two
// branching nodes (a lone LEAF, by contrast, sits directly under its parent branching node).
struct RadixNode {
    uint64_t child_hash[2][4];
    int parent;     // internal node index, -1 for the root
    int side;       // which child of the parent this node is
    int depth;      // length of the common prefix of the keys below (= depth of this branching node)
    int lo;         // first key of the node's range (any key below it gives the bits of the unary chain above it)
    unsigned arrivals;
    int cur;        // depth of the running hash while it climbs the unary chain above the node (-1: not hashed yet)
    uint64_t running[4];
    int child[2];   // >= 0: internal node index; < 0: ~(entry index) of a leaf  (the top-down walk of smt_find_kernel)
};

__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ rk, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const uint64_t x = rk[(size_t)i * 4 + l] ^ rk[(size_t)j * 4 + l];
        if (x) return 64 * l + __clzll(x);
    }
    return 256;
}

// one lane per internal node: find its key range, split, children; record parent links
__global__ __launch_bounds__(THREADS) void radix_build_kernel(const uint64_t* __restrict__ rk, int n, RadixNode* nodes,
                                                              int* __restrict__ leaf_parent, int* __restrict__ leaf_side) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= n - 1) return;
    const int d = common_prefix(rk, n, i, i + 1) - common_prefix(rk, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(rk, n, i, i - d);
    int lmax = 2;
    while (common_prefix(rk, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (common_prefix(rk, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(rk, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2; ; t = (t + 1) / 2) {
        if (common_prefix(rk, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    nodes[i].depth = dnode;
    nodes[i].lo = lo;
    nodes[i].arrivals = 0;
    nodes[i].cur = -1;
    if (i == 0) {
        nodes[0].parent = -1;
        nodes[0].side = 0;
    }
    if (lo == gamma) {
        leaf_parent[gamma] = i;
        leaf_side[gamma] = 0;
        nodes[i].child[0] = ~gamma;
    } else {
        nodes[gamma].parent = i;
        nodes[gamma].side = 0;
        nodes[i].child[0] = gamma;
    }
    if (hi == gamma + 1) {
        leaf_parent[gamma + 1] = i;
        leaf_side[gamma + 1] = 1;
        nodes[i].child[1] = ~(gamma + 1);
    } else {
        nodes[gamma + 1].parent = i;
        nodes[gamma + 1].side = 1;
        nodes[i].child[1] = gamma + 1;
    }
}

// Level-synchronous hashing of the radix tree: a node is READY once both children have delivered their hashes.  Every
// round performs exactly ONE permutation per queued node — either the two_to_one at the branching node or one step of
// the unary chain above it (H(h, 0) / H(0, h) at the depths between it and its parent) — so the lanes of a wave never
// wait for a neighbour's longer chain; a node with chain steps left re-queues itself, a finished one delivers to its
// parent, and the second delivery to a parent queues it.  Rounds = height of the reference's tree (<= 256; about
// log2(n) + 20 for spread keys); one lane per node, or 16 lanes when the round is small.
__device__ __forceinline__ void radix_deliver(RadixNode* nodes, int parent, int side, const uint64_t (&d)[4], int* queue_out,
                                              int* count_out) {
    RadixNode& nd = nodes[parent];
#pragma unroll
    for (int j = 0; j < 4; ++j) nd.child_hash[side][j] = d[j];
    const unsigned prev = __hip_atomic_fetch_add(&nd.arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == 1) queue_out[__hip_atomic_fetch_add(count_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = parent;
}

// round 0: the leaves deliver their hashes to their parents
__global__ __launch_bounds__(THREADS) void radix_deliver_leaves_kernel(const uint64_t* __restrict__ leaf_hash, int n, RadixNode* nodes,
                                                                       const int* __restrict__ leaf_parent,
                                                                       const int* __restrict__ leaf_side, int* queue_out,
                                                                       int* count_out) {
    const int i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    uint64_t d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = leaf_hash[(size_t)i * 4 + j];
    radix_deliver(nodes, leaf_parent[i], leaf_side[i], d, queue_out, count_out);
}

template <class H>
__global__ __launch_bounds__(THREADS) void radix_round_kernel(const uint64_t* __restrict__ rk, RadixNode* nodes,
                                                              const int* __restrict__ queue_in, const int* __restrict__ n_in_dev,
                                                              int* queue_out, int* count_out, uint64_t* __restrict__ root_out) {
    // the round's width is read where the previous round counted it: the host launches several rounds on an upper bound (the
    // widths never grow: a node comes back itself or two children make one parent) without waiting for each count
    const int n_in = *n_in_dev;
    const size_t item = item_index<H>();
    if (item >= (size_t)n_in) return;
    const int node = queue_in[item];
    RadixNode& nd = nodes[node];
    const int parent = nd.parent, side = nd.side;
    const int pdepth = parent >= 0 ? nodes[parent].depth : -1;
    const bool leader = H::W == 1 || threadIdx.x % H::W == 0;
    int cur = nd.cur;
    H h;
    if (cur < 0) {  // two_to_one(left, right): the hash AT the branching node
        h.load(8, [&](int j) { return nd.child_hash[j >> 2][j & 3]; });
        h.permute();
        cur = nd.depth;
    } else {  // one unary node: the subtree hangs on the side given by key bit cur - 1
        h.load(4, [&](int j) { return nd.running[j]; });
        const int t = cur - 1;
        const uint64_t limb = __brevll(rk[(size_t)nd.lo * 4 + (t >> 6)]);
        h.chain((int)((limb >> (t & 63)) & 1), [](int) -> uint64_t { return 0; });
        cur = t;
    }
    uint64_t d[4];
    h.digest(d);
    if (cur - 1 > pdepth) {  // more unary nodes below the parent: come back next round
        if (leader) {
#pragma unroll
            for (int j = 0; j < 4; ++j) nd.running[j] = d[j];
            nd.cur = cur;
            queue_out[__hip_atomic_fetch_add(count_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = node;
        }
    } else if (parent < 0) {
        h.store(root_out);
    } else if (leader) {
        radix_deliver(nodes, parent, side, d, queue_out, count_out);
    }
}

// ---- `find` for a batch of query keys against the hashed radix tree (reference src/sparse_merkle_tree/tree.rs:588-690) ----
// One lane per query walks the tree top-down the way find_rec does, in radix-tree steps: below a branching node of depth pd the
// reference's tree has unary nodes at depths pd + 1 .. D - 1 (D = depth of the next branching node X on the way) and X at D.
//   * the query agrees with X's keys on all of those bits: the unary nodes contribute 0^4 siblings, X contributes the hash of the
//     child not taken (what that child DELIVERED to X: its own hash climbed up its unary chain to depth D + 1), and the walk goes
//     on in the child taken — a leaf ends it (found, or "another leaf": not_found_key / value);
//   * the query leaves the chain at depth c < D, into the empty side of a unary node: the sibling at c is the chain's running hash
//     at depth c + 1 — recomputed here from X's two children (the build keeps only what X delivered to its parent) — and the
//     walk ends in an empty subtree (is_old0).
// flags: bit 0 found, bit 1 is_old0.  siblings: [n_q][max_sib][4], row 0 just under the root; a proof longer than max_sib
// sets *overflow (its siblings beyond are not written).  sib_out may be null (lengths and answers only).
__global__ __launch_bounds__(THREADS) void smt_find_kernel(const uint64_t* __restrict__ rk, const uint64_t* __restrict__ values,
                                                           const RadixNode* __restrict__ nodes, int n_entries, const uint64_t* __restrict__ queries, size_t n_q,
                                                           size_t max_sib, uint64_t* __restrict__ sib_out,
                                                           uint32_t* __restrict__ n_sib_out, uint32_t* __restrict__ flags_out,
                                                           uint64_t* __restrict__ value_out, uint64_t* __restrict__ nf_key_out,
                                                           uint64_t* __restrict__ nf_value_out, unsigned* overflow,
                                                           // optional (glp_smt_apply_updates): per sibling 1 + the entry index when it is a
                                                           // lone LEAF, -1 when it is an internal node (0 = the zero sibling, as
                                                           // pre-filled), [n_q][max_sib]; per query the entry the walk ended on (-1: none)
                                                           int* __restrict__ sib_entry_out = nullptr, int* __restrict__ end_entry_out = nullptr) {
    const size_t qi = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (qi >= n_q) return;
    int end_entry = -1;
    uint64_t qr[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) qr[l] = __brevll(gl::canon(queries[qi * 4 + l]));
    uint64_t* sib = sib_out ? sib_out + qi * max_sib * 4 : nullptr;
    uint64_t val[4] = {0, 0, 0, 0}, nfk[4] = {0, 0, 0, 0}, nfv[4] = {0, 0, 0, 0};
    uint32_t flags = 0;
    int level = 0;  // siblings written so far = depth reached
    // the host zeroes the sibling array before the launch: only non-zero siblings are written (the 0^4 siblings of unary nodes
    // and the rows behind a proof's end are already there); a proof longer than max_sib is flagged once, at the end
    auto put = [&](int d, uint64_t a, uint64_t b, uint64_t c2, uint64_t e) {
        if (!sib || (size_t)d >= max_sib) return;
        sib[d * 4 + 0] = a, sib[d * 4 + 1] = b, sib[d * 4 + 2] = c2, sib[d * 4 + 3] = e;
    };
    auto prefix_with = [&](int entry) {  // common prefix (in key bits) of the query and entry `entry`; 256 = the same key
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const uint64_t x = qr[l] ^ rk[(size_t)entry * 4 + l];
            if (x) return 64 * l + __clzll(x);
        }
        return 256;
    };
    auto end_on_leaf = [&](int entry) {
        end_entry = entry;
        if (prefix_with(entry) == 256) {
            flags = 1;
#pragma unroll
            for (int l = 0; l < 4; ++l) val[l] = values[(size_t)entry * 4 + l];
        } else {
#pragma unroll
            for (int l = 0; l < 4; ++l) nfk[l] = __brevll(rk[(size_t)entry * 4 + l]), nfv[l] = values[(size_t)entry * 4 + l];
        }
    };
    if (n_entries == 0) {
        flags = 2;
    } else if (n_entries == 1) {
        end_on_leaf(0);
    } else {
        int node = 0;
        for (;;) {
            const RadixNode& nd = nodes[node];
            const int D = nd.depth;
            const int c = prefix_with(nd.lo);
            // (the depths between the parent branching node and min(c, D) are unary nodes passed on their populated side: 0^4 siblings)
            if (c < D) {  // into the empty side of the unary node at depth c
                LaneHasher h;
                h.load(8, [&](int j) { return nd.child_hash[j >> 2][j & 3]; });
                h.permute();  // the hash AT the branching node (depth D)
                for (int t = D - 1; t > c; --t) {
                    const uint64_t limb = rk[(size_t)nd.lo * 4 + (t >> 6)];
                    h.chain((int)((limb >> (63 - (t & 63))) & 1), [](int) -> uint64_t { return 0; });
                }
                put(c, gl::canon(h.st[0]), gl::canon(h.st[1]), gl::canon(h.st[2]), gl::canon(h.st[3]));
                if (sib_entry_out && (size_t)c < max_sib) sib_entry_out[qi * max_sib + c] = -1;  // the rest of the chain: >= 2 entries
                level = c + 1;
                flags = 2;
                break;
            }
            const int b = (int)((qr[D >> 6] >> (63 - (D & 63))) & 1);
            const uint64_t* o = nd.child_hash[1 - b];
            put(D, gl::canon(o[0]), gl::canon(o[1]), gl::canon(o[2]), gl::canon(o[3]));
            if (sib_entry_out && (size_t)D < max_sib) {
                const int oc = nd.child[1 - b];
                sib_entry_out[qi * max_sib + D] = oc < 0 ? 1 + ~oc : -1;
            }
            level = D + 1;
            const int ch = nd.child[b];
            if (ch < 0) {
                end_on_leaf(~ch);
                break;
            }
            node = ch;
        }
    }
    if (sib && (size_t)level > max_sib) atomicOr(overflow, 1u);
    n_sib_out[qi] = (uint32_t)level;
    flags_out[qi] = flags;
    if (end_entry_out) end_entry_out[qi] = end_entry;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        value_out[qi * 4 + l] = val[l];
        nf_key_out[qi * 4 + l] = nfk[l];
        nf_value_out[qi * 4 + l] = nfv[l];
    }
}

// ---- a tree that stays on the device (glp_smt_open / glp_smt_update, trees.hip): the entry set back in input form ----
// keys_out[j] = the key whose bit-reversed limbs are rk[j] (j < m), values_out[j] = v[j]
__global__ __launch_bounds__(THREADS) void smt_entries_out_kernel(const uint64_t* __restrict__ rk, const uint64_t* __restrict__ v, size_t m,
                                                                  uint64_t* __restrict__ keys_out, uint64_t* __restrict__ values_out) {
    const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= 4 * m) return;
    keys_out[i] = __brevll(rk[i]);
    values_out[i] = v[i];
}
// values[idx[i]] = new_values[i] (0^4 = the entry leaves the set at the next sort)
__global__ __launch_bounds__(THREADS) void smt_patch_values_kernel(uint64_t* __restrict__ values, const int* __restrict__ idx,
                                                                   const uint64_t* __restrict__ new_values, size_t n) {
    const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= 4 * n) return;
    values[(size_t)idx[i >> 2] * 4 + (i & 3)] = new_values[i];
}

// ---- sequential `set`s as a levelled batch of hash jobs (glp_smt_apply_updates, trees.hip) ----
// The host replays the k updates on a small pointer trie holding only the paths they touch, with value IDS in place of hashes
// (the reference's walks never look at a hash value except to compare it with 0^4); every new node is a job
//     table[id] = two_to_one(table[a], table[b])      or      table[id] = leaf_hash(key, value)
// whose level is one more than its inputs'.  A launch runs all jobs of one level — of ALL updates at once: the dependency chain
// is the height of the tree, not the number of updates.
// (struct SmtNodeJob {id, a, b}: smt_replay.h)
template <class H>
__global__ __launch_bounds__(THREADS) void smt_node_jobs_kernel(uint64_t* __restrict__ table, const SmtNodeJob* __restrict__ jobs, size_t n) {
    const size_t i = item_index<H>();
    if (i >= n) return;
    const SmtNodeJob j = jobs[i];
    const uint64_t* l = table + (size_t)j.a * 4;
    const uint64_t* r = table + (size_t)j.b * 4;
    H h;
    h.load(8, [&](int w) { return w < 4 ? l[w] : r[w - 4]; });
    h.permute();
    h.store(table + (size_t)j.id * 4);
}
// leaf jobs: kv[i] = key ++ value (8 words, any order of ids); table[ids[i]] = hash_pad(key ++ value ++ [1])
template <class H>
__global__ __launch_bounds__(THREADS) void smt_leaf_jobs_kernel(uint64_t* __restrict__ table, const uint64_t* __restrict__ kv,
                                                                const int* __restrict__ ids, size_t n) {
    const size_t i = item_index<H>();
    if (i >= n) return;
    const uint64_t* x = kv + i * 8;
    H h;
    h.load(8, [&](int w) { return x[w]; });
    h.permute();
    h.load(4, [](int w) -> uint64_t { return w == 2 ? 0 : 1; });  // words 8..11 of hash_pad's input: the leaf marker 1, pad 1, 0, 1
    h.permute();
    h.store(table + (size_t)ids[i] * 4);
}
// entries by index out of the sorted arrays: out[i] = key (limbs restored) ++ value ++ leaf hash, 12 words
__global__ __launch_bounds__(THREADS) void smt_gather_entries_kernel(const uint64_t* __restrict__ rk, const uint64_t* __restrict__ values,
                                                                     const uint64_t* __restrict__ leaf_hash, const int* __restrict__ idx,
                                                                     size_t n, uint64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t e = (size_t)idx[i];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        out[i * 12 + l] = __brevll(rk[e * 4 + l]);
        out[i * 12 + 4 + l] = values[e * 4 + l];
        out[i * 12 + 8 + l] = leaf_hash[e * 4 + l];
    }
}

}  // namespace trees
