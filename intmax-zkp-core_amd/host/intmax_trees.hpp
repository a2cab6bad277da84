// intmax_trees.hpp — C++ host-side mirror of the reference's NATIVE tree hashing, over the batched
// entry points of include/glprover.h.  Same names and error behaviour as the Rust functions they stand in for:
//
//   reference (Rust)                                                       here (C++, batched where the Rust is scalar)
//   merkle_tree::tree::get_merkle_proof_with_zero(leaves, index, depth,    intmax::get_merkle_proof_with_zero(ctx, …)
//       zero) / get_merkle_proof            [src/merkle_tree/tree.rs:49-109]   (asserts -> std::logic_error, as panics)
//   merkle_tree::tree::get_merkle_root(index, value, siblings) [:111-128]  intmax::get_merkle_root(ctx, indices, values, siblings)
//   PoseidonHash::two_to_one                                               intmax::two_to_one(ctx, lefts, rights)
//   PoseidonNodeHash::calc_node_hash(Node::Leaf(k, v))                     intmax::smt_leaf_hash(ctx, keys, values)
//       [src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:161-184]
//   SparseMerkleTree root after a sequence of set()s [tree.rs:255-586]     intmax::smt_root_of_set(ctx, keys, values)
//   SparseMerkleTree::find, batched [tree.rs:157-163,588-690]              intmax::smt_find_batch(ctx, keys, values, queries)
//   transaction::block_header::get_block_hash [block_header.rs:157-174]    intmax::get_block_hash(ctx, headers)
#pragma once
#include <cstring>

#include "plonky2_gpu.hpp"

namespace intmax {

using plonky2::Context;
using plonky2::F;
using plonky2::HashOut;

// device scratch owned for one call (the context pools the memory)
class DeviceWords {
  public:
    DeviceWords(Context& ctx, size_t words) : ctx_(ctx.raw()) {  // the block keeps the glp_ctx alive (library refcount)
        plonky2::check(glp_dev_alloc(ctx_, (words ? words : 1) * 8, &p_), "glp_dev_alloc");
    }
    DeviceWords(Context& ctx, const void* host, size_t words) : DeviceWords(ctx, words) {
        if (words) plonky2::check(glp_dev_upload(ctx.raw(), p_, host, words * 8), "glp_dev_upload");
    }
    ~DeviceWords() { glp_dev_free(ctx_, p_); }
    DeviceWords(const DeviceWords&) = delete;
    uint64_t* get() const { return static_cast<uint64_t*>(p_); }
    void download(void* host, size_t words) const {
        if (words) plonky2::check(glp_dev_download(ctx_, host, p_, words * 8), "glp_dev_download");
    }

  private:
    glp_ctx* ctx_;
    void* p_ = nullptr;
};

struct MerkleProof {  // src/merkle_tree/tree.rs:27-33
    size_t index;
    HashOut value;
    std::vector<HashOut> siblings;  // leaf level first
    HashOut root;
};

inline MerkleProof get_merkle_proof_with_zero(Context& ctx, const std::vector<HashOut>& leaves, size_t index, size_t depth,
                                              const HashOut& zero) {
    DeviceWords dev(ctx, leaves.data(), leaves.size() * 4);
    MerkleProof p;
    p.index = index;
    p.siblings.resize(glp_merkle_proof_n_siblings(depth));
    // the reference panics (assert!) on a bad index / too many leaves
    plonky2::check(glp_merkle_proof_with_zero_dev(ctx.raw(), leaves.empty() ? nullptr : dev.get(), leaves.size(), index, depth,
                                                  zero.elements, &p.siblings[0].elements[0], p.root.elements),
                   "get_merkle_proof_with_zero", true);
    p.value = leaves.empty() ? zero : leaves[index];
    return p;
}

inline MerkleProof get_merkle_proof(Context& ctx, const std::vector<HashOut>& leaves, size_t index, size_t depth) {
    return get_merkle_proof_with_zero(ctx, leaves, index, depth, HashOut{{0, 0, 0, 0}});
}

// n proofs of one depth: siblings[i * depth + l]
inline std::vector<HashOut> get_merkle_root(Context& ctx, const std::vector<uint64_t>& indices, const std::vector<HashOut>& values,
                                            const std::vector<HashOut>& siblings, size_t depth) {
    const size_t n = indices.size();
    if (values.size() != n || siblings.size() != n * depth) throw std::logic_error("get_merkle_root: shape mismatch");
    std::vector<HashOut> out(n);
    if (!n) return out;
    DeviceWords i_dev(ctx, indices.data(), n), v_dev(ctx, values.data(), n * 4), s_dev(ctx, siblings.data(), n * depth * 4),
        o_dev(ctx, n * 4);
    plonky2::check(glp_merkle_root_batch_dev(ctx.raw(), i_dev.get(), v_dev.get(), s_dev.get(), n, depth, o_dev.get()),
                   "get_merkle_root");
    o_dev.download(out.data(), n * 4);
    return out;
}

inline std::vector<HashOut> two_to_one(Context& ctx, const std::vector<HashOut>& left, const std::vector<HashOut>& right) {
    const size_t n = left.size();
    if (right.size() != n) throw std::logic_error("two_to_one: shape mismatch");
    std::vector<HashOut> out(n);
    if (!n) return out;
    DeviceWords l(ctx, left.data(), n * 4), r(ctx, right.data(), n * 4), o(ctx, n * 4);
    plonky2::check(glp_two_to_one_batch_dev(ctx.raw(), l.get(), r.get(), n, o.get()), "two_to_one");
    o.download(out.data(), n * 4);
    return out;
}

inline std::vector<HashOut> smt_leaf_hash(Context& ctx, const std::vector<HashOut>& keys, const std::vector<HashOut>& values) {
    const size_t n = keys.size();
    if (values.size() != n) throw std::logic_error("smt_leaf_hash: shape mismatch");
    std::vector<HashOut> out(n);
    if (!n) return out;
    DeviceWords k(ctx, keys.data(), n * 4), v(ctx, values.data(), n * 4), o(ctx, n * 4);
    plonky2::check(glp_smt_leaf_hash_batch_dev(ctx.raw(), k.get(), v.get(), n, o.get()), "calc_node_hash(Leaf)");
    o.download(out.data(), n * 4);
    return out;
}

inline HashOut smt_root_of_set(Context& ctx, const std::vector<HashOut>& keys, const std::vector<HashOut>& values) {
    if (values.size() != keys.size()) throw std::logic_error("smt_root_of_set: shape mismatch");
    HashOut root{};
    plonky2::check(glp_smt_root_of_set(ctx.raw(), keys.empty() ? nullptr : &keys[0].elements[0],
                                       values.empty() ? nullptr : &values[0].elements[0], keys.size(), root.elements),
                   "smt_root_of_set");
    return root;
}

// SparseMerkleTree::find for a batch of keys (src/sparse_merkle_tree/tree.rs:157-163,588-690): one
// SparseMerkleInclusionProof per query, fields as find_rec fills them
struct SparseMerkleInclusionProof {
    HashOut root{}, key{}, value{}, not_found_key{}, not_found_value{};
    bool found = false, is_old0 = false;
    std::vector<HashOut> siblings;  // [0] just under the root
};

inline std::vector<SparseMerkleInclusionProof> smt_find_batch(Context& ctx, const std::vector<HashOut>& keys,
                                                              const std::vector<HashOut>& values,
                                                              const std::vector<HashOut>& queries) {
    if (values.size() != keys.size()) throw std::logic_error("smt_find_batch: shape mismatch");
    const size_t m = queries.size();
    std::vector<uint32_t> ns(m), fl(m);
    std::vector<HashOut> val(m), nfk(m), nfv(m);
    HashOut root{};
    const uint64_t* kp = keys.empty() ? nullptr : &keys[0].elements[0];
    const uint64_t* vp = values.empty() ? nullptr : &values[0].elements[0];
    const uint64_t* qp = m ? &queries[0].elements[0] : nullptr;
    auto call = [&](size_t max_sib, uint64_t* sib) {
        plonky2::check(glp_smt_proofs_of_set(ctx.raw(), kp, vp, keys.size(), qp, m, max_sib, sib, ns.data(), fl.data(),
                                             m ? &val[0].elements[0] : nullptr, m ? &nfk[0].elements[0] : nullptr,
                                             m ? &nfv[0].elements[0] : nullptr, root.elements),
                       "smt_find_batch");
    };
    call(0, nullptr);  // lengths first: the sibling array is sized to the longest proof
    size_t max_sib = 1;
    for (uint32_t x : ns) max_sib = x > max_sib ? x : max_sib;
    std::vector<uint64_t> sib(m * max_sib * 4);
    if (m) call(max_sib, sib.data());
    std::vector<SparseMerkleInclusionProof> out(m);
    for (size_t i = 0; i < m; ++i) {
        out[i].root = root;
        out[i].key = queries[i];
        out[i].value = val[i];
        out[i].not_found_key = nfk[i];
        out[i].not_found_value = nfv[i];
        out[i].found = fl[i] & GLP_SMT_FOUND;
        out[i].is_old0 = fl[i] & GLP_SMT_IS_OLD0;
        out[i].siblings.resize(ns[i]);
        if (ns[i]) std::memcpy(&out[i].siblings[0].elements[0], &sib[i * max_sib * 4], (size_t)ns[i] * 32);
    }
    return out;
}

struct BlockHeader {  // src/transaction/block_header.rs:17-33; laid out as the 29 words glp_block_hash_batch_dev reads
    uint64_t block_number;  // u32 in the reference
    HashOut prev_block_hash, block_headers_digest, transactions_digest, deposit_digest, proposed_world_state_digest,
        approved_world_state_digest, latest_account_digest;
};
static_assert(sizeof(BlockHeader) == 29 * 8, "BlockHeader must be 29 packed words");

inline std::vector<HashOut> get_block_hash(Context& ctx, const std::vector<BlockHeader>& headers) {
    const size_t n = headers.size();
    std::vector<HashOut> out(n);
    if (!n) return out;
    DeviceWords h(ctx, headers.data(), n * 29), o(ctx, n * 4);
    plonky2::check(glp_block_hash_batch_dev(ctx.raw(), h.get(), n, o.get()), "get_block_hash");
    o.download(out.data(), n * 4);
    return out;
}

}  // namespace intmax
