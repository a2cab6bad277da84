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
//   SparseMerkleTree::set, k times in order [tree.rs:143-151,561-585]      intmax::smt_apply_updates(ctx, keys, values, upd_keys, upd_values)
//   roots of many small trees (layered trees, one call per layer)          intmax::smt_roots_of_sets(ctx, sets)
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

// the roots of many small independent trees in one call (a block's per-transaction trees, one call per layer of the
// LayeredLayered / user-asset trees: src/sparse_merkle_tree/layered_layered_tree.rs:124-160, src/transaction/tree/user_asset.rs:99-150)
inline std::vector<HashOut> smt_roots_of_sets(Context& ctx, const std::vector<std::vector<std::pair<HashOut, HashOut>>>& sets) {
    std::vector<HashOut> keys, values, roots(sets.size());
    std::vector<size_t> off(sets.size() + 1, 0);
    for (size_t s = 0; s < sets.size(); ++s) {
        for (const auto& kv : sets[s]) keys.push_back(kv.first), values.push_back(kv.second);
        off[s + 1] = keys.size();
    }
    plonky2::check(glp_smt_roots_of_sets(ctx.raw(), keys.empty() ? nullptr : &keys[0].elements[0],
                                         values.empty() ? nullptr : &values[0].elements[0], off.data(), sets.size(),
                                         roots.empty() ? nullptr : &roots[0].elements[0]),
                   "smt_roots_of_sets");
    return roots;
}

// SparseMerkleTree::set, k times in order, as one batch (src/sparse_merkle_tree/tree.rs:143-151 -> calc_process_proof :561-585):
// one SparseMerkleProcessProof (proof/process.rs:13-23) per update, fields filled as update / insert / remove / noop fill them
enum class ProcessMerkleProofRole : uint8_t { ProcessNoOp = 0, ProcessUpdate = 1, ProcessInsert = 2, ProcessDelete = 3 };
struct SparseMerkleProcessProof {
    HashOut old_root{}, old_key{}, old_value{}, new_root{}, new_key{}, new_value{};
    std::vector<HashOut> siblings;
    bool is_old0 = true;
    ProcessMerkleProofRole fnc = ProcessMerkleProofRole::ProcessNoOp;
};

namespace detail {
// the library's per-update outputs (glp_smt_apply_updates / glp_smt_update: same layouts) -> the reference's process proofs.
// raw(max_siblings, siblings, n_siblings, flags, old_value, other_key, other_value, last_key, last_value, roots) makes the call;
// lengths_first: a call with max_siblings = 0 and no sibling buffer only measures (true for the stateless entry point; a resident
// tree would APPLY the batch, so it is asked again with 256 siblings instead).
template <class Raw>
std::vector<SparseMerkleProcessProof> process_proofs(const char* what, const std::vector<HashOut>& upd_keys,
                                                     const std::vector<HashOut>& upd_values, std::vector<HashOut>* roots_out, bool lengths_first,
                                                     Raw raw) {
    if (upd_values.size() != upd_keys.size()) throw std::logic_error("smt updates: shape mismatch");
    const size_t k = upd_keys.size();
    std::vector<uint32_t> ns(k), fl(k);
    std::vector<HashOut> oldv(k), ok(k), ov(k), lk(k), lv(k), roots(k + 1);
    auto p = [](std::vector<HashOut>& v) { return v.empty() ? nullptr : &v[0].elements[0]; };
    auto call = [&](size_t max_sib, uint64_t* sib) {
        return raw(max_sib, sib, ns.data(), fl.data(), p(oldv), p(ok), p(ov), p(lk), p(lv), &roots[0].elements[0]);
    };
    size_t max_sib = 48;
    std::vector<uint64_t> sib(k * max_sib * 4);
    int32_t st = call(max_sib, sib.data());
    if (st == GLP_ERR_INVALID) {  // a proof longer than the guess (or bad input: the second call tells)
        if (lengths_first) {
            plonky2::check(call(0, nullptr), what);
            max_sib = 1;
            for (uint32_t x : ns) max_sib = x > max_sib ? x : max_sib;
        } else {
            max_sib = 256;
        }
        sib.assign(k * max_sib * 4, 0);
        st = call(max_sib, sib.data());
    }
    plonky2::check(st, what);
    auto is_zero = [](const uint64_t* w) { return !(w[0] | w[1] | w[2] | w[3]); };
    auto canon = [](HashOut h) {
        for (auto& w : h.elements) w %= 0xFFFFFFFF00000001ull;
        return h;
    };
    std::vector<SparseMerkleProcessProof> out(k);
    for (size_t i = 0; i < k; ++i) {
        SparseMerkleProcessProof& q = out[i];
        const HashOut key = canon(upd_keys[i]), value = canon(upd_values[i]);
        const bool found = fl[i] & GLP_SMT_FOUND, old0 = fl[i] & GLP_SMT_IS_OLD0, last_leaf = fl[i] & GLP_SMT_LAST_SIBLING_IS_LEAF;
        const uint64_t* s = &sib[i * max_sib * 4];
        size_t cnt = ns[i];
        q.old_root = roots[i];
        q.new_root = roots[i + 1];
        if (is_zero(value.elements) && !found) {  // noop, tree.rs:537-547
            q.old_key = q.new_key = key;
            q.new_root = roots[i];
        } else if (!is_zero(value.elements) && found) {  // update, tree.rs:242-252
            q.fnc = ProcessMerkleProofRole::ProcessUpdate;
            q.old_key = q.new_key = key;
            q.old_value = oldv[i];
            q.new_value = value;
            q.is_old0 = false;
        } else if (!is_zero(value.elements)) {  // insert, tree.rs:369-379: trailing 0^4 siblings popped
            q.fnc = ProcessMerkleProofRole::ProcessInsert;
            while (cnt && is_zero(s + 4 * (cnt - 1))) --cnt;
            q.old_key = ok[i];
            q.old_value = ov[i];
            q.new_key = key;
            q.new_value = value;
            q.is_old0 = old0;
        } else {  // remove, tree.rs:508-518 (the removed entry is reported as "old")
            q.fnc = ProcessMerkleProofRole::ProcessDelete;
            q.old_key = key;
            q.old_value = oldv[i];
            q.is_old0 = !last_leaf;
            q.new_key = last_leaf ? lk[i] : key;
            if (last_leaf) q.new_value = lv[i];
            bool mixed = cnt && !last_leaf;
            size_t first = cnt;  // the siblings from the first (deepest) level on which `mixed` holds, up to the root
            for (size_t level = 0; level < cnt; ++level) {
                const size_t lvl = cnt - 1 - level;
                const bool new_sibling_zero = (level == 0 && last_leaf) || is_zero(s + 4 * lvl);
                if (!new_sibling_zero) mixed = true;
                if (mixed) {
                    first = lvl + 1;
                    break;
                }
            }
            cnt = mixed ? first : 0;
        }
        if (q.fnc != ProcessMerkleProofRole::ProcessNoOp) {
            q.siblings.resize(cnt);
            if (cnt) std::memcpy(&q.siblings[0].elements[0], s, cnt * 32);
        }
    }
    if (roots_out) *roots_out = roots;
    return out;
}
}  // namespace detail

inline std::vector<SparseMerkleProcessProof> smt_apply_updates(Context& ctx, const std::vector<HashOut>& keys,
                                                               const std::vector<HashOut>& values,
                                                               const std::vector<HashOut>& upd_keys,
                                                               const std::vector<HashOut>& upd_values,
                                                               std::vector<HashOut>* roots_out = nullptr) {
    if (values.size() != keys.size()) throw std::logic_error("smt_apply_updates: shape mismatch");
    auto cp = [](const std::vector<HashOut>& v) { return v.empty() ? nullptr : &v[0].elements[0]; };
    return detail::process_proofs("smt_apply_updates", upd_keys, upd_values, roots_out, true,
                                  [&](size_t max_sib, uint64_t* sib, uint32_t* ns, uint32_t* fl, uint64_t* oldv, uint64_t* ok, uint64_t* ov,
                                      uint64_t* lk, uint64_t* lv, uint64_t* roots) {
                                      return glp_smt_apply_updates(ctx.raw(), cp(keys), cp(values), keys.size(), cp(upd_keys), cp(upd_values),
                                                                   upd_keys.size(), max_sib, sib, ns, fl, oldv, ok, ov, lk, lv, roots);
                                  });
}

// The reference's `SparseMerkleTree` as a node keeps it (src/sparse_merkle_tree/tree.rs:29-60): built once, resident on the GPU
// (glp_smt_open), then `find` for batches of keys and `set` for batches of updates, block after block.
class SparseMerkleTree {
public:
    SparseMerkleTree(Context& ctx, const std::vector<HashOut>& keys, const std::vector<HashOut>& values) {
        if (values.size() != keys.size()) throw std::logic_error("SparseMerkleTree: shape mismatch");
        plonky2::check(glp_smt_open(ctx.raw(), keys.empty() ? nullptr : &keys[0].elements[0], values.empty() ? nullptr : &values[0].elements[0],
                                    keys.size(), &h_), "glp_smt_open");
    }
    SparseMerkleTree(const SparseMerkleTree&) = delete;
    SparseMerkleTree& operator=(const SparseMerkleTree&) = delete;
    ~SparseMerkleTree() { glp_smt_close(h_); }
    size_t len() const {
        size_t n = 0;
        plonky2::check(glp_smt_len(h_, &n), "glp_smt_len");
        return n;
    }
    HashOut get_root() {
        HashOut r{};
        plonky2::check(glp_smt_root(h_, r.elements), "glp_smt_root");
        return r;
    }
    // `set(key_i, value_i)` in order; the tree holds the updated set afterwards (tree.rs:143-151 for every update)
    std::vector<SparseMerkleProcessProof> set_batch(const std::vector<HashOut>& upd_keys, const std::vector<HashOut>& upd_values,
                                                    std::vector<HashOut>* roots_out = nullptr) {
        auto cp = [](const std::vector<HashOut>& v) { return v.empty() ? nullptr : &v[0].elements[0]; };
        return detail::process_proofs("glp_smt_update", upd_keys, upd_values, roots_out, false,
                                      [&](size_t max_sib, uint64_t* sib, uint32_t* ns, uint32_t* fl, uint64_t* oldv, uint64_t* ok, uint64_t* ov,
                                          uint64_t* lk, uint64_t* lv, uint64_t* roots) {
                                          return glp_smt_update(h_, cp(upd_keys), cp(upd_values), upd_keys.size(), max_sib, sib, ns, fl, oldv, ok,
                                                                ov, lk, lv, roots);
                                      });
    }

private:
    glp_smt* h_ = nullptr;
};

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
