// plonky2_gpu.hpp — C++ host-side mirror of the plonky2 interface that libglprover.so replaces.
//
// The reference is compiled code (Rust) and its toolchain is absent from the build image, so the host side above the
// C ABI (include/glprover.h) is written in C++ with plonky2's names, argument meaning and error behaviour:
//
//   plonky2 (Rust)                                                 here (C++)
//   PolynomialBatch::from_values(values, rate_bits, blinding,      plonky2::PolynomialBatch::from_values(ctx, values,
//       cap_height, timing, fft_root_table)  [fri/oracle.rs]           rate_bits, blinding, cap_height)
//   PolynomialBatch::from_coeffs(...)                              PolynomialBatch::from_coeffs(...)
//   batch.get_lde_values(index, step)                              batch.get_lde_values(index, step)
//   batch.merkle_tree.cap / .prove(i)                              batch.merkle_tree_cap() / .prove(i)
//   Challenger::{observe_elements, observe_cap, get_challenge,     plonky2::Challenger (same method names)
//       get_n_challenges, get_extension_challenge, get_hash}
//   PolynomialBatch::prove_openings(instance, oracles,             plonky2::prove_openings(ctx, instance, oracles,
//       challenger, fri_params, timing)                                challenger, fri_params)
//   CircuitData::prove(inputs) after witness generation            plonky2::CircuitData::prove(wire_values, public_inputs)
//   (reference call sites: src/rollup/circuits/mod.rs:605,1247; src/zkdsa/circuits/mod.rs:326)
//
// Error behaviour: from_values / from_coeffs are infallible in plonky2 (they panic on misuse) -> std::logic_error;
// CircuitData::prove returns anyhow::Result -> std::runtime_error carrying glp_strerror().
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <array>
#include <utility>
#include <vector>

#include "../../include/glprover.h"

namespace plonky2 {

using F = uint64_t;                 // GoldilocksField(u64), possibly non-canonical on input
struct HashOut { F elements[4]; };
using MerkleCap = std::vector<HashOut>;
using Ext = std::pair<F, F>;        // QuadraticExtension<GoldilocksField>

inline void check(int32_t st, const char* what, bool infallible = false) {
    if (st == GLP_OK) return;
    std::string msg = std::string(what) + ": " + glp_strerror(st);
    if (infallible) throw std::logic_error(msg);
    throw std::runtime_error(msg);
}

// plonky2::hash::poseidon::PoseidonHash on the HOST (no context, no GPU: the library's own host permutation) — what the
// reference calls natively for one-off hashes (src/merkle_tree/tree.rs:70,79,90,123; src/zkdsa/circuits/mod.rs:85-100;
// hash_pad in PoseidonNodeHash::calc_node_hash, src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:161-184)
struct PoseidonHash {
    static HashOut hash_no_pad(const std::vector<F>& input) {
        HashOut h;
        glp_hash_no_pad(input.data(), input.size(), h.elements);
        return h;
    }
    static HashOut hash_pad(const std::vector<F>& input) {
        HashOut h;
        glp_hash_pad(input.data(), input.size(), h.elements);
        return h;
    }
    static HashOut hash_or_noop(const std::vector<F>& input) {
        HashOut h;
        glp_hash_or_noop(input.data(), input.size(), h.elements);
        return h;
    }
    static HashOut two_to_one(const HashOut& left, const HashOut& right) {
        HashOut h;
        glp_two_to_one(left.elements, right.elements, h.elements);
        return h;
    }
    static std::array<F, 12> permute(std::array<F, 12> state) {  // Poseidon::poseidon
        glp_poseidon_permute(state.data());
        return state;
    }
};

class Context {
  public:
    explicit Context(int device = 0) { check(glp_ctx_create(device, &c_), "glp_ctx_create"); }
    ~Context() { glp_ctx_destroy(c_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    glp_ctx* raw() const { return c_; }

  private:
    glp_ctx* c_ = nullptr;
};

class PolynomialBatch {
  public:
    // values: one Vec<F> per polynomial (Vec<PolynomialValues<F>>), all of length 2^degree_log
    static PolynomialBatch from_values(Context& ctx, const std::vector<std::vector<F>>& values, size_t rate_bits,
                                       bool blinding, size_t cap_height) {
        return make(ctx, values, rate_bits, blinding, cap_height, false);
    }
    static PolynomialBatch from_coeffs(Context& ctx, const std::vector<std::vector<F>>& polynomials, size_t rate_bits,
                                       bool blinding, size_t cap_height) {
        return make(ctx, polynomials, rate_bits, blinding, cap_height, true);
    }
    PolynomialBatch(PolynomialBatch&& o) noexcept : ctx_(o.ctx_), b_(o.b_), cap_(std::move(o.cap_)) { o.b_ = nullptr; }
    PolynomialBatch(const PolynomialBatch&) = delete;
    ~PolynomialBatch() { glp_batch_free(b_); }  // Drop

    size_t degree_log() const { return shape().log_n; }
    size_t rate_bits() const { return shape().rate_bits; }
    size_t num_polys() const { return shape().n_cols; }
    const MerkleCap& merkle_tree_cap() const { return cap_; }
    const glp_batch* raw() const { return b_; }

    // leaves[reverse_bits(index * step, degree_log + rate_bits)]
    std::vector<F> get_lde_values(size_t index, size_t step) const {
        auto s = shape();
        size_t bits = s.log_n + s.rate_bits, x = index * step, r = 0;
        for (size_t k = 0; k < bits; ++k) r |= ((x >> k) & 1) << (bits - 1 - k);
        // glp_batch_row hands back the WHOLE leaf (MerkleTree::get): a blinded batch carries GLP_SALT_SIZE salt words behind the
        // polynomial values, which get_lde_values slices off (`&leaves[index][..self.polynomials.len()]`)
        std::vector<F> row(s.n_cols + glp_batch_salt_cols(b_));
        check(glp_batch_row(ctx_->raw(), b_, r, row.data(), nullptr), "get_lde_values");
        row.resize(s.n_cols);
        return row;
    }
    // merkle_tree.leaves[leaf_index]: the whole leaf, salts included (what a Merkle opening of a blinded oracle carries)
    std::vector<F> leaf(size_t leaf_index) const {
        std::vector<F> row(shape().n_cols + glp_batch_salt_cols(b_));
        check(glp_batch_row(ctx_->raw(), b_, leaf_index, row.data(), nullptr), "MerkleTree::get");
        return row;
    }
    bool blinding() const { return glp_batch_salt_cols(b_) != 0; }
    // merkle_tree.prove(leaf_index).siblings
    std::vector<HashOut> prove(size_t leaf_index) const {
        auto s = shape();
        std::vector<HashOut> path(s.log_n + s.rate_bits - s.cap_height);
        check(glp_batch_row(ctx_->raw(), b_, leaf_index, nullptr, path.empty() ? nullptr : &path[0].elements[0]),
              "MerkleTree::prove");
        return path;
    }
    // polynomials[i].coeffs
    std::vector<F> coeffs(size_t i) const {
        std::vector<F> out((size_t)1 << shape().log_n);
        check(glp_batch_coeffs(ctx_->raw(), b_, i, out.data()), "polynomials[i]");
        return out;
    }
    // OpeningSet::new's eval_commitment(z, batch)
    std::vector<Ext> eval_commitment(const Ext& z) const {
        size_t n = shape().n_cols;
        std::vector<F> flat(2 * n);
        const F zz[2] = {z.first, z.second};
        check(glp_batch_eval_ext(ctx_->raw(), b_, 0, n, zz, flat.data()), "eval_commitment");
        std::vector<Ext> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {flat[2 * i], flat[2 * i + 1]};
        return out;
    }

  private:
    struct Shape { size_t n_cols; uint32_t log_n, rate_bits, cap_height; };
    Shape shape() const {
        Shape s{};
        glp_batch_shape(b_, &s.n_cols, &s.log_n, &s.rate_bits, &s.cap_height);
        return s;
    }
    PolynomialBatch(Context* c, glp_batch* b, MerkleCap cap) : ctx_(c), b_(b), cap_(std::move(cap)) {}
    static PolynomialBatch make(Context& ctx, const std::vector<std::vector<F>>& cols, size_t rate_bits, bool blinding,
                                size_t cap_height, bool coeffs) {
        if (cols.empty()) throw std::logic_error("PolynomialBatch: no polynomials");
        const size_t n = cols[0].size();
        uint32_t log_n = 0;
        while (((size_t)1 << log_n) < n) ++log_n;
        if (((size_t)1 << log_n) != n) throw std::logic_error("PolynomialBatch: length is not a power of two");
        for (auto& c : cols)
            if (c.size() != n) throw std::logic_error("PolynomialBatch: ragged polynomials");
        MerkleCap cap((size_t)1 << cap_height);
        glp_batch* b = nullptr;
        if (!coeffs) {  // Vec<PolynomialValues<F>> as it is: one pointer per polynomial, no flattening (glp_batch_from_values_cols)
            std::vector<const uint64_t*> ptrs;
            for (auto& c : cols) ptrs.push_back(c.data());
            check(glp_batch_from_values_cols(ctx.raw(), ptrs.data(), ptrs.size(), log_n, (uint32_t)rate_bits, (uint32_t)cap_height,
                                             blinding ? 1u : 0u, &b, &cap[0].elements[0]),
                  "PolynomialBatch::from_values", /*infallible=*/true);
            return PolynomialBatch(&ctx, b, std::move(cap));
        }
        std::vector<F> flat;
        flat.reserve(cols.size() * n);
        for (auto& c : cols) flat.insert(flat.end(), c.begin(), c.end());
        check(glp_batch_from_coeffs(ctx.raw(), flat.data(), cols.size(), log_n, (uint32_t)rate_bits, (uint32_t)cap_height, blinding ? 1u : 0u,
                                    &b, &cap[0].elements[0]),
              "PolynomialBatch::from_coeffs", /*infallible=*/true);
        return PolynomialBatch(&ctx, b, std::move(cap));
    }
    Context* ctx_;
    glp_batch* b_;
    MerkleCap cap_;
};

class Challenger {
  public:
    Challenger() { glp_challenger_init(&c_); }
    void observe_element(F e) { glp_challenger_observe(&c_, &e, 1); }
    void observe_elements(const std::vector<F>& es) { glp_challenger_observe(&c_, es.data(), es.size()); }
    void observe_hash(const HashOut& h) { glp_challenger_observe(&c_, h.elements, 4); }
    void observe_cap(const MerkleCap& cap) {
        for (auto& h : cap) observe_hash(h);
    }
    void observe_extension_element(const Ext& e) {
        observe_element(e.first);
        observe_element(e.second);
    }
    void observe_extension_elements(const std::vector<Ext>& es) {
        for (auto& e : es) observe_extension_element(e);
    }
    F get_challenge() { return glp_challenger_get(&c_); }
    std::vector<F> get_n_challenges(size_t n) {
        std::vector<F> out(n);
        glp_challenger_get_n(&c_, out.data(), n);
        return out;
    }
    Ext get_extension_challenge() {
        F a = get_challenge(), b = get_challenge();
        return {a, b};
    }
    HashOut get_hash() {
        HashOut h;
        glp_challenger_get_n(&c_, h.elements, 4);
        return h;
    }
    glp_challenger* raw() { return &c_; }

  private:
    glp_challenger c_;
};

struct FriPolynomialInfo { uint32_t oracle_index, polynomial_index; };
struct FriBatchInfo {
    Ext point;
    std::vector<FriPolynomialInfo> polynomials;
    std::vector<Ext> opened_values;  // the matching FriOpeningBatch (already observed by the challenger)
};
using FriInstanceInfo = std::vector<FriBatchInfo>;

struct FriParams : glp_fri_params {
    // standard_recursion_config: rate 3, cap 4, 16 PoW bits, 28 queries, ConstantArityBits(4, 5)
    static FriParams standard(uint32_t degree_bits) {
        FriParams p{};
        p.degree_bits = degree_bits;
        p.rate_bits = 3;
        p.cap_height = 4;
        p.proof_of_work_bits = 16;
        p.num_query_rounds = 28;
        uint32_t d = degree_bits;
        while (d > 5 && d + p.rate_bits - 4 >= p.cap_height) {
            p.reduction_arity_bits[p.n_reduction_rounds++] = 4;
            d -= 4;
        }
        return p;
    }
};

// PolynomialBatch::prove_openings -> flat FriProof (layout in include/glprover.h)
inline std::vector<F> prove_openings(Context& ctx, const FriInstanceInfo& instance, const std::vector<const PolynomialBatch*>& oracles,
                                     Challenger& challenger, const FriParams& params) {
    std::vector<const glp_batch*> raw;
    std::vector<size_t> cols;
    for (auto* o : oracles) {
        raw.push_back(o->raw());
        cols.push_back(o->num_polys());
    }
    std::vector<std::vector<uint32_t>> oi(instance.size()), pi(instance.size());
    std::vector<std::vector<F>> ov(instance.size());
    std::vector<glp_fri_batch> batches(instance.size());
    for (size_t b = 0; b < instance.size(); ++b) {
        for (auto& p : instance[b].polynomials) {
            oi[b].push_back(p.oracle_index);
            pi[b].push_back(p.polynomial_index);
        }
        for (auto& e : instance[b].opened_values) {
            ov[b].push_back(e.first);
            ov[b].push_back(e.second);
        }
        batches[b].point[0] = instance[b].point.first;
        batches[b].point[1] = instance[b].point.second;
        batches[b].n_polys = oi[b].size();
        batches[b].oracle_index = oi[b].data();
        batches[b].poly_index = pi[b].data();
        batches[b].opened_values = ov[b].empty() ? nullptr : ov[b].data();
    }
    std::vector<F> proof(glp_fri_proof_len(&params, cols.data(), cols.size()));
    check(glp_fri_prove(ctx.raw(), raw.data(), raw.size(), batches.data(), batches.size(), &params, challenger.raw(),
                        proof.data()),
          "PolynomialBatch::prove_openings");
    return proof;
}

// CircuitData: `build` commits to constants and sigmas; `prove` is plonk::prover::prove after witness generation
class CircuitData {
  public:
    CircuitData(Context& ctx, glp_circuit common, std::vector<glp_gate> gates, std::vector<F> k_is,
                const std::vector<std::vector<F>>& constants_sigmas_values)
        : ctx_(&ctx), gates_(std::move(gates)), k_is_(std::move(k_is)) {
        common.gates = gates_.data();
        common.k_is = k_is_.data();
        std::vector<F> flat;
        for (auto& c : constants_sigmas_values) flat.insert(flat.end(), c.begin(), c.end());
        cap_.resize((size_t)1 << common.fri.cap_height);
        check(glp_circuit_build(ctx.raw(), &common, flat.data(), &d_, &cap_[0].elements[0], digest_.elements),
              "CircuitBuilder::build", true);
        common_ = common;
        for (int i = 0; i < 4; ++i) common_.circuit_digest[i] = digest_.elements[i];
    }
    ~CircuitData() { glp_circuit_free(d_); }
    CircuitData(const CircuitData&) = delete;
    const MerkleCap& constants_sigmas_cap() const { return cap_; }
    const HashOut& circuit_digest() const { return digest_; }
    const glp_circuit& common() const { return common_; }
    // returns the flat ProofWithPublicInputs::proof (layout in include/glprover.h); Err -> std::runtime_error
    // wire_values = witness.wire_values as plonky2 holds it (one Vec<F> per wire column): handed over as column pointers
    // (glp_prove_cols), no flattening on the host
    std::vector<F> prove(const std::vector<std::vector<F>>& wire_values, const std::vector<F>& public_inputs) const {
        const size_t n = (size_t)1 << common_.degree_bits;
        std::vector<const uint64_t*> cols;
        for (auto& c : wire_values) {
            if (c.size() != n) throw std::runtime_error("CircuitData::prove: a wire column is not 2^degree_bits long");
            cols.push_back(c.data());
        }
        std::vector<F> proof(glp_proof_len(&common_));
        check(glp_prove_cols(ctx_->raw(), d_, cols.data(), cols.size(), public_inputs.data(), public_inputs.size(), proof.data()),
              "CircuitData::prove");
        return proof;
    }

    // K proofs of this circuit in one call (glp_prove_batch_dev): the shape of the reference's per-sender loops
    // (src/bin/block_circuit.rs:293-361,410-434).  witnesses[k] = witness.wire_values of proof k; -> K flat proofs, each equal
    // to prove(witnesses[k], public_inputs[k]).
    std::vector<std::vector<F>> prove_many(const std::vector<std::vector<std::vector<F>>>& witnesses,
                                           const std::vector<std::vector<F>>& public_inputs) const {
        const size_t K = witnesses.size(), n = (size_t)1 << common_.degree_bits, nw = common_.num_wires;
        if (K == 0 || public_inputs.size() != K) throw std::runtime_error("prove_many: one public-input vector per witness");
        const size_t n_pi = public_inputs[0].size(), len = glp_proof_len(&common_);
        // every witness as plonky2 holds it — K x num_wires column pointers, no flattened copy (glp_prove_batch_cols)
        std::vector<const uint64_t*> cols;
        std::vector<F> pis;
        for (size_t k = 0; k < K; ++k) {
            if (witnesses[k].size() != nw || public_inputs[k].size() != n_pi) throw std::runtime_error("prove_many: ragged input");
            for (auto& c : witnesses[k]) {
                if (c.size() != n) throw std::runtime_error("prove_many: a wire column is not 2^degree_bits long");
                cols.push_back(c.data());
            }
            pis.insert(pis.end(), public_inputs[k].begin(), public_inputs[k].end());
        }
        std::vector<F> out(K * len);
        check(glp_prove_batch_cols(ctx_->raw(), d_, K, cols.data(), nw, pis.data(), n_pi, out.data()), "CircuitData::prove (batch)");
        std::vector<std::vector<F>> proofs(K);
        for (size_t k = 0; k < K; ++k) proofs[k].assign(out.begin() + k * len, out.begin() + (k + 1) * len);
        return proofs;
    }

  private:
    Context* ctx_;
    std::vector<glp_gate> gates_;
    std::vector<F> k_is_;
    glp_circuit common_{};
    glp_circuit_data* d_ = nullptr;
    MerkleCap cap_;
    HashOut digest_{};
};

}  // namespace plonky2
