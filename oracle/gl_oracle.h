/*
 * gl_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the algorithms on the hot path that intmax-zkp-core reaches through
 * `CircuitData::prove` (reference call sites: src/zkdsa/circuits/mod.rs:326,
 * src/transaction/circuits/mod.rs:453, src/rollup/circuits/mod.rs:1247).  The arithmetic itself lives in the
 * un-vendored dependency plonky2 @ f99ed9cc3ad62ee2d526f668adfa45fdeafed5a7 (reference Cargo.toml:12,
 * Cargo.lock:333-335), which is absent from /root/reference; its published algorithm is restated here.
 *
 * PARITY STATUS
 *   - Poseidon permutation / two_to_one / hash_no_pad / hash_pad / HashOut hex codec: PINNED by the reference's
 *     own golden values (tests/golden/poseidon_kat.json, extracted from src/zkdsa/circuits/mod.rs:85-100,
 *     src/rollup/circuits/mod.rs:104, test_cases/block1_info.json).
 *   - NTT ordering, LDE / bit-reversal conventions, Merkle cap ordering, challenger, FRI: "parity unpinned" —
 *     the reference holds no golden vectors for them; they are checked by mathematical
 *     properties (NTT == naive DFT, Merkle proofs verify, FRI proofs accepted by the restated verifier).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this library.
 */
#ifndef GL_ORACLE_H
#define GL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Goldilocks field, p = 2^64 - 2^32 + 1 (plonky2 field/src/goldilocks_field.rs) ---- */
#define GLO_P 0xFFFFFFFF00000001ULL
#define GLO_EPS 0xFFFFFFFFULL

uint64_t glo_canon(uint64_t x);
uint64_t glo_add(uint64_t a, uint64_t b);
uint64_t glo_sub(uint64_t a, uint64_t b);
uint64_t glo_mul(uint64_t a, uint64_t b);
uint64_t glo_mul_slow(uint64_t a, uint64_t b); /* via 128-bit % — cross-check for glo_mul */
void glo_mul_vec(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n); /* out[i] = a[i] b[i] mod p (128-bit %) */
uint64_t glo_pow(uint64_t a, uint64_t e);
uint64_t glo_inv(uint64_t a);
uint64_t glo_root_of_unity(unsigned log_n); /* POWER_OF_TWO_GENERATOR^(2^(32-log_n)) */

/* quadratic extension F[X]/(X^2 - 7)  (plonky2 field/src/extension/quadratic.rs) */
void glo_ext_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void glo_ext_add(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void glo_ext_sub(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void glo_ext_inv(const uint64_t a[2], uint64_t out[2]);

/* ---- Poseidon-Goldilocks, width 12 (plonky2 src/hash/poseidon.rs, poseidon_goldilocks.rs) ---- */
const uint64_t *glo_poseidon_round_constants(void); /* 360 values, derived from ChaCha8Rng::seed_from_u64(0) */
void glo_poseidon_permute(uint64_t state[12]);
void glo_poseidon_permute_slow(uint64_t state[12]); /* textbook MDS with integer %, cross-check */
/* plonky2 src/hash/hashing.rs: hash_n_to_m_no_pad (overwrite-mode sponge, rate 8) */
void glo_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]);
/* hash_n_to_hash_pad... as used natively by src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:170 */
void glo_hash_pad(const uint64_t *in, size_t len, uint64_t out[4]);
void glo_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);
void glo_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]);
/* HashOut <-> "0x…" hex, reference src/sparse_merkle_tree/goldilocks_poseidon/hash/mod.rs:84-119 */
void glo_hashout_to_hex(const uint64_t h[4], char out[67]);
int glo_hashout_from_hex(const char *hex, uint64_t h[4]);

/* ---- native tree hashing of the reference itself (smt.c) ---- */
int glo_smt_key_bit(const uint64_t key[4], unsigned i);
void glo_smt_leaf_hash(const uint64_t key[4], const uint64_t value[4], uint64_t out[4]);
void glo_smt_root_from_path(const uint64_t key[4], const uint64_t start[4], const uint64_t *siblings, size_t n_siblings,
                            uint64_t out[4]);
void glo_merkle_root(size_t index, const uint64_t value[4], const uint64_t *siblings, size_t depth, uint64_t out[4]);
size_t glo_merkle_proof_n_siblings(size_t depth);
int glo_merkle_proof_with_zero(const uint64_t *leaves, size_t n_leaves, size_t index, size_t depth, const uint64_t zero[4],
                               uint64_t *siblings_out, uint64_t root_out[4]);
void glo_block_hash(const uint64_t header[29], uint64_t out[4]);
int glo_smt_root_of_set(const uint64_t *keys, const uint64_t *values, size_t n, uint64_t out[4]);
/* `find` for m query keys against the tree of the set (reference src/sparse_merkle_tree/tree.rs:588-690) */
int glo_smt_proofs_of_set(const uint64_t *keys, const uint64_t *values, size_t n, const uint64_t *queries, size_t m,
                          size_t max_siblings, uint64_t *siblings_out, uint32_t *n_siblings_out, uint32_t *flags_out,
                          uint64_t *value_out, uint64_t *not_found_key_out, uint64_t *not_found_value_out, uint64_t root_out[4]);

/* ---- NTT (plonky2 field/src/fft.rs, polynomial/mod.rs) — natural order in and out ---- */
void glo_dft_naive(const uint64_t *coeffs, uint64_t *values, unsigned log_n);
void glo_fft(uint64_t *a, unsigned log_n);                 /* coeffs -> values, in place */
void glo_ifft(uint64_t *a, unsigned log_n);                /* values -> coeffs, in place */
void glo_coset_fft(uint64_t *a, unsigned log_n, uint64_t shift);
void glo_coset_ifft(uint64_t *a, unsigned log_n, uint64_t shift);
size_t glo_reverse_bits(size_t x, unsigned bits);
uint64_t glo_poly_eval(const uint64_t *coeffs, size_t n, uint64_t x); /* Horner, base-field point */

/* ---- Merkle tree with cap (plonky2 src/hash/merkle_tree.rs) ---- */
/* leaves: n_leaves rows of leaf_len u64, row-major. digests_out (optional) receives every level bottom-up:
 * level 0 = leaf digests (n_leaves*4), level 1 = n_leaves/2 * 4, ... down to the cap level (2^cap_height * 4).
 * cap_out receives 2^cap_height * 4 u64. */
size_t glo_merkle_levels_len(size_t n_leaves, unsigned cap_height); /* u64 count of digests_out */
void glo_merkle_build(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height,
                      uint64_t *digests_out, uint64_t *cap_out);
/* siblings bottom-up, (log2(n_leaves) - cap_height) * 4 u64 */
void glo_merkle_prove(const uint64_t *digests, size_t n_leaves, unsigned cap_height, size_t leaf_index,
                      uint64_t *siblings_out);
int glo_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t leaf_index, const uint64_t *siblings,
                      size_t n_siblings, const uint64_t *cap, unsigned cap_height);

/* ---- PolynomialBatch::from_values / from_coeffs (plonky2 src/fri/oracle.rs) ---- */
/* cols: n_cols columns of 2^log_n u64, column-major (one contiguous column after another).
 * coeffs_out: n_cols x 2^log_n (column-major, natural order).
 * leaves_out: 2^(log_n+rate_bits) rows x n_cols, row-major, rows in bit-reversed order.
 * digests_out: as glo_merkle_build (may be NULL). cap_out: 2^cap_height x 4. */
void glo_batch_from_values(const uint64_t *cols, size_t n_cols, unsigned log_n, unsigned rate_bits,
                           unsigned cap_height, uint64_t *coeffs_out, uint64_t *leaves_out,
                           uint64_t *digests_out, uint64_t *cap_out);
void glo_batch_from_coeffs(const uint64_t *coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits,
                           unsigned cap_height, uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out);
/* blinding = true (plonky2 src/fri/oracle.rs::lde_values: `.chain((0..salt_size).map(|_| F::rand_vec(degree << rate_bits)))`):
 * SALT_SIZE = 4 random columns of 2^(log_n + rate_bits) values are appended to the LDE values before the transpose, so every
 * leaf row is [polynomial values ..., salt_0[i], .., salt_3[i]] with i the row's NATURAL index (rows then bit-reversed like the
 * rest).  The salts are the caller's (plonky2 draws them from thread_rng): salts = 4 columns of 2^(log_n + rate_bits), or NULL
 * for an unblinded batch.  leaves_out: rows of n_cols + 4 words when salted. */
#define GLO_SALT_SIZE 4
void glo_batch_from_values_salted(const uint64_t *cols, size_t n_cols, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                                  const uint64_t *salts, uint64_t *coeffs_out, uint64_t *leaves_out, uint64_t *digests_out,
                                  uint64_t *cap_out);
void glo_batch_from_coeffs_salted(const uint64_t *coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                                  const uint64_t *salts, uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out);


/* ---- Fiat-Shamir challenger (plonky2 src/iop/challenger.rs) ---- */
typedef struct {
    uint64_t state[12];
    uint64_t in[8];
    uint64_t out[8];
    int n_in, n_out;
} glo_challenger;
void glo_challenger_init(glo_challenger *c);
void glo_challenger_observe(glo_challenger *c, const uint64_t *elems, size_t n);
uint64_t glo_challenger_get(glo_challenger *c);
void glo_challenger_get_n(glo_challenger *c, uint64_t *out, size_t n);

/* p(z) for base-field coefficients and an extension point (building block of OpeningSet::new, plonk/proof.rs) */
void glo_poly_eval_ext(const uint64_t *coeffs, size_t n, const uint64_t z[2], uint64_t out[2]);

/* ---- FRI (plonky2 src/fri/{oracle,prover,verifier,challenges}.rs) ---- */
/* Transcript variants: the three places where plonky2 revisions differ and this restatement of f99ed9c (Dec 2022) is
 * from memory.  flags = 0 is the restated f99ed9c behaviour; the other settings exist so that the first run against real
 * plonky2 (integration/rust/parity_dump.rs + tools/compare_plonky2_dump.py) can settle each choice by flipping a bit.
 * The product (include/glprover.h: GLP_FLAG_*) takes the same bits and must equal this oracle under every setting. */
#define GLO_FLAG_FRI_NO_MUL_X 1u   /* prove_openings does NOT multiply the final polynomial by X (plonky2 before PR 436) */
#define GLO_FLAG_POW_DUPLEX 2u     /* PoW response = challenger.observe(w); get_challenge()  (plonky2 >= mid 2023) instead of
                                      hash_no_pad(get_hash() || w)[0] with w not observed                                */
#define GLO_FLAG_DIGEST_MASK 0x300u
#define GLO_FLAG_DIGEST_CAP 0x000u        /* circuit_digest = hash_no_pad(cap)                          (f99ed9c) */
#define GLO_FLAG_DIGEST_CAP_DEG 0x100u    /* hash_no_pad(cap || degree_bits)                            (early 2023) */
#define GLO_FLAG_DIGEST_CAP_DOM_DEG 0x200u /* hash_no_pad(cap || hash_pad([]) || degree_bits)           (>= mid 2023) */
/* config.zero_knowledge (FriParams::hiding): the wires, Z / partial-product and quotient oracles are blinded (PlonkOracle::
 * {WIRES, ZS_PARTIAL_PRODUCTS, QUOTIENT}.blinding; CONSTANTS_SIGMAS never is) — their leaves carry GLO_SALT_SIZE salt words
 * that the query openings include and FriInitialTreeProof::unsalted_eval skips */
#define GLO_FLAG_ZERO_KNOWLEDGE 0x1000u
typedef struct {
    unsigned degree_bits, rate_bits, cap_height, pow_bits, num_queries;
    unsigned n_rounds;
    unsigned arity_bits[16]; /* reduction_arity_bits */
    unsigned flags;          /* GLO_FLAG_* */
} glo_fri_params;
/* circuit_builder.rs::build: the circuit digest under the composition the flags select */
void glo_circuit_digest(const uint64_t *cap, size_t cap_words, unsigned degree_bits, unsigned flags, uint64_t out[4]);
typedef struct {
    uint64_t point[2];            /* opening point (extension) */
    size_t n_polys;
    const uint32_t *oracle_index; /* FriPolynomialInfo */
    const uint32_t *poly_index;
} glo_fri_batch;
typedef struct {
    size_t n_batches;
    const glo_fri_batch *batches;
} glo_fri_instance;
/* u64 words of the flat proof: caps[n_rounds][2^cap][4] | queries[num_queries]{ per oracle: row, path ; per round:
 * evals[arity][2], path } | final_poly[][2] | pow_witness  (field order of plonky2's FriProof struct) */
size_t glo_fri_proof_len(const glo_fri_params *p, const size_t *oracle_cols, size_t n_oracles);
/* PolynomialBatch::prove_openings: oracle_coeffs[o] = n_cols x n column-major, oracle_leaves[o] = N x n_cols row-major
 * (bit-reversed rows), oracle_digests[o] as glo_merkle_build. The challenger must already have observed the openings. */
int glo_fri_prove(const glo_fri_instance *inst, const uint64_t *const *oracle_coeffs, const uint64_t *const *oracle_leaves,
                  const uint64_t *const *oracle_digests, const size_t *oracle_cols, size_t n_oracles,
                  const glo_fri_params *p, glo_challenger *ch, uint64_t *proof_out);
/* verify_fri_proof: 0 = accept, negative = the failing check. batch_openings[b] = n_polys ext values. */
int glo_fri_verify(const glo_fri_instance *inst, const uint64_t *const *batch_openings, const uint64_t *const *oracle_caps,
                   const size_t *oracle_cols, size_t n_oracles, const glo_fri_params *p, glo_challenger *ch,
                   const uint64_t *proof);


/* ---- PLONK layer (plonky2 src/plonk/{prover,vanishing_poly,verifier,get_challenges}.rs, src/gates/) ---- */
#define GLO_GATE_NOOP 0
#define GLO_GATE_CONSTANT 1      /* param = num_consts */
#define GLO_GATE_PUBLIC_INPUT 2
#define GLO_GATE_ARITHMETIC 3    /* param = num_ops */
#define GLO_GATE_POSEIDON 4
#define GLO_GATE_BASE_SUM 5      /* BaseSumGate<2>, param = num_limbs */
#define GLO_GATE_ARITHMETIC_EXT 6 /* ArithmeticExtensionGate, param = num_ops */
#define GLO_GATE_MUL_EXT 7       /* MulExtensionGate, param = num_ops */
/* the gates that builder.verify_proof adds (reference src/recursion/gadgets/mod.rs:114-127) */
#define GLO_GATE_RANDOM_ACCESS 8   /* RandomAccessGate, param = bits | num_copies << 8 | num_extra_constants << 16 */
#define GLO_GATE_REDUCING 9        /* ReducingGate<2>, param = num_coeffs */
#define GLO_GATE_REDUCING_EXT 10   /* ReducingExtensionGate<2>, param = num_coeffs */
#define GLO_GATE_EXPONENTIATION 11 /* ExponentiationGate, param = num_power_bits */
#define GLO_GATE_POSEIDON_MDS 12   /* PoseidonMdsGate */
#define GLO_GATE_COSET_INTERPOLATION 13 /* CosetInterpolationGate, param = subgroup_bits | degree << 8 */
#define GLO_GATE_ID_MAX 13
typedef struct {
    uint32_t id;                       /* GLO_GATE_* */
    uint32_t selector_index;           /* selectors_info.selector_indices[i] */
    uint32_t group_start, group_end;   /* selectors_info.groups[selector_index] (range of gate indices) */
    uint32_t param;
} glo_gate;
typedef struct {                       /* the parts of CommonCircuitData the prover/verifier arithmetic needs */
    uint32_t degree_bits;
    uint32_t num_wires, num_routed_wires;
    uint32_t num_constants;            /* selectors + config.num_constants */
    uint32_t num_selectors;
    uint32_t num_challenges, quotient_degree_factor, num_partial_products;
    uint32_t num_gates;
    uint32_t num_gate_constraints;
    const glo_gate *gates;             /* sorted by (degree, id) as common_data.gates */
    const uint64_t *k_is;              /* num_routed_wires coset shifts */
    uint64_t circuit_digest[4];
    glo_fri_params fri;
} glo_circuit;
unsigned glo_gate_num_constraints(unsigned id, unsigned param);
/* CosetInterpolationGate constants: the 2^bits subgroup points g^i and their barycentric weights */
void glo_coset_interpolation_tables(unsigned bits, uint64_t *domain, uint64_t *weights);
void glo_poseidon_gate_witness(const uint64_t in[12], uint64_t wires[135]); /* PoseidonGenerator, swap = 0 */
void glo_poseidon_gate_witness_many(const uint64_t *in, size_t k, uint64_t *wires); /* k rows: in [k][12] -> wires [k][135] */
void glo_partial_products_and_zs(const glo_circuit *cc, const uint64_t *wires, const uint64_t *sigmas,
                                 const uint64_t *betas, const uint64_t *gammas, uint64_t *out);
void glo_quotient_chunks(const glo_circuit *cc, const uint64_t *cs_leaves, const uint64_t *wires_leaves,
                         const uint64_t *zs_leaves, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                         const uint64_t pi_hash[4], uint64_t *out);
/* flat proof: wires_cap | zs_partial_products_cap | quotient_cap | openings (constants, sigmas, wires, zs, zs_next,
 * partial_products, quotient_polys; extension values) | FriProof (layout of glo_fri_proof_len) */
size_t glo_plonk_proof_len(const glo_circuit *cc);
/* prove: cs_* = the constants_sigmas oracle made at build time (glo_batch_from_values); sigma_values = the sigma
 * polynomials' values on H [num_routed][n]; wire_values = the full witness [num_wires][n] */
int glo_plonk_prove(const glo_circuit *cc, const uint64_t *cs_coeffs, const uint64_t *cs_leaves, const uint64_t *cs_digests,
                    const uint64_t *sigma_values, const uint64_t *wire_values, const uint64_t *public_inputs,
                    size_t n_public_inputs, uint64_t *proof_out);
/* the same under config.zero_knowledge (cc->fri.flags & GLO_FLAG_ZERO_KNOWLEDGE): salts = 3 x GLO_SALT_SIZE columns of
 * 2^(degree_bits + rate_bits) values, for the wires, Z / partial-product and quotient commitments in this order (NULL allowed
 * only without the flag) */
int glo_plonk_prove_salted(const glo_circuit *cc, const uint64_t *cs_coeffs, const uint64_t *cs_leaves, const uint64_t *cs_digests,
                           const uint64_t *sigma_values, const uint64_t *wire_values, const uint64_t *public_inputs,
                           size_t n_public_inputs, const uint64_t *salts, uint64_t *proof_out);
int glo_last_phase_seconds(double out[8]); /* per-phase wall time of the last glo_plonk_prove call (returns the count) */
/* verify: 0 = accept; -20-i = vanishing identity i fails; other negatives = FRI check (glo_fri_verify) */
int glo_plonk_verify(const glo_circuit *cc, const uint64_t *cs_cap, const uint64_t *public_inputs, size_t n_public_inputs,
                     const uint64_t *proof);
/* First-contact diagnostics: the vanishing-term vector at zeta before the alpha reduction and every gate's filter and
 * eval_unfiltered output, from an OpeningSet (flat extension values: constants ++ sigmas, wires, zs, zs_next, partial
 * products) and given challenges.  terms_out [n_terms][2] (returns n_terms), filters_out [num_gates][2],
 * unfiltered_out [num_gates][128][2], n_cons_out [num_gates] */
size_t glo_vanishing_diagnose(const glo_circuit *cc, const uint64_t zeta[2], const uint64_t *o_cs, const uint64_t *o_w,
                              const uint64_t *o_z, const uint64_t *o_zn, const uint64_t *o_pp, const uint64_t *betas,
                              const uint64_t *gammas, const uint64_t pi_hash[4], uint64_t *terms_out, uint64_t *filters_out,
                              uint64_t *unfiltered_out, uint32_t *n_cons_out);

/* threads used by the OpenMP-parallel loops (per-column NTT, per-leaf hashing — plonky2's rayon split) */
void glo_set_threads(int n);
int glo_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
