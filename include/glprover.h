/*
 * glprover.h — C ABI of libglprover.so: the MI355X (gfx950) backend for the Plonky2 prover path that
 * intmax-zkp-core reaches through `CircuitData::prove` / `CircuitBuilder::build`.
 *
 * Drop-in boundary.  The reference (Rust) never calls PolynomialBatch itself; it calls
 *     data.prove(pw)            src/zkdsa/circuits/mod.rs:326, src/transaction/circuits/mod.rs:453,
 *                               src/rollup/circuits/mod.rs:1247, src/bin/block_circuit.rs:706, ...
 *     builder.build::<C>()      src/rollup/circuits/mod.rs:605 and 19 more sites
 * and everything below happens inside its pinned dependency plonky2 @ f99ed9c (Cargo.toml:12).  The entry points
 * here are what a patched plonky2 `fri/oracle.rs`, `hash/merkle_tree.rs` and `field/fft.rs` would bind through
 * `extern "C"` (INTEGRATION.md shows the Rust stub for each).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns GLP_OK (0) or a negative GLP_ERR_*; nothing throws or unwinds across the boundary;
 *     out-parameters are written only on success.
 *   - field elements are little-endian u64 words of GoldilocksField (p = 2^64 - 2^32 + 1); inputs may be
 *     non-canonical (>= p), outputs are always canonical.
 *   - `*_dev` arguments are device (HBM) pointers valid on the context's GPU; others are host pointers.
 *   - a context is single-threaded; different contexts (one per GPU / per stream) may run concurrently.
 *   - a context may be used from any host thread (one at a time).  HIP's current device is per thread and starts at 0: every
 *     entry point makes the context's device current for the duration of the call and puts the caller's back on return
 *     (glp_ctx_create included), so a context of GPU 3 works from a worker thread that never called hipSetDevice.
 *   - circuit data and batches are read-only once built: they may be used through ANY context of the same GPU (a host builds a
 *     circuit once and proves it from each thread's own context, concurrently); with a context of another GPU -> GLP_ERR_INVALID.
 *   - there is NO CPU fallback: without a usable GPU, glp_ctx_create fails with GLP_ERR_NO_DEVICE.
 */
#ifndef GLPROVER_H
#define GLPROVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLP_OK 0
#define GLP_ERR_INVALID (-1)      /* bad argument (null pointer, size not supported, ...) */
#define GLP_ERR_NO_DEVICE (-2)    /* no HIP device / device index out of range */
#define GLP_ERR_HIP (-3)          /* a HIP runtime call failed; glp_last_hip_error() has the hipError_t */
#define GLP_ERR_OOM (-4)          /* device allocation failed */
#define GLP_ERR_INTERNAL (-6)     /* an internal consistency check failed (a bug in the library, never a caller error) */
#define GLP_ERR_UNSUPPORTED (-5)  /* a shape or option outside what the kernels are written for (blinding = 1 is supported since round 5;
                                     the one case left: the library is asked to draw the salts itself and the context got no entropy —
                                     getrandom() and /dev/urandom both failed at glp_ctx_create, no glp_ctx_set_blinding_seed since)
                                     (TODO comments at src/zkdsa/circuits/mod.rs:412, src/rollup/circuits/mod.rs:1283) */

typedef struct glp_ctx glp_ctx;     /* one GPU worker: stream, twiddle tables, device memory pool */
typedef struct glp_batch glp_batch; /* device-resident PolynomialBatch: coefficients + LDE + Merkle tree */

/* ---- library / context ---- */
uint32_t glp_abi_version(void);
const char *glp_strerror(int32_t status);
int32_t glp_last_hip_error(void);
int32_t glp_device_count(int32_t *count_out);
/* Host worker threads of this process (the caller included) behind the batched prover's per-proof transcript / assembly work:
 * GLP_HOST_THREADS if set; else the CPUs the process is pinned to when that is a subset of the host; else hardware threads /
 * LOCAL_WORLD_SIZE (one process per GPU: eight ranks share one host), floor 2; at most 32.  plonky2's equivalent is the rayon
 * pool size under the reference's `parallel` feature (Cargo.toml:18-21). */
uint32_t glp_host_threads(void);
/* One context per GPU worker (plonky2 has no equivalent: it owns the rayon pool implicitly). */
int32_t glp_ctx_create(int32_t device, glp_ctx **ctx_out);
void glp_ctx_destroy(glp_ctx *ctx);
/* Run all work of this context on an existing hipStream_t (e.g. the caller's PyTorch stream) instead of the
 * non-blocking stream the context created for itself; NULL = the device's default (null) stream. */
int32_t glp_ctx_set_stream(glp_ctx *ctx, void *hip_stream);
int32_t glp_ctx_synchronize(glp_ctx *ctx);
/* Device memory of the context's pool: bytes handed out to live objects / in-flight calls, and bytes of freed blocks kept for reuse
 * (at most GLP_POOL_CACHE_GB, default 96).  glp_ctx_trim_cache returns the cached blocks to the driver (it waits for the device);
 * the library does the same by itself — to this context's cache, then to the other contexts' of the GPU — before it reports
 * GLP_ERR_OOM.  May be called from any thread while the context is idle. */
int32_t glp_ctx_memory(glp_ctx *ctx, uint64_t *live_bytes_out, uint64_t *cached_bytes_out);
int32_t glp_ctx_trim_cache(glp_ctx *ctx);
/* Debug aid.  With GLP_POOL_CANARY=1 in the environment every device block of the context's pool carries a 256-byte guard behind
 * the bytes that were asked for, checked when the block is returned (the stand-in for a compute sanitizer: a kernel that ran
 * past its buffer is reported on stderr and counted; GLP_POOL_CANARY=2 aborts at the first one, for runs whose stderr is
 * captured).  count_out = violations seen so far (always 0 without the switch). */
int32_t glp_ctx_canary_violations(glp_ctx *ctx, uint64_t *count_out);
/* Fault injection (debug aid; SURVEY.md section 5 "GPU OOM / hipError -> Err, caller may fall back": the reference surfaces prover
 * errors as `Err`, src/rollup/circuits/mod.rs:1296,1303,1319,1323).  kind = GLP_FAULT_POOL: the k-th device-pool allocation of
 * this context from now on fails with GLP_ERR_OOM; kind = GLP_FAULT_HIP: the k-th checked HIP runtime call made on the CALLING
 * thread from now on is not made and fails with GLP_ERR_HIP.  One shot (the counter disarms when it fires); k = 0 disarms.
 * The same from the environment, armed when a context is created: GLP_POOL_FAIL_AT=k, GLP_HIP_FAIL_AT=k.
 * After an injected (or real) failure the entry point has returned its pool blocks, its out-parameters are untouched and the
 * context is usable for the next call. */
#define GLP_FAULT_POOL 0u
#define GLP_FAULT_HIP 1u
int32_t glp_ctx_debug_fail_at(glp_ctx *ctx, uint32_t kind, uint64_t k);
/* pool allocations made by this context so far, checked HIP calls made on the calling thread so far, pool blocks / bytes handed
 * out and not yet returned (batches, circuits and glp_dev_alloc blocks count while they live).  Any pointer may be NULL. */
int32_t glp_ctx_debug_counters(glp_ctx *ctx, uint64_t *pool_allocs_out, uint64_t *hip_checks_out,
                               uint64_t *pool_blocks_in_use_out, uint64_t *pool_bytes_in_use_out);
/* Poseidon MDS code path: 0 = integer (v_mad_u64_u32; the default and the measured configuration), 1 = fp64 FMA, 2 = OPT-IN
 * experiment: the six full 12 x 12 layers of a permutation as int8 matrix-core products on byte planes in the wide
 * lane-per-state hashing launches (the north_star of this backend is a non-MFMA path: never selected by default; measured in
 * DESIGN.md).  Results are bit-identical. */
int32_t glp_ctx_set_mds_variant(glp_ctx *ctx, int32_t variant);

/* Per-kernel device timing (hipEvent pairs on the context's stream), used by bench.py for the roofline numbers. */
#define GLP_K_NTT 0          /* NTT / LDE passes */
#define GLP_K_LEAF_HASH 1    /* Poseidon sponge over leaves */
#define GLP_K_MERKLE_LEVELS 2
#define GLP_K_REORDER 3      /* bit-reversal / scaling copies */
#define GLP_K_OPENINGS 4     /* polynomial evaluation at zeta */
#define GLP_K_FRI 5          /* combine / fold / pack / proof-of-work kernels */
#define GLP_K_PLONK 6        /* partial products / Z scan / quotient evaluation */
#define GLP_K_TREES 7        /* native tree hashing (SMT / fixed-depth Merkle / block hash batches) */
#define GLP_K_COUNT 8
int32_t glp_ctx_timing_enable(glp_ctx *ctx, int32_t on);
int32_t glp_ctx_timing_reset(glp_ctx *ctx);
/* ms_out[GLP_K_COUNT], launches_out[GLP_K_COUNT]; synchronises the stream */
int32_t glp_ctx_timing_read(glp_ctx *ctx, double *ms_out, uint64_t *launches_out);

/* ---- PolynomialBatch (plonky2 src/fri/oracle.rs) ---- */
/* PolynomialBatch::from_values(values, rate_bits, blinding, cap_height, timing, fft_root_table):
 * `cols` = n_cols polynomials of 2^log_n evaluations each, one contiguous column after another
 * (= Vec<PolynomialValues<F>>).  Per column: iNTT -> zero-pad x 2^rate_bits -> coset NTT (shift 7) ->
 * leaves (row = all columns at one point, row index bit-reversed) -> Poseidon Merkle tree with 2^cap_height roots.
 * cap_out (optional, host): 2^cap_height x 4 u64. */
int32_t glp_batch_from_values(glp_ctx *ctx, const uint64_t *cols, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, uint32_t blinding, glp_batch **batch_out, uint64_t *cap_out);
/* The same from `values: Vec<PolynomialValues<F>>` as plonky2 passes it (fri/oracle.rs::from_values; the reference reaches it through
 * CircuitBuilder::build, src/rollup/circuits/mod.rs:605): cols[j] = host pointer to polynomial j's 2^log_n values, one allocation
 * per polynomial, pageable or pinned.  The columns cross PCIe in chunks through the library's pinned ring and the commitment
 * follows them (see glp_prove_cols). */
int32_t glp_batch_from_values_cols(glp_ctx *ctx, const uint64_t *const *cols, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                                   uint32_t cap_height, uint32_t blinding, glp_batch **batch_out, uint64_t *cap_out);
/* same, input already resident in HBM (column c starts at cols_dev + c * col_stride) */
int32_t glp_batch_from_values_dev(glp_ctx *ctx, const uint64_t *cols_dev, size_t col_stride, size_t n_cols,
                                  uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, uint32_t blinding,
                                  glp_batch **batch_out, uint64_t *cap_out);
/* PolynomialBatch::from_coeffs: input columns are coefficient vectors (natural order) */
int32_t glp_batch_from_coeffs(glp_ctx *ctx, const uint64_t *coeffs, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, uint32_t blinding, glp_batch **batch_out, uint64_t *cap_out);
int32_t glp_batch_from_coeffs_dev(glp_ctx *ctx, const uint64_t *coeffs_dev, size_t col_stride, size_t n_cols,
                                  uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, uint32_t blinding,
                                  glp_batch **batch_out, uint64_t *cap_out);
/* blinding = 1 (plonky2 fri/oracle.rs::lde_values: SALT_SIZE = 4 columns of F::rand_vec(degree << rate_bits) chained behind the
 * LDE values, so every Merkle leaf is [polynomial values .., salt_0[i], .., salt_3[i]] with i the row's natural index; the
 * reference leaves zero_knowledge off — TODOs at src/zkdsa/circuits/mod.rs:412, src/rollup/circuits/mod.rs:1283,1298 — and
 * standard_recursion_zk_config turns it on).  The four entry points above draw the salts themselves: ChaCha12 in counter mode
 * under a per-context key from getrandom() (glp_ctx_set_blinding_seed fixes the key: reproducible runs), one nonce per
 * commitment.  The *_salted forms take the CALLER's salts — GLP_SALT_SIZE columns of 2^(log_n + rate_bits) values in natural row
 * order, host (device for *_dev) — which is what makes a blinded commitment comparable bit for bit with another implementation
 * given the same salts.  glp_batch_rows / glp_batch_row return the whole leaf (n_cols + GLP_SALT_SIZE words:
 * MerkleTree::get), glp_batch_shape reports the polynomials only (get_lde_values slices the salts off). */
#define GLP_SALT_SIZE 4
int32_t glp_batch_from_values_salted(glp_ctx *ctx, const uint64_t *cols, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                                     uint32_t cap_height, const uint64_t *salts, glp_batch **batch_out, uint64_t *cap_out);
int32_t glp_batch_from_values_salted_dev(glp_ctx *ctx, const uint64_t *cols_dev, size_t col_stride, size_t n_cols, uint32_t log_n,
                                         uint32_t rate_bits, uint32_t cap_height, const uint64_t *salts_dev, glp_batch **batch_out,
                                         uint64_t *cap_out);
int32_t glp_batch_from_coeffs_salted(glp_ctx *ctx, const uint64_t *coeffs, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                                     uint32_t cap_height, const uint64_t *salts, glp_batch **batch_out, uint64_t *cap_out);
int32_t glp_batch_from_coeffs_salted_dev(glp_ctx *ctx, const uint64_t *coeffs_dev, size_t col_stride, size_t n_cols, uint32_t log_n,
                                         uint32_t rate_bits, uint32_t cap_height, const uint64_t *salts_dev, glp_batch **batch_out,
                                         uint64_t *cap_out);
int32_t glp_ctx_set_blinding_seed(glp_ctx *ctx, const uint64_t seed[4]);
uint32_t glp_batch_salt_cols(const glp_batch *batch); /* 0, or GLP_SALT_SIZE for a blinded batch */
void glp_batch_free(glp_batch *batch); /* Rust `Drop` */

/* shape: n_cols, log_n, rate_bits, cap_height */
int32_t glp_batch_shape(const glp_batch *batch, size_t *n_cols, uint32_t *log_n, uint32_t *rate_bits,
                        uint32_t *cap_height);
/* merkle_tree.cap : 2^cap_height x 4 u64 to host */
int32_t glp_batch_cap(glp_ctx *ctx, const glp_batch *batch, uint64_t *cap_out);
/* merkle_tree.leaves[leaf_index] (= get_lde_values: leaf_index = reverse_bits(index * step, log_n + rate_bits)) and
 * merkle_tree.prove(leaf_index): row_out n_cols u64, path_out (log_n + rate_bits - cap_height) x 4 u64, bottom-up.
 * Either output may be NULL. */
int32_t glp_batch_row(glp_ctx *ctx, const glp_batch *batch, size_t leaf_index, uint64_t *row_out, uint64_t *path_out);
/* many rows at once (FRI query phase): rows_out [n_idx][n_cols], paths_out [n_idx][path_len][4] */
int32_t glp_batch_rows(glp_ctx *ctx, const glp_batch *batch, const uint64_t *leaf_indices, size_t n_idx,
                       uint64_t *rows_out, uint64_t *paths_out);
/* .polynomials[col].coeffs (natural order, 2^log_n u64) to host */
int32_t glp_batch_coeffs(glp_ctx *ctx, const glp_batch *batch, size_t col, uint64_t *coeffs_out);
/* device views (valid until glp_batch_free): LDE values, column-major, rows in bit-reversed order:
 * elem(leaf i, column c) = lde_dev[c * col_stride + i]; digests: all levels bottom-up, 4 u64 each */
int32_t glp_batch_lde_dev(const glp_batch *batch, const uint64_t **lde_dev, size_t *col_stride);
int32_t glp_batch_coeffs_dev(const glp_batch *batch, const uint64_t **coeffs_dev, size_t *col_stride);
int32_t glp_batch_digests_dev(const glp_batch *batch, const uint64_t **digests_dev, size_t *n_u64);


/* ---- Fiat-Shamir transcript (plonky2 src/iop/challenger.rs::Challenger<F, PoseidonHash>) ----
 * Plain data so that the host (Rust) can hold it by value; the sponge is the overwrite-mode duplex of plonky2,
 * challenges are popped from the END of the rate part.  Runs on the host: it is the serialisation point between GPU
 * phases and handles a few hundred field elements per proof. */
typedef struct {
    uint64_t sponge_state[12];
    uint64_t input_buffer[8];
    uint64_t output_buffer[8];
    uint32_t n_input, n_output;
} glp_challenger;
void glp_challenger_init(glp_challenger *ch);
void glp_challenger_observe(glp_challenger *ch, const uint64_t *elements, size_t n); /* observe_elements / _hash / _cap */
uint64_t glp_challenger_get(glp_challenger *ch);                                      /* get_challenge */
void glp_challenger_get_n(glp_challenger *ch, uint64_t *out, size_t n); /* get_n_challenges; 2 = extension, 4 = hash */

/* ---- host-side Poseidon utilities (SURVEY.md 8b last row: known-answer tests and the reference's NATIVE call sites) ----
 * Pure host code, no context and no GPU: the permutation the challenger above runs.  They serve the places where the
 * reference hashes a handful of elements on the host (PoseidonHash::two_to_one at src/merkle_tree/tree.rs:70,79,90,123 and
 * src/zkdsa/circuits/mod.rs:85-100; hash_pad in PoseidonNodeHash::calc_node_hash, src/sparse_merkle_tree/goldilocks_poseidon/
 * mod.rs:161-184; hash_no_pad of the public inputs) and let a host check the library against its golden digests before it
 * touches a device.  Batches belong on the *_dev entry points below (a CPU core: ~1 us per permutation).
 * Inputs may be any u64 representative; outputs are canonical. */
void glp_poseidon_permute(uint64_t state[12]);                                   /* Poseidon::poseidon, in place */
void glp_poseidon_round_constants(uint64_t out[360]);                            /* ALL_ROUND_CONSTANTS, round * 12 + lane (what a host-built
                                                                                  * recursive verifier circuit puts into its constant gates) */
void glp_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]);             /* PoseidonHash::hash_no_pad */
void glp_hash_pad(const uint64_t *in, size_t n, uint64_t out[4]);                /* PoseidonHash::hash_pad (pad 1, 0.., 1 to a multiple of 12) */
void glp_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]);            /* PoseidonHash::hash_or_noop (n <= 4: packed, not hashed) */
void glp_two_to_one(const uint64_t left[4], const uint64_t right[4], uint64_t out[4]); /* PoseidonHash::two_to_one */

/* ---- openings (plonky2 src/plonk/proof.rs::OpeningSet::new -> eval_commitment) ----
 * out[(k)*2 .. +2] = polynomial (col_begin + k) of the batch, evaluated at the extension point z = z[0] + z[1] X. */
int32_t glp_batch_eval_ext(glp_ctx *ctx, const glp_batch *batch, size_t col_begin, size_t n_cols, const uint64_t z[2],
                           uint64_t *out);

/* ---- FRI (plonky2 src/fri/oracle.rs::PolynomialBatch::prove_openings -> src/fri/prover.rs::fri_proof) ---- */
/* Transcript variants.  The prover path lives in plonky2 @ f99ed9c, which is absent from the reference tree (Cargo.toml:12),
 * so three choices where plonky2 revisions differ are restated from memory; flags = 0 is the f99ed9c behaviour as restated.
 * Each is one bit here so that the first run against real plonky2 (integration/rust/parity_dump.rs +
 * tests/first_contact/compare_plonky2_dump.py) settles it without a code change. */
#define GLP_FLAG_FRI_NO_MUL_X 1u   /* prove_openings does NOT multiply the final polynomial by X (plonky2 before PR 436) */
#define GLP_FLAG_POW_DUPLEX 2u     /* PoW response = challenger.observe(w); get_challenge() (plonky2 >= mid 2023) instead of
                                      hash_no_pad(get_hash() || w)[0] with w not observed */
#define GLP_FLAG_DIGEST_MASK 0x300u        /* circuit_digest composition in glp_circuit_build: */
#define GLP_FLAG_DIGEST_CAP 0x000u         /*   hash_no_pad(cap)                                   (f99ed9c)      */
#define GLP_FLAG_DIGEST_CAP_DEG 0x100u     /*   hash_no_pad(cap || degree_bits)                    (early 2023)   */
#define GLP_FLAG_DIGEST_CAP_DOM_DEG 0x200u /*   hash_no_pad(cap || hash_pad([]) || degree_bits)    (>= mid 2023)  */
/* config.zero_knowledge (FriParams::hiding): the wires, Z / partial-product and quotient commitments of a proof are blinded
 * (PlonkOracle::{WIRES, ZS_PARTIAL_PRODUCTS, QUOTIENT}.blinding; CONSTANTS_SIGMAS never is): their query openings carry
 * GLP_SALT_SIZE salt words per leaf, which FriInitialTreeProof::unsalted_eval skips.  (The blinding ROWS of a zero-knowledge
 * circuit — CircuitBuilder::blind_and_pad — are circuit construction: the host's, like every other gate row.) */
#define GLP_FLAG_ZERO_KNOWLEDGE 0x1000u
typedef struct {
    uint32_t degree_bits, rate_bits, cap_height, proof_of_work_bits, num_query_rounds;
    uint32_t n_reduction_rounds;
    uint32_t reduction_arity_bits[16]; /* FriParams::reduction_arity_bits */
    uint32_t flags;                    /* GLP_FLAG_* */
} glp_fri_params;
typedef struct {                 /* FriBatchInfo */
    uint64_t point[2];           /* opening point */
    size_t n_polys;
    const uint32_t *oracle_index; /* FriPolynomialInfo::{oracle_index, polynomial_index} */
    const uint32_t *poly_index;
    const uint64_t *opened_values; /* n_polys extension values f(point) — the FriOpeningBatch the caller has already
                                      computed and observed; NULL = evaluate here */
} glp_fri_batch;
/* u64 words of the flat FriProof (field order of plonky2's struct):
 *   commit_phase_merkle_caps[rounds][2^cap][4] | query_round_proofs[num_query_rounds]{ per oracle: leaf row, path ;
 *   per round: evals[arity][2], path } | final_poly[][2] | pow_witness */
size_t glp_fri_proof_len(const glp_fri_params *params, const size_t *oracle_n_cols, size_t n_oracles);
/* The challenger must have observed the openings; on return it is in the state plonky2's prover leaves it in.
 * pow_witness: plonky2 accepts any valid witness (rayon find_any); this returns the smallest. */
int32_t glp_fri_prove(glp_ctx *ctx, const glp_batch *const *oracles, size_t n_oracles, const glp_fri_batch *batches,
                      size_t n_batches, const glp_fri_params *params, glp_challenger *challenger, uint64_t *proof_out);


/* ---- PLONK layer (plonky2 src/plonk/prover.rs::prove; CircuitBuilder::build's commitment step) ----
 * Gate set supported by the quotient kernel: NoopGate, ConstantGate, PublicInputGate, ArithmeticGate, PoseidonGate,
 * BaseSumGate<2>, ArithmeticExtensionGate, MulExtensionGate — the stock gates of the reference's non-recursive circuits
 * (zkdsa, SMT / Merkle gadgets, merge-and-purge) — and the six gates `verify_proof` adds to the recursive ones
 * (RandomAccessGate, ReducingGate, ReducingExtensionGate, ExponentiationGate, PoseidonMdsGate, CosetInterpolationGate).
 * Up to 16 distinct gates per circuit; anything else -> GLP_ERR_UNSUPPORTED. */
#define GLP_GATE_NOOP 0
#define GLP_GATE_CONSTANT 1     /* param = num_consts */
#define GLP_GATE_PUBLIC_INPUT 2
#define GLP_GATE_ARITHMETIC 3   /* param = num_ops */
#define GLP_GATE_POSEIDON 4
#define GLP_GATE_BASE_SUM 5       /* BaseSumGate<2>, param = num_limbs */
#define GLP_GATE_ARITHMETIC_EXT 6 /* ArithmeticExtensionGate, param = num_ops */
#define GLP_GATE_MUL_EXT 7        /* MulExtensionGate, param = num_ops */
/* the gates builder.verify_proof adds (reference src/recursion/gadgets/mod.rs:114-127) */
#define GLP_GATE_RANDOM_ACCESS 8   /* RandomAccessGate, param = bits | num_copies << 8 | num_extra_constants << 16, bits <= 4 */
#define GLP_GATE_REDUCING 9        /* ReducingGate<2>, param = num_coeffs */
#define GLP_GATE_REDUCING_EXT 10   /* ReducingExtensionGate<2>, param = num_coeffs */
#define GLP_GATE_EXPONENTIATION 11 /* ExponentiationGate, param = num_power_bits */
#define GLP_GATE_POSEIDON_MDS 12   /* PoseidonMdsGate */
#define GLP_GATE_COSET_INTERPOLATION 13 /* CosetInterpolationGate, param = subgroup_bits | degree << 8, subgroup_bits <= 4 */
typedef struct {
    uint32_t id;                     /* GLP_GATE_* */
    uint32_t selector_index;         /* SelectorsInfo::selector_indices[i] */
    uint32_t group_start, group_end; /* SelectorsInfo::groups[selector_index] */
    uint32_t param;
} glp_gate;
typedef struct {                     /* the arithmetic-relevant part of CommonCircuitData */
    uint32_t degree_bits;
    uint32_t num_wires, num_routed_wires;
    uint32_t num_constants;          /* selectors + config.num_constants */
    uint32_t num_selectors;
    uint32_t num_challenges, quotient_degree_factor, num_partial_products;
    uint32_t num_gates;
    uint32_t num_gate_constraints;
    const glp_gate *gates;           /* common_data.gates order: sorted by (degree, id) */
    const uint64_t *k_is;            /* num_routed_wires coset shifts */
    uint64_t circuit_digest[4];      /* filled by glp_circuit_build */
    glp_fri_params fri;
} glp_circuit;
typedef struct glp_circuit_data glp_circuit_data; /* ProverOnlyCircuitData on the device: constants_sigmas batch + sigmas */
/* The GPU part of CircuitBuilder::build: constants_sigmas_commitment = PolynomialBatch::from_values(constants ++ sigmas),
 * circuit_digest = hash_no_pad(cap) (or the composition common->fri.flags selects).  constants_sigmas_values: host, (num_constants + num_routed_wires) columns of 2^degree_bits. */
/* The argument check glp_circuit_build applies (host only, no GPU needed): every wire / constant index a gate of `common`
 * touches must exist, parameters must be in the range the quotient kernel is written for.  GLP_OK, GLP_ERR_INVALID or
 * GLP_ERR_UNSUPPORTED. */
int32_t glp_circuit_validate(const glp_circuit *common);
int32_t glp_circuit_build(glp_ctx *ctx, const glp_circuit *common, const uint64_t *constants_sigmas_values,
                          glp_circuit_data **data_out, uint64_t *cap_out, uint64_t digest_out[4]);
/* Replace the circuit digest the prover observes first (CircuitData::prove reads common_data.circuit_digest, which the Rust
 * host computes itself in CircuitBuilder::build): a host that has its own digest hands it over instead of relying on the
 * composition selected by GLP_FLAG_DIGEST_*. */
int32_t glp_circuit_set_digest(glp_circuit_data *data, const uint64_t digest[4]);
void glp_circuit_free(glp_circuit_data *data);
const glp_batch *glp_circuit_constants_sigmas(const glp_circuit_data *data);
/* flat proof: wires_cap | plonk_zs_partial_products_cap | quotient_polys_cap | OpeningSet (constants, plonk_sigmas,
 * wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys; extension values) | FriProof (glp_fri_proof_len) */
size_t glp_proof_len(const glp_circuit *common);
/* plonk::prover::prove after witness generation: wire_values = the full witness, num_wires columns of 2^degree_bits
 * (witness.wire_values), host or device resident. */
int32_t glp_prove(glp_ctx *ctx, const glp_circuit_data *data, const uint64_t *wire_values, const uint64_t *public_inputs,
                  size_t n_public_inputs, uint64_t *proof_out);
/* The same with the witness as plonky2 holds it when witness generation ends — `witness.wire_values: Vec<Vec<F>>`, ONE ALLOCATION PER
 * WIRE COLUMN (PartitionWitness::full_witness; consumed by prove at reference src/rollup/circuits/mod.rs:1247,
 * src/zkdsa/circuits/mod.rs:326, src/transaction/circuits/mod.rs:453): wire_columns[j] = host pointer to column j's 2^degree_bits
 * values, n_columns = num_wires (anything else: GLP_ERR_INVALID).  Pageable memory is the expected case: the library's worker pool
 * copies the columns through a ring of pinned slots while the previous slot is on the bus, and the wires commitment (inverse NTT,
 * LDE, leaf sponge in 8-column blocks) follows the chunks as they land, so a 2^20-row witness (1.13 GB) costs the proof little more
 * than its first sixteen columns' transfer.  No host-side flattening is needed.  Same proof as glp_prove of the flattened witness. */
int32_t glp_prove_cols(glp_ctx *ctx, const glp_circuit_data *data, const uint64_t *const *wire_columns, size_t n_columns,
                       const uint64_t *public_inputs, size_t n_public_inputs, uint64_t *proof_out);
/* col_stride (words between the starts of two witness columns) must be 2^degree_bits — columns back to back, as witness.wire_values
 * flattened; any other value: GLP_ERR_UNSUPPORTED before any work is done.  8-byte alignment of the pointer is enough. */
int32_t glp_prove_dev(glp_ctx *ctx, const glp_circuit_data *data, const uint64_t *wire_values_dev, size_t col_stride,
                      const uint64_t *public_inputs, size_t n_public_inputs, uint64_t *proof_out);
/* The same under config.zero_knowledge (common.fri.flags & GLP_FLAG_ZERO_KNOWLEDGE) with the CALLER's salts: salts =
 * [3][GLP_SALT_SIZE][2^(degree_bits + rate_bits)] — the salt columns of the wires, Z / partial-product and quotient commitments, each
 * in natural row order (what plonky2's three lde_values calls draw from thread_rng, in proof order); host pointers for
 * glp_prove_salted, device pointers for glp_prove_salted_dev.  glp_prove / glp_prove_dev / _batch_dev / _stream on such a circuit
 * draw the salts themselves (glp_ctx_set_blinding_seed).  GLP_ERR_INVALID when the circuit is not configured for zero knowledge. */
int32_t glp_prove_salted(glp_ctx *ctx, const glp_circuit_data *data, const uint64_t *wire_values, const uint64_t *public_inputs,
                         size_t n_public_inputs, const uint64_t *salts, uint64_t *proof_out);
int32_t glp_prove_salted_dev(glp_ctx *ctx, const glp_circuit_data *data, const uint64_t *wire_values_dev, size_t col_stride,
                             const uint64_t *public_inputs, size_t n_public_inputs, const uint64_t *salts_dev, uint64_t *proof_out);
/* n_proofs proofs of ONE circuit in one call — the reference proves streams of small same-circuit proofs (per-tx user and
 * signature proofs, src/bin/block_circuit.rs:293-361,410-434, feeding src/rollup/circuits/mod.rs:1272-1326), whose circuits
 * (2^4 .. 2^16 rows) leave a GPU mostly idle one at a time.  The witnesses are proven together: one launch sequence over
 * n_proofs x C columns (each commitment a forest of n_proofs Merkle trees), the n_proofs transcripts advanced side by side.
 * Proof k's witness starts at wire_values_dev + k * proof_stride (num_wires columns, col_stride apart); public_inputs:
 * n_proofs x n_public_inputs; proofs_out: n_proofs x glp_proof_len(common) words.  Each proof is word for word the proof
 * glp_prove_dev gives for that witness (glp_prove_dev IS this call with n_proofs = 1). */
int32_t glp_prove_batch_dev(glp_ctx *ctx, const glp_circuit_data *data, size_t n_proofs, const uint64_t *wire_values_dev,
                            size_t col_stride, size_t proof_stride, const uint64_t *public_inputs, size_t n_public_inputs,
                            uint64_t *proofs_out);
/* The same with every witness as plonky2 holds it: wire_columns = n_proofs x n_columns host pointers (witness k's columns at
 * wire_columns + k * n_columns, each 2^degree_bits values, pageable or pinned), n_columns = num_wires.  The library stages the columns
 * through its pinned ring into one device array and proves them together; no flattened copy on the caller's side. */
int32_t glp_prove_batch_cols(glp_ctx *ctx, const glp_circuit_data *data, size_t n_proofs, const uint64_t *const *wire_columns,
                             size_t n_columns, const uint64_t *public_inputs, size_t n_public_inputs, uint64_t *proofs_out);
/* A stream of host-resident witnesses (the reference generates the witness on the host and then proves,
 * src/rollup/circuits/mod.rs:1223-1260): proof k is proven while witness k + 1 travels to the GPU on a second stream into the
 * other of two staging buffers, so the H2D copy (1.13 GB for a 2^20-row witness) leaves the critical path.  wire_values[k]:
 * host pointer to proof k's num_wires x 2^degree_bits witness — PINNED memory (glp_host_alloc, hipHostMalloc, hipHostRegister)
 * for the copy to overlap; pageable memory works but copies synchronously.  Small circuits are proven in groups of up to
 * 2^19 / 2^degree_bits (at most 1024) witnesses through glp_prove_batch_dev, group g + 1 uploaded while group g proves —
 * the reference's per-transaction proofs (src/bin/block_circuit.rs:293-361) at the batched rate without device-resident
 * witnesses.  Same proofs as n_proofs calls of glp_prove. */
int32_t glp_prove_stream(glp_ctx *ctx, const glp_circuit_data *data, size_t n_proofs, const uint64_t *const *wire_values,
                         const uint64_t *public_inputs, size_t n_public_inputs, uint64_t *proofs_out);
/* pinned host memory for witnesses (hipHostMalloc / hipHostFree) */
int32_t glp_host_alloc(size_t bytes, void **ptr_out);
int32_t glp_host_free(void *ptr);
/* building blocks, exposed for parity tests and for hosts that keep plonky2's own control flow:
 * all_wires_permutation_partial_products: out_dev = [Z_0.., pp(ch 0).., pp(ch 1)..] columns of 2^degree_bits */
int32_t glp_partial_products_dev(glp_ctx *ctx, const glp_circuit_data *data, const uint64_t *wire_values_dev,
                                 size_t col_stride, const uint64_t *betas, const uint64_t *gammas, uint64_t *out_dev);
/* compute_quotient_polys + chunking: out_dev = num_challenges * quotient_degree_factor coefficient columns */
int32_t glp_quotient_dev(glp_ctx *ctx, const glp_circuit_data *data, const glp_batch *wires, const glp_batch *zs_pp,
                         const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                         const uint64_t public_inputs_hash[4], uint64_t *out_dev);
/* PoseidonGate witness rows (gates/poseidon.rs::PoseidonGenerator, swap = 0): inputs_dev[n_rows][12] ->
 * elem(row r, wire c) = rows_dev[r * row_stride + c * col_stride], 135 wires */
int32_t glp_poseidon_gate_witness_dev(glp_ctx *ctx, const uint64_t *inputs_dev, size_t n_rows, uint64_t *rows_dev,
                                      size_t row_stride, size_t col_stride);

/* ---- primitives on device buffers (BASELINE configs[1] microbench; also what fft.rs / merkle_tree.rs bind) ---- */
/* fft / ifft / coset_fft / coset_ifft of field/src/fft.rs + polynomial/mod.rs, natural order in and out, in place.
 * inverse = 0: values[i] = sum_j c_j (shift * w^i)^j ; inverse = 1: the inverse map.  shift = 1 for plain (i)fft. */
int32_t glp_ntt_dev(glp_ctx *ctx, uint64_t *data_dev, size_t col_stride, size_t n_cols, uint32_t log_n,
                    int32_t inverse, uint64_t shift);
/* MerkleTree::new(leaves, cap_height): elem(i, c) = leaves_dev[i * row_stride + c * col_stride];
 * digests_dev must hold glp_merkle_digests_len(n_leaves, cap_height) u64; cap_out optional host buffer. */
size_t glp_merkle_digests_len(size_t n_leaves, uint32_t cap_height);
int32_t glp_merkle_build_dev(glp_ctx *ctx, const uint64_t *leaves_dev, size_t n_leaves, size_t leaf_len,
                             size_t row_stride, size_t col_stride, uint32_t cap_height, uint64_t *digests_dev,
                             uint64_t *cap_out);
/* n independent Poseidon permutations, states_dev[i][12] in place, each applied `reps` times */
int32_t glp_poseidon_permute_dev(glp_ctx *ctx, uint64_t *states_dev, size_t n, uint32_t reps);
/* Element-wise GoldilocksField arithmetic (field/src/goldilocks_field.rs: Add / Sub / Mul on GoldilocksField(u64), any u64
 * representative accepted) with the very device routines every kernel uses — the hand-scheduled gfx950 multiply among them:
 * out[i] = a[i] op b[i], canonical.  GLP_FIELD_SHL: a[i] * 2^b[i] through the compile-time shift multiply (b[i] must be one
 * of 1, 3, 12, 24, 31, 32, 33, 36, 48, 60, 63, 64, 65, 72, 84, 95);  GLP_FIELD_MDS_COMBINE: a[i] + 2^32 b[i] (a, b < 2^57).
 * The CANONICAL forms the NTT passes compute with (inputs of BFLY_* must be < p; every result is < p before the final
 * canonicalisation, which the test therefore cannot hide behind): GLP_FIELD_BFLY_SUM / _DIFF = the two outputs of the
 * ten-instruction butterfly (a + b, a - b), GLP_FIELD_MULC = a * b, GLP_FIELD_SHLC = a * 2^b (same shifts as GLP_FIELD_SHL);
 * for these four the kernel stores the raw result — NOT canonicalised — so a value >= p shows. */
#define GLP_FIELD_ADD 0
#define GLP_FIELD_SUB 1
#define GLP_FIELD_MUL 2
#define GLP_FIELD_SHL 3
#define GLP_FIELD_MDS_COMBINE 4
#define GLP_FIELD_BFLY_SUM 5
#define GLP_FIELD_BFLY_DIFF 6
#define GLP_FIELD_MULC 7
#define GLP_FIELD_SHLC 8
int32_t glp_field_op_dev(glp_ctx *ctx, uint32_t op, const uint64_t *a_dev, const uint64_t *b_dev, size_t n,
                         uint64_t *out_dev);

/* ---- native tree hashing: the Poseidon work the reference does on the HOST to build witnesses ----
 * All digests are 4 u64; *_dev pointers are device memory of this context's GPU; batches of any size (0 is a no-op).
 * Results are canonical field elements; inputs may be any u64 representative. */
#define GLP_HASH_NO_PAD 0  /* PoseidonHash::hash_no_pad                                                            */
#define GLP_HASH_PAD 1     /* PoseidonHash::hash_pad  (reference src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:170) */
#define GLP_HASH_OR_NOOP 2 /* PoseidonHash::hash_or_noop (<= 4 elements are packed, not hashed)                     */
/* out[i] = hash(in[i * item_stride .. + len)) */
int32_t glp_hash_batch_dev(glp_ctx *ctx, const uint64_t *in_dev, size_t n, size_t len, size_t item_stride, uint32_t mode,
                           uint64_t *out_dev);
/* out[i] = PoseidonHash::two_to_one(left[i], right[i])   (reference src/merkle_tree/tree.rs:79,90,123) */
int32_t glp_two_to_one_batch_dev(glp_ctx *ctx, const uint64_t *left_dev, const uint64_t *right_dev, size_t n,
                                 uint64_t *out_dev);
/* PoseidonNodeHash::calc_node_hash(Node::Leaf(key, value)) = hash_pad(key ++ value ++ [1])   (…/goldilocks_poseidon/mod.rs:167-181) */
int32_t glp_smt_leaf_hash_batch_dev(glp_ctx *ctx, const uint64_t *keys_dev, const uint64_t *values_dev, size_t n,
                                    uint64_t *out_dev);
/* climb n sparse-Merkle paths to their roots: item i starts from start[i] at depth n_siblings[i] (NULL: all have
 * max_siblings), siblings[i][max_siblings][4] with [0] just under the root; key bit lvl (LE limbs, LSB first) = 1 puts the
 * running hash on the right   (reference src/sparse_merkle_tree/tree.rs:208-225, …/goldilocks_poseidon/mod.rs:27-47) */
int32_t glp_smt_root_batch_dev(glp_ctx *ctx, const uint64_t *keys_dev, const uint64_t *start_dev, const uint64_t *siblings_dev,
                               const uint32_t *n_siblings_dev, size_t n, size_t max_siblings, uint64_t *out_dev);
/* get_merkle_root(index, value, siblings) for n proofs of one depth; siblings[i][depth][4], leaf level first
 * (reference src/merkle_tree/tree.rs:111-128) */
int32_t glp_merkle_root_batch_dev(glp_ctx *ctx, const uint64_t *index_dev, const uint64_t *value_dev,
                                  const uint64_t *siblings_dev, size_t n, size_t depth, uint64_t *out_dev);
/* get_block_hash for n headers of 29 u64: block_number, then prev_block_hash, block_headers_digest, transactions_digest,
 * deposit_digest, proposed_world_state_digest, approved_world_state_digest, latest_account_digest
 * (reference src/transaction/block_header.rs:157-174) */
int32_t glp_block_hash_batch_dev(glp_ctx *ctx, const uint64_t *headers_dev, size_t n, uint64_t *out_dev);
/* get_merkle_proof_with_zero(leaves, index, depth, zero) (reference src/merkle_tree/tree.rs:49-100): leaves_dev[n_leaves][4]
 * on the device (n_leaves = 0: one `zero` leaf); siblings_out = glp_merkle_proof_n_siblings(depth) digests (host, leaf level
 * first), root_out = 4 u64 (host); either may be NULL.  GLP_ERR_INVALID where the reference asserts. */
size_t glp_merkle_proof_n_siblings(size_t depth);
int32_t glp_merkle_proof_with_zero_dev(glp_ctx *ctx, const uint64_t *leaves_dev, size_t n_leaves, size_t index, size_t depth,
                                       const uint64_t zero[4], uint64_t *siblings_out, uint64_t *root_out);
/* Root of the sparse Merkle tree holding exactly the given entries (HOST arrays keys[n][4], values[n][4]; entries with
 * value 0^4 are absent, as after a `set` to zero).  Equals the root after any sequence of insert/update/remove leading
 * to that set (reference src/sparse_merkle_tree/tree.rs:255-520).  GLP_ERR_INVALID on a duplicate key. */
int32_t glp_smt_root_of_set(glp_ctx *ctx, const uint64_t *keys, const uint64_t *values, size_t n, uint64_t root_out[4]);

/* SparseMerkleTree::find (calc_inclusion_proof; reference src/sparse_merkle_tree/tree.rs:157-163,588-690) for n_queries keys
 * against the tree holding exactly the given entries (HOST arrays, as for glp_smt_root_of_set) — the sibling lists the
 * reference's witness builders need per transaction (inclusion / non-inclusion of sender and asset keys; the `siblings` the
 * process proofs of proof/process.rs:153-337 carry are these), produced for a whole batch by one walk kernel over the tree
 * glp_smt_root_of_set builds on the device.  Per query i (every output is a HOST array and may be NULL):
 *   siblings_out[i][max_siblings][4]  SparseMerkleInclusionProof::siblings, [0] just under the root (the order
 *                                     glp_smt_root_batch_dev takes), unary nodes' 0^4 siblings included; rows >= n_siblings_out[i]
 *                                     are 0^4
 *   n_siblings_out[i]                 number of siblings (<= 256)
 *   flags_out[i]                      bit 0 = found, bit 1 = is_old0 (the walk ended in an empty subtree)
 *   value_out[i][4]                   the value stored under the key (found), else 0^4
 *   not_found_key_out / not_found_value_out [i][4]   the OTHER leaf the walk ended on (not found, is_old0 = 0), else 0^4
 * root_out: the root (0^4 for an empty set).  max_siblings <= 256; GLP_ERR_INVALID when a proof needs more than max_siblings
 * siblings — NO output array is written in that case, not even n_siblings_out: call with siblings_out = NULL first (the
 * lengths-only form never overflows) to learn n_siblings_out and size the array — or on a duplicate key. */
#define GLP_SMT_FOUND 1u
#define GLP_SMT_IS_OLD0 2u
int32_t glp_smt_proofs_of_set(glp_ctx *ctx, const uint64_t *keys, const uint64_t *values, size_t n, const uint64_t *query_keys,
                              size_t n_queries, size_t max_siblings, uint64_t *siblings_out, uint32_t *n_siblings_out,
                              uint32_t *flags_out, uint64_t *value_out, uint64_t *not_found_key_out,
                              uint64_t *not_found_value_out, uint64_t root_out[4]);

/* k sequential SparseMerkleTree::set(key_i, value_i) calls (reference src/sparse_merkle_tree/tree.rs:143-151 -> calc_process_proof
 * :561-585: a zero value removes or is a no-op, a non-zero one updates or inserts) applied IN ORDER to the tree holding the base
 * entries — what a block producer does to its world-state and account trees, once per transaction — as one batch: the tree of the
 * set is built on the device, the k walks are replayed structurally on the host, and all the hashing they need runs as
 * ~tree-height launches for the whole batch instead of k x depth dependent permutations.  Per update i (HOST arrays, any may be
 * NULL) the `find` the reference's `set` starts from, on the tree BEFORE the update:
 *   siblings_out[i][max_siblings][4], n_siblings_out[i]   as for glp_smt_proofs_of_set (rows behind the proof are 0^4)
 *   flags_out[i]            GLP_SMT_FOUND | GLP_SMT_IS_OLD0 | GLP_SMT_LAST_SIBLING_IS_LEAF
 *   old_value_out[i][4]     the value stored under key_i (found)
 *   other_key_out / other_value_out [i][4]                  the other leaf the walk ended on (not found, not is_old0)
 *   last_sibling_key_out / last_sibling_value_out [i][4]    the entry of the deepest sibling when that sibling is a lone leaf
 *                           (what `remove` reports as new_key / new_value: tree.rs:410-421)
 * and roots_out[k + 1][4]: the root before update 0, then after every update (old_root / new_root of update i = rows i, i + 1).
 * From these the fields of SparseMerkleProcessProof follow without hashing (host mirrors: trees.py::smt_apply_updates,
 * intmax_trees.hpp::smt_apply_updates).  GLP_ERR_INVALID: duplicate BASE key, or a proof longer than
 * max_siblings (pass siblings_out = NULL to learn the lengths). */
#define GLP_SMT_LAST_SIBLING_IS_LEAF 4u
int32_t glp_smt_apply_updates(glp_ctx *ctx, const uint64_t *keys, const uint64_t *values, size_t n, const uint64_t *upd_keys,
                              const uint64_t *upd_values, size_t k, size_t max_siblings, uint64_t *siblings_out,
                              uint32_t *n_siblings_out, uint32_t *flags_out, uint64_t *old_value_out, uint64_t *other_key_out,
                              uint64_t *other_value_out, uint64_t *last_sibling_key_out, uint64_t *last_sibling_value_out,
                              uint64_t *roots_out);

/* glp_smt_apply_updates for n_trees INDEPENDENT small trees in one call: tree s holds the base entries
 * [set_offsets[s], set_offsets[s + 1]) of keys / values and receives the updates [upd_offsets[s], upd_offsets[s + 1]) of upd_keys /
 * upd_values, in order (both offset arrays have n_trees + 1 entries; HOST arrays).  The per-update outputs are laid out in the order
 * of the update arrays (K = upd_offsets[n_trees] rows, as for glp_smt_apply_updates); roots_out has K + n_trees rows: tree s
 * occupies rows upd_offsets[s] + s .. upd_offsets[s + 1] + s (its root before its first update, then after each).  For the
 * per-transaction trees of block production — a transaction's outputs inserted into its (empty) diff tree, its spent assets
 * removed from the sender's asset tree: the block circuit consumes the process proof of every such `set` (reference
 * src/transaction/gadgets/purge/mod.rs, src/sparse_merkle_tree/layered_layered_tree.rs:124-160); layered trees take one call per
 * layer (a lower layer's roots_out are the next layer's update values).  Shapes are replayed on the host, all hashing runs as one
 * levelled pass on the device. */
int32_t glp_smt_apply_updates_multi(glp_ctx *ctx, const uint64_t *keys, const uint64_t *values, const size_t *set_offsets,
                                    const uint64_t *upd_keys, const uint64_t *upd_values, const size_t *upd_offsets, size_t n_trees,
                                    size_t max_siblings, uint64_t *siblings_out, uint32_t *n_siblings_out, uint32_t *flags_out,
                                    uint64_t *old_value_out, uint64_t *other_key_out, uint64_t *other_value_out,
                                    uint64_t *last_sibling_key_out, uint64_t *last_sibling_value_out, uint64_t *roots_out);

/* The roots of n_sets INDEPENDENT sparse trees in one call: set s holds the entries [set_offsets[s], set_offsets[s + 1]) of the HOST
 * arrays keys[.][4] / values[.][4] (value 0^4 = absent; an empty set has root 0^4); set_offsets has n_sets + 1 entries.  For the
 * many small trees of block production — a transaction's LayeredLayered diff tree is three layers of them (reference
 * src/sparse_merkle_tree/layered_layered_tree.rs:124-160; tx_hash = two_to_one(diff_root, nonce), src/transaction/gadgets/purge/
 * mod.rs:296) and so is a user's asset tree (src/transaction/tree/user_asset.rs:99-150): one call per LAYER for all transactions of
 * a block (a lower layer's roots are the next layer's values) instead of one tree build per (transaction, key).  Each root equals
 * glp_smt_root_of_set of that set; the shapes are worked out on the host and all trees are hashed together, level by level.
 * GLP_ERR_INVALID on a duplicate key inside a set.  roots_out: n_sets x 4 (host). */
int32_t glp_smt_roots_of_sets(glp_ctx *ctx, const uint64_t *keys, const uint64_t *values, const size_t *set_offsets, size_t n_sets,
                              uint64_t *roots_out);

/* A sparse Merkle tree that STAYS on the device between calls — the reference's world-state and latest-account trees live as long
 * as the node (`SparseMerkleTree` over its node store, src/sparse_merkle_tree/tree.rs:29-60) and take one batch of `set`s per block
 * (src/bin/block_circuit.rs:444-648).  glp_smt_open = glp_smt_root_of_set's build, kept; glp_smt_find = glp_smt_proofs_of_set
 * without the build; glp_smt_update = glp_smt_apply_updates without uploading, sorting and hashing the whole entry set again (same
 * outputs, same layouts) — afterwards the tree holds the updated set: changed values, removed entries (value 0^4) and new keys are
 * patched into the entry arrays on the device and the tree is rebuilt from them.  On an error the tree is unchanged (bulk output
 * buffers may have been written).  glp_smt_entries returns the current set in tree order (n = glp_smt_len entries): a checkpoint.
 * The tree belongs to the context it was opened with (single-threaded like it) and keeps it alive until glp_smt_close. */
typedef struct glp_smt glp_smt;
int32_t glp_smt_open(glp_ctx *ctx, const uint64_t *keys, const uint64_t *values, size_t n, glp_smt **tree_out);
void glp_smt_close(glp_smt *tree);
int32_t glp_smt_len(const glp_smt *tree, size_t *n_entries_out);
int32_t glp_smt_root(glp_smt *tree, uint64_t root_out[4]);
int32_t glp_smt_entries(glp_smt *tree, uint64_t *keys_out, uint64_t *values_out);
int32_t glp_smt_find(glp_smt *tree, const uint64_t *query_keys, size_t n_queries, size_t max_siblings, uint64_t *siblings_out,
                     uint32_t *n_siblings_out, uint32_t *flags_out, uint64_t *value_out, uint64_t *not_found_key_out,
                     uint64_t *not_found_value_out, uint64_t root_out[4]);
int32_t glp_smt_update(glp_smt *tree, const uint64_t *upd_keys, const uint64_t *upd_values, size_t n_updates, size_t max_siblings,
                       uint64_t *siblings_out, uint32_t *n_siblings_out, uint32_t *flags_out, uint64_t *old_value_out,
                       uint64_t *other_key_out, uint64_t *other_value_out, uint64_t *last_sibling_key_out,
                       uint64_t *last_sibling_value_out, uint64_t *roots_out);

/* plain device memory helpers for non-HIP hosts (Rust): the Python host uses torch tensors instead */
int32_t glp_dev_alloc(glp_ctx *ctx, size_t bytes, void **ptr_out);
int32_t glp_dev_free(glp_ctx *ctx, void *ptr);
int32_t glp_dev_upload(glp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int32_t glp_dev_download(glp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* GLPROVER_H */
