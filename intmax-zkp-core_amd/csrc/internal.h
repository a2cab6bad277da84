// internal.h — shared internals of libglprover.so (context, device pool, per-kernel timing, NTT drivers).
// Not part of the C ABI (include/glprover.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <chrono>
#include <cstdio>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/glprover.h"
#include "gl_field.h"

namespace glpi {

extern thread_local int32_t g_last_hip;
// A failed HIP call also stays behind as the runtime's per-thread "last error" until somebody reads it: the next
// `hipGetLastError()` after a kernel launch — of this call or of the NEXT one on the thread — would report it as that launch's
// failure.  Every error the library handles (returned as a status, or tolerated: a hipMalloc retried after dropping the cache, a pinned
// allocation replaced by a pageable one) is therefore read off here.
inline void forget_hip_error() { (void)hipGetLastError(); }
inline void note_hip_error(hipError_t e) {
    g_last_hip = (int32_t)e;
    forget_hip_error();
}
// Fault injection (debug aid, glp_ctx_debug_fail_at / GLP_HIP_FAIL_AT): the k-th HIP call checked by HIP_TRY on this thread from
// now on is NOT made and reports hipErrorLaunchFailure instead — one shot.  A context is single-threaded, so "this thread" is
// "this context's caller".  g_hip_checks counts every checked call (tests read it to know how far a sweep has to go).
extern thread_local uint64_t g_hip_fail_at;
extern thread_local uint64_t g_hip_checks;
inline bool hip_fault_due() {
    ++g_hip_checks;
    return g_hip_fail_at && --g_hip_fail_at == 0;
}

#define HIP_TRY(expr)                              \
    do {                                           \
        hipError_t _e = glpi::hip_fault_due() ? hipErrorLaunchFailure : (expr); \
        if (_e != hipSuccess) {                    \
            glpi::note_hip_error(_e);              \
            return _e == hipErrorOutOfMemory ? GLP_ERR_OOM : GLP_ERR_HIP; \
        }                                          \
    } while (0)

#define GLP_TRY(expr)                 \
    do {                              \
        int32_t _s = (expr);          \
        if (_s != GLP_OK) return _s;  \
    } while (0)

constexpr int MASTER_K = 24;       // master root table w = w_{2^24}: transforms up to 2^24 points
constexpr int SMALL_LOG_MAX = 11;  // whole transform in one workgroup up to 2^11 points
constexpr int TOP_LEVEL_LOG = 5;    // tree levels with at most 2^this children are fused into one workgroup
constexpr int COOP_LEVEL_LOG = 15;  // levels with at most 2^this parents use the cooperative (16 lanes per node) permutation
constexpr int SUBTREE_LOG = 5;      // a cap subtree with at most 2^this children on a level finishes in ONE workgroup (subtree_levels_kernel,
                                    // GLP_SUBTREE_LOG; -1: never).  Swept in profiles/r05_subtree_levels.md: 5 is the best, 8 and up lose ...
constexpr int SUBTREE_FOREST_LOG = 13;  // ... when the forest has at most 2^this parents on that level (more: throughput matters, the wide forms stay)
constexpr int COOP4_LEVEL_LOG = 13; // ... of which those with more than 2^this parents the four-lanes-per-node form
constexpr size_t COOP_LEAF_MAX = 4096;    // leaf sets up to this size use the cooperative (16 lanes per leaf) sponge
constexpr size_t COOP4_LEAF_MAX = 65536;  // ... and up to this size the four-lanes-per-leaf sponge (poseidon_coop4.h); round 6: 32768 -> 65536
                                          // (tools/threshold_ab.py: three 2^11-row proofs per call — 49 152 leaves — 3.15 -> 2.90-2.98 ms, 2^13 rows even; 2^17: loses)

struct CosetPlan {
    int log_n, rate_bits;
    uint64_t shift;
    int s_lo, r_bits;       // geometry of the first DIF pass
    uint64_t* scale_lo_dev; // [2^rate_bits][2^s_lo]
    uint64_t* scale_hi_dev; // [2^rate_bits][2^r_bits]
    uint64_t* scale_full_dev;  // [2^rate_bits][2^log_n] = scale_hi * scale_lo, or null when it would be too large
};

// where an arriving witness comes from (glp_ctx::feed)
constexpr int RING_SLOTS = 8;
constexpr size_t RING_SLOT_BYTES = (size_t)16 << 20;
struct ArrivalFeed {
    const uint64_t* const* cols = nullptr;  // one host pointer per witness column (glp_prove_cols), or
    const uint64_t* flat = nullptr;         // one contiguous num_wires x n host array (glp_prove, glp_prove_stream)
    bool pinned = false;                    // flat and pinned: copied straight out of the caller's buffer
    uint64_t* dst = nullptr;                // the device array, columns n words apart
    char* ring_base = nullptr;              // the pinned slots this transfer is staged through (the context's ring, or its two transfer halves)
    size_t ring_slots = 0;
    size_t n = 0, fed = 0;                  // words per column; chunks enqueued so far
    std::chrono::steady_clock::time_point t0;  // GLP_TRACE_FEED: when the witness was announced
};

struct PassPlan {
    int log_e;
    std::vector<std::pair<int, int>> passes;  // (s_lo, r_bits), DIF order (high bits first)
};

}  // namespace glpi

struct glp_ctx {
    // Lifetime: the creator holds one reference; every glp_batch, glp_circuit_data and glp_dev_alloc block holds one more.
    // glp_ctx_destroy drops the creator's reference; the context is torn down when the last object is gone, so a handle
    // dropped late (or on another thread: Rust `Drop` under rayon) never frees through a dead context.
    std::atomic<int> refs{1};
    std::mutex pool_mu;  // guards pool_free / pool_live / pool_cached_bytes (frees may come from any thread)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int mds = 0;
    uint64_t* w_lo_dev = nullptr;
    uint64_t* w_hi_dev = nullptr;
    std::vector<uint64_t> w_lo, w_hi;  // host copies
    std::unordered_multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_live;
    std::unordered_map<void*, size_t> pool_asked;  // GLP_POOL_CANARY=1: bytes asked for (the guard sits right behind them)
    uint64_t canary_violations = 0;
    uint64_t pool_fail_at = 0;  // fault injection: the k-th pool_alloc from now returns GLP_ERR_OOM (one shot; 0 = off)
    uint64_t pool_allocs = 0;   // pool_alloc calls so far
    size_t pool_cached_bytes = 0;
    hipStream_t copy_stream = nullptr;  // witness uploads of glp_prove_stream (created on first use)
    bool fq_attr_set = false;           // fused quotient kernel: dynamic-LDS limit raised on this context's device
    bool fqp_attr_set = false;  // the same for quotient_fused_persistent_kernel
    bool ab_attr_set = false;   // the same for ntt_pass_ab_kernel
    int n_cus = 0;              // compute units of this context's device (read on first use by the persistent quotient launch)
    // A host witness arriving in column chunks (glp_prove_cols / glp_prove / the first witness of glp_prove_stream, large circuits):
    // columns [arrival_bounds[k], arrival_bounds[k + 1]) of the array at arrival_base are in HBM once arrival_ev[k] has fired.  The
    // next commitment over exactly that array (the wires commitment, the first device work of a proof) transforms AND HASHES chunk k
    // as soon as it is there (the leaf sponge runs in column chunks: merkle_leaf_chunk), so inverse NTT, LDE and most of the
    // hashing run under the upload instead of after it; consumed (cleared) by that commitment.  The copies themselves are
    // enqueued lazily (arrival_feed): a source in the caller's PAGEABLE memory — plonky2's witness.wire_values is one Vec per
    // column — goes through a ring of pinned slots filled by the worker pool, and the next chunk is staged while the GPU works
    // on the previous one.
    std::vector<hipEvent_t> arrival_ev;
    std::vector<size_t> arrival_bounds;
    const uint64_t* arrival_base = nullptr;
    glpi::ArrivalFeed feed;
    char* ring_buf = nullptr;  // RING_SLOTS x RING_SLOT_BYTES of pinned memory (allocated on first use, kept)
    hipEvent_t ring_ev[glpi::RING_SLOTS] = {nullptr};
    bool ring_busy[glpi::RING_SLOTS] = {false};
    size_t ring_next = 0;
    hipStream_t side_stream = nullptr;  // GLP_SUBTREE_PIPELINE (closed experiment): the level chains of finished cap subtrees
    std::vector<hipEvent_t> side_ev;
    // pinned host staging for the small per-call uploads and read-backs (challenge tables, pointer tables, query indices,
    // openings): blocks stay valid until the next arena_reset, so no upload has to be waited for just because a host
    // vector would go out of scope — that alone was ~20 of the 33 stream synchronisations of a proof
    char* arena_base = nullptr;
    size_t arena_cap = 0, arena_off = 0, arena_need = 0;  // need: bytes asked for since the last reset (sizes the next arena)
    std::vector<void*> arena_overflow;
    std::map<std::pair<int, uint64_t>, uint64_t*> power_tables;  // scale_by_powers: (log_n, s) -> device table
    // blinding salts the library draws itself (blinding = 1 without caller salts): ChaCha12 in counter mode under this key
    // (getrandom at creation, glp_ctx_set_blinding_seed replaces it), one nonce per blinded commitment
    uint32_t blinding_key[8] = {0};
    bool blinding_key_ok = false;  // the key came from the kernel's generator or from glp_ctx_set_blinding_seed
    uint64_t blinding_nonce = 0;
    const uint64_t* next_salts_dev = nullptr;  // caller-supplied salts of the NEXT blinded commitment (consumed by it)
    const uint64_t* zk_salts_dev = nullptr;    // caller-supplied salts of the NEXT zero-knowledge proof: [3][GLP_SALT_SIZE][N] (consumed by it)
    int* pinned_word = nullptr;  // one pinned host word for small device -> host read-backs (round widths of the tree builder)
    char* rb_buf = nullptr;      // 64 KB of pinned memory: small read-backs (Merkle caps, final polynomials) land here first — a copy to
                                 // PAGEABLE host memory goes through the runtime's own staging and costs ~10 us more per round trip
                                 // (tools/roundtrip_bench.hip: 29.6 us against 20.1), and a proof makes a dozen
    char* xfer_buf = nullptr;    // 2 x XFER_CHUNK (16 MB) bytes of pinned memory: the caller's PAGEABLE buffers cross PCIe through it (upload_staged)
    std::map<std::tuple<int, int, uint64_t>, glpi::CosetPlan> coset_plans;
    std::map<std::tuple<int, int, int>, uint64_t*> pass_twiddles;  // (s_lo, r_bits, inverse) -> [2^(s_lo + r_bits)]
    // cached powers table of the last evaluation point (openings)
    uint64_t zpow_z[2] = {0, 0};
    int zpow_log_n = -1;
    uint64_t* zpow_dev = nullptr;  // [2][n] limbs (a pool block)
    // timing
    bool timing = false;
    struct Rec { int kind; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> ev_free;
    double ms_acc[GLP_K_COUNT] = {0};
    uint64_t launches[GLP_K_COUNT] = {0};
};

struct glp_batch {
    glp_ctx* ctx;
    size_t n_cols;      // polynomials per tree
    size_t n_trees;     // 1, or K for the batched prover: K independent PolynomialBatches of equal shape in one allocation
    uint32_t log_n, rate_bits, cap_height;
    uint32_t salt_cols; // blinding: GLP_SALT_SIZE extra LDE columns (random, not polynomials) behind the n_cols ones; n_trees == 1 then
    uint64_t* coeffs;   // [n_trees * n_cols][n]
    uint64_t* lde;      // [n_trees * n_cols (+ salt_cols)][N], rows bit-reversed
    uint64_t* digests;  // levels bottom-up, each level across the trees: [level][tree][node][4] (merkle_kernels.h)
    size_t digests_len; // of all trees together
};

namespace glpi {

// words per Merkle leaf of a batch: its polynomials' values and, when blinded, the salt words behind them
inline size_t leaf_width(const glp_batch* b) { return b->n_cols + b->salt_cols; }

// A caller's host buffer -> device and back.  Pinned memory (and anything small) is copied directly; a large PAGEABLE buffer goes
// through the context's pinned transfer buffer in chunks (host copies by the worker pool): asynchronous copies to / from pageable memory
// that is freed soon afterwards were followed by 10 ms stalls of the next device call in the tree entry points; nothing
// of the caller's memory stays registered with the driver this way.  Both return after the data has arrived.
constexpr size_t XFER_CHUNK = (size_t)16 << 20;
int32_t upload_staged(glp_ctx* c, void* dst_dev, const void* src_host, size_t bytes);
int32_t download_staged(glp_ctx* c, void* dst_host, const void* src_dev, size_t bytes);
int32_t read_small(glp_ctx* c, void* dst_host, const void* src_dev, size_t bytes);  // D2H + wait, through c->rb_buf when it fits
// Results a kernel stores straight into pinned host memory (Merkle caps, openings, the final polynomial): the producer writes canonical
// field elements, the host fills the words with MAPPED_SENTINEL (not canonical) before the launch and spins on them afterwards — no copy
// kernel, no stream synchronisation on the Fiat-Shamir round trip (tools/roundtrip_bench.hip: 11 us against 20).  GLP_MAPPED_CAPS=0: off.
constexpr uint64_t MAPPED_SENTINEL = ~(uint64_t)0;
int mapped_readbacks_level();   // GLP_MAPPED_CAPS: 0 off, 1 (default) Merkle caps only, 2 caps + openings + final polynomial (measured slower)
bool mapped_readbacks_enabled();  // level >= 2
bool arena_is_pinned(const glp_ctx* c, const void* p, size_t bytes);  // inside the context's pinned arena (an overflow block may be pageable)
void mapped_prefill(uint64_t* host_words, size_t words);
// all `words` are there: spins (bounded: ~50 ms), then falls back to a stream synchronisation, after which they must be
int32_t mapped_wait(glp_ctx* c, const uint64_t* host_words, size_t words);
void* arena_alloc(glp_ctx* c, size_t bytes);  // pinned host memory, 64-byte aligned, valid until the next arena_reset; NULL only when the host is out of memory (callers return GLP_ERR_OOM)
int32_t arena_reset(glp_ctx* c);              // waits for the stream (pending copies read the arena), then rewinds it
template <typename T>
inline T* arena_alloc_t(glp_ctx* c, size_t count) {
    return static_cast<T*>(arena_alloc(c, (count ? count : 1) * sizeof(T)));
}
int32_t pool_alloc(glp_ctx* c, size_t bytes, void** out);
void pool_free(glp_ctx* c, void* p);
void ctx_retain(glp_ctx* c);
void ctx_release(glp_ctx* c);  // tears the context down when the last reference goes
template <typename T>
inline int32_t pool_alloc_t(glp_ctx* c, size_t count, T** out) {
    void* p = nullptr;
    GLP_TRY(pool_alloc(c, count * sizeof(T), &p));
    *out = static_cast<T*>(p);
    return GLP_OK;
}

// Every entry point that takes a context (or an object that belongs to one) runs with the context's device current and puts the
// caller's device back when it returns.  HIP's current device is PER HOST THREAD and starts at 0: a context of device 3 driven from
// a thread that never called hipSetDevice (a rayon worker of the Rust host, a Python side thread) would otherwise allocate its pool
// blocks and create its events on device 0.  hipGetDevice is a thread-local read; nothing else happens when the device already matches.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device) {
        if (device < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    explicit DeviceScope(const glp_ctx* c) : DeviceScope(c ? c->device : -1) {}
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
    ~DeviceScope() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
};
#define GLP_ON_DEVICE_OF(ctx) glpi::DeviceScope glp_device_scope_(ctx)

// pool blocks that go back to the pool when the scope ends (error paths included)
struct Scratch {
    glp_ctx* c;
    std::vector<void*> ptrs;
    explicit Scratch(glp_ctx* ctx) : c(ctx) {}
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() {
        for (void* p : ptrs) pool_free(c, p);
    }
    int32_t alloc(size_t words, uint64_t** out) {
        GLP_TRY(pool_alloc_t(c, words ? words : 1, out));
        ptrs.push_back(*out);
        return GLP_OK;
    }
};

struct Timed {
    glp_ctx* c;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(glp_ctx* ctx, int k);
    ~Timed();
};

uint64_t host_root_pow(const glp_ctx* c, uint64_t e);
// DIF passes natural -> bit-reversed (optionally the coset-LDE pass), DIT passes bit-reversed -> natural
// natural_dst: the last pass writes NATURAL order there (out of place; only where dif_natural_supported(log_n))
int32_t run_dif(glp_ctx* c, const uint64_t* src, size_t src_stride, uint64_t* dst, size_t dst_stride, size_t n_cols,
                int log_n, int inverse, const CosetPlan* cp, uint64_t out_scale, bool canon_out, uint64_t* natural_dst = nullptr,
                size_t natural_stride = 0);
bool dif_natural_supported(int log_n);
int32_t run_dit(glp_ctx* c, uint64_t* data, size_t stride, size_t n_cols, int log_n, int inverse, uint64_t out_scale);
int32_t get_coset_plan(glp_ctx* c, int log_n, int rate_bits, uint64_t shift, const CosetPlan** out);
int32_t launch_small(glp_ctx* c, const uint64_t* src, size_t src_stride, uint64_t* dst, size_t dst_stride, size_t n_cols,
                     int log_in, int log_out, int inverse, uint64_t pre_shift, uint64_t post_shift, uint64_t out_scale,
                     int bitrev_out, int bitrev_in = 0);
int32_t scale_by_powers(glp_ctx* c, uint64_t* data, size_t stride, size_t n_cols, int log_n, uint64_t s, uint64_t c0);
// coefficients [n_cols][n] (natural) -> LDE [n_cols][n << rate_bits] on the coset shift*<w>, rows bit-reversed
int32_t lde_from_coeffs(glp_ctx* c, const uint64_t* coeffs, uint64_t* lde, size_t n_cols, int log_n, int rate_bits,
                        uint64_t shift);
size_t digests_len(size_t n_leaves, uint32_t cap_height);
// a forest of n_trees trees of n_leaves leaves each (tree t's leaves at leaves + t * tree_stride); digests: n_trees *
// digests_len(n_leaves, cap_height) words, level-major across the trees; cap_out (host, optional): n_trees caps
int32_t merkle_build(glp_ctx* c, const uint64_t* leaves, size_t n_leaves, size_t leaf_len, size_t row_stride,
                     size_t col_stride, uint32_t cap_height, uint64_t* digests, uint64_t* cap_out, size_t n_trees = 1,
                     size_t tree_stride = 0, bool leaves_done = false);
// the leaf sponge in column chunks (one tree): c0 a multiple of 8; every launch but the last absorbs whole 8-word blocks and leaves
// at least 8 words for later; the last one (c0 + cnt == leaf_len) writes the digests.  Then merkle_build(.., leaves_done = true).
bool merkle_chunks_supported(const glp_ctx* c, size_t n_leaves, size_t leaf_len);
int32_t merkle_leaf_chunk(glp_ctx* c, const uint64_t* leaves, size_t n_leaves, size_t leaf_len, size_t c0, size_t cnt,
                          size_t row_stride, size_t col_stride, uint64_t* digests);
// rows + Merkle paths of `indices` (host) into host buffers (either may be null); tree_of (host, optional): the tree each
// index refers to in a forest of n_trees trees (null: tree 0)
int32_t gather_rows(glp_ctx* c, const uint64_t* leaves, size_t row_stride, size_t col_stride, size_t n_cols,
                    const uint64_t* digests, size_t n_leaves, int path_len, const uint64_t* indices, size_t n_idx,
                    uint64_t* rows_out, uint64_t* paths_out, size_t n_trees = 1, size_t tree_stride = 0,
                    const uint32_t* tree_of = nullptr);
// the same without any wait: idx_dev (device) holds the packed indices (tree << 40 | leaf), results go to device buffers
int32_t gather_rows_launch(glp_ctx* c, const uint64_t* leaves, size_t row_stride, size_t col_stride, size_t n_cols,
                           const uint64_t* digests, size_t n_leaves, int path_len, const uint64_t* idx_dev, size_t n_idx,
                           uint64_t* rows_dev, uint64_t* paths_dev, size_t n_trees, size_t tree_stride);
// Announce a host witness (nw columns of n words) that is to arrive in the device array dst (internal.h, glp_ctx::arrival_*):
// exactly one of cols / flat is set.  Large witnesses are cut into column chunks whose copies arrival_feed enqueues on the
// context's copy stream; small ones are copied at once and c->stream waits for them.  arrival_finish: after the proof (error
// paths too) — waits for the copy stream (the caller's buffers are not read after return) and forgets the announcement.
int32_t arrival_announce(glp_ctx* c, uint64_t* dst, const uint64_t* const* cols, const uint64_t* flat, size_t nw, size_t n,
                         bool small_flat_direct);  // small_flat_direct: a small pageable flat array is copied without staging
int32_t arrival_feed(glp_ctx* c, size_t k);  // the copies of chunks 0 .. k are enqueued and their events recorded
void arrival_finish(glp_ctx* c);
// PolynomialBatch of n_trees independent column sets (device input, one allocation out): the batched prover's commitments
int32_t batch_from_values_forest(glp_ctx* c, const uint64_t* cols_dev, size_t col_stride, size_t tree_stride, size_t n_cols,
                                 size_t n_trees, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, glp_batch** batch_out,
                                 uint64_t* caps_out, uint32_t blinding = 0);
int32_t batch_from_coeffs_forest(glp_ctx* c, const uint64_t* coeffs_dev, size_t n_cols, size_t n_trees, uint32_t log_n,
                                 uint32_t rate_bits, uint32_t cap_height, glp_batch** batch_out, uint64_t* caps_out, uint32_t blinding = 0);
// blinded commitments (n_trees == 1) take their salt columns from ctx->next_salts_dev when it is set ([GLP_SALT_SIZE][N], natural
// row order; consumed by the next commitment), else from the context's ChaCha12 stream

// openings / FRI for K proofs side by side (fri.hip)
int32_t build_zpow(glp_ctx* c, int log_n, size_t K, const uint64_t* zs, uint64_t** zpow_out);
int32_t eval_with_zpow(glp_ctx* c, const uint64_t* coeffs, size_t col_stride, size_t n_cols, size_t tree_cols, int log_n,
                       size_t K, const uint64_t* zpow, uint64_t* out_host);
// launches only: out_dev [K][n_cols][2]; partial: scratch of K * n_cols * eval_partial_blocks(log_n) * 2 words
size_t eval_partial_blocks(int log_n);
int32_t eval_launch(glp_ctx* c, const uint64_t* coeffs, size_t col_stride, size_t n_cols, size_t tree_cols, int log_n, size_t K,
                    const uint64_t* zpow, uint64_t* partial, uint64_t* out_dev);
int32_t fri_prove_forest(glp_ctx* c, const glp_batch* const* oracles, size_t n_oracles, size_t K, const glp_fri_batch* batches,
                         size_t n_batches, const glp_fri_params* p, glp_challenger* chs, uint64_t* const* proof_out);

// fn(k) for k in [0, n) on the library's host worker pool: the per-proof transcript work of the batched prover (a few hundred
// Poseidon permutations per proof) and the assembly of the proofs are independent across proofs and would otherwise bound
// small-circuit batches.  host_pool_run (glprover.hip) keeps the workers alive between calls — creating and joining 31 threads
// costs ~0.5 ms, and a batch makes 15 such calls; a second caller arriving while the pool is busy runs its loop itself.
unsigned host_pool_width();  // worker threads incl. the caller (GLP_HOST_THREADS, default: hardware threads, at most 32)
void host_pool_run(size_t n_items, size_t width, void (*fn)(void*, size_t), void* arg);
// The pool takes a loop from n / grain >= 2 on (every sleeping worker then joins: host_pool_run wakes them all — limiting a job to
// "its" n / grain threads was measured and LOSES: 32 signature-sized proofs 3.08 -> 3.47 ms, profiles/r06_grain_ab.txt).  grain = 8: loops
// of a few microseconds per item (waking the workers costs ~10 us).  grain = 2: the transcript steps of the batched prover (an item =
// tens of permutations, 10-80 us) — round 6: four proofs per call 1.57 -> 1.43 ms, eight 2.21 -> 1.57 ms, 2^11 rows x 8 4.83 -> 4.19 ms;
// from sixteen proofs per call on the loops were in the pool already.
template <typename F>
inline void parallel_for(size_t n, F&& fn, size_t grain = 8) {
    const unsigned hw = host_pool_width();
    size_t nt = n / (grain ? grain : 1);
    if (nt > hw) nt = hw;
    if (nt <= 1) {
        for (size_t k = 0; k < n; ++k) fn(k);
        return;
    }
    using Fn = typename std::remove_reference<F>::type;
    host_pool_run(n, nt, [](void* a, size_t k) { (*static_cast<Fn*>(a))(k); }, (void*)&fn);
}

// The small host tables of one step (challenge powers, pointer lists, per-proof parameters) go up in ONE copy: a block of the
// pinned arena is reserved up front, take<T>() hands out 64-byte aligned pieces of it, flush() uploads what was taken, dev(p)
// is the device address of piece p.  (A copy per table was ~25 blits and their launch gaps per proof: most of a small
// proof's copy traffic.)
struct StagedUpload {
    glp_ctx* c;
    char* host;
    char* dev_base = nullptr;
    size_t cap, off = 0;
    static size_t need(size_t bytes) { return (bytes + 63) & ~(size_t)63; }
    StagedUpload(glp_ctx* ctx, size_t bytes) : c(ctx), host(static_cast<char*>(arena_alloc(ctx, bytes ? bytes : 64))), cap(bytes ? bytes : 64) {}
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(host + off);
        off += need(count * sizeof(T));
        return host && off <= cap ? p : nullptr;  // the caller reserved too little: a programming error, surfaces as GLP_ERR_INVALID in flush
    }
    template <typename Alloc>
    int32_t flush(Alloc& scratch) {
        if (!host) return GLP_ERR_OOM;
        if (off > cap) return GLP_ERR_INVALID;
        uint64_t* d = nullptr;
        GLP_TRY(scratch.alloc(off / 8 + 8, &d));
        dev_base = reinterpret_cast<char*>(d);
        HIP_TRY(hipMemcpyAsync(dev_base, host, off, hipMemcpyHostToDevice, c->stream));
        return GLP_OK;
    }
    template <typename T>
    T* dev(T* host_piece) const {
        return reinterpret_cast<T*>(dev_base + (reinterpret_cast<char*>(host_piece) - host));
    }
};

// GLP_TRACE_STEPS=1: wall-clock time between the named steps of a prove call on stderr (host-side view: what the transcripts,
// the proof assembly and the waits for the GPU cost; the per-kernel view is glp_ctx_timing_*)
inline double trace_clock_ms() {  // one clock for every trace line of the process (GLP_TRACE_STEPS, GLP_TRACE_FEED)
    static const std::chrono::steady_clock::time_point epoch = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - epoch).count();
}
struct StepTrace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    explicit StepTrace(const char* what) {
        const char* e = getenv("GLP_TRACE_STEPS");
        on = e && atoi(e) != 0;
        if (on) {
            t0 = last = std::chrono::steady_clock::now();
            fprintf(stderr, "[glp trace] %s (clock %.3f ms)\n", what, trace_clock_ms());
        }
    }
    void mark(const char* step) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[glp trace]   %-28s %9.3f ms  (at %9.3f)\n", step, std::chrono::duration<double, std::milli>(now - last).count(),
                std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

// trees: sparse-Merkle entries (host arrays) -> device arrays in tree order, absent entries dropped (tree_sort.hip)
int32_t smt_sort_device(glp_ctx* c, const uint64_t* keys, const uint64_t* values, size_t n, uint64_t** rk_dev,
                        uint64_t** v_dev, size_t* m_out, bool inputs_on_device = false);

}  // namespace glpi
