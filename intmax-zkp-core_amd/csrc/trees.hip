// trees.hip — C-ABI drivers of the native tree hashing kernels (tree_kernels.h).
#include <algorithm>
#include <functional>
#include <numeric>

#include "internal.h"
#include "smt_replay.h"
#include "tree_kernels.h"

using namespace glpi;

namespace {

constexpr size_t COOP_ITEMS_MAX = 8192;  // batches up to this size: 16 lanes per item (latency), above: 1 lane (throughput)

template <class H>
unsigned grid_for(size_t n) {
    return (unsigned)((n * H::W + trees::THREADS - 1) / trees::THREADS);
}

// launch KERNEL<LaneHasher> or KERNEL<CoopHasher> by batch size
#define LAUNCH_BY_SIZE(c, n, KERNEL, ...)                                                                             \
    do {                                                                                                              \
        Timed t_((c), GLP_K_TREES);                                                                                   \
        if ((n) <= COOP_ITEMS_MAX)                                                                                    \
            hipLaunchKernelGGL((trees::KERNEL<trees::CoopHasher>), dim3(grid_for<trees::CoopHasher>(n)),              \
                               dim3(trees::THREADS), 0, (c)->stream, __VA_ARGS__);                                    \
        else                                                                                                          \
            hipLaunchKernelGGL((trees::KERNEL<trees::LaneHasher>), dim3(grid_for<trees::LaneHasher>(n)),              \
                               dim3(trees::THREADS), 0, (c)->stream, __VA_ARGS__);                                    \
        HIP_TRY(hipGetLastError());                                                                                   \
    } while (0)

// The hashed radix tree of a key set on the device (glp_smt_root_of_set, glp_smt_proofs_of_set): entries sorted into tree order
// (tree_sort.hip), leaf hashes, the n - 1 branching nodes with parent / child links (radix_build_kernel) and the level-synchronous
// hashing rounds; afterwards every node holds what its two children delivered, and root_dev the root.  Frees everything it
// allocated when it goes out of scope (error paths included).
// Host <-> device copies of the tree entry points go through the context's PINNED arena (blocks stay valid until the next
// arena_reset), as the prover's do (StagedUpload, internal.h).  Observed with hipMemcpyAsync to / from pageable std::vectors that are
// freed at the end of the call: glp_smt_update took 60 / 70 / 80 / 90 ms — steps of 10 ms — and the wait sat in whatever touched the
// device first in the NEXT call (a 32-byte read-back: 21 ms).  With every transfer staged here the steps are gone (50-55 ms flat).
// tools/pageable_stall_repro.hip reproduces it without any library code (13 - 25 ms against 0.06 ms with pinned staging).
struct Staging {
    glp_ctx* c;
    struct Back {
        void* dst;
        const void* block;
        size_t bytes;
    };
    std::vector<Back> back;
    explicit Staging(glp_ctx* ctx) : c(ctx) {}
    int32_t up(void* dst_dev, const void* src_host, size_t bytes) {
        if (!bytes) return GLP_OK;
        void* p = arena_alloc(c, bytes);
        if (!p) return GLP_ERR_OOM;
        std::memcpy(p, src_host, bytes);
        HIP_TRY(hipMemcpyAsync(dst_dev, p, bytes, hipMemcpyHostToDevice, c->stream));
        return GLP_OK;
    }
    int32_t down(void* dst_host, const void* src_dev, size_t bytes) {  // dst_host is filled by finish(), after the stream was waited for
        if (!bytes) return GLP_OK;
        void* p = arena_alloc(c, bytes);
        if (!p) return GLP_ERR_OOM;
        HIP_TRY(hipMemcpyAsync(p, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        back.push_back({dst_host, p, bytes});
        return GLP_OK;
    }
    void finish() {
        for (const Back& b : back) std::memcpy(b.dst, b.block, b.bytes);
        back.clear();
    }
};

struct SmtTree {
    glp_ctx* c;
    uint64_t *rk_dev = nullptr, *v_dev = nullptr, *lh_dev = nullptr, *root_dev = nullptr;
    trees::RadixNode* nodes = nullptr;
    int *lp = nullptr, *ls = nullptr, *queue = nullptr, *counts = nullptr;
    size_t m = 0;  // present entries
    explicit SmtTree(glp_ctx* ctx) : c(ctx) {}
    SmtTree(const SmtTree&) = delete;
    SmtTree& operator=(const SmtTree&) = delete;
    ~SmtTree() { release(); }
    void release() {
        pool_free(c, rk_dev);
        pool_free(c, v_dev);
        pool_free(c, lh_dev);
        pool_free(c, root_dev);
        pool_free(c, nodes);
        pool_free(c, lp);
        pool_free(c, ls);
        pool_free(c, queue);
        pool_free(c, counts);
        rk_dev = v_dev = lh_dev = root_dev = nullptr;
        nodes = nullptr;
        lp = ls = queue = counts = nullptr;
        m = 0;
    }
    void swap(SmtTree& o) {
        std::swap(rk_dev, o.rk_dev);
        std::swap(v_dev, o.v_dev);
        std::swap(lh_dev, o.lh_dev);
        std::swap(root_dev, o.root_dev);
        std::swap(nodes, o.nodes);
        std::swap(lp, o.lp);
        std::swap(ls, o.ls);
        std::swap(queue, o.queue);
        std::swap(counts, o.counts);
        std::swap(m, o.m);
    }
    // keys / values: host arrays, or (on_device) device arrays that are only read; the tree must be empty (fresh or released)
    int32_t build(const uint64_t* keys, const uint64_t* values, size_t n, bool on_device = false) {
        // drop absent entries (value 0^4), order by key bit string, reject duplicate keys: all on the device (tree_sort.hip)
        GLP_TRY(smt_sort_device(c, keys, values, n, &rk_dev, &v_dev, &m, on_device));
        if (m == 0) return GLP_OK;
        GLP_TRY(pool_alloc_t(c, 4 * m, &lh_dev));
        GLP_TRY(pool_alloc_t(c, 4, &root_dev));
        if (m > 1) {
            GLP_TRY(pool_alloc_t(c, m - 1, &nodes));
            GLP_TRY(pool_alloc_t(c, m, &lp));
            GLP_TRY(pool_alloc_t(c, m, &ls));
            GLP_TRY(pool_alloc_t(c, 2 * m, &queue));  // two ready queues of up to m - 1 nodes
            GLP_TRY(pool_alloc_t(c, 2, &counts));
        }
        {
            Timed t(c, GLP_K_TREES);
            if (m <= COOP_ITEMS_MAX)
                hipLaunchKernelGGL((trees::smt_leaf_hash_kernel<trees::CoopHasher>), dim3(grid_for<trees::CoopHasher>(m)),
                                   dim3(trees::THREADS), 0, c->stream, rk_dev, v_dev, m, 1, m > 1 ? lh_dev : root_dev);
            else
                hipLaunchKernelGGL((trees::smt_leaf_hash_kernel<trees::LaneHasher>), dim3(grid_for<trees::LaneHasher>(m)),
                                   dim3(trees::THREADS), 0, c->stream, rk_dev, v_dev, m, 1, lh_dev);
            HIP_TRY(hipGetLastError());
        }
        if (m == 1) return GLP_OK;
        if (!c->pinned_word) HIP_TRY(hipHostMalloc((void**)&c->pinned_word, 64, hipHostMallocDefault));
        int& n_ready = *c->pinned_word;  // pinned: the 4-byte read-back per round is a ~10 us DMA, not a staged copy
        n_ready = 0;
        {
            Timed t(c, GLP_K_TREES);
            HIP_TRY(hipMemsetAsync(counts, 0, 2 * sizeof(int), c->stream));
            hipLaunchKernelGGL(trees::radix_build_kernel, dim3((unsigned)((m - 1 + trees::THREADS - 1) / trees::THREADS)),
                               dim3(trees::THREADS), 0, c->stream, rk_dev, (int)m, nodes, lp, ls);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(trees::radix_deliver_leaves_kernel, dim3((unsigned)((m + trees::THREADS - 1) / trees::THREADS)),
                               dim3(trees::THREADS), 0, c->stream, lh_dev, (int)m, nodes, lp, ls, queue, counts);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(&n_ready, counts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        // one launch per level of the radix tree (<= 256).  A round reads its width from the counter the previous round
        // filled; the host only needs an upper bound for the grid, and widths never grow, so eight rounds are launched on
        // the last width it has seen before it looks again (a 4-byte pinned read-back): ~6 waits at 2^20 entries, not ~45.
        int round = 0;
        while (n_ready > 0) {
            Timed t(c, GLP_K_TREES);
            const int bound = n_ready;
            for (int b = 0; b < 8; ++b, ++round) {
                int *q_in = queue + (round & 1) * m, *q_out = queue + ((round & 1) ^ 1) * m;
                int *cnt_in = counts + (round & 1), *cnt_out = counts + ((round & 1) ^ 1);
                HIP_TRY(hipMemsetAsync(cnt_out, 0, sizeof(int), c->stream));
                if ((size_t)bound <= COOP_ITEMS_MAX)
                    hipLaunchKernelGGL((trees::radix_round_kernel<trees::CoopHasher>), dim3(grid_for<trees::CoopHasher>(bound)),
                                       dim3(trees::THREADS), 0, c->stream, rk_dev, nodes, q_in, cnt_in, q_out, cnt_out, root_dev);
                else
                    hipLaunchKernelGGL((trees::radix_round_kernel<trees::LaneHasher>), dim3(grid_for<trees::LaneHasher>(bound)),
                                       dim3(trees::THREADS), 0, c->stream, rk_dev, nodes, q_in, cnt_in, q_out, cnt_out, root_dev);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipMemcpyAsync(&n_ready, counts + (round & 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        return GLP_OK;
    }
};

// `find` of every query key against a built tree (glp_smt_proofs_of_set builds one for the call, glp_smt_find uses a resident one)
static int32_t smt_proofs_from_tree(glp_ctx* c, SmtTree& t, const uint64_t* query_keys, size_t n_queries, size_t max_siblings,
                                    uint64_t* siblings_out, uint32_t* n_siblings_out, uint32_t* flags_out, uint64_t* value_out,
                                    uint64_t* not_found_key_out, uint64_t* not_found_value_out, uint64_t root_out[4]) {
    if ((!query_keys && n_queries) || max_siblings > 256 || (siblings_out && max_siblings == 0 && n_queries)) return GLP_ERR_INVALID;
    Scratch sc(c);
    uint64_t root[4] = {0, 0, 0, 0};
    if (n_queries == 0) {
        if (t.m) {
            HIP_TRY(hipMemcpyAsync(root, t.root_dev, 32, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        if (root_out) std::memcpy(root_out, root, 32);
        return GLP_OK;
    }
    // results are staged on the device side by side and come back in ONE copy: [siblings | values | nf keys | nf values | n_sib | flags | overflow]
    const size_t sib_words = siblings_out ? n_queries * max_siblings * 4 : 0;
    const size_t words = sib_words + 3 * n_queries * 4 + n_queries + 1;  // (two u32 arrays = n_queries words, + the overflow word)
    uint64_t *q_dev = nullptr, *res = nullptr;
    GLP_TRY(sc.alloc(n_queries * 4, &q_dev));
    GLP_TRY(sc.alloc(words, &res));
    uint64_t* sib_dev = siblings_out ? res : nullptr;
    uint64_t* val_dev = res + sib_words;
    uint64_t* nfk_dev = val_dev + n_queries * 4;
    uint64_t* nfv_dev = nfk_dev + n_queries * 4;
    uint32_t* ns_dev = reinterpret_cast<uint32_t*>(nfv_dev + n_queries * 4);
    uint32_t* fl_dev = ns_dev + n_queries;
    unsigned* ovf_dev = reinterpret_cast<unsigned*>(res + words - 1);
    GLP_TRY(arena_reset(c));
    Staging stg(c);  // (query keys in, results out: through the pinned arena)
    GLP_TRY(stg.up(q_dev, query_keys, n_queries * 32));
    HIP_TRY(hipMemsetAsync(ovf_dev, 0, 8, c->stream));
    if (sib_words) HIP_TRY(hipMemsetAsync(sib_dev, 0, sib_words * 8, c->stream));  // rows behind a proof's last sibling read 0^4
    {
        Timed tm(c, GLP_K_TREES);
        hipLaunchKernelGGL(trees::smt_find_kernel, dim3((unsigned)((n_queries + trees::THREADS - 1) / trees::THREADS)), dim3(trees::THREADS),
                           0, c->stream, t.rk_dev, t.v_dev, t.nodes, (int)t.m, q_dev, n_queries, max_siblings, sib_dev, ns_dev, fl_dev,
                           val_dev, nfk_dev, nfv_dev, ovf_dev);
        HIP_TRY(hipGetLastError());
    }
    std::vector<uint64_t> host(words - sib_words);
    // the sibling array is the bulk (134 MB for 2^16 proofs of 64 siblings): it crosses PCIe into the context's PINNED arena (the
    // caller's buffer is pageable: a direct copy runs at a tenth of the link) and is copied out by the host worker pool
    uint64_t* stage = nullptr;
    if (sib_words) {
        stage = arena_alloc_t<uint64_t>(c, sib_words);
        if (!stage) return GLP_ERR_OOM;
        HIP_TRY(hipMemcpyAsync(stage, res, sib_words * 8, hipMemcpyDeviceToHost, c->stream));
    }
    GLP_TRY(stg.down(host.data(), res + sib_words, host.size() * 8));
    if (t.m) GLP_TRY(stg.down(root, t.root_dev, 32));  // through the pinned arena like every other read-back: no copy into this frame in flight
    HIP_TRY(hipStreamSynchronize(c->stream));
    stg.finish();
    if (reinterpret_cast<const unsigned*>(&host[host.size() - 1])[0]) return GLP_ERR_INVALID;  // a proof longer than max_siblings
    if (sib_words) {
        const size_t piece = (size_t)1 << 18;  // 2 MB per item
        const size_t n_pieces = (sib_words + piece - 1) / piece;
        parallel_for(n_pieces, [&](size_t k) {
            const size_t o = k * piece, cnt = o + piece < sib_words ? piece : sib_words - o;
            std::memcpy(siblings_out + o, stage + o, cnt * 8);
        });
    }
    const uint32_t* ns_h = reinterpret_cast<const uint32_t*>(host.data() + 3 * n_queries * 4);
    if (value_out) std::memcpy(value_out, host.data(), n_queries * 32);
    if (not_found_key_out) std::memcpy(not_found_key_out, host.data() + n_queries * 4, n_queries * 32);
    if (not_found_value_out) std::memcpy(not_found_value_out, host.data() + 2 * n_queries * 4, n_queries * 32);
    if (n_siblings_out) std::memcpy(n_siblings_out, ns_h, n_queries * 4);
    if (flags_out) std::memcpy(flags_out, ns_h + n_queries, n_queries * 4);
    if (root_out) std::memcpy(root_out, root, 32);
    return GLP_OK;
}

}  // namespace

extern "C" {

int32_t glp_hash_batch_dev(glp_ctx* c, const uint64_t* in_dev, size_t n, size_t len, size_t item_stride, uint32_t mode,
                           uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !out_dev || (!in_dev && len) || mode > GLP_HASH_OR_NOOP || len > (1u << 24) || item_stride < len)
        return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, hash_batch_kernel, in_dev, n, (int)len, item_stride, (int)mode, out_dev);
    return GLP_OK;
}

int32_t glp_two_to_one_batch_dev(glp_ctx* c, const uint64_t* left_dev, const uint64_t* right_dev, size_t n,
                                 uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !left_dev || !right_dev || !out_dev) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, two_to_one_batch_kernel, left_dev, right_dev, n, out_dev);
    return GLP_OK;
}

int32_t glp_smt_leaf_hash_batch_dev(glp_ctx* c, const uint64_t* keys_dev, const uint64_t* values_dev, size_t n,
                                    uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !keys_dev || !values_dev || !out_dev) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, smt_leaf_hash_kernel, keys_dev, values_dev, n, 0, out_dev);
    return GLP_OK;
}

int32_t glp_smt_root_batch_dev(glp_ctx* c, const uint64_t* keys_dev, const uint64_t* start_dev, const uint64_t* siblings_dev,
                               const uint32_t* n_siblings_dev, size_t n, size_t max_siblings, uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !keys_dev || !start_dev || !out_dev || max_siblings > 256 || (!siblings_dev && max_siblings))
        return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, smt_root_batch_kernel, keys_dev, start_dev, siblings_dev, n_siblings_dev, n, (int)max_siblings,
                   out_dev);
    return GLP_OK;
}

int32_t glp_merkle_root_batch_dev(glp_ctx* c, const uint64_t* index_dev, const uint64_t* value_dev,
                                  const uint64_t* siblings_dev, size_t n, size_t depth, uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !index_dev || !value_dev || !out_dev || depth > 256 || (!siblings_dev && depth)) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, merkle_root_batch_kernel, index_dev, value_dev, siblings_dev, n, (int)depth, out_dev);
    return GLP_OK;
}

int32_t glp_block_hash_batch_dev(glp_ctx* c, const uint64_t* headers_dev, size_t n, uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !headers_dev || !out_dev) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, block_hash_batch_kernel, headers_dev, n, out_dev);
    return GLP_OK;
}

size_t glp_merkle_proof_n_siblings(size_t depth) { return depth ? depth : 1; }

int32_t glp_merkle_proof_with_zero_dev(glp_ctx* c, const uint64_t* leaves_dev, size_t n_leaves, size_t index, size_t depth,
                                       const uint64_t zero[4], uint64_t* siblings_out, uint64_t* root_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !zero || (!leaves_dev && n_leaves) || depth > 64) return GLP_ERR_INVALID;
    const size_t len = n_leaves ? n_leaves : 1;
    // the reference asserts both (src/merkle_tree/tree.rs:60-61)
    if (index >= len) return GLP_ERR_INVALID;
    if (depth < 64 && len > ((size_t)1 << depth)) return GLP_ERR_INVALID;
    size_t m = 1;
    int log_m = 0;
    while (m < len) m <<= 1, ++log_m;
    if (log_m > 30) return GLP_ERR_UNSUPPORTED;
    const size_t n_sib = glp_merkle_proof_n_siblings(depth);
    const size_t dlen = digests_len(m, 0);
    uint64_t *padded = nullptr, *dig = nullptr, *out = nullptr;
    int32_t s = pool_alloc_t(c, m * 4, &padded);
    if (s == GLP_OK) s = pool_alloc_t(c, dlen, &dig);
    if (s == GLP_OK) s = pool_alloc_t(c, (n_sib + 1) * 4, &out);
    hipError_t e = hipSuccess;
    if (s == GLP_OK) {
        {
            Timed t(c, GLP_K_TREES);
            hipLaunchKernelGGL(trees::pad_leaves_kernel, dim3((unsigned)((m + trees::THREADS - 1) / trees::THREADS)),
                               dim3(trees::THREADS), 0, c->stream, leaves_dev, n_leaves, m, zero[0], zero[1], zero[2], zero[3],
                               padded);
            e = hipGetLastError();
        }
        if (e == hipSuccess) s = merkle_build(c, padded, m, 4, 4, 1, 0, dig, nullptr);
        if (e == hipSuccess && s == GLP_OK) {
            Timed t(c, GLP_K_TREES);
            hipLaunchKernelGGL(trees::merkle_proof_finish_kernel, dim3(1), dim3(64), 0, c->stream, dig, log_m, index, (int)n_sib,
                               zero[0], zero[1], zero[2], zero[3], out, out + n_sib * 4);
            e = hipGetLastError();
        }
        if (e == hipSuccess && s == GLP_OK && siblings_out)
            e = hipMemcpyAsync(siblings_out, out, n_sib * 32, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && s == GLP_OK && root_out)
            e = hipMemcpyAsync(root_out, out + n_sib * 4, 32, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && s == GLP_OK) e = hipStreamSynchronize(c->stream);
    }
    pool_free(c, padded);
    pool_free(c, dig);
    pool_free(c, out);
    if (s != GLP_OK) return s;
    if (e != hipSuccess) {
        note_hip_error(e);
        return e == hipErrorOutOfMemory ? GLP_ERR_OOM : GLP_ERR_HIP;
    }
    return GLP_OK;
}

int32_t glp_smt_root_of_set(glp_ctx* c, const uint64_t* keys, const uint64_t* values, size_t n, uint64_t root_out[4]) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !root_out || ((!keys || !values) && n) || n > (1u << 30)) return GLP_ERR_INVALID;
    SmtTree t(c);
    GLP_TRY(t.build(keys, values, n));
    if (t.m == 0) {
        std::memset(root_out, 0, 32);
        return GLP_OK;
    }
    HIP_TRY(hipMemcpyAsync(root_out, t.root_dev, 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int32_t glp_smt_proofs_of_set(glp_ctx* c, const uint64_t* keys, const uint64_t* values, size_t n, const uint64_t* query_keys,
                              size_t n_queries, size_t max_siblings, uint64_t* siblings_out, uint32_t* n_siblings_out,
                              uint32_t* flags_out, uint64_t* value_out, uint64_t* not_found_key_out, uint64_t* not_found_value_out,
                              uint64_t root_out[4]) {
    GLP_ON_DEVICE_OF(c);
    if (!c || ((!keys || !values) && n) || n > (1u << 30) || (!query_keys && n_queries) || max_siblings > 256 ||
        (siblings_out && max_siblings == 0 && n_queries))
        return GLP_ERR_INVALID;
    SmtTree t(c);
    GLP_TRY(t.build(keys, values, n));
    return smt_proofs_from_tree(c, t, query_keys, n_queries, max_siblings, siblings_out, n_siblings_out, flags_out, value_out,
                                not_found_key_out, not_found_value_out, root_out);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// glp_smt_apply_updates: k sequential `set(key, value)` calls on the tree of a set (reference src/sparse_merkle_tree/tree.rs:143-151
// -> calc_process_proof :561-585 -> update :174-253, insert :255-381, remove :384-520), as ONE batch.
//
// Sequential on a CPU — every `set` walks and re-hashes its path, ~depth Poseidon permutations that depend on the previous
// `set` — but the walks never look at a hash VALUE (only "is it 0^4", "is it a leaf"), so the structure can be replayed first and
// hashed afterwards:
//   1. the tree of the base set is built on the device (SmtTree) and the walk kernel (smt_find_kernel) extracts, for every distinct
//      key that will be touched, its path: the sibling hashes, which of them are lone leaves (and which entries), where it ends;
//   2. the host merges those paths into a small pointer trie in which a hash is a VALUE ID — static ids for what the device gave,
//      fresh ids for nodes that have to be hashed — and replays the k updates on it by path copying, exactly the canonical-form
//      rules of the reference (a one-entry subtree IS its leaf; a removed leaf's lone sibling leaf moves up; an inserted leaf
//      pushes the leaf it meets down to the first differing bit); every new node is a job (id <- H(a, b) or id <- leaf hash) whose
//      level is one more than its inputs'; the `find` each `set` starts with is read off the trie as a list of ids;
//   3. the device runs the jobs level by level — one launch per level, all updates at once: ~tree-height launches instead of
//      k x depth dependent permutations — and the table of values comes back in one copy.
namespace {

// all jobs of a replay, level by level -> the table of values (host)
int32_t smt_replay_run(glp_ctx* c, Scratch& sc, const SmtReplay& rp, std::vector<uint64_t>& tab, size_t* n_jobs_out = nullptr) {
    const auto& init = rp.init;
    const auto& leaf_ids = rp.leaf_ids;
    const auto& leaf_kv = rp.leaf_kv;
    const auto& level_jobs = rp.level_jobs;
    const size_t ids = rp.n_ids();
    uint64_t* table = nullptr;
    GLP_TRY(sc.alloc(ids * 4, &table));
    Staging stg(c);
    GLP_TRY(stg.up(table, init.data(), ids * 32));
    if (!leaf_ids.empty()) {
        uint64_t *kv_dev = nullptr, *ids_dev = nullptr;
        GLP_TRY(sc.alloc(leaf_kv.size(), &kv_dev));
        GLP_TRY(sc.alloc((leaf_ids.size() + 1) / 2, &ids_dev));
        GLP_TRY(stg.up(kv_dev, leaf_kv.data(), leaf_kv.size() * 8));
        GLP_TRY(stg.up(ids_dev, leaf_ids.data(), leaf_ids.size() * 4));
        LAUNCH_BY_SIZE(c, leaf_ids.size(), smt_leaf_jobs_kernel, table, kv_dev, reinterpret_cast<const int*>(ids_dev), leaf_ids.size());
    }
    size_t n_jobs = 0;
    for (auto& lv : level_jobs) n_jobs += lv.size();
    if (n_jobs) {
        trees::SmtNodeJob* flat = arena_alloc_t<trees::SmtNodeJob>(c, n_jobs);  // the levels side by side, written straight into pinned memory
        if (!flat) return GLP_ERR_OOM;
        {
            size_t at = 0;
            for (auto& lv : level_jobs) {
                if (!lv.empty()) std::memcpy(flat + at, lv.data(), lv.size() * sizeof(trees::SmtNodeJob));
                at += lv.size();
            }
        }
        uint64_t* jobs_dev = nullptr;
        GLP_TRY(sc.alloc((n_jobs * sizeof(trees::SmtNodeJob) + 7) / 8, &jobs_dev));
        HIP_TRY(hipMemcpyAsync(jobs_dev, flat, n_jobs * sizeof(trees::SmtNodeJob), hipMemcpyHostToDevice, c->stream));
        size_t off = 0;
        for (auto& lv : level_jobs) {
            if (lv.empty()) continue;
            LAUNCH_BY_SIZE(c, lv.size(), smt_node_jobs_kernel, table, reinterpret_cast<const trees::SmtNodeJob*>(jobs_dev) + off, lv.size());
            off += lv.size();
        }
    }
    tab.resize(ids * 4);
    GLP_TRY(stg.down(tab.data(), table, ids * 32));
    HIP_TRY(hipStreamSynchronize(c->stream));
    stg.finish();
    if (n_jobs_out) *n_jobs_out = n_jobs;
    return GLP_OK;
}

}  // namespace

// What a resident tree needs to follow a batch of updates (glp_smt_update): per distinct updated key its entry in the tree before the
// batch (-1: not in it) and its value after the batch (the last `set` of the key wins; 0^4 = not in the set afterwards)
struct SmtNetChanges {
    std::vector<uint64_t> keys;    // [n][4] canonical, distinct
    std::vector<uint64_t> values;  // [n][4] canonical
    std::vector<int> entry;        // [n]
};

// k sequential `set`s on a BUILT tree (glp_smt_apply_updates builds one for the call, glp_smt_update keeps one); the tree itself is
// not modified
static int32_t smt_apply_updates_on_tree(glp_ctx* c, SmtTree& t, StepTrace& trace, const uint64_t* upd_keys, const uint64_t* upd_values,
                                         size_t k, size_t max_siblings, uint64_t* siblings_out, uint32_t* n_siblings_out,
                                         uint32_t* flags_out, uint64_t* old_value_out, uint64_t* other_key_out,
                                         uint64_t* other_value_out, uint64_t* last_sibling_key_out, uint64_t* last_sibling_value_out,
                                         uint64_t* roots_out, SmtNetChanges* net,
                                         const std::function<int32_t(const SmtNetChanges&)>* after_paths = nullptr) {
    if (((!upd_keys || !upd_values) && k) || k > (1u << 24) || max_siblings > 256 || (siblings_out && max_siblings == 0 && k))
        return GLP_ERR_INVALID;
    GLP_TRY(arena_reset(c));  // every copy below is staged in the pinned arena; its blocks live until the next call
    uint64_t root0[4] = {0, 0, 0, 0};
    if (t.m) {  // staged in pinned memory: a failed (or injected-away) synchronize must not leave a copy into this frame in flight
        Staging st0(c);
        GLP_TRY(st0.down(root0, t.root_dev, 32));
        HIP_TRY(hipStreamSynchronize(c->stream));
        st0.finish();
    }
    if (k == 0) {
        if (roots_out) std::memcpy(roots_out, root0, 32);
        return GLP_OK;
    }
    // canonical copies of the updates; the distinct keys among them
    std::vector<uint64_t> uk(k * 4), uv(k * 4);
    for (size_t i = 0; i < 4 * k; ++i) uk[i] = gl::canon(upd_keys[i]), uv[i] = gl::canon(upd_values[i]);
    std::vector<uint32_t> order(k);
    std::iota(order.begin(), order.end(), 0u);
    auto key_less = [&](uint32_t a, uint32_t b) {
        return std::lexicographical_compare(&uk[4 * a], &uk[4 * a] + 4, &uk[4 * b], &uk[4 * b] + 4);
    };
    std::sort(order.begin(), order.end(), key_less);
    std::vector<uint64_t> dq;  // distinct keys
    for (size_t j = 0; j < k; ++j)
        if (j == 0 || !key_eq(&uk[4 * order[j]], &uk[4 * order[j - 1]])) dq.insert(dq.end(), &uk[4 * order[j]], &uk[4 * order[j]] + 4);
    const size_t nq = dq.size() / 4;
    if (net) {  // the value each distinct key holds after the batch: its last update in sequence order
        net->keys = dq;
        net->values.assign(nq * 4, 0);
        net->entry.assign(nq, -1);
        size_t qi = 0;
        for (size_t j = 0; j < k;) {
            size_t e = j, last = order[j];
            while (e < k && key_eq(&uk[4 * order[e]], &uk[4 * order[j]])) {
                if (order[e] > last) last = order[e];
                ++e;
            }
            std::memcpy(&net->values[4 * qi], &uv[4 * last], 32);
            ++qi;
            j = e;
        }
    }

    // ---- 1. the base paths of the distinct keys ----
    Scratch sc(c);
    uint64_t* q_dev = nullptr;
    GLP_TRY(sc.alloc(nq * 4, &q_dev));
    Staging stg(c);
    GLP_TRY(stg.up(q_dev, dq.data(), nq * 32));
    std::vector<uint32_t> ns(nq), fl(nq);
    std::vector<int> end_entry(nq);
    std::vector<uint64_t> sib;      // [nq][msib][4]
    std::vector<int> sib_entry;     // [nq][msib]
    size_t msib = 0;
    for (int pass = 0; pass < 2; ++pass) {  // lengths first, then the siblings in an array sized to the longest path
        const size_t sw = pass ? nq * msib * 4 : 0, ew = pass ? (nq * msib + 1) / 2 : 0;
        const size_t words = sw + ew + 3 * nq * 4 + nq + (nq + 1) / 2 + 1;
        uint64_t* res = nullptr;
        GLP_TRY(sc.alloc(words, &res));
        uint64_t* sib_dev = pass ? res : nullptr;
        int* se_dev = pass ? reinterpret_cast<int*>(res + sw) : nullptr;
        uint64_t* val_dev = res + sw + ew;  // (value / other key / other value: not read back — the gathered entries carry them)
        uint32_t* ns_dev = reinterpret_cast<uint32_t*>(val_dev + 3 * nq * 4);
        uint32_t* fl_dev = ns_dev + nq;
        int* ee_dev = reinterpret_cast<int*>(val_dev + 3 * nq * 4 + nq);
        unsigned* ovf_dev = reinterpret_cast<unsigned*>(res + words - 1);
        HIP_TRY(hipMemsetAsync(res, 0, words * 8, c->stream));
        {
            Timed tm(c, GLP_K_TREES);
            hipLaunchKernelGGL(trees::smt_find_kernel, dim3((unsigned)((nq + trees::THREADS - 1) / trees::THREADS)), dim3(trees::THREADS), 0,
                               c->stream, t.rk_dev, t.v_dev, t.nodes, (int)t.m, q_dev, nq, msib, sib_dev, ns_dev, fl_dev, val_dev,
                               val_dev + nq * 4, val_dev + 2 * nq * 4, ovf_dev, se_dev, ee_dev);
            HIP_TRY(hipGetLastError());
        }
        GLP_TRY(stg.down(ns.data(), ns_dev, nq * 4));
        GLP_TRY(stg.down(fl.data(), fl_dev, nq * 4));
        GLP_TRY(stg.down(end_entry.data(), ee_dev, nq * 4));
        if (pass) {
            sib.resize(nq * msib * 4);
            sib_entry.resize(nq * msib);
            GLP_TRY(stg.down(sib.data(), sib_dev, sib.size() * 8));
            GLP_TRY(stg.down(sib_entry.data(), se_dev, sib_entry.size() * 4));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        stg.finish();
        if (!pass) {
            for (uint32_t x : ns) msib = x > msib ? x : msib;
            if (msib == 0) msib = 1;
        }
    }
    if (net) {
        for (size_t qi = 0; qi < nq; ++qi) net->entry[qi] = (fl[qi] & GLP_SMT_FOUND) ? end_entry[qi] : -1;
        // the resident tree builds its successor HERE, while the GPU is in the middle of this call's device work: the ~30 ms of host
        // work below leave it idle, and the first large kernels after an idle stretch were measured 10 - 30 ms late (clocks come
        // back in 10 ms steps)
        if (after_paths) GLP_TRY((*after_paths)(*net));
    }
    // the entries those paths meet as lone leaves (siblings, ends): key, value, leaf hash by index
    std::vector<int> want;
    for (size_t qi = 0; qi < nq; ++qi) {
        if (end_entry[qi] >= 0) want.push_back(end_entry[qi]);
        for (uint32_t d = 0; d < ns[qi]; ++d)
            if (sib_entry[qi * msib + d] > 0) want.push_back(sib_entry[qi * msib + d] - 1);
    }
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    std::vector<uint64_t> ent(want.size() * 12);
    if (!want.empty()) {
        uint64_t *idx_dev = nullptr, *ent_dev = nullptr;
        GLP_TRY(sc.alloc((want.size() + 1) / 2, &idx_dev));
        GLP_TRY(sc.alloc(want.size() * 12, &ent_dev));
        GLP_TRY(stg.up(idx_dev, want.data(), want.size() * 4));
        {
            Timed tm(c, GLP_K_TREES);
            hipLaunchKernelGGL(trees::smt_gather_entries_kernel, dim3((unsigned)((want.size() + trees::THREADS - 1) / trees::THREADS)),
                               dim3(trees::THREADS), 0, c->stream, t.rk_dev, t.v_dev, t.m > 1 ? t.lh_dev : t.root_dev,
                               reinterpret_cast<const int*>(idx_dev), want.size(), ent_dev);
            HIP_TRY(hipGetLastError());
        }
        GLP_TRY(stg.down(ent.data(), ent_dev, ent.size() * 8));
        HIP_TRY(hipStreamSynchronize(c->stream));
        stg.finish();
    }
    auto entry_rec = [&](int e) -> const uint64_t* {  // key[4] value[4] leaf hash[4]
        const size_t pos = (size_t)(std::lower_bound(want.begin(), want.end(), e) - want.begin());
        return &ent[pos * 12];
    };

    trace.mark("base paths + entries");
    // ---- 2. the host trie: value ids, jobs ----
    SmtReplay rp;
    {   // one allocation each instead of ~17 doublings: the replay below is a few milliseconds of pure pointer work
        size_t path_nodes = 0;
        for (uint32_t x : ns) path_nodes += 2 * (size_t)x + 2;
        const size_t est = path_nodes + k * (msib + 10);
        rp.nodes.reserve(est);
        rp.leaves.reserve(want.size() + k + 16);
        rp.init.reserve(4 * (est + 16));
        rp.leaf_kv.reserve(8 * k);
        rp.leaf_ids.reserve(k);
    }
    std::vector<UpdNode>& nodes = rp.nodes;
    auto static_leaf = [&](int entry) {  // a leaf of the base set: key, value and hash came from the device
        const uint64_t* r = entry_rec(entry);
        return rp.known_leaf(r, r + 4, r + 8);
    };
    // merge the base paths
    int root = -3;
    auto get_slot = [&](int parent, int side) { return parent < 0 ? root : (side ? nodes[parent].r : nodes[parent].l); };
    auto set_slot = [&](int parent, int side, int v) {
        if (parent < 0)
            root = v;
        else
            (side ? nodes[parent].r : nodes[parent].l) = v;
    };
    for (size_t qi = 0; qi < nq; ++qi) {
        const uint64_t* u = &dq[qi * 4];
        int parent = -1, side = 0;
        for (uint32_t d = 0; d < ns[qi]; ++d) {
            int nd = get_slot(parent, side);
            if (nd == -3) {
                nodes.push_back({2, -3, -3, -1, 0, -1, false});
                nd = (int)nodes.size() - 1;
                set_slot(parent, side, nd);
            } else if (nd >= 0 && nodes[nd].kind == 2 && nodes[nd].opaque) {  // another key's static sibling: this key enters it
                nodes[nd].opaque = false;
                nodes[nd].l = nodes[nd].r = -3;  // (its value stays the static one)
            } else if (nd < 0 || nodes[nd].kind != 2) {
                return GLP_ERR_INTERNAL;  // the paths of one tree cannot disagree
            }
            const int b = key_bit_at(u, (int)d);
            if ((b ? nodes[nd].l : nodes[nd].r) == -3) {
                const int code = sib_entry[qi * msib + d];
                int sn = -1;
                if (code > 0) {
                    sn = static_leaf(code - 1);
                } else if (code < 0) {
                    const int id = rp.new_id(&sib[(qi * msib + d) * 4]);
                    nodes.push_back({2, -2, -2, id, 0, -1, true});
                    sn = (int)nodes.size() - 1;
                }
                (b ? nodes[nd].l : nodes[nd].r) = sn;
            }
            parent = nd;
            side = b;
        }
        if (get_slot(parent, side) == -3) {
            const int leaf = (fl[qi] & 2) || end_entry[qi] < 0 ? -1 : static_leaf(end_entry[qi]);
            set_slot(parent, side, leaf);
        }
    }
    if (root == -3) root = -1;
    {   // values of the path nodes at time 0: jobs, children first (iterative post-order)
        std::vector<std::pair<int, int>> st;
        if (root >= 0) st.push_back({root, 0});
        while (!st.empty()) {
            auto [nd, phase] = st.back();
            st.pop_back();
            if (nd < 0 || nodes[nd].kind != 2 || nodes[nd].opaque) continue;
            if (nodes[nd].l == -3 || nodes[nd].r == -3) return GLP_ERR_INTERNAL;
            if (phase == 0) {
                st.push_back({nd, 1});
                st.push_back({nodes[nd].l, 0});
                st.push_back({nodes[nd].r, 0});
            } else if (nodes[nd].id < 0) {
                rp.job_for(nd);
            }
        }
    }
    trace.mark("merge paths");
    {   // the replay: the key prefixes side by side on the worker pool where the tree's shape allows it (smt_replay.h), else in sequence
        static const int rbits = [] {
            const char* e = getenv("GLP_SMT_REPLAY_BITS");  // 0 = always sequential
            return e ? atoi(e) : 4;
        }();
        static const size_t rmin = [] {
            const char* e = getenv("GLP_SMT_REPLAY_MIN_K");
            return e ? (size_t)atol(e) : (size_t)6144;  // below that the subtree copies and the stitching pass cost what the pool saves (profiles/r05_trees_bench.md)
        }();
        int done = 0;
        if (rbits > 0 && k >= rmin) {
            done = replay_partitioned(rp, root, uk.data(), uv.data(), k, rbits, [](size_t n_items, auto&& fn) { parallel_for(n_items, fn); });
            if (done < 0) return done;
        }
        if (!done)
            for (size_t i = 0; i < k; ++i) GLP_TRY(rp.set(root, &uk[4 * i], &uv[4 * i]));
        if (trace.on) fprintf(stderr, "[glp trace]   replay: %s\n", done ? "partitioned by key prefix" : "sequential");
    }
    if (siblings_out && rp.too_long(max_siblings)) return GLP_ERR_INVALID;
    trace.mark("replay");
    // ---- 3. the jobs, level by level ----
    std::vector<uint64_t> tab;
    size_t n_jobs = 0;
    GLP_TRY(smt_replay_run(c, sc, rp, tab, &n_jobs));
    trace.mark("jobs on the device");
    rp.fill(tab, k, max_siblings, siblings_out, n_siblings_out, flags_out, old_value_out, other_key_out, other_value_out,
            last_sibling_key_out, last_sibling_value_out);
    if (roots_out) {
        std::memcpy(roots_out, &tab[(size_t)rp.root_before[0] * 4], 32);
        for (size_t i = 0; i < k; ++i) std::memcpy(roots_out + 4 * (i + 1), &tab[(size_t)rp.root_after[i] * 4], 32);
    }
    trace.mark("outputs");
    if (trace.on)
        fprintf(stderr, "[glp trace]   %zu value ids, %zu node jobs in %zu levels, %zu leaf jobs\n", rp.n_ids(), n_jobs, rp.level_jobs.size(),
                rp.leaf_ids.size());
    return GLP_OK;
}

extern "C" int32_t glp_smt_apply_updates(glp_ctx* c, const uint64_t* keys, const uint64_t* values, size_t n, const uint64_t* upd_keys,
                                         const uint64_t* upd_values, size_t k, size_t max_siblings, uint64_t* siblings_out,
                                         uint32_t* n_siblings_out, uint32_t* flags_out, uint64_t* old_value_out,
                                         uint64_t* other_key_out, uint64_t* other_value_out, uint64_t* last_sibling_key_out,
                                         uint64_t* last_sibling_value_out, uint64_t* roots_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || ((!keys || !values) && n) || n > (1u << 30) || ((!upd_keys || !upd_values) && k) || k > (1u << 24) || max_siblings > 256 ||
        (siblings_out && max_siblings == 0 && k))
        return GLP_ERR_INVALID;
    StepTrace trace("glp_smt_apply_updates");
    SmtTree t(c);
    GLP_TRY(t.build(keys, values, n));
    trace.mark("base tree");
    return smt_apply_updates_on_tree(c, t, trace, upd_keys, upd_values, k, max_siblings, siblings_out, n_siblings_out, flags_out,
                                     old_value_out, other_key_out, other_value_out, last_sibling_key_out, last_sibling_value_out,
                                     roots_out, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------------
// A tree that STAYS on the device between calls (glp_smt_open .. glp_smt_close): the reference's world-state and latest-account trees
// live as long as the node and take a batch of `set`s per block (src/bin/block_circuit.rs:444-648).  glp_smt_apply_updates uploads,
// sorts and hashes the whole entry set on every call (7.3 of 17.5 ms for 4 096 updates of 2^20 entries); here the set is on the
// device already: glp_smt_find = the walk, glp_smt_update = the same replay, after which the entry arrays are patched ON THE DEVICE
// (changed values, removed entries, new keys appended) and the tree is rebuilt from them.
struct glp_smt {
    glp_ctx* ctx;
    SmtTree t;
    explicit glp_smt(glp_ctx* c) : ctx(c), t(c) {}
};

extern "C" {

int32_t glp_smt_open(glp_ctx* c, const uint64_t* keys, const uint64_t* values, size_t n, glp_smt** tree_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !tree_out || ((!keys || !values) && n) || n > (1u << 30)) return GLP_ERR_INVALID;
    glp_smt* s = new (std::nothrow) glp_smt(c);
    if (!s) return GLP_ERR_OOM;
    ctx_retain(c);
    const int32_t st = s->t.build(keys, values, n);
    if (st != GLP_OK) {
        delete s;
        ctx_release(c);
        return st;
    }
    *tree_out = s;
    return GLP_OK;
}

void glp_smt_close(glp_smt* s) {
    if (!s) return;
    GLP_ON_DEVICE_OF(s->ctx);
    glp_ctx* c = s->ctx;
    delete s;
    ctx_release(c);
}

int32_t glp_smt_len(const glp_smt* s, size_t* n_entries_out) {
    if (!s || !n_entries_out) return GLP_ERR_INVALID;
    *n_entries_out = s->t.m;
    return GLP_OK;
}

int32_t glp_smt_root(glp_smt* s, uint64_t root_out[4]) {
    if (!s || !root_out) return GLP_ERR_INVALID;
    glp_ctx* c = s->ctx;
    GLP_ON_DEVICE_OF(c);
    if (s->t.m == 0) {
        std::memset(root_out, 0, 32);
        return GLP_OK;
    }
    HIP_TRY(hipMemcpyAsync(root_out, s->t.root_dev, 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

// the current entry set, in tree order (keys by their bit strings, bit 0 first): a checkpoint, or the input of another glp_smt_open
int32_t glp_smt_entries(glp_smt* s, uint64_t* keys_out, uint64_t* values_out) {
    if (!s || ((!keys_out || !values_out) && s->t.m)) return GLP_ERR_INVALID;
    glp_ctx* c = s->ctx;
    GLP_ON_DEVICE_OF(c);
    const size_t m = s->t.m;
    if (m == 0) return GLP_OK;
    Scratch sc(c);
    uint64_t *k = nullptr, *v = nullptr;
    GLP_TRY(sc.alloc(4 * m, &k));
    GLP_TRY(sc.alloc(4 * m, &v));
    {
        Timed tm(c, GLP_K_TREES);
        hipLaunchKernelGGL(trees::smt_entries_out_kernel, dim3((unsigned)((4 * m + trees::THREADS - 1) / trees::THREADS)), dim3(trees::THREADS),
                           0, c->stream, s->t.rk_dev, s->t.v_dev, m, k, v);
        HIP_TRY(hipGetLastError());
    }
    GLP_TRY(download_staged(c, keys_out, k, m * 32));
    GLP_TRY(download_staged(c, values_out, v, m * 32));
    return GLP_OK;
}

int32_t glp_smt_find(glp_smt* s, const uint64_t* query_keys, size_t n_queries, size_t max_siblings, uint64_t* siblings_out,
                     uint32_t* n_siblings_out, uint32_t* flags_out, uint64_t* value_out, uint64_t* not_found_key_out,
                     uint64_t* not_found_value_out, uint64_t root_out[4]) {
    if (!s) return GLP_ERR_INVALID;
    GLP_ON_DEVICE_OF(s->ctx);
    return smt_proofs_from_tree(s->ctx, s->t, query_keys, n_queries, max_siblings, siblings_out, n_siblings_out, flags_out, value_out,
                                not_found_key_out, not_found_value_out, root_out);
}

int32_t glp_smt_update(glp_smt* s, const uint64_t* upd_keys, const uint64_t* upd_values, size_t k, size_t max_siblings,
                       uint64_t* siblings_out, uint32_t* n_siblings_out, uint32_t* flags_out, uint64_t* old_value_out,
                       uint64_t* other_key_out, uint64_t* other_value_out, uint64_t* last_sibling_key_out,
                       uint64_t* last_sibling_value_out, uint64_t* roots_out) {
    if (!s) return GLP_ERR_INVALID;
    glp_ctx* c = s->ctx;
    GLP_ON_DEVICE_OF(c);
    StepTrace trace("glp_smt_update");
    SmtNetChanges net;
    SmtTree nt(c);  // the tree after the batch; takes the resident tree's place only if the whole call succeeds
    const std::function<int32_t(const SmtNetChanges&)> build_successor = [&](const SmtNetChanges& net) -> int32_t {
        // the set after the batch, still on the device: the old entries in input form, changed values patched in (0^4 drops the entry at
        // the sort), new keys appended; a new tree is built from it and takes the old one's place only when that succeeded
        const size_t m_old = s->t.m, nq = net.entry.size();
        std::vector<int> patch_idx;
        std::vector<uint64_t> patch_val, add_k, add_v;
        for (size_t q = 0; q < nq; ++q) {
            const uint64_t* v = &net.values[4 * q];
            if (net.entry[q] >= 0) {
                patch_idx.push_back(net.entry[q]);
                patch_val.insert(patch_val.end(), v, v + 4);
            } else if (v[0] | v[1] | v[2] | v[3]) {
                add_k.insert(add_k.end(), &net.keys[4 * q], &net.keys[4 * q] + 4);
                add_v.insert(add_v.end(), v, v + 4);
            }
        }
        const size_t n_add = add_k.size() / 4, n_all = m_old + n_add;
        if (n_all) {
            Scratch sc(c);
            uint64_t *k_in = nullptr, *v_in = nullptr;
            GLP_TRY(sc.alloc(4 * n_all, &k_in));
            GLP_TRY(sc.alloc(4 * n_all, &v_in));
            if (m_old) {
                Timed tm(c, GLP_K_TREES);
                hipLaunchKernelGGL(trees::smt_entries_out_kernel, dim3((unsigned)((4 * m_old + trees::THREADS - 1) / trees::THREADS)),
                                   dim3(trees::THREADS), 0, c->stream, s->t.rk_dev, s->t.v_dev, m_old, k_in, v_in);
                HIP_TRY(hipGetLastError());
            }
            // (no arena_reset here: the blocks the caller staged before this point are still read after it)
            auto staged = [&](const void* src, size_t bytes) -> const void* {
                void* p = arena_alloc(c, bytes);
                if (p) std::memcpy(p, src, bytes);
                return p;
            };
            if (n_add) {
                const void *ak = staged(add_k.data(), n_add * 32), *av = staged(add_v.data(), n_add * 32);
                if (!ak || !av) return GLP_ERR_OOM;
                HIP_TRY(hipMemcpyAsync(k_in + 4 * m_old, ak, n_add * 32, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipMemcpyAsync(v_in + 4 * m_old, av, n_add * 32, hipMemcpyHostToDevice, c->stream));
            }
            if (!patch_idx.empty()) {
                uint64_t *idx_dev = nullptr, *val_dev = nullptr;
                GLP_TRY(sc.alloc((patch_idx.size() + 1) / 2, &idx_dev));
                GLP_TRY(sc.alloc(patch_val.size(), &val_dev));
                const void *pi = staged(patch_idx.data(), patch_idx.size() * 4), *pv = staged(patch_val.data(), patch_val.size() * 8);
                if (!pi || !pv) return GLP_ERR_OOM;
                HIP_TRY(hipMemcpyAsync(idx_dev, pi, patch_idx.size() * 4, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipMemcpyAsync(val_dev, pv, patch_val.size() * 8, hipMemcpyHostToDevice, c->stream));
                Timed tm(c, GLP_K_TREES);
                hipLaunchKernelGGL(trees::smt_patch_values_kernel, dim3((unsigned)((patch_val.size() + trees::THREADS - 1) / trees::THREADS)),
                                   dim3(trees::THREADS), 0, c->stream, v_in, reinterpret_cast<const int*>(idx_dev), val_dev, patch_idx.size());
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipStreamSynchronize(c->stream));  // the host vectors above are done being read
            GLP_TRY(nt.build(k_in, v_in, n_all, true));
        }
        trace.mark("successor tree built on the device");
        return GLP_OK;
    };
    GLP_TRY(smt_apply_updates_on_tree(c, s->t, trace, upd_keys, upd_values, k, max_siblings, siblings_out, n_siblings_out, flags_out,
                                      old_value_out, other_key_out, other_value_out, last_sibling_key_out, last_sibling_value_out,
                                      roots_out, &net, &build_successor));
    if (k == 0) return GLP_OK;
    s->t.swap(nt);  // (nt's destructor releases the old tree)
    return GLP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// glp_smt_apply_updates_multi: the same for MANY small independent trees, each with its own base entries and its own update
// sequence, in one call — the per-transaction trees of block production: a transaction inserts its outputs into an empty diff
// tree and removes its spent assets from the sender's asset tree, and the block circuit consumes the process proof of every one
// of those `set`s (reference src/transaction/gadgets/purge/mod.rs, src/sparse_merkle_tree/layered_layered_tree.rs:124-160).  The
// trees are small, so their base tries are built right here (sorted entries, leaf jobs) — no device walk — and every tree's
// updates are replayed into ONE job table: one levelled pass hashes all of them.
extern "C" int32_t glp_smt_apply_updates_multi(glp_ctx* c, const uint64_t* keys, const uint64_t* values, const size_t* set_offsets,
                                               const uint64_t* upd_keys, const uint64_t* upd_values, const size_t* upd_offsets,
                                               size_t n_trees, size_t max_siblings, uint64_t* siblings_out, uint32_t* n_siblings_out,
                                               uint32_t* flags_out, uint64_t* old_value_out, uint64_t* other_key_out,
                                               uint64_t* other_value_out, uint64_t* last_sibling_key_out,
                                               uint64_t* last_sibling_value_out, uint64_t* roots_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !set_offsets || !upd_offsets || max_siblings > 256) return GLP_ERR_INVALID;
    if (n_trees == 0) return GLP_OK;
    // the replay records and every output row are indexed by the ABSOLUTE update / entry index: both offset tables start at 0
    if (set_offsets[0] != 0 || upd_offsets[0] != 0) return GLP_ERR_INVALID;
    const size_t n = set_offsets[n_trees], k = upd_offsets[n_trees];
    if (((!keys || !values) && n) || ((!upd_keys || !upd_values) && k) || n > ((size_t)1 << 26) || k > ((size_t)1 << 24) ||
        (siblings_out && max_siblings == 0 && k))
        return GLP_ERR_INVALID;
    for (size_t s2 = 0; s2 < n_trees; ++s2)
        if (set_offsets[s2] > set_offsets[s2 + 1] || upd_offsets[s2] > upd_offsets[s2 + 1]) return GLP_ERR_INVALID;
    std::vector<uint64_t> ck(4 * n), cv(4 * n), uk(4 * k), uv(4 * k);
    for (size_t i = 0; i < 4 * n; ++i) ck[i] = gl::canon(keys[i]), cv[i] = gl::canon(values[i]);
    for (size_t i = 0; i < 4 * k; ++i) uk[i] = gl::canon(upd_keys[i]), uv[i] = gl::canon(upd_values[i]);
    auto bit_order = [&](uint32_t a, uint32_t b) {  // by the key's bit string, bit 0 first
        const uint64_t *x = &ck[4 * (size_t)a], *y = &ck[4 * (size_t)b];
        for (int l = 0; l < 4; ++l)
            if (const uint64_t d = x[l] ^ y[l]) return ((y[l] >> __builtin_ctzll(d)) & 1) != 0;
        return false;
    };
    SmtReplay rp;
    rp.nodes.reserve(2 * n + 24 * k + 16);
    rp.init.reserve(4 * (2 * n + 24 * k + 16));
    std::vector<int> first_root(n_trees, 0);
    std::vector<uint32_t> perm;
    for (size_t s2 = 0; s2 < n_trees; ++s2) {
        std::vector<uint32_t> all(set_offsets[s2 + 1] - set_offsets[s2]);
        std::iota(all.begin(), all.end(), (uint32_t)set_offsets[s2]);
        std::sort(all.begin(), all.end(), bit_order);
        for (size_t j = 1; j < all.size(); ++j)
            if (key_eq(&ck[4 * (size_t)all[j]], &ck[4 * (size_t)all[j - 1]])) return GLP_ERR_INVALID;  // a duplicate key in a base set
        perm.clear();
        for (uint32_t e : all) {
            const uint64_t* v = &cv[4 * (size_t)e];
            if (v[0] | v[1] | v[2] | v[3]) perm.push_back(e);  // value 0^4 = absent
        }
        int root = rp.build_sorted(ck.data(), cv.data(), perm.data(), 0, perm.size(), 0);
        first_root[s2] = rp.id_of(root);
        for (size_t i = upd_offsets[s2]; i < upd_offsets[s2 + 1]; ++i) GLP_TRY(rp.set(root, &uk[4 * i], &uv[4 * i]));
    }
    if (siblings_out && rp.too_long(max_siblings)) return GLP_ERR_INVALID;
    Scratch sc(c);
    GLP_TRY(arena_reset(c));  // smt_replay_run stages its copies in the pinned arena
    std::vector<uint64_t> tab;
    GLP_TRY(smt_replay_run(c, sc, rp, tab));
    rp.fill(tab, k, max_siblings, siblings_out, n_siblings_out, flags_out, old_value_out, other_key_out, other_value_out,
            last_sibling_key_out, last_sibling_value_out);
    if (roots_out)  // tree s: its root before its first update, then the root after each of its updates
        for (size_t s2 = 0; s2 < n_trees; ++s2) {
            uint64_t* o = roots_out + 4 * (upd_offsets[s2] + s2);
            std::memcpy(o, &tab[(size_t)first_root[s2] * 4], 32);
            for (size_t i = upd_offsets[s2]; i < upd_offsets[s2 + 1]; ++i)
                std::memcpy(o + 4 * (i - upd_offsets[s2] + 1), &tab[(size_t)rp.root_after[i] * 4], 32);
        }
    return GLP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// glp_smt_roots_of_sets: the roots of MANY small independent sparse trees in one call — the per-transaction trees of block
// production (the LayeredLayered diff tree of a transaction is three layers of them: reference
// src/sparse_merkle_tree/layered_layered_tree.rs:124-160, src/transaction/tree/user_asset.rs:99-150; tx_hash =
// two_to_one(diff_root, nonce), src/transaction/gadgets/purge/mod.rs:296).  glp_smt_root_of_set builds ONE tree with device-wide
// launches (sort, radix tree, ~40 rounds): right for 2^16+ entries, ~0.3 ms of fixed cost for a five-entry tree.  Here the shape of
// every tree is worked out on the host (sorted keys, split by the next bit, unary chains from common-prefix lengths — no hashing)
// as the same value-id jobs glp_smt_apply_updates uses, and ALL trees are hashed together, level by level.
namespace {
struct SetsBuilder {
    const uint64_t* keys;    // canonical, [n][4]
    const uint64_t* values;  // canonical
    std::vector<uint32_t>& perm;
    std::vector<std::vector<trees::SmtNodeJob>>& level_jobs;
    std::vector<uint64_t>& leaf_kv;
    std::vector<int>& leaf_ids;
    int& next_id;
    const uint64_t* key(size_t j) const { return keys + 4 * (size_t)perm[j]; }
    static int lcp(const uint64_t* a, const uint64_t* b) {  // common prefix in key bits (bit 0 first)
        for (int l = 0; l < 4; ++l)
            if (const uint64_t x = a[l] ^ b[l]) return 64 * l + __builtin_ctzll(x);
        return 256;
    }
    std::pair<int, int> node_job(int a, int la, int b, int lb) {
        const int id = next_id++, level = 1 + std::max(la, lb);
        if ((size_t)level >= level_jobs.size()) level_jobs.resize(level + 1);
        level_jobs[level].push_back({id, a, b});
        return {id, level};
    }
    // (value id, level) of the subtree holding the sorted entries [lo, hi) as it hangs at trie depth `depth` (hi - lo >= 1)
    std::pair<int, int> build(size_t lo, size_t hi, int depth) {
        if (hi - lo == 1) {
            const int id = next_id++;
            const uint64_t* kk = key(lo);
            leaf_kv.insert(leaf_kv.end(), kk, kk + 4);
            leaf_kv.insert(leaf_kv.end(), values + 4 * (size_t)perm[lo], values + 4 * (size_t)perm[lo] + 4);
            leaf_ids.push_back(id);
            return {id, 1};
        }
        const int d = lcp(key(lo), key(hi - 1));  // the branching depth of this range (sorted: first and last differ the earliest)
        size_t mid = lo;                           // first entry whose bit d is 1
        {
            size_t a = lo, b = hi;
            while (a < b) {
                const size_t m = (a + b) / 2;
                if (key_bit_at(key(m), d)) b = m; else a = m + 1;
            }
            mid = a;
        }
        auto l = build(lo, mid, d + 1), r = build(mid, hi, d + 1);
        auto cur = node_job(l.first, l.second, r.first, r.second);
        for (int t = d - 1; t >= depth; --t)  // the unary nodes between this branching node and where the range hangs
            cur = key_bit_at(key(lo), t) ? node_job(0, 0, cur.first, cur.second) : node_job(cur.first, cur.second, 0, 0);
        return cur;
    }
};
}  // namespace

extern "C" int32_t glp_smt_roots_of_sets(glp_ctx* c, const uint64_t* keys, const uint64_t* values, const size_t* set_offsets,
                                         size_t n_sets, uint64_t* roots_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !set_offsets || (!roots_out && n_sets)) return GLP_ERR_INVALID;
    if (n_sets == 0) return GLP_OK;
    const size_t n = set_offsets[n_sets];
    if ((!keys || !values) && n) return GLP_ERR_INVALID;
    if (n > ((size_t)1 << 26)) return GLP_ERR_INVALID;
    for (size_t s2 = 0; s2 < n_sets; ++s2)
        if (set_offsets[s2] > set_offsets[s2 + 1]) return GLP_ERR_INVALID;
    std::vector<uint64_t> ck(4 * n), cv(4 * n);
    for (size_t i = 0; i < 4 * n; ++i) ck[i] = gl::canon(keys[i]), cv[i] = gl::canon(values[i]);
    std::vector<uint32_t> perm;
    perm.reserve(n);
    std::vector<std::vector<trees::SmtNodeJob>> level_jobs;
    std::vector<uint64_t> leaf_kv;
    std::vector<int> leaf_ids, root_ids(n_sets, 0);
    leaf_kv.reserve(8 * n);
    leaf_ids.reserve(n);
    int next_id = 1;  // id 0 = 0^4
    SetsBuilder sb{ck.data(), cv.data(), perm, level_jobs, leaf_kv, leaf_ids, next_id};
    auto bit_order = [&](uint32_t a, uint32_t b) {  // by the key's bit string, bit 0 first
        const uint64_t *x = &ck[4 * (size_t)a], *y = &ck[4 * (size_t)b];
        for (int l = 0; l < 4; ++l)
            if (const uint64_t d = x[l] ^ y[l]) return ((y[l] >> __builtin_ctzll(d)) & 1) != 0;
        return false;
    };
    for (size_t s2 = 0; s2 < n_sets; ++s2) {
        const size_t lo = perm.size();
        for (size_t i = set_offsets[s2]; i < set_offsets[s2 + 1]; ++i) {
            const uint64_t* v = &cv[4 * i];
            if (v[0] | v[1] | v[2] | v[3]) perm.push_back((uint32_t)i);  // value 0^4 = absent
        }
        // duplicates are checked among ALL entries of the set, present or absent (the argument is a set)
        std::vector<uint32_t> all(set_offsets[s2 + 1] - set_offsets[s2]);
        std::iota(all.begin(), all.end(), (uint32_t)set_offsets[s2]);
        std::sort(all.begin(), all.end(), bit_order);
        for (size_t j = 1; j < all.size(); ++j)
            if (key_eq(&ck[4 * (size_t)all[j]], &ck[4 * (size_t)all[j - 1]])) return GLP_ERR_INVALID;
        std::sort(perm.begin() + lo, perm.end(), bit_order);
        if (perm.size() > lo) root_ids[s2] = sb.build(lo, perm.size(), 0).first;
    }
    const size_t n_ids = (size_t)next_id;
    Scratch sc(c);
    GLP_TRY(arena_reset(c));
    Staging stg(c);
    uint64_t* table = nullptr;
    GLP_TRY(sc.alloc(n_ids * 4, &table));
    HIP_TRY(hipMemsetAsync(table, 0, 32, c->stream));  // id 0; every other id is written before it is read
    if (!leaf_ids.empty()) {
        uint64_t *kv_dev = nullptr, *ids_dev = nullptr;
        GLP_TRY(sc.alloc(leaf_kv.size(), &kv_dev));
        GLP_TRY(sc.alloc((leaf_ids.size() + 1) / 2, &ids_dev));
        GLP_TRY(stg.up(kv_dev, leaf_kv.data(), leaf_kv.size() * 8));
        GLP_TRY(stg.up(ids_dev, leaf_ids.data(), leaf_ids.size() * 4));
        LAUNCH_BY_SIZE(c, leaf_ids.size(), smt_leaf_jobs_kernel, table, kv_dev, reinterpret_cast<const int*>(ids_dev), leaf_ids.size());
    }
    size_t n_jobs = 0;
    for (auto& lv : level_jobs) n_jobs += lv.size();
    std::vector<trees::SmtNodeJob> flat;
    if (n_jobs) {
        flat.reserve(n_jobs);
        for (auto& lv : level_jobs) flat.insert(flat.end(), lv.begin(), lv.end());
        uint64_t* jobs_dev = nullptr;
        GLP_TRY(sc.alloc((n_jobs * sizeof(trees::SmtNodeJob) + 7) / 8, &jobs_dev));
        GLP_TRY(stg.up(jobs_dev, flat.data(), n_jobs * sizeof(trees::SmtNodeJob)));
        size_t off = 0;
        for (auto& lv : level_jobs) {
            if (lv.empty()) continue;
            LAUNCH_BY_SIZE(c, lv.size(), smt_node_jobs_kernel, table, reinterpret_cast<const trees::SmtNodeJob*>(jobs_dev) + off, lv.size());
            off += lv.size();
        }
    }
    // the roots, gathered on the host side from one read-back of the table (n_ids * 32 bytes: a few MB for 2^16 entries)
    std::vector<uint64_t> tab(n_ids * 4);
    GLP_TRY(stg.down(tab.data(), table, n_ids * 32));
    HIP_TRY(hipStreamSynchronize(c->stream));
    stg.finish();
    for (size_t s2 = 0; s2 < n_sets; ++s2) std::memcpy(roots_out + 4 * s2, &tab[(size_t)root_ids[s2] * 4], 32);
    return GLP_OK;
}
