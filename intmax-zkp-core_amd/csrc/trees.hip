// trees.hip — C-ABI drivers of the native tree hashing kernels (tree_kernels.h).
#include <algorithm>
#include <numeric>

#include "internal.h"
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
struct SmtTree {
    glp_ctx* c;
    uint64_t *rk_dev = nullptr, *v_dev = nullptr, *lh_dev = nullptr, *root_dev = nullptr;
    trees::RadixNode* nodes = nullptr;
    int *lp = nullptr, *ls = nullptr, *queue = nullptr, *counts = nullptr;
    size_t m = 0;  // present entries
    explicit SmtTree(glp_ctx* ctx) : c(ctx) {}
    SmtTree(const SmtTree&) = delete;
    SmtTree& operator=(const SmtTree&) = delete;
    ~SmtTree() {
        pool_free(c, rk_dev);
        pool_free(c, v_dev);
        pool_free(c, lh_dev);
        pool_free(c, root_dev);
        pool_free(c, nodes);
        pool_free(c, lp);
        pool_free(c, ls);
        pool_free(c, queue);
        pool_free(c, counts);
    }
    int32_t build(const uint64_t* keys, const uint64_t* values, size_t n) {
        // drop absent entries (value 0^4), order by key bit string, reject duplicate keys: all on the device (tree_sort.hip)
        GLP_TRY(smt_sort_device(c, keys, values, n, &rk_dev, &v_dev, &m));
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

}  // namespace

extern "C" {

int32_t glp_hash_batch_dev(glp_ctx* c, const uint64_t* in_dev, size_t n, size_t len, size_t item_stride, uint32_t mode,
                           uint64_t* out_dev) {
    if (!c || !out_dev || (!in_dev && len) || mode > GLP_HASH_OR_NOOP || len > (1u << 24) || item_stride < len)
        return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, hash_batch_kernel, in_dev, n, (int)len, item_stride, (int)mode, out_dev);
    return GLP_OK;
}

int32_t glp_two_to_one_batch_dev(glp_ctx* c, const uint64_t* left_dev, const uint64_t* right_dev, size_t n,
                                 uint64_t* out_dev) {
    if (!c || !left_dev || !right_dev || !out_dev) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, two_to_one_batch_kernel, left_dev, right_dev, n, out_dev);
    return GLP_OK;
}

int32_t glp_smt_leaf_hash_batch_dev(glp_ctx* c, const uint64_t* keys_dev, const uint64_t* values_dev, size_t n,
                                    uint64_t* out_dev) {
    if (!c || !keys_dev || !values_dev || !out_dev) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, smt_leaf_hash_kernel, keys_dev, values_dev, n, 0, out_dev);
    return GLP_OK;
}

int32_t glp_smt_root_batch_dev(glp_ctx* c, const uint64_t* keys_dev, const uint64_t* start_dev, const uint64_t* siblings_dev,
                               const uint32_t* n_siblings_dev, size_t n, size_t max_siblings, uint64_t* out_dev) {
    if (!c || !keys_dev || !start_dev || !out_dev || max_siblings > 256 || (!siblings_dev && max_siblings))
        return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, smt_root_batch_kernel, keys_dev, start_dev, siblings_dev, n_siblings_dev, n, (int)max_siblings,
                   out_dev);
    return GLP_OK;
}

int32_t glp_merkle_root_batch_dev(glp_ctx* c, const uint64_t* index_dev, const uint64_t* value_dev,
                                  const uint64_t* siblings_dev, size_t n, size_t depth, uint64_t* out_dev) {
    if (!c || !index_dev || !value_dev || !out_dev || depth > 256 || (!siblings_dev && depth)) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, merkle_root_batch_kernel, index_dev, value_dev, siblings_dev, n, (int)depth, out_dev);
    return GLP_OK;
}

int32_t glp_block_hash_batch_dev(glp_ctx* c, const uint64_t* headers_dev, size_t n, uint64_t* out_dev) {
    if (!c || !headers_dev || !out_dev) return GLP_ERR_INVALID;
    if (n == 0) return GLP_OK;
    LAUNCH_BY_SIZE(c, n, block_hash_batch_kernel, headers_dev, n, out_dev);
    return GLP_OK;
}

size_t glp_merkle_proof_n_siblings(size_t depth) { return depth ? depth : 1; }

int32_t glp_merkle_proof_with_zero_dev(glp_ctx* c, const uint64_t* leaves_dev, size_t n_leaves, size_t index, size_t depth,
                                       const uint64_t zero[4], uint64_t* siblings_out, uint64_t* root_out) {
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
        g_last_hip = (int32_t)e;
        return e == hipErrorOutOfMemory ? GLP_ERR_OOM : GLP_ERR_HIP;
    }
    return GLP_OK;
}

int32_t glp_smt_root_of_set(glp_ctx* c, const uint64_t* keys, const uint64_t* values, size_t n, uint64_t root_out[4]) {
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
    if (!c || ((!keys || !values) && n) || n > (1u << 30) || (!query_keys && n_queries) || max_siblings > 256 ||
        (siblings_out && max_siblings == 0 && n_queries))
        return GLP_ERR_INVALID;
    SmtTree t(c);
    GLP_TRY(t.build(keys, values, n));
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
    HIP_TRY(hipMemcpyAsync(q_dev, query_keys, n_queries * 32, hipMemcpyHostToDevice, c->stream));
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
        GLP_TRY(arena_reset(c));
        stage = arena_alloc_t<uint64_t>(c, sib_words);
        if (!stage) return GLP_ERR_OOM;
        HIP_TRY(hipMemcpyAsync(stage, res, sib_words * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(host.data(), res + sib_words, host.size() * 8, hipMemcpyDeviceToHost, c->stream));
    if (t.m) HIP_TRY(hipMemcpyAsync(root, t.root_dev, 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
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

}  // extern "C"
