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
    // host side: drop absent entries (value 0^4), order by key bit string = lexicographic order of the bit-reversed limbs
    std::vector<uint32_t> order;
    order.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const uint64_t* v = values + 4 * i;
        if (v[0] | v[1] | v[2] | v[3]) order.push_back((uint32_t)i);
    }
    const size_t m = order.size();
    std::vector<uint64_t> rk(4 * (m ? m : 1)), vv(4 * (m ? m : 1));
    {
        // LSD radix sort of (bit-reversed limb 0, entry) pairs, 11-bit digits; the rare runs that agree on all of limb 0
        // are finished with a comparison sort on the remaining limbs
        struct Pair { uint64_t k; uint32_t i; };
        std::vector<Pair> a(m), b(m);
        for (size_t t = 0; t < m; ++t) a[t] = {__builtin_bitreverse64(keys[4 * (size_t)order[t]]), order[t]};
        // only the top `bits` bits are radix-sorted: enough to leave (for well-spread keys) almost no ties
        int bits = 8;
        while (((size_t)1 << (bits - 8)) < m && bits < 64) ++bits;
        bits = std::min(64, (bits + 10) / 11 * 11);
        const int low = 64 - bits;
        for (int shift = low; shift < 64; shift += 11) {
            size_t cnt[2049] = {0};
            for (size_t t = 0; t < m; ++t) ++cnt[((a[t].k >> shift) & 2047) + 1];
            for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
            for (size_t t = 0; t < m; ++t) b[cnt[(a[t].k >> shift) & 2047]++] = a[t];
            a.swap(b);
        }
        auto rest_less = [&](const Pair& x, const Pair& y) {
            if (x.k != y.k) return x.k < y.k;
            for (int l = 1; l < 4; ++l) {
                const uint64_t u = __builtin_bitreverse64(keys[4 * (size_t)x.i + l]), v = __builtin_bitreverse64(keys[4 * (size_t)y.i + l]);
                if (u != v) return u < v;
            }
            return false;
        };
        for (size_t t = 0; t < m;) {
            size_t e = t + 1;
            while (e < m && (a[e].k >> low) == (a[t].k >> low)) ++e;
            if (e - t > 1) std::sort(a.begin() + t, a.begin() + e, rest_less);
            t = e;
        }
        for (size_t t = 0; t < m; ++t) {
            for (int l = 0; l < 4; ++l) rk[4 * t + l] = __builtin_bitreverse64(keys[4 * (size_t)a[t].i + l]);
            std::memcpy(&vv[4 * t], values + 4 * (size_t)a[t].i, 32);
            if (t && std::memcmp(&rk[4 * t], &rk[4 * (t - 1)], 32) == 0) return GLP_ERR_INVALID;  // duplicate key
        }
    }
    if (m == 0) {
        std::memset(root_out, 0, 32);
        return GLP_OK;
    }
    uint64_t *rk_dev = nullptr, *v_dev = nullptr, *lh_dev = nullptr, *root_dev = nullptr;
    trees::RadixNode* nodes = nullptr;
    int *lp = nullptr, *ls = nullptr;
    int32_t s = pool_alloc_t(c, 4 * m, &rk_dev);
    if (s == GLP_OK) s = pool_alloc_t(c, 4 * m, &v_dev);
    if (s == GLP_OK) s = pool_alloc_t(c, 4 * m, &lh_dev);
    if (s == GLP_OK) s = pool_alloc_t(c, 4, &root_dev);
    if (s == GLP_OK && m > 1) s = pool_alloc_t(c, m - 1, &nodes);
    if (s == GLP_OK && m > 1) s = pool_alloc_t(c, m, &lp);
    if (s == GLP_OK && m > 1) s = pool_alloc_t(c, m, &ls);
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    if (s == GLP_OK) {
        step(hipMemcpyAsync(rk_dev, rk.data(), 32 * m, hipMemcpyHostToDevice, c->stream));
        step(hipMemcpyAsync(v_dev, vv.data(), 32 * m, hipMemcpyHostToDevice, c->stream));
        if (e == hipSuccess) {
            Timed t(c, GLP_K_TREES);
            if (m <= COOP_ITEMS_MAX)
                hipLaunchKernelGGL((trees::smt_leaf_hash_kernel<trees::CoopHasher>), dim3(grid_for<trees::CoopHasher>(m)),
                                   dim3(trees::THREADS), 0, c->stream, rk_dev, v_dev, m, 1, m > 1 ? lh_dev : root_dev);
            else
                hipLaunchKernelGGL((trees::smt_leaf_hash_kernel<trees::LaneHasher>), dim3(grid_for<trees::LaneHasher>(m)),
                                   dim3(trees::THREADS), 0, c->stream, rk_dev, v_dev, m, 1, lh_dev);
            step(hipGetLastError());
        }
        if (e == hipSuccess && m > 1) {
            Timed t(c, GLP_K_TREES);
            hipLaunchKernelGGL(trees::radix_build_kernel, dim3((unsigned)((m - 1 + trees::THREADS - 1) / trees::THREADS)),
                               dim3(trees::THREADS), 0, c->stream, rk_dev, (int)m, nodes, lp, ls);
            step(hipGetLastError());
            hipLaunchKernelGGL(trees::radix_hash_kernel, dim3((unsigned)((m + trees::THREADS - 1) / trees::THREADS)),
                               dim3(trees::THREADS), 0, c->stream, rk_dev, lh_dev, (int)m, nodes, lp, ls, root_dev);
            step(hipGetLastError());
        }
        step(hipMemcpyAsync(root_out, root_dev, 32, hipMemcpyDeviceToHost, c->stream));
        step(hipStreamSynchronize(c->stream));
    }
    pool_free(c, rk_dev);
    pool_free(c, v_dev);
    pool_free(c, lh_dev);
    pool_free(c, root_dev);
    pool_free(c, nodes);
    pool_free(c, lp);
    pool_free(c, ls);
    if (s != GLP_OK) return s;
    if (e != hipSuccess) {
        g_last_hip = (int32_t)e;
        return e == hipErrorOutOfMemory ? GLP_ERR_OOM : GLP_ERR_HIP;
    }
    return GLP_OK;
}

}  // extern "C"
