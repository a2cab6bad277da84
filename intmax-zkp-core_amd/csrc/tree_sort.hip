// tree_sort.hip — device-side ordering of sparse-Merkle entries for glp_smt_root_of_set (trees.hip): the entries are
// sorted by the key's bit string (bit 0 first) = lexicographic order of the bit-reversed limbs, with rocPRIM's radix sort
// as a 4-pass LSD sort over the 64-bit limbs (stable pair sorts of (limb, entry index)); entries whose value is 0^4 are
// absent from the tree and moved behind the present ones by a final 1-bit pass.  Own translation unit: rocPRIM is
// header-only and heavy to compile.
#include <cstring>  // before the HIP headers: rocPRIM's host code calls ::memset

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "internal.h"

namespace glpi {

namespace {
constexpr int T = 256;

__global__ __launch_bounds__(T) void iota_kernel(uint32_t* perm, size_t n) {
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    if (i < n) perm[i] = (uint32_t)i;
}

// inputs may be any u64 representative (include/glprover.h): absence (value == 0^4), order and duplicate detection are
// decided on canonical words
__global__ __launch_bounds__(T) void canon_kernel(uint64_t* __restrict__ words, size_t n_words) {
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    if (i < n_words) words[i] = gl::canon(words[i]);
}

// equal neighbours in the key order (all n entries, present or absent: a key may appear once in a set)
__global__ __launch_bounds__(T) void dup_check_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ perm, size_t n,
                                                      unsigned* dup_flag) {
    const size_t j = (size_t)blockIdx.x * T + threadIdx.x;
    if (j == 0 || j >= n) return;
    const uint64_t* a = keys + (size_t)perm[j] * 4;
    const uint64_t* b = keys + (size_t)perm[j - 1] * 4;
    if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3]) atomicOr(dup_flag, 1u);
}

// sort key of pass `limb`: bit-reversed limb of the entry at position j of the current order; limb 4 = absence flag
__global__ __launch_bounds__(T) void sort_key_kernel(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ values,
                                                     const uint32_t* __restrict__ perm, size_t n, int limb,
                                                     uint64_t* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * T + threadIdx.x;
    if (j >= n) return;
    const size_t i = perm[j];
    if (limb < 4) {
        out[j] = __brevll(keys[i * 4 + limb]);
    } else {
        const uint64_t* v = values + i * 4;
        out[j] = (v[0] | v[1] | v[2] | v[3]) ? 0 : 1;
    }
}

__global__ __launch_bounds__(T) void count_absent_kernel(const uint64_t* __restrict__ values, size_t n, unsigned* count) {
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    const bool absent = i < n && !(values[i * 4] | values[i * 4 + 1] | values[i * 4 + 2] | values[i * 4 + 3]);
    const unsigned long long b = __ballot(absent);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned)__popcll(b));
}

// rk[j] = bit-reversed key of the j-th entry in tree order, v[j] its value
__global__ __launch_bounds__(T) void gather_sorted_kernel(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ values,
                                                          const uint32_t* __restrict__ perm, size_t m, uint64_t* __restrict__ rk,
                                                          uint64_t* __restrict__ v) {
    const size_t j = (size_t)blockIdx.x * T + threadIdx.x;
    if (j >= m) return;
    const size_t i = perm[j];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        rk[j * 4 + l] = __brevll(keys[i * 4 + l]);
        v[j * 4 + l] = values[i * 4 + l];
    }
}
}  // namespace

// keys / values: HOST arrays [n][4] (inputs_on_device: device arrays, read only).  On success *rk_dev / *v_dev (pool allocations
// owned by the caller) hold the *m_out present entries in tree order; GLP_ERR_INVALID on a duplicate key.
int32_t smt_sort_device(glp_ctx* c, const uint64_t* keys, const uint64_t* values, size_t n, uint64_t** rk_dev,
                        uint64_t** v_dev, size_t* m_out, bool inputs_on_device) {
    *rk_dev = *v_dev = nullptr;
    *m_out = 0;
    if (n == 0) return GLP_OK;
    uint64_t *k_in = nullptr, *v_in = nullptr, *sk_a = nullptr, *sk_b = nullptr;
    uint32_t *perm_a = nullptr, *perm_b = nullptr;
    unsigned* flags = nullptr;  // [0] absent count, [1] duplicate flag
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, sk_a, sk_b, perm_a, perm_b, n, 0, 64, c->stream);
    int32_t s = e == hipSuccess ? GLP_OK : GLP_ERR_HIP;
    if (s == GLP_OK) s = pool_alloc_t(c, 4 * n, &k_in);
    if (s == GLP_OK) s = pool_alloc_t(c, 4 * n, &v_in);
    if (s == GLP_OK) s = pool_alloc_t(c, n, &sk_a);
    if (s == GLP_OK) s = pool_alloc_t(c, n, &sk_b);
    if (s == GLP_OK) s = pool_alloc_t(c, n, &perm_a);
    if (s == GLP_OK) s = pool_alloc_t(c, n, &perm_b);
    if (s == GLP_OK) s = pool_alloc_t(c, 2, &flags);
    if (s == GLP_OK) s = pool_alloc(c, tmp_bytes ? tmp_bytes : 8, &tmp);
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    const unsigned grid = (unsigned)((n + T - 1) / T);
    unsigned host_flags[2] = {0, 0};
    size_t m = 0;
    if (s == GLP_OK) {
        Timed t(c, GLP_K_TREES);
        if (inputs_on_device) {
            step(hipMemcpyAsync(k_in, keys, 32 * n, hipMemcpyDeviceToDevice, c->stream));
            step(hipMemcpyAsync(v_in, values, 32 * n, hipMemcpyDeviceToDevice, c->stream));
        } else {  // the caller's arrays: through pinned memory when they are pageable (internal.h: upload_staged)
            s = upload_staged(c, k_in, keys, 32 * n);
            if (s == GLP_OK) s = upload_staged(c, v_in, values, 32 * n);
        }
    }
    if (s == GLP_OK) {
        Timed t(c, GLP_K_TREES);
        step(hipMemsetAsync(flags, 0, 2 * sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(canon_kernel, dim3((unsigned)((4 * n + T - 1) / T)), dim3(T), 0, c->stream, k_in, 4 * n);
        hipLaunchKernelGGL(canon_kernel, dim3((unsigned)((4 * n + T - 1) / T)), dim3(T), 0, c->stream, v_in, 4 * n);
        hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(T), 0, c->stream, perm_a, n);
        hipLaunchKernelGGL(count_absent_kernel, dim3(grid), dim3(T), 0, c->stream, v_in, n, flags);
        step(hipGetLastError());
        step(hipMemcpyAsync(host_flags, flags, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        step(hipStreamSynchronize(c->stream));
        m = n - host_flags[0];
        for (int pass = 0; pass < 5 && e == hipSuccess; ++pass) {  // limb 3, 2, 1, 0, then the absence flag
            const int limb = pass < 4 ? 3 - pass : 4;
            if (limb == 4 && m == n) break;
            hipLaunchKernelGGL(sort_key_kernel, dim3(grid), dim3(T), 0, c->stream, k_in, v_in, perm_a, n, limb, sk_a);
            step(hipGetLastError());
            step(rocprim::radix_sort_pairs(tmp, tmp_bytes, sk_a, sk_b, perm_a, perm_b, n, 0, limb == 4 ? 1 : 64, c->stream));
            std::swap(perm_a, perm_b);
            if (limb == 0) {  // key order established: duplicates are neighbours, whether present or absent
                hipLaunchKernelGGL(dup_check_kernel, dim3(grid), dim3(T), 0, c->stream, k_in, perm_a, n, flags + 1);
                step(hipGetLastError());
            }
        }
        step(hipMemcpyAsync(host_flags + 1, flags + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        step(hipStreamSynchronize(c->stream));
    }
    if (s == GLP_OK && e == hipSuccess && m > 0 && !host_flags[1]) {
        s = pool_alloc_t(c, 4 * m, rk_dev);
        if (s == GLP_OK) s = pool_alloc_t(c, 4 * m, v_dev);
        if (s == GLP_OK) {
            Timed t(c, GLP_K_TREES);
            hipLaunchKernelGGL(gather_sorted_kernel, dim3((unsigned)((m + T - 1) / T)), dim3(T), 0, c->stream, k_in, v_in, perm_a, m,
                               *rk_dev, *v_dev);
            step(hipGetLastError());
            step(hipStreamSynchronize(c->stream));
        }
    }
    pool_free(c, k_in);
    pool_free(c, v_in);
    pool_free(c, sk_a);
    pool_free(c, sk_b);
    pool_free(c, perm_a);
    pool_free(c, perm_b);
    pool_free(c, flags);
    pool_free(c, tmp);
    if (s == GLP_OK && e != hipSuccess) {
        note_hip_error(e);
        s = e == hipErrorOutOfMemory ? GLP_ERR_OOM : GLP_ERR_HIP;
    }
    if (s == GLP_OK && host_flags[1]) s = GLP_ERR_INVALID;  // duplicate key
    if (s != GLP_OK) {
        pool_free(c, *rk_dev);
        pool_free(c, *v_dev);
        *rk_dev = *v_dev = nullptr;
        return s;
    }
    *m_out = m;
    return GLP_OK;
}

}  // namespace glpi
