// fri_kernels.h — kernels of the opening / FRI phase (extension-field arithmetic on limb-separated columns).
//
// Replaces plonky2 @ f99ed9c src/plonk/proof.rs::OpeningSet::new (eval_commitment), src/fri/oracle.rs::prove_openings
// (reduce_polys_base, divide_by_linear, lde + coset_fft) and src/fri/prover.rs (fri_committed_trees fold,
// fri_proof_of_work), reached from the reference through CircuitData::prove (src/rollup/circuits/mod.rs:1247).
//
// An extension element (a + bX, X^2 = 7) is stored as two base-field words in two separate columns ("limbs"), so that
// every extension NTT is just the batched base-field NTT on 2 columns.
#pragma once
#include <hip/hip_runtime.h>

#include "poseidon.h"

namespace fri {

constexpr int THREADS = 256;

// zpow[0][j], zpow[1][j] = limbs of z^j, j < n.  z2k[k] = z^(2^k) (host-computed, k < 32)
// blockIdx.y = point (the batched prover evaluates at one zeta per proof): table y at out + y * 2n
struct PowArgs {
    uint64_t z2k[32][2];
};
__global__ __launch_bounds__(THREADS) void ext_powers_kernel(const PowArgs* __restrict__ args, uint64_t* __restrict__ out, size_t n) {
    // thread t produces z^(16 t) .. z^(16 t + 15)
    const PowArgs& a = args[blockIdx.y];
    uint64_t* out0 = out + (size_t)blockIdx.y * 2 * n;
    uint64_t* out1 = out0 + n;
    size_t t = (size_t)blockIdx.x * THREADS + threadIdx.x;
    size_t j0 = t * 16;
    if (j0 >= n) return;
    gl::ext2 acc = {1, 0};
    for (int k = 0; k < 32; ++k)
        if ((j0 >> k) & 1) acc = gl::ext_mul(acc, gl::ext2{a.z2k[k][0], a.z2k[k][1]});
    const gl::ext2 z = {a.z2k[0][0], a.z2k[0][1]};
#pragma unroll 1
    for (int i = 0; i < 16 && j0 + i < n; ++i) {
        out0[j0 + i] = gl::canon(acc.a);
        out1[j0 + i] = gl::canon(acc.b);
        acc = gl::ext_mul(acc, z);
    }
}

// partial[col][blk] = sum_{j in chunk blk} coeffs[col][j] * zpow[j]     (chunk = THREADS * PER elements)
constexpr int EVAL_PER = 16;
// blockIdx.z = point k: columns [k * tree_cols, k * tree_cols + gridDim.y) of `coeffs` (tree_cols = 0: every point evaluates the
// same columns — the circuit's constants_sigmas batch is shared by all proofs) against the power table zpow + k * 2n
__global__ __launch_bounds__(THREADS) void eval_partial_kernel(const uint64_t* __restrict__ coeffs, size_t col_stride,
                                                               size_t n, size_t tree_cols, const uint64_t* __restrict__ zpow,
                                                               uint64_t* __restrict__ partial /* [point][cols][blocks][2] */) {
    __shared__ uint64_t red[2][THREADS];
    const size_t col = blockIdx.y + (size_t)blockIdx.z * gridDim.y;  // flat output column
    const uint64_t* p = coeffs + ((size_t)blockIdx.z * tree_cols + blockIdx.y) * col_stride;
    const uint64_t* zp0 = zpow + (size_t)blockIdx.z * 2 * n;
    const uint64_t* zp1 = zp0 + n;
    const size_t base = (size_t)blockIdx.x * THREADS * EVAL_PER;
    uint64_t a0 = 0, a1 = 0;
#pragma unroll 4
    for (int k = 0; k < EVAL_PER; ++k) {
        size_t j = base + (size_t)k * THREADS + threadIdx.x;
        if (j < n) {
            uint64_t cj = p[j];
            a0 = gl::add(a0, gl::mul(cj, zp0[j]));
            a1 = gl::add(a1, gl::mul(cj, zp1[j]));
        }
    }
    red[0][threadIdx.x] = a0;
    red[1][threadIdx.x] = a1;
    __syncthreads();
    for (int s = THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = gl::add(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = gl::add(red[1][threadIdx.x], red[1][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        size_t o = (col * gridDim.x + blockIdx.x) * 2;
        partial[o] = gl::canon(red[0][0]);
        partial[o + 1] = gl::canon(red[1][0]);
    }
}

__global__ __launch_bounds__(64) void eval_final_kernel(const uint64_t* __restrict__ partial, int n_blocks,
                                                        uint64_t* __restrict__ out /* [cols][2] */) {
    const size_t col = blockIdx.x;
    uint64_t a0 = 0, a1 = 0;
    for (int b = threadIdx.x; b < n_blocks; b += 64) {
        a0 = gl::add(a0, partial[(col * n_blocks + b) * 2]);
        a1 = gl::add(a1, partial[(col * n_blocks + b) * 2 + 1]);
    }
    __shared__ uint64_t red[2][64];
    red[0][threadIdx.x] = a0;
    red[1][threadIdx.x] = a1;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 64; ++i) {
            a0 = gl::add(a0, red[0][i]);
            a1 = gl::add(a1, red[1][i]);
        }
        out[col * 2] = gl::canon(a0);
        out[col * 2 + 1] = gl::canon(a1);
    }
}

// One FRI batch on the size-n coset shift*H (= chunk 0 of every LDE column, bit-reversed positions):
//   comp(x_p) = sum_k alpha^k f_k(x_p);  acc[p] = acc[p] * alpha^(n_polys) + (comp(x_p) - comp(z)) / (x_p - z)
// cols[k] points at position 0 of LDE column k; apow = alpha^k limbs [n_polys][2]; x_p = shift * w_n^rev(p).
struct CombineProof {      // per proof of the batch (blockIdx.y)
    uint64_t opened[2];    // comp(z) = sum_k alpha^k f_k(z)
    uint64_t z[2];
    uint64_t ashift[2];    // alpha^(n_polys)
};
struct CombineArgs {
    const uint64_t* const* cols;  // [proof][n_polys]
    const uint64_t* apow;         // [proof][n_polys][2]
    const CombineProof* proofs;   // [proof]
    int n_polys;
    int log_n;
    uint64_t shift;
    int first;             // 1: acc starts at zero
    int last;              // 1: multiply the result by x_p (prove_openings' "multiply by X")
    int K;
    const uint64_t* w_lo;
    const uint64_t* w_hi;
};
// acc: [proof][2][n] limb columns
// COOP = false: one lane per point walks the n_polys columns (every step waits for a pointer and a value: ~0.3 us per column
// whatever the size — fine when there are points enough to hide it).  COOP = true, for launches of <= 2^14 points (the
// reference's small circuits): one WAVE per point, the lanes share the columns and the partial sums meet in a butterfly of
// shuffles; lane 0 finishes the point (2^4 rows: 74 -> 16 us for the 257 columns of a proof's first batch).
template <bool COOP>
__global__ __launch_bounds__(THREADS) void fri_combine_kernel(CombineArgs g, uint64_t* __restrict__ acc) {
    const int lane = COOP ? (int)(threadIdx.x & 63) : 0;
    const size_t p = COOP ? (size_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64 : (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= ((size_t)1 << g.log_n)) return;
    struct {
        const uint64_t* const* cols;
        const uint64_t* apow;
        uint64_t opened[2], z[2], ashift[2], shift;
        int n_polys, log_n, first, last, K;
        const uint64_t *w_lo, *w_hi;
    } a;
    {
        const size_t pk = blockIdx.y;
        const CombineProof pr = g.proofs[pk];
        a.cols = g.cols + pk * g.n_polys;
        a.apow = g.apow + pk * 2 * (size_t)g.n_polys;
        for (int i = 0; i < 2; ++i) a.opened[i] = pr.opened[i], a.z[i] = pr.z[i], a.ashift[i] = pr.ashift[i];
        a.shift = g.shift, a.n_polys = g.n_polys, a.log_n = g.log_n, a.first = g.first, a.last = g.last, a.K = g.K;
        a.w_lo = g.w_lo, a.w_hi = g.w_hi;
    }
    uint64_t* acc0 = acc + (size_t)blockIdx.y * 2 * ((size_t)1 << g.log_n);
    uint64_t* acc1 = acc0 + ((size_t)1 << g.log_n);
    uint64_t c0 = 0, c1 = 0;
    for (int k = lane; k < a.n_polys; k += COOP ? 64 : 1) {
        uint64_t v = a.cols[k][p];
        c0 = gl::add(c0, gl::mul(v, a.apow[2 * k]));
        c1 = gl::add(c1, gl::mul(v, a.apow[2 * k + 1]));
    }
    if (COOP) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t o0 = ((uint64_t)(uint32_t)__shfl_xor((int)(c0 >> 32), off) << 32) | (uint32_t)__shfl_xor((int)c0, off);
            const uint64_t o1 = ((uint64_t)(uint32_t)__shfl_xor((int)(c1 >> 32), off) << 32) | (uint32_t)__shfl_xor((int)c1, off);
            c0 = gl::add(c0, o0);
            c1 = gl::add(c1, o1);
        }
        if (lane != 0) return;
    }
    // x_p
    uint32_t e = a.log_n ? __brev((uint32_t)p) >> (32 - a.log_n) : 0;
    uint64_t ex = (uint64_t)e << (a.K - a.log_n);
    uint64_t x = gl::mul(a.shift, gl::mul(a.w_lo[ex & 4095], a.w_hi[ex >> 12]));
    gl::ext2 num = {gl::sub(c0, a.opened[0]), gl::sub(c1, a.opened[1])};
    gl::ext2 den = {gl::sub(x, a.z[0]), gl::neg(a.z[1])};
    gl::ext2 q = gl::ext_mul(num, gl::ext_inv(den));
    gl::ext2 r = q;
    if (!a.first) {
        gl::ext2 prev = {acc0[p], acc1[p]};
        r = gl::ext_add(gl::ext_mul(prev, gl::ext2{a.ashift[0], a.ashift[1]}), q);
    }
    if (a.last) r = gl::ext_mul_base(r, x);
    acc0[p] = gl::canon(r.a);
    acc1[p] = gl::canon(r.b);
}

// coefficient fold: out[j] = sum_{i < arity} beta^i c[arity*j + i]   (reduce_with_powers per chunk)
// blockIdx.y = proof: coefficients [proof][2][arity * n_out] -> [proof][2][n_out], beta = betas[proof]
__global__ __launch_bounds__(THREADS) void fri_fold_kernel(const uint64_t* __restrict__ coeffs, size_t n_out, int arity,
                                                           const uint64_t* __restrict__ betas, uint64_t* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (j >= n_out) return;
    const size_t n_in = n_out * (size_t)arity;
    const uint64_t* c0 = coeffs + (size_t)blockIdx.y * 2 * n_in;
    const uint64_t* c1 = c0 + n_in;
    uint64_t* o0 = out + (size_t)blockIdx.y * 2 * n_out;
    uint64_t* o1 = o0 + n_out;
    const gl::ext2 beta = {betas[2 * blockIdx.y], betas[2 * blockIdx.y + 1]};
    gl::ext2 acc = {0, 0};
    for (int i = arity - 1; i >= 0; --i) {
        acc = gl::ext_mul(acc, beta);
        acc = gl::ext_add(acc, gl::ext2{c0[(size_t)arity * j + i], c1[(size_t)arity * j + i]});
    }
    o0[j] = gl::canon(acc.a);
    o1[j] = gl::canon(acc.b);
}

// leaves of a commit-phase tree: values (bit-reversed order, limb columns) chunked by arity and flattened =
// simply the two limb columns interleaved: out[2p] = v0[p], out[2p+1] = v1[p]
// blockIdx.y = proof: vals [proof][2][n] -> out [proof][n][2]
__global__ __launch_bounds__(THREADS) void fri_interleave_kernel(const uint64_t* __restrict__ vals, size_t n,
                                                                 uint64_t* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= n) return;
    const uint64_t* v0 = vals + (size_t)blockIdx.y * 2 * n;
    reinterpret_cast<ulonglong2*>(out + (size_t)blockIdx.y * 2 * n)[p] = make_ulonglong2(v0[p], v0[n + p]);
}

// proof of work: smallest i >= start whose response has `bits` leading zeros (atomicMin on *best).  The response is word
// `out_word` of permute(state with state[pos] = i): the hash_no_pad(get_hash() || i) rule of f99ed9c is
// state = (h0..h3, i, 0..), word 0; the duplex rule of later revisions is the challenger's sponge state with its pending
// inputs written in, i behind them, word 7 (the challenge popped right after observing i).
// The candidates of one proof differ in ONE word of the permutation's input (`pos`), so everything the other eleven words
// contribute to round 0 is the same for all of them and computed once per proof on the host (pow_precompute, fri.hip):
//   base[r] = sum_{i != pos} MDS[r][i] * sbox(state[i] + RC[i]) + RC[12 + r]      (the round-1 constants folded in)
// and a candidate x enters round 1 as  base[r] + MDS[r][pos] * sbox(x + RC[pos]):  one S-box and 24 multiply-adds instead of
// twelve S-boxes and a 288-multiply layer.  Only word `out_word` of the result is read.
struct PowArgs2 {
    uint64_t base[12];
    int pos, out_word, bits;
};
// blockIdx.y = entry of `ids` (the proofs still without a witness; nullptr = all): args[proof], best[proof]; the grid scans
// the candidates start .. start + count - 1 of each.  A proof settled by an EARLIER launch (smaller candidates) is skipped by a
// plain read of its slot (coherent across kernel boundaries; inside one kernel a plain read may or may not see the atomicMin of a
// workgroup on another XCD — an in-kernel "stop when found" loop was measured and never stopped).  The host issues many short launches
// back to back, so no work is done for a settled proof beyond the launch that settled it: the grind is 2^proof_of_work_bits
// permutations per proof in expectation, which for the reference's small circuits is most of a proof's arithmetic.
__global__ __launch_bounds__(THREADS) void fri_pow_kernel(const PowArgs2* __restrict__ args, const uint32_t* __restrict__ ids, uint64_t start,
                                                          uint64_t count, unsigned long long* best_all) {
    const uint32_t proof = ids ? ids[blockIdx.y] : blockIdx.y;
    unsigned long long* best = best_all + proof;
    // Skip only what cannot matter: a workgroup whose FIRST candidate lies above a witness that is already known.  (Until round 4
    // this read `*best != ~0`: workgroups are not started in candidate order when another context's kernels hold some of the XCDs —
    // a workgroup of low candidates that started after one of high candidates had scored gave up its own, smaller witness.  Every
    // witness is valid, plonky2 itself takes any (`find_any`), but the proof was no longer a function of its inputs: one proof in
    // ~1 000 under five busy contexts differed from the sequential one in this word, tests/concurrent_mix.py.)
    if (*best < start + (uint64_t)blockIdx.x * THREADS) return;
    const uint64_t t = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    if (t >= count) return;
    const PowArgs2 a = args[proof];
    const uint64_t i = start + t;
    if (i >= gl::P) return;
    const uint64_t y = poseidon::sbox7(gl::add(i, poseidon::RC[a.pos]));
    const uint32_t ylo = (uint32_t)y, yhi = (uint32_t)(y >> 32);
    uint64_t st[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        const uint32_t coef = poseidon::MDS_CIRC_TAB[(a.pos + 12 - r) % 12] + (r == 0 && a.pos == 0 ? 8u : 0u);  // MDS[r][pos]
        st[r] = poseidon::mds_combine((uint32_t)a.base[r] + (uint64_t)ylo * coef, (a.base[r] >> 32) + (uint64_t)yhi * coef);
    }
    poseidon::permute_folded(st, a.out_word == 0 ? poseidon::OUT_WORD0 : a.out_word == 7 ? poseidon::OUT_WORD7 : poseidon::OUT_ALL, 1);
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k)
        if (k == a.out_word) v = st[k];
    v = gl::canon(v);
    if (a.bits == 0 || (v >> (64 - a.bits)) == 0) atomicMin(best, (unsigned long long)i);
}

}  // namespace fri
