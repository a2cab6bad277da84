// plonk_kernels.h — kernels of the PLONK layer: permutation partial products / Z, constraint-quotient evaluation on the
// 8n coset, PoseidonGate witness rows.
//
// Replaces plonky2 @ f99ed9c src/plonk/prover.rs::{wires_permutation_partial_products_and_zs, compute_quotient_polys},
// src/plonk/vanishing_poly.rs::eval_vanishing_poly_base_batch and the gates' eval_unfiltered_base_batch
// (src/gates/{noop,constant,public_input,arithmetic_base,poseidon}.rs), reached from the reference through
// CircuitData::prove (reference src/rollup/circuits/mod.rs:1247, src/zkdsa/circuits/mod.rs:326).
//
// Layout: every oracle is column-major with bit-reversed rows (see merkle_kernels.h), so a lane = one evaluation point
// reads every column with a coalesced wave load; the 135 wires of a point are streamed (each is consumed once per gate
// that uses it), never held in registers at the same time.
#pragma once
#include <hip/hip_runtime.h>

#include "poseidon.h"

namespace plonk {

#ifndef GLP_QW
#define GLP_QW 5  // waves per SIMD of quotient_kernel (register budget 512 / this): 5 measured best once the PoseidonGate left it
#endif

constexpr int THREADS = 256;
constexpr int MAX_GATES = 16;
constexpr int GATE_NOOP = 0, GATE_CONSTANT = 1, GATE_PUBLIC_INPUT = 2, GATE_ARITHMETIC = 3, GATE_POSEIDON = 4;
constexpr int GATE_BASE_SUM = 5, GATE_ARITHMETIC_EXT = 6, GATE_MUL_EXT = 7;
// the gates builder.verify_proof adds (reference src/recursion/gadgets/mod.rs:114-127)
constexpr int GATE_RANDOM_ACCESS = 8, GATE_REDUCING = 9, GATE_REDUCING_EXT = 10, GATE_EXPONENTIATION = 11;
constexpr int GATE_POSEIDON_MDS = 12, GATE_COSET_INTERPOLATION = 13;
constexpr int MAX_INTERP_POINTS = 16;  // CosetInterpolationGate subgroup_bits <= 4, RandomAccessGate bits <= 4

struct Gate {
    uint32_t id, selector_index, group_start, group_end, param;
};

// ------------------------------------------------------------------ partial products and Z
// quot[(ch * n_chunks + ck)][i] = prod_{j in chunk ck} (w_j + beta k_j x_i + gamma) / (w_j + beta sigma_j(x_i) + gamma)
// tot[ch][i] = product over the chunks (the factor Z(g x)/Z(x))
// Batched over proofs (blockIdx.z = proof; the single-proof entry points run a batch of one): proof k's witness starts at
// wires + k * wires_proof_stride, its challenges are chal[k] = (beta_0, beta_1, gamma_0, gamma_1).
struct PPArgs {
    const uint64_t* wires;   // [num_wires][n] witness values on H (natural order), per proof
    size_t wires_proof_stride;
    const uint64_t* sigmas;  // [num_routed][n] sigma values on H (shared by all proofs)
    const uint64_t* bk;      // [proof][nch][num_routed]  beta_ch * k_j
    const uint64_t* chal;    // [proof][4]
    uint64_t* quot;          // [proof][nch * n_chunks][n]
    uint64_t* tot;           // [proof][nch][n]
    int log_n, num_routed, chunk, n_chunks, nch;
    int K;
    const uint64_t* w_lo;
    const uint64_t* w_hi;
};

constexpr int MAX_PP_CHUNKS = 16;  // chunk quotients inverted together (80 routed wires / 8 = 10 under the standard config)

// one lane per (row, challenge): the chunk denominators are inverted together (Montgomery's trick: one field inversion
// and 3 multiplications per chunk instead of one ~95-multiplication inversion per chunk); a zero denominator gives a
// zero quotient for that chunk only, as the per-chunk inversion would
// MAXC = chunks inverted together: 10 for the standard configuration (three arrays of 10 instead of 16 words per lane: more waves
// for a kernel that waits on HBM), 16 for anything else
template <int MAXC>
__global__ __launch_bounds__(THREADS) void pp_quotients_kernel(PPArgs a) {
    const size_t n = (size_t)1 << a.log_n;
    // Both challenges of a row live in the SAME workgroup (waves 0..1 challenge 0, waves 2..3 challenge 1 under the standard
    // configuration): they read the same 80 wire and 80 sigma values, so the second read is served by the CU's cache instead of
    // HBM (one lane per (row, challenge) in separate workgroups read 2.69 GB for 1.51 algorithmic, profiles/r03s_summary.md).
    const int rows_per_block = THREADS / a.nch;  // nch is 1 or 2
    const int ch = (int)threadIdx.x / rows_per_block;
    const size_t i = (size_t)blockIdx.x * rows_per_block + (threadIdx.x - ch * rows_per_block);
    const size_t pk = blockIdx.z;  // proof
    if (i >= n) return;
    const uint64_t ex = (uint64_t)i << (a.K - a.log_n);
    const uint64_t x = gl::mul(a.w_lo[ex & 4095], a.w_hi[ex >> 12]);
    const uint64_t beta = a.chal[pk * 4 + ch], gamma = a.chal[pk * 4 + 2 + ch];
    const uint64_t* wires = a.wires + pk * a.wires_proof_stride;
    const uint64_t* bk = a.bk + pk * a.nch * a.num_routed;
    uint64_t* quot = a.quot + pk * (size_t)a.nch * a.n_chunks * n;
    uint64_t total = 1;  // product of the chunk quotients: the factor Z(g x) / Z(x)
    for (int ck0 = 0; ck0 < a.n_chunks; ck0 += MAXC) {
        uint64_t nump[MAXC], denp[MAXC], pre[MAXC];
        uint64_t run = 1;
#pragma unroll
        for (int t = 0; t < MAXC; ++t) {
            const int ck = ck0 + t;
            uint64_t nm = 1, dn = 1;
            if (ck < a.n_chunks)
                for (int j = ck * a.chunk; j < (ck + 1) * a.chunk && j < a.num_routed; ++j) {
                    const uint64_t wv = wires[(size_t)j * n + i];
                    const uint64_t wg = gl::add(wv, gamma);
                    nm = gl::mul(nm, gl::add(wg, gl::mul(bk[ch * a.num_routed + j], x)));
                    dn = gl::mul(dn, gl::add(wg, gl::mul(beta, a.sigmas[(size_t)j * n + i])));
                }
            nump[t] = nm;
            denp[t] = gl::canon(dn);
            pre[t] = run;  // product of the non-zero denominators before this chunk
            run = gl::mul(run, denp[t] ? denp[t] : 1);
        }
        uint64_t inv_run = gl::inv(run);
#pragma unroll
        for (int t = MAXC - 1; t >= 0; --t) {
            const int ck = ck0 + t;
            if (ck < a.n_chunks) {
                const uint64_t q = denp[t] ? gl::mul(nump[t], gl::mul(inv_run, pre[t])) : 0;
                quot[((size_t)ch * a.n_chunks + ck) * n + i] = gl::canon(q);
                total = gl::mul(total, q);
            }
            inv_run = gl::mul(inv_run, denp[t] ? denp[t] : 1);
        }
    }
    a.tot[(pk * a.nch + ch) * n + i] = gl::canon(total);
}

// multiplicative scan helpers: block-wide inclusive product scan of one value per thread (Hillis-Steele in LDS)
__device__ __forceinline__ uint64_t block_scan_mul(uint64_t v, uint64_t* lds) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < THREADS; off <<= 1) {
        uint64_t other = t >= off ? lds[t - off] : 1;
        __syncthreads();
        v = gl::mul(v, other);
        lds[t] = v;
        __syncthreads();
    }
    return v;
}

constexpr int SCAN_PER = 4;  // rows per thread: one workgroup scans THREADS * SCAN_PER rows

// phase A: block products of tot[ch][*]
__global__ __launch_bounds__(THREADS) void pp_block_products_kernel(const uint64_t* tot, size_t n, uint64_t* block_prod) {
    __shared__ uint64_t lds[THREADS];
    const int ch = blockIdx.y;
    const size_t base = ((size_t)blockIdx.x * THREADS + threadIdx.x) * SCAN_PER;
    uint64_t v = 1;
    for (int k = 0; k < SCAN_PER; ++k)
        if (base + k < n) v = gl::mul(v, tot[(size_t)ch * n + base + k]);
    v = block_scan_mul(v, lds);
    if (threadIdx.x == THREADS - 1) block_prod[(size_t)ch * gridDim.x + blockIdx.x] = gl::canon(v);
}

// phase B: exclusive scan of the block products (one workgroup per challenge; n_blocks is small)
__global__ __launch_bounds__(THREADS) void pp_scan_blocks_kernel(uint64_t* block_prod, int n_blocks) {
    __shared__ uint64_t lds[THREADS];
    __shared__ uint64_t carry;
    uint64_t* bp = block_prod + (size_t)blockIdx.x * n_blocks;
    if (threadIdx.x == 0) carry = 1;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += THREADS) {
        int idx = base + threadIdx.x;
        uint64_t v = idx < n_blocks ? bp[idx] : 1;
        uint64_t incl = block_scan_mul(v, lds);
        uint64_t c = carry;
        uint64_t excl = threadIdx.x == 0 ? c : gl::mul(c, lds[threadIdx.x - 1]);
        __syncthreads();
        if (idx < n_blocks) bp[idx] = gl::canon(excl);
        if (threadIdx.x == THREADS - 1) carry = gl::mul(c, incl);
        __syncthreads();
    }
}

// phase C: Z(x_i) = exclusive prefix of tot; pp_k(x_i) = Z(x_i) * prod_{m <= k} quot_m(x_i)
// out columns: [Z_0 .. Z_{nch-1}, pp(ch 0) x npp, pp(ch 1) x npp, ...]  (plonky2's zs_partial_products order)
__global__ __launch_bounds__(THREADS) void pp_finalize_kernel(const uint64_t* tot, const uint64_t* quot,
                                                              const uint64_t* block_excl, size_t n, int n_chunks, int nch,
                                                              uint64_t* out) {
    __shared__ uint64_t lds[THREADS];
    const int ch = blockIdx.y, npp = n_chunks - 1;
    {   // proof blockIdx.z: its slices of the inputs and of the output
        const size_t pk = blockIdx.z;
        tot += pk * nch * n;
        quot += pk * (size_t)nch * n_chunks * n;
        block_excl += pk * nch * gridDim.x;
        out += pk * (size_t)nch * (1 + npp) * n;
    }
    const size_t base = ((size_t)blockIdx.x * THREADS + threadIdx.x) * SCAN_PER;
    uint64_t loc[SCAN_PER], v = 1;
    for (int k = 0; k < SCAN_PER; ++k) {
        loc[k] = base + k < n ? tot[(size_t)ch * n + base + k] : 1;
        v = gl::mul(v, loc[k]);
    }
    uint64_t incl = block_scan_mul(v, lds);
    (void)incl;
    uint64_t z = block_excl[(size_t)ch * gridDim.x + blockIdx.x];
    if (threadIdx.x > 0) z = gl::mul(z, lds[threadIdx.x - 1]);
    for (int k = 0; k < SCAN_PER; ++k) {
        const size_t i = base + k;
        if (i >= n) break;
        out[(size_t)ch * n + i] = gl::canon(z);
        uint64_t acc = z;
        for (int ck = 0; ck < npp; ++ck) {
            acc = gl::mul(acc, quot[((size_t)ch * n_chunks + ck) * n + i]);
            out[((size_t)nch + (size_t)ch * npp + ck) * n + i] = gl::canon(acc);
        }
        z = gl::mul(z, loc[k]);
    }
}

// ------------------------------------------------------------------ quotient evaluation
// what differs from proof to proof in a batch (device array, one entry per proof)
struct QuotProof {
    uint64_t betas[2], gammas[2];
    uint64_t pi_hash[4];
};

struct QuotArgs {
    const uint64_t* cs;   // constants_sigmas LDE [num_constants + num_routed][N] (shared by all proofs)
    const uint64_t* w;    // wires LDE [proof][num_wires][N]
    const uint64_t* z;    // zs_partial_products LDE [proof][nch * (1 + npp)][N]
    uint64_t* out;        // [proof][nch][N] quotient values at bit-reversed positions
    const uint64_t* apow; // [proof][2][n_terms] alpha_c^t (row 1 = zeros when nch = 1)
    const uint64_t* bk;   // [proof][nch][num_routed] beta_ch * k_j
    const QuotProof* proofs;
    int n_terms;
    int log_n, lde_bits;
    int num_constants, num_selectors, num_routed, num_wires, nch, npp, chunk;
    int num_gates;
    Gate gates[MAX_GATES];
    uint64_t zh[8], zh_inv[8];  // Z_H on the coset: 8 distinct values (index = natural point index mod 8)
    uint64_t n_field;           // n as a field element
    uint64_t interp_dom[MAX_INTERP_POINTS], interp_wts[MAX_INTERP_POINTS];  // CosetInterpolationGate: g^i, barycentric weights
    int K;
    const uint64_t* w_lo;
    const uint64_t* w_hi;
    const uint64_t* ga_poseidon;  // [proof][nch][N] alpha-combination of the PoseidonGate constraints (poseidon_gate_kernel), unfiltered
};

// the view of proof `k` (blockIdx.y of the quotient kernels): its slices and challenges, everything wave-uniform
// The challenges stay in the per-proof device struct and are read through `pr` where they are used (wave-uniform address:
// scalar or broadcast loads): copied into array members here, a run-time index (the challenge loop, the public-input
// constraints of the fused kernel) sent the whole view to scratch — 160 bytes per lane written and re-read, 10.6 GB of HBM
// writes per 2^20-row launch (PMC, profiles/r02e_summary.md).
struct QuotView {
    const uint64_t *w, *z, *apow, *bk, *ga;
    uint64_t* out;
    const QuotProof* pr;
    __device__ __forceinline__ uint64_t beta(int c) const { return pr->betas[c]; }
    __device__ __forceinline__ uint64_t gamma(int c) const { return pr->gammas[c]; }
    __device__ __forceinline__ uint64_t pi_hash(int k) const { return pr->pi_hash[k]; }
};
__device__ __forceinline__ QuotView quot_view(const QuotArgs& a, size_t k) {
    const size_t N = (size_t)1 << a.lde_bits;
    QuotView v;
    v.w = a.w + k * (size_t)a.num_wires * N;
    v.z = a.z + k * (size_t)a.nch * (1 + a.npp) * N;
    v.apow = a.apow + k * 2 * (size_t)a.n_terms;
    v.bk = a.bk + k * (size_t)a.nch * a.num_routed;
    v.ga = a.ga_poseidon ? a.ga_poseidon + k * (size_t)a.nch * N : nullptr;
    v.out = a.out + k * (size_t)a.nch * N;
    v.pr = a.proofs + k;
    return v;
}

// accumulate one vanishing term: res[c] += term * alpha_c^t
#define PLONK_ACC(term, t)                                                                         \
    do {                                                                                           \
        uint64_t _tm = (term);                                                                     \
        _Pragma("unroll") for (int _c = 0; _c < 2; ++_c) if (_c < a.nch)                           \
            res[_c] = gl::add(res[_c], gl::mul(_tm, v.apow[(size_t)_c * a.n_terms + (t)]));        \
    } while (0)

// PoseidonGate::eval_unfiltered_base: 123 constraints streamed into ga[c] += cons * alpha_c^(t0 + k), evaluated with the round
// structure of the production permutation (poseidon.h::permute_folded): the MDS layer
// of full round 3 fused with partial round 4, partial rounds 5..25 as seven linear maps of three rounds.  Every S-box input
// the maps produce is the true one (the pushed constants only move what is added where), so it is compared with its wire
// and replaced by it exactly as in the round-by-round form: 5 054 multiplies in the linear layers instead of 8 640.
__device__ __forceinline__ void poseidon_gate_constraints_fast(const QuotArgs& a, const uint64_t* __restrict__ apow,
                                                               const uint64_t* __restrict__ wp, size_t N, int t0,
                                                               uint64_t (&ga)[2]) {
    auto W = [&](int col) { return wp[(size_t)col * N]; };
    int k = t0;
    auto emit = [&](uint64_t cons) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < a.nch) ga[c] = gl::add(ga[c], gl::mul(cons, apow[(size_t)c * a.n_terms + k]));
        ++k;
    };
    const uint64_t swap = W(24);
    emit(gl::mul(swap, gl::sub(swap, 1)));
    uint64_t st[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t lhs = W(i), rhs = W(i + 4), delta = W(25 + i);
        emit(gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta));
        st[i] = gl::add(lhs, delta);
        st[i + 4] = gl::sub(rhs, delta);
    }
#pragma unroll
    for (int i = 8; i < 12; ++i) st[i] = W(i);
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = gl::add(st[i], poseidon::RC[i]);
    // full rounds 0..3 (the MDS layer of round 3 is part of the fused step below)
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        if (r != 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                uint64_t sin = W(29 + 12 * (r - 1) + i);
                emit(gl::sub(st[i], sin));
                st[i] = sin;
            }
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = poseidon::sbox7(st[i]);
        if (r != 3) poseidon::mds_layer_add(st, &poseidon::RC[(r + 1) * 12]);
    }
    int pr = 0;  // partial round 4 + pr
    auto wire_in = [&](uint64_t computed) {
        const uint64_t sin = W(65 + pr++);
        emit(gl::sub(computed, sin));
        return sin;
    };
    poseidon::mds_partial1_fused(st, wire_in);
#pragma unroll 1
    for (int g = 0; g < 6; ++g)
        poseidon::partial_rounds3<true>(st, poseidon::RC_PARTIAL_S0[3 * g + 1], poseidon::RC_PARTIAL_S0[3 * g + 2],
                                        &poseidon::RC_PARTIAL_S0[3 * g + 3], wire_in);
    poseidon::partial_rounds3<false>(st, poseidon::RC_PARTIAL_S0[19], poseidon::RC_PARTIAL_S0[20], poseidon::RC_AFTER_PARTIAL, wire_in);
    // second full rounds
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            uint64_t sin = W(87 + 12 * r + i);
            emit(gl::sub(st[i], sin));
            st[i] = sin;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = poseidon::sbox7(st[i]);
        poseidon::mds_layer_add(st, &poseidon::RC[(26 + r + 1) * 12]);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) emit(gl::sub(st[i], W(12 + i)));
}

// PoseidonGate constraints of every LDE point, alpha-combined and unfiltered, as a kernel of its own: it has the register
// budget the grouped rounds need (inside quotient_kernel, under its 128-register cap, they spill and are slower than the
// round-by-round form), and quotient_kernel without the permutation code is a much smaller program.
__global__ __launch_bounds__(THREADS) void poseidon_gate_kernel(QuotArgs a, int t_gates, uint64_t* __restrict__ ga_out) {
    const size_t N = (size_t)1 << a.lde_bits;
    const size_t p = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= N) return;
    uint64_t ga[2] = {0, 0};
    const size_t pk = blockIdx.y;  // proof
    poseidon_gate_constraints_fast(a, a.apow + pk * 2 * (size_t)a.n_terms, a.w + pk * (size_t)a.num_wires * N + p, N, t_gates, ga);
    for (int c = 0; c < a.nch; ++c) ga_out[(pk * a.nch + c) * N + p] = ga[c];
}

__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(GLP_QW, GLP_QW))) void quotient_kernel(QuotArgs a) {
    const size_t N = (size_t)1 << a.lde_bits;
    const size_t p = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= N) return;
    const QuotView v = quot_view(a, blockIdx.y);
    const uint32_t nat = __brev((uint32_t)p) >> (32 - a.lde_bits);  // natural index of the point
    const size_t p_next = __brev((uint32_t)((nat + (1u << (a.lde_bits - a.log_n))) & (uint32_t)(N - 1))) >> (32 - a.lde_bits);
    const uint64_t ex = (uint64_t)nat << (a.K - a.lde_bits);
    const uint64_t x = gl::mul(gl::COSET_SHIFT, gl::mul(a.w_lo[ex & 4095], a.w_hi[ex >> 12]));
    const int rem = nat & ((1 << (a.lde_bits - a.log_n)) - 1);
    uint64_t res[2] = {0, 0};

    // ---- L_0(x) (Z(x) - 1) ----
    {
        uint64_t l0 = gl::mul(a.zh[rem], gl::inv(gl::mul(a.n_field, gl::sub(x, 1))));
        for (int c2 = 0; c2 < a.nch; ++c2) PLONK_ACC(gl::mul(l0, gl::sub(v.z[(size_t)c2 * N + p], 1)), c2);
    }
    // ---- partial products: prev * prod(numerators) - next * prod(denominators) per chunk ----
    {
        const int n_chunks = a.npp + 1;
        uint64_t prev[2];
        for (int c2 = 0; c2 < a.nch; ++c2) prev[c2] = v.z[(size_t)c2 * N + p];
        for (int ck = 0; ck < n_chunks; ++ck) {
            uint64_t nump[2] = {1, 1}, denp[2] = {1, 1};
            for (int j = ck * a.chunk; j < (ck + 1) * a.chunk && j < a.num_routed; ++j) {
                const uint64_t wv = v.w[(size_t)j * N + p];
                const uint64_t sg = a.cs[(size_t)(a.num_constants + j) * N + p];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
                    if (c2 < a.nch) {
                        const uint64_t wg = gl::add(wv, v.gamma(c2));  // shared by numerator and denominator
                        uint64_t num = gl::add(wg, gl::mul(v.bk[c2 * a.num_routed + j], x));
                        uint64_t den = gl::add(wg, gl::mul(v.beta(c2), sg));
                        nump[c2] = gl::mul(nump[c2], num);
                        denp[c2] = gl::mul(denp[c2], den);
                    }
            }
            for (int c2 = 0; c2 < a.nch; ++c2) {
                uint64_t next = ck < a.npp ? v.z[(size_t)(a.nch + c2 * a.npp + ck) * N + p] : v.z[(size_t)c2 * N + p_next];
                uint64_t term = gl::sub(gl::mul(prev[c2], nump[c2]), gl::mul(next, denp[c2]));
                PLONK_ACC(term, a.nch + c2 * n_chunks + ck);
                prev[c2] = next;
            }
        }
    }
    // ---- gate constraints, filtered by the selector polynomials ----
    const int t_gates = a.nch + a.nch * (a.npp + 1);
    for (int g = 0; g < a.num_gates; ++g) {
        const Gate gt = a.gates[g];
        const uint64_t s = a.cs[(size_t)gt.selector_index * N + p];
        uint64_t filter = 1;
        for (uint32_t r = gt.group_start; r < gt.group_end; ++r)
            if ((int)r != g) filter = gl::mul(filter, gl::sub((uint64_t)r, s));
        if (a.num_selectors > 1) filter = gl::mul(filter, gl::sub(0xFFFFFFFFull, s));
        uint64_t ga[2] = {0, 0};
        const uint64_t* gc = a.cs + (size_t)a.num_selectors * N + p;  // the gate's own constants
        const uint64_t* wp = v.w + p;
        auto emit = [&](uint64_t cons, int k) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (c < a.nch) ga[c] = gl::add(ga[c], gl::mul(cons, v.apow[(size_t)c * a.n_terms + t_gates + k]));
        };
        if (gt.id == GATE_CONSTANT) {
            for (uint32_t k = 0; k < gt.param; ++k) emit(gl::sub(gc[(size_t)k * N], wp[(size_t)k * N]), k);
        } else if (gt.id == GATE_PUBLIC_INPUT) {
            for (int k = 0; k < 4; ++k) emit(gl::sub(wp[(size_t)k * N], v.pi_hash(k)), k);
        } else if (gt.id == GATE_ARITHMETIC) {
            const uint64_t c0 = gc[0], c1 = gc[N];
            for (uint32_t k = 0; k < gt.param; ++k) {
                uint64_t m0 = wp[(size_t)(4 * k) * N], m1 = wp[(size_t)(4 * k + 1) * N];
                uint64_t ad = wp[(size_t)(4 * k + 2) * N], out = wp[(size_t)(4 * k + 3) * N];
                uint64_t comp = gl::add(gl::mul(gl::mul(m0, m1), c0), gl::mul(ad, c1));
                emit(gl::sub(out, comp), k);
            }
        } else if (gt.id == GATE_POSEIDON) {
            for (int c = 0; c < a.nch; ++c) ga[c] = v.ga[(size_t)c * N + p];  // poseidon_gate_kernel ran first
        } else if (gt.id == GATE_BASE_SUM) {  // BaseSumGate<2>: wire 0 = sum, wires 1.. = limbs
            uint64_t acc = 0;
            for (int k = (int)gt.param - 1; k >= 0; --k) acc = gl::add(gl::add(acc, acc), wp[(size_t)(1 + k) * N]);
            emit(gl::sub(acc, wp[0]), 0);
            for (uint32_t k = 0; k < gt.param; ++k) {
                uint64_t limb = wp[(size_t)(1 + k) * N];
                emit(gl::mul(limb, gl::sub(limb, 1)), 1 + k);
            }
        } else if (gt.id == GATE_ARITHMETIC_EXT || gt.id == GATE_MUL_EXT) {
            const bool arith = gt.id == GATE_ARITHMETIC_EXT;
            const int per = arith ? 8 : 6;
            const uint64_t c0 = gc[0], c1 = gc[N];
            for (uint32_t k = 0; k < gt.param; ++k) {
                const uint64_t* q = wp + (size_t)(per * k) * N;
                gl::ext2 m0 = {q[0], q[N]}, m1 = {q[2 * N], q[3 * N]};
                gl::ext2 pr = gl::ext_mul_base(gl::ext_mul(m0, m1), c0);
                gl::ext2 out = {q[4 * N], q[5 * N]};
                if (arith) {
                    pr = gl::ext_add(pr, gl::ext_mul_base(out, c1));  // q[4], q[5] are the addend here
                    out = gl::ext2{q[6 * N], q[7 * N]};
                }
                emit(gl::sub(out.a, pr.a), 2 * k);
                emit(gl::sub(out.b, pr.b), 2 * k + 1);
            }
        } else if (gt.id == GATE_RANDOM_ACCESS) {  // gates/random_access.rs
            const int bits = gt.param & 0xFF, copies = (gt.param >> 8) & 0xFF, extra = (gt.param >> 16) & 0xFF;
            const int vec = 1 << bits, routed = (2 + vec) * copies + extra;
            int k = 0;
            for (int cp = 0; cp < copies; ++cp) {
                const uint64_t* q = wp + (size_t)((2 + vec) * cp) * N;       // access index, claimed element, list items
                const uint64_t* b = wp + (size_t)(routed + cp * bits) * N;   // selector bits, little-endian
                uint64_t rec = 0;
                for (int i = bits - 1; i >= 0; --i) rec = gl::add(gl::add(rec, rec), b[(size_t)i * N]);
                for (int i = 0; i < bits; ++i) {
                    const uint64_t bi = b[(size_t)i * N];
                    emit(gl::mul(bi, gl::sub(bi, 1)), k++);
                }
                emit(gl::sub(rec, q[0]), k++);
                uint64_t items[MAX_INTERP_POINTS];  // folded pairwise by bit 0, then bit 1, ... (statically unrolled: registers)
#pragma unroll
                for (int j = 0; j < MAX_INTERP_POINTS; ++j) items[j] = j < vec ? q[(size_t)(2 + j) * N] : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < bits) {
                        const uint64_t bi = b[(size_t)i * N];
#pragma unroll
                        for (int j = 0; j < (MAX_INTERP_POINTS / 2) >> i; ++j)
                            items[j] = gl::add(items[2 * j], gl::mul(bi, gl::sub(items[2 * j + 1], items[2 * j])));
                    }
                emit(gl::sub(items[0], q[N]), k++);
            }
            for (int i = 0; i < extra; ++i)
                emit(gl::sub(gc[(size_t)i * N], wp[(size_t)((2 + vec) * copies + i) * N]), k++);
        } else if (gt.id == GATE_REDUCING || gt.id == GATE_REDUCING_EXT) {  // gates/reducing.rs, reducing_extension.rs
            const bool is_ext = gt.id == GATE_REDUCING_EXT;
            const int nc = (int)gt.param, start_accs = 6 + (is_ext ? 2 * nc : nc);
            const gl::ext2 alpha = {wp[2 * N], wp[3 * N]};
            gl::ext2 acc = {wp[4 * N], wp[5 * N]};
            for (int i = 0; i < nc; ++i) {
                gl::ext2 coeff;
                if (is_ext)
                    coeff = gl::ext2{wp[(size_t)(6 + 2 * i) * N], wp[(size_t)(7 + 2 * i) * N]};
                else
                    coeff = gl::ext2{wp[(size_t)(6 + i) * N], 0};
                const int nx = i == nc - 1 ? 0 : start_accs + 2 * i;  // the last accumulator is the output
                const gl::ext2 next = {wp[(size_t)nx * N], wp[(size_t)(nx + 1) * N]};
                const gl::ext2 d = gl::ext_sub(gl::ext_add(gl::ext_mul(acc, alpha), coeff), next);
                emit(d.a, 2 * i);
                emit(d.b, 2 * i + 1);
                acc = next;
            }
        } else if (gt.id == GATE_EXPONENTIATION) {  // gates/exponentiation.rs: square-and-multiply, MSB first
            const int nb = (int)gt.param;
            const uint64_t base = wp[0];
            uint64_t last = 0;
            for (int i = 0; i < nb; ++i) {
                const uint64_t prev = i == 0 ? 1 : gl::mul(last, last);
                const uint64_t bit = wp[(size_t)(1 + nb - 1 - i) * N];
                const uint64_t factor = gl::add(gl::mul(bit, base), gl::sub(1, bit));
                last = wp[(size_t)(2 + nb + i) * N];
                emit(gl::sub(gl::mul(prev, factor), last), i);
            }
            emit(gl::sub(wp[(size_t)(1 + nb) * N], last), nb);
        } else if (gt.id == GATE_POSEIDON_MDS) {  // gates/poseidon_mds.rs: MDS layer on 12 extension elements
            for (int r = 0; r < 12; ++r)
#pragma unroll
                for (int comp = 0; comp < 2; ++comp) {
                    unsigned __int128 acc = 0;  // < 2^64 * 500: reduced once
                    for (int i = 0; i < 12; ++i) {
                        int src = i + r;
                        src = src >= 12 ? src - 12 : src;
                        acc += (unsigned __int128)gl::canon(wp[(size_t)(2 * src + comp) * N]) * poseidon::MDS_CIRC_TAB[i];
                    }
                    if (r == 0) acc += (unsigned __int128)gl::canon(wp[(size_t)comp * N]) * 8u;
                    const uint64_t got = gl::reduce128((uint64_t)acc, (uint64_t)(acc >> 64));
                    emit(gl::sub(wp[(size_t)(24 + 2 * r + comp) * N], got), 2 * r + comp);
                }
        } else if (gt.id == GATE_COSET_INTERPOLATION) {  // gates/coset_interpolation.rs
            const int bits = gt.param & 0xFF, degree = (gt.param >> 8) & 0xFF, npts = 1 << bits;
            const int ni = (npts - 2) / (degree - 1);
            const int sp = 1 + 2 * npts, sv = sp + 2, si = sv + 2;
            const uint64_t shift = wp[0];
            const gl::ext2 point = {wp[(size_t)sp * N], wp[(size_t)(sp + 1) * N]};
            const gl::ext2 shifted = {wp[(size_t)(si + 4 * ni) * N], wp[(size_t)(si + 4 * ni + 1) * N]};
            int k = 0;
            gl::ext2 d = gl::ext_sub(point, gl::ext_mul_base(shifted, shift));
            emit(d.a, k++);
            emit(d.b, k++);
            gl::ext2 eval = {0, 0}, prod = {1, 0};
            auto run = [&](int from, int to) {  // partial_interpolate_ext_algebra over points [from, to)
                for (int j = from; j < to; ++j) {
                    const gl::ext2 term = {gl::sub(shifted.a, a.interp_dom[j]), shifted.b};
                    const gl::ext2 val = gl::ext_mul_base(gl::ext2{wp[(size_t)(1 + 2 * j) * N], wp[(size_t)(2 + 2 * j) * N]},
                                                          a.interp_wts[j]);
                    const gl::ext2 ne = gl::ext_add(gl::ext_mul(eval, term), gl::ext_mul(val, prod));
                    prod = gl::ext_mul(prod, term);
                    eval = ne;
                }
            };
            run(0, degree);
            for (int i = 0; i < ni; ++i) {
                const gl::ext2 ie = {wp[(size_t)(si + 2 * i) * N], wp[(size_t)(si + 2 * i + 1) * N]};
                const gl::ext2 ip = {wp[(size_t)(si + 2 * (ni + i)) * N], wp[(size_t)(si + 2 * (ni + i) + 1) * N]};
                d = gl::ext_sub(ie, eval);
                emit(d.a, k++);
                emit(d.b, k++);
                d = gl::ext_sub(ip, prod);
                emit(d.a, k++);
                emit(d.b, k++);
                const int from = 1 + (degree - 1) * (i + 1);
                const int to = from + degree - 1 < npts ? from + degree - 1 : npts;
                eval = ie;
                prod = ip;
                run(from, to);
            }
            d = gl::ext_sub(gl::ext2{wp[(size_t)sv * N], wp[(size_t)(sv + 1) * N]}, eval);
            emit(d.a, k++);
            emit(d.b, k++);
        }
        for (int c = 0; c < a.nch; ++c) res[c] = gl::add(res[c], gl::mul(filter, ga[c]));
    }
    for (int c = 0; c < a.nch; ++c) v.out[(size_t)c * N + p] = gl::canon(gl::mul(res[c], a.zh_inv[rem]));
}

// ------------------------------------------------------------------ witness helper (PoseidonGenerator, swap = 0)
// in[r][12] -> rows[r][135] (row-major, one PoseidonGate row each)
__global__ __launch_bounds__(THREADS) void poseidon_gate_witness_kernel(const uint64_t* __restrict__ in, size_t n_rows,
                                                                        uint64_t* __restrict__ rows, size_t row_stride,
                                                                        size_t col_stride) {
    const size_t r = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rows) return;
    uint64_t* o = rows + r * row_stride;
    auto put = [&](int col, uint64_t v) { o[(size_t)col * col_stride] = gl::canon(v); };
    uint64_t st[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        st[i] = in[r * 12 + i];
        put(i, st[i]);
    }
    for (int c = 24; c < 29; ++c) put(c, 0);  // swap, deltas
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = gl::add(st[i], poseidon::RC[i]);
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
        if (rr != 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) put(29 + 12 * (rr - 1) + i, st[i]);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = poseidon::sbox7(st[i]);
        poseidon::mds_layer_add(st, &poseidon::RC[(rr + 1) * 12]);
    }
#pragma unroll 1
    for (int rr = 0; rr < 22; ++rr) {
        put(65 + rr, st[0]);
        st[0] = poseidon::sbox7(st[0]);
        poseidon::mds_layer_add(st, &poseidon::RC[(4 + rr + 1) * 12]);
    }
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int i = 0; i < 12; ++i) put(87 + 12 * rr + i, st[i]);
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = poseidon::sbox7(st[i]);
        poseidon::mds_layer_add(st, &poseidon::RC[(26 + rr + 1) * 12]);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) put(12 + i, st[i]);
}

}  // namespace plonk
