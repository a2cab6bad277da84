// ntt_kernels.h — gfx950 kernels for the batched Goldilocks NTT / LDE (see ntt_index.h for the algebra).
//
// Layout in HBM: one polynomial per column, columns contiguous (`col_stride` elements apart) — the layout of
// plonky2's Vec<PolynomialValues<F>> / Vec<PolynomialCoeffs<F>> that PolynomialBatch::from_values receives
// (reached from reference src/rollup/circuits/mod.rs:1247).  grid = (tiles per column, columns); each 256-thread
// workgroup stages one 2^LOGE-element tile through LDS: coalesced 128-byte runs in, R butterfly layers as register-blocked
// radix-8 rounds (3 layers per LDS round trip), one twiddle multiply, coalesced runs out.  HBM traffic per pass: 16 B per element (8 read + 8 written).
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_index.h"
#include "ntt_canon.h"

namespace ntt {

constexpr int THREADS = 256;

struct PassArgs {
    const uint64_t* src;     // column 0 of the input
    uint64_t* dst;           // column 0 of the output (== src for in-place passes)
    size_t src_col_stride;   // elements between columns
    size_t dst_col_stride;
    int log_n;               // transform size per column
    int s_lo, r_bits;        // bits [s_lo, s_lo + r_bits) are transformed by this pass
    int inverse;             // 1: use inverse roots
    int K;                   // master root table: w = w_{2^K}
    const uint64_t* w_lo;    // [4096]      w^j
    const uint64_t* w_hi;    // [2^(K-12)]  w^(4096 j)
    // coset / LDE support (first DIF pass only, s_lo + r_bits == log_n): the tile is loaded once and transformed
    // n_cosets times; coset t scales element j = (r << s_lo) | B by scale_hi[t][r] * scale_lo[t][B] and writes its
    // result to dst + dst_coset_stride * coset_slot[t]
    int n_cosets;            // 0: plain pass
    const uint64_t* scale_lo;  // [n_cosets][2^s_lo]
    const uint64_t* scale_hi;  // [n_cosets][2^r_bits]
    size_t dst_coset_stride;
    uint64_t out_scale;      // every output multiplied by this (1/n on the last inverse pass); 0 = none
    // precomputed tables shared by all columns (L2-resident: every XCD sees 1/8 of the tiles), either may be null:
    const uint64_t* tw_pass;     // [2^(s_lo + r_bits)] inter-pass twiddle of position p at index p mod 2^(s_lo + r_bits)
    const uint64_t* scale_full;  // [n_cosets][2^log_n] coset scale sigma_t^j (replaces scale_hi * scale_lo)
    int canon_out;           // canonicalise on store (last pass)
    int src_canonical;       // the source holds values < p (written by a pass that computes on canonical values): ntt_pass_ab_kernel
                             // in its canonical form skips the canonicalisation on load
};

// raw buffer access (gfx950): descriptor = base pointer + byte range (reads beyond it return 0, writes are dropped); address =
// base + lane offset (VGPR, bytes) + scalar offset (SGPR, bytes)
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned int buf_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ BufRsrc buf_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint64_t buf_load(BufRsrc r, uint32_t lane_off, uint32_t uniform_off) {
    const buf_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_off, (int)uniform_off, 0);
    return ((uint64_t)v.y << 32) | v.x;
}
__device__ __forceinline__ void buf_store(uint64_t x, BufRsrc r, uint32_t lane_off, uint32_t uniform_off) {
    buf_v2u v;
    v.x = (uint32_t)x;
    v.y = (uint32_t)(x >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)lane_off, (int)uniform_off, 0);
}

__device__ __forceinline__ uint64_t root_pow(const PassArgs& a, uint64_t e) {
    uint64_t lo = a.w_lo[e & ((1u << TW_LO_BITS) - 1)];
    uint64_t hi = a.w_hi[e >> TW_LO_BITS];
    return gl::mul(lo, hi);
}

// inter-pass twiddle of local element e at position p: from the per-pass table when there is one
__device__ __forceinline__ uint64_t pass_twiddle(const PassArgs& a, const TileGeom& g, size_t p, uint32_t e) {
    const size_t m_mask = ((size_t)1 << (g.s_lo + g.r_bits)) - 1;
    if (a.tw_pass) return a.tw_pass[p & m_mask];
    const size_t b = p & (((size_t)1 << g.s_lo) - 1);
    return root_pow(a, pass_twiddle_exp(g, tile_r(g, e), b, a.K, a.inverse));
}

// table fills (once per context and geometry)
__global__ __launch_bounds__(THREADS) void fill_pass_twiddles_kernel(PassArgs a, uint64_t* tw) {
    const TileGeom g = make_geom(11, a.s_lo, a.r_bits);  // only s_lo / r_bits matter here
    const size_t idx = (size_t)blockIdx.x * THREADS + threadIdx.x;
    if (idx >> (a.s_lo + a.r_bits)) return;
    const uint32_t r = (uint32_t)(idx >> a.s_lo);
    const size_t b = idx & (((size_t)1 << a.s_lo) - 1);
    tw[idx] = gl::canon(root_pow(a, pass_twiddle_exp(g, r, b, a.K, a.inverse)));
}
__global__ __launch_bounds__(THREADS) void fill_scale_full_kernel(const uint64_t* scale_lo, const uint64_t* scale_hi, int s_lo,
                                                                  int r_bits, int n_cosets, uint64_t* out) {
    const size_t idx = (size_t)blockIdx.x * THREADS + threadIdx.x;
    const int log_n = s_lo + r_bits;
    if ((idx >> log_n) >= (size_t)n_cosets) return;
    const size_t t = idx >> log_n, j = idx & (((size_t)1 << log_n) - 1);
    out[idx] = gl::canon(gl::mul(scale_hi[(t << r_bits) + (j >> s_lo)], scale_lo[(t << s_lo) + (j & (((size_t)1 << s_lo) - 1))]));
}

// one register-blocked round over the whole tile: 2^Q elements per group, groups dealt round-robin to the threads
template <int LOGE, int Q, bool DIT, bool T0>
__device__ __forceinline__ void tile_round_t(uint64_t* tile, const uint64_t* ltw, const TileGeom& g, int t, uint32_t tid) {
    constexpr uint32_t N_GROUPS = (1u << LOGE) >> Q;
    const int stride = 1 << (g.rshift + t);
#pragma unroll
    for (uint32_t grp = tid; grp < N_GROUPS; grp += THREADS) {
        const uint32_t base = round_base(g, grp, t, Q);
        uint64_t x[1 << Q];
#pragma unroll
        for (int j = 0; j < (1 << Q); ++j) x[j] = tile[base + j * stride];
        radix_round<Q, DIT, T0>(x, ltw, g, base, t);
#pragma unroll
        for (int j = 0; j < (1 << Q); ++j) tile[base + j * stride] = x[j];
    }
}
template <int LOGE, int Q, bool DIT>
__device__ __forceinline__ void tile_round(uint64_t* tile, const uint64_t* ltw, const TileGeom& g, int t, uint32_t tid) {
    if (t == 0)
        tile_round_t<LOGE, Q, DIT, true>(tile, ltw, g, t, tid);
    else
        tile_round_t<LOGE, Q, DIT, false>(tile, ltw, g, t, tid);
}

// DIT = false: decimation in frequency (natural in -> bit-reversed out); DIT = true: the mirror image.
template <int LOGE, bool DIT>
__global__ __launch_bounds__(THREADS) void ntt_pass_kernel(PassArgs a) {
    constexpr uint32_t E = 1u << LOGE;
    constexpr int PER_THREAD = E / THREADS;
    __shared__ uint64_t tile[E];
    __shared__ uint64_t ltw[E / 2];  // local twiddles w_{2^R}^i, i < 2^(R-1)

    const TileGeom g = make_geom(LOGE, a.s_lo, a.r_bits);
    const uint32_t tid = threadIdx.x;
    const size_t tile_id = blockIdx.x;
    const size_t col = blockIdx.y;
    const uint64_t* src = a.src + col * a.src_col_stride;
    uint64_t* dst = a.dst + col * a.dst_col_stride;

    // local twiddle table
    for (uint32_t i = tid; i < (1u << (g.r_bits - 1)); i += THREADS) {
        uint64_t e = (uint64_t)i << (a.K - g.r_bits);
        if (a.inverse) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        ltw[i] = root_pow(a, e);
    }

    uint64_t raw[PER_THREAD];
#pragma unroll
    for (int q = 0; q < PER_THREAD; ++q) raw[q] = src[tile_pos(g, tile_id, tid + q * THREADS)];

    // the cosets of the LDE pass: all of them one after the other in this workgroup, or — a launch too small to fill the chip
    // (a 2^11-row circuit: 135 workgroups at 8 x 4 LDS rounds each, 87 us of pure latency per commitment) — split over gridDim.z
    const int n_rep = a.n_cosets > 0 ? a.n_cosets : 1;
    const int per_z = n_rep / (int)gridDim.z, t_first = (int)blockIdx.z * per_z;
    for (int t = t_first; t < t_first + per_z; ++t) {
        if (t > t_first) __syncthreads();  // previous coset's stores have read the tile
#pragma unroll 2
        for (int q = 0; q < PER_THREAD; ++q) {
            uint32_t e = tid + q * THREADS;
            uint64_t v = raw[q];
            size_t p = tile_pos(g, tile_id, e);
            if (a.n_cosets > 0) {
                if (a.scale_full) {
                    v = gl::mul(v, a.scale_full[((size_t)t << a.log_n) + p]);
                } else {
                    uint32_t r = tile_r(g, e);
                    size_t b = p & (((size_t)1 << g.s_lo) - 1);
                    uint64_t s = gl::mul(a.scale_hi[((size_t)t << g.r_bits) + r], a.scale_lo[((size_t)t << g.s_lo) + b]);
                    v = gl::mul(v, s);
                }
            }
            if (DIT && g.s_lo > 0) v = gl::mul(v, pass_twiddle(a, g, p, e));
            tile[e] = v;
        }
        __syncthreads();

        // R butterfly stages in rounds of up to 3 stages held in registers (one LDS round trip per round)
        {
            int done = 0;
            while (done < g.r_bits) {
                const int left = g.r_bits - done;
                const int q = left >= 3 ? 3 : left;
                const int tt = DIT ? done : g.r_bits - done - q;  // lowest local bit of this round
                if (q == 3)
                    tile_round<LOGE, 3, DIT>(tile, ltw, g, tt, tid);
                else if (q == 2)
                    tile_round<LOGE, 2, DIT>(tile, ltw, g, tt, tid);
                else
                    tile_round<LOGE, 1, DIT>(tile, ltw, g, tt, tid);
                __syncthreads();
                done += q;
            }
        }

        uint64_t* out = dst + (a.n_cosets > 0 ? a.dst_coset_stride * (size_t)rev_bits((uint32_t)t, 31 - __clz(a.n_cosets)) : 0);
#pragma unroll 2
        for (int q = 0; q < PER_THREAD; ++q) {
            uint32_t e = tid + q * THREADS;
            uint64_t v = tile[e];
            size_t p = tile_pos(g, tile_id, e);
            if (!DIT && g.s_lo > 0) v = gl::mul(v, pass_twiddle(a, g, p, e));
            if (a.out_scale) v = gl::mul(v, a.out_scale);
            if (a.canon_out) v = gl::canon(v);
            out[p] = v;
        }
    }
}

// Specialised DIF pass for the geometry every large transform uses on its upper bits (R = 7, s_lo >= 4; see ntt_index.h
// "pass A"): global -> registers -> radix-16 (shift twiddles) -> w_128 twiddle -> LDS -> radix-8 (shift twiddles) ->
// inter-pass twiddle -> global.  One LDS
// round trip and one barrier per tile (the generic kernel: four and four), geometry folded at compile time.
// CANON (coset form only): the arithmetic on canonical values (ntt_canon.h); everything it stores is < p
template <bool INVERSE, bool COSETS, bool CANON = false>
__global__ __launch_bounds__(A_THREADS, COSETS ? A_WAVES_PER_SIMD : 6) void ntt_pass_a_kernel(PassArgs a) {
    // the plain pass exchanges the tile in two halves of 8 rows (9 KB of LDS per workgroup: six waves per SIMD instead of
    // four); the coset pass is register-bound at four waves anyway and keeps the whole tile (one barrier less per coset)
    constexpr int LDS_ROWS = COSETS ? 16 : 8;
    constexpr int G2_UNROLL = COSETS ? 1 : 2;  // the plain form indexes its registers by the half: unrolled
    __shared__ uint64_t tile[LDS_ROWS * A_ROW];
    __shared__ uint64_t ltw[128];  // w_128^i
    const uint32_t t = threadIdx.x, b_lo = t & 15, t_hi = t >> 4;
    const int s_lo = a.s_lo;
    // The coset pass walks the COLUMNS fastest (grid = columns x tiles): the eight 16 KB slices of the coset-scale table that a
    // tile position needs are the same for every column, and workgroups that run at the same time on an XCD then share them in
    // its L2.  With tiles fastest each column swept the whole 64 MB table (12.2 GB fetched per 135-column launch for 1.13 GB
    // of input, profiles/r02a_summary.md).
    const size_t col = COSETS ? blockIdx.x : blockIdx.y;
    const size_t tile_id = COSETS ? blockIdx.y : blockIdx.x;
    const uint64_t* src = a.src + col * a.src_col_stride;
    uint64_t* dst = a.dst + col * a.dst_col_stride;
    {
        uint64_t e = (uint64_t)t << (a.K - A_R_BITS);
        if (INVERSE) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        ltw[t] = root_pow(a, e);
    }
    const size_t base = a_tile_base(tile_id, s_lo);
    const size_t m_mask = ((size_t)1 << (s_lo + A_R_BITS)) - 1;
    // Addressing: every global access is a buffer access — (descriptor of the column / table) + one 32-bit lane offset + a
    // wave-uniform scalar offset — so the sixteen addresses of a load group cost ONE VGPR and no address arithmetic (with flat
    // pointers: sixteen register pairs and a 64-bit add each, in a kernel that is short of registers).
    const uint32_t col_bytes = (uint32_t)8 << a.log_n;
    const uint32_t off_in = ((t_hi << s_lo) + b_lo) * 8;          // bytes; r = (q << 3) | t_hi at + q * in_step
    const uint32_t off_out = ((t_hi << (s_lo + 3)) + b_lo) * 8;   // bytes; r = (hi << 3) | j at + (8 g2 << (s_lo + 3)) + (j << s_lo)
    const uint32_t base_b = (uint32_t)base * 8, in_step = (uint32_t)64 << s_lo;
    const BufRsrc r_src = buf_rsrc(src, col_bytes);

    uint64_t raw[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) raw[q] = buf_load(r_src, off_in, base_b + q * in_step);

    // cosets of this workgroup: all of them, or the share of blockIdx.z when the host split a small launch (a 2^12-row
    // transform is 135 x 2 workgroups, each walking 8 cosets one after the other: 96 us of latency on a mostly idle chip)
    const int per_z = COSETS ? a.n_cosets / (int)gridDim.z : 1;
    const int c_lo = COSETS ? (int)blockIdx.z * per_z : 0, c_hi = c_lo + per_z;
    // Loads are issued TOGETHER and pinned there (sched_barrier): left to itself the compiler, short of registers in the coset
    // form, sinks every scale / twiddle load to just before the multiplication that uses it and waits for it — 32 serialised L2
    // round trips per coset and wave; 57 % of the wave-cycles of the LDE pass were spent in s_waitcnt (profiles/r03k_lde_pmc.md).
    // Now the sixteen scales of a coset are in flight at once, and the eight inter-pass twiddles of a radix-8 half are fetched
    // before its butterflies.  (Fetching the next coset's scales under the radix-8 halves as well — 4, 8, 12 or 16 of them, at three
    // or four waves per SIMD — measured the same 13.6-13.7 ms for the transforms of a 135-column commitment as this form, 15.4 before.)
    const bool have_tw = a.tw_pass != nullptr;
    const BufRsrc r_tw = buf_rsrc(have_tw ? a.tw_pass : a.w_lo, have_tw ? (uint32_t)((m_mask + 1) * 8) : 8u);
    const uint32_t tw_base_b = (uint32_t)(base & m_mask) * 8;  // (p & m_mask) = (base & m_mask) + (p - base): the fields do not overlap
    __syncthreads();  // ltw
#pragma unroll 1
    for (int c = c_lo; c < c_hi; ++c) {
        uint64_t x[16];
        if (COSETS) {
            const BufRsrc r_sc = buf_rsrc(a.scale_full + ((size_t)c << a.log_n), col_bytes);
            uint64_t sc[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) sc[q] = buf_load(r_sc, off_in, base_b + q * in_step);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
#if defined(__HIP_DEVICE_COMPILE__)
                if (CANON)
                    x[q] = glc::mulc(raw[q], sc[q]);
                else
#endif
                    x[q] = gl::mul(raw[q], sc[q]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = raw[q];
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::radix16_dif_ct<INVERSE>(x, ltw, t_hi);
        else
#endif
            radix16_dif_ct<INVERSE>(x, ltw, t_hi);
        if (COSETS) {
            if (c > c_lo) __syncthreads();  // the previous coset's radix-8 reads are done
#pragma unroll
            for (int q = 0; q < 16; ++q) tile[q * A_ROW + t] = x[q];  // = a_lds_index((q << 3) | t_hi, b_lo)
            __syncthreads();
        }
        const BufRsrc r_out = buf_rsrc(dst + (COSETS ? a.dst_coset_stride * (size_t)rev_bits((uint32_t)c, 31 - __clz(a.n_cosets)) : 0), col_bytes);
#pragma unroll G2_UNROLL
        for (int g2 = 0; g2 < 2; ++g2) {
            const uint32_t hi = t_hi + 8 * g2;  // r = (hi << 3) | j
            const uint32_t u_out = (uint32_t)(64 * g2) << (s_lo + 3);  // bytes, uniform part of p_out - base
            uint64_t tw[8];
            if (have_tw) {
#pragma unroll
                for (int j = 0; j < 8; ++j) tw[j] = buf_load(r_tw, off_out, tw_base_b + u_out + ((uint32_t)(8 * j) << s_lo));
            }
            if (!COSETS) {  // rows 8 g2 .. 8 g2 + 7 of the tile go through the eight LDS rows
                if (g2) __syncthreads();
#pragma unroll
                for (int q = 0; q < 8; ++q) tile[q * A_ROW + t] = x[8 * g2 + q];
                __syncthreads();
            }
            const uint32_t row = COSETS ? hi : t_hi;
            uint64_t y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = tile[row * A_ROW + (j << 4) + b_lo];
            __builtin_amdgcn_sched_barrier(0);
#if defined(__HIP_DEVICE_COMPILE__)
            if (CANON)
                nttc::radix8_dif_const<INVERSE>(y);
            else
#endif
                radix8_dif_const<INVERSE>(y);
            if (!have_tw) {  // geometries whose table would not stay in L2 (get_pass_twiddles)
                const TileGeom g = make_geom(11, s_lo, A_R_BITS);
                const size_t p_out = base + ((size_t)hi << (s_lo + 3)) + b_lo;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    tw[j] = root_pow(a, pass_twiddle_exp(g, (hi << 3) | j, (p_out + ((size_t)j << s_lo)) & (((size_t)1 << s_lo) - 1), a.K, INVERSE));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint64_t o;
#if defined(__HIP_DEVICE_COMPILE__)
                if (CANON)
                    o = glc::mulc(y[j], tw[j]);
                else
#endif
                    o = gl::mul(y[j], tw[j]);
                buf_store(o, r_out, off_out, base_b + u_out + ((uint32_t)(8 * j) << s_lo));
            }
        }
    }
}


// Specialised DIF pass for the lowest six bits (s_lo = 0, R = 6; ntt_index.h "pass B"), in place or out of place.
template <bool INVERSE>
__global__ __launch_bounds__(A_THREADS, 4) void ntt_pass_b_kernel(PassArgs a) {
    __shared__ uint64_t tile[256 * B_PITCH];
    __shared__ uint64_t ltw[B_W64_ENTRIES];
    const uint32_t t = threadIdx.x, r3 = t & 7, blk0 = t >> 3;
    const size_t col = blockIdx.y;
    const uint64_t* src = a.src + col * a.src_col_stride + ((size_t)blockIdx.x << 11);
    uint64_t* dst = a.dst + col * a.dst_col_stride + ((size_t)blockIdx.x << 11);
    if (t < (uint32_t)B_W64_ENTRIES) {
        uint64_t e = (uint64_t)t << (a.K - B_R_BITS);
        if (INVERSE) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        ltw[t] = root_pow(a, e);
    }
    uint64_t x[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[g2][j] = src[((blk0 + 16 * g2) << 6) + (j << 3) + r3];
    __syncthreads();
    uint64_t tw[7];
    b_round_twiddles(ltw, r3, tw);
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        b_round<INVERSE>(x[g2], tw);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[((blk0 + 16 * g2) * 8 + j) * B_PITCH + r3] = x[g2][j];  // b_lds_index(blk, (j << 3) | r3)
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        uint64_t* row = tile + (t + 128 * g2) * B_PITCH;  // row = blk * 8 + (r >> 3): eight consecutive values of r
        uint64_t y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = row[j];
        radix8_dif_const<INVERSE>(y);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t v = y[j];
            if (a.out_scale) v = gl::mul(v, a.out_scale);
            if (a.canon_out) v = gl::canon(v);
            row[j] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[((blk0 + 16 * g2) << 6) + (j << 3) + r3] = tile[((blk0 + 16 * g2) * 8 + j) * B_PITCH + r3];
}

// Passes A (bits 6..12) and B (bits 0..5) of a transform as ONE sweep (round 3): a workgroup of 512 threads owns 8192 CONTIGUOUS
// elements (64 KB, the best access pattern there is), runs pass A's radix-16 / w_128 twiddle / radix-8 / inter-pass twiddle on them,
// regroups through LDS instead of HBM, and runs pass B's two radix-8 rounds.  Same per-thread routines, same arithmetic, one
// 16 n C-byte sweep less per transform (the two passes it replaces: 4.4 + 3.5 ms for the 135 LDE columns of a 2^20-row commitment).
// LDS: one buffer of 9216 words used in three layouts — [r][64] for the exchange inside pass A, [r'][72] for the regrouping
// (the eight 64-point blocks a wave reads are 72 words apart: two-way conflicts at most), pass B's rows of 8 padded to 9.
constexpr int AB_THREADS = 512;
constexpr int AB_LOG_TILE = 13;
constexpr int AB_PITCH = 72;
constexpr size_t AB_LDS_BYTES = (128 * AB_PITCH + 128 + B_W64_ENTRIES) * sizeof(uint64_t);  // 75 264: dynamic (above the 64 KB static limit)
// CANON: the arithmetic on canonical values (ntt_canon.h); the tile is canonicalised on load unless PassArgs::src_canonical
template <bool INVERSE, bool CANON>
__global__ __launch_bounds__(AB_THREADS, 4) void ntt_pass_ab_kernel(PassArgs a) {
    extern __shared__ uint64_t ab_lds[];
    uint64_t* tile = ab_lds;                    // 128 * AB_PITCH = 9216 words
    uint64_t* w128 = ab_lds + 128 * AB_PITCH;   // w_128^i
    uint64_t* w64 = w128 + 128;                 // w_64^i, i < 32
    const uint32_t t = threadIdx.x;
    const size_t col = blockIdx.y;
    const uint64_t* src = a.src + col * a.src_col_stride + ((size_t)blockIdx.x << AB_LOG_TILE);
    uint64_t* dst = a.dst + col * a.dst_col_stride + ((size_t)blockIdx.x << AB_LOG_TILE);
    if (t < 128) {
        uint64_t e = (uint64_t)t << (a.K - A_R_BITS);
        if (INVERSE) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        w128[t] = root_pow(a, e);
    } else if (t < 128 + (uint32_t)B_W64_ENTRIES) {
        uint64_t e = (uint64_t)(t - 128) << (a.K - B_R_BITS);
        if (INVERSE) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        w64[t - 128] = root_pow(a, e);
    }
    // ---- pass A, s_lo = 6: p = (r : 7 | b : 6);  thread (t_hi : 3 | b : 6) holds r = (q << 3) | t_hi ----
    const uint32_t b = t & 63, t_hi = t >> 6;
    uint64_t x[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = src[((((uint32_t)q << 3) | t_hi) << 6) | b];  // 512-byte runs
    // the eight inter-pass twiddles of each radix-8 half are fetched now, ahead of everything (table of 2^13 words, L2)
    uint64_t twa[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r = ((t_hi + 8 * g2) << 3) | j;
            twa[g2][j] = a.tw_pass[(r << 6) | b];
        }
    if (CANON && !a.src_canonical) {
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = gl::canon(x[q]);
    }
    __syncthreads();  // w128, w64
#if defined(__HIP_DEVICE_COMPILE__)
    if (CANON)
        nttc::radix16_dif_ct<INVERSE>(x, w128, t_hi);
    else
#endif
        radix16_dif_ct<INVERSE>(x, w128, t_hi);
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[((((uint32_t)q << 3) | t_hi) << 6) | b] = x[q];
    __syncthreads();
    uint64_t y[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) y[g2][j] = tile[((((t_hi + 8 * g2) << 3) | j) << 6) | b];
    __syncthreads();  // everybody has read layout 1
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::radix8_dif_const<INVERSE>(y[g2]);
        else
#endif
            radix8_dif_const<INVERSE>(y[g2]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r = ((t_hi + 8 * g2) << 3) | j;
            uint64_t o;
#if defined(__HIP_DEVICE_COMPILE__)
            if (CANON)
                o = glc::mulc(y[g2][j], twa[g2][j]);
            else
#endif
                o = gl::mul(y[g2][j], twa[g2][j]);
            tile[r * AB_PITCH + b] = o;  // position (r << 6) | b of the block, regrouping layout
        }
    }
    __syncthreads();
    // ---- pass B on the 128 contiguous 64-point blocks: thread (blk0 : 6 | r3 : 3) holds local indices (j << 3) | r3 of blocks blk0, blk0 + 64 ----
    const uint32_t r3 = t & 7, blk0 = t >> 3;
    uint64_t z[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) z[g2][j] = tile[(blk0 + 64 * g2) * AB_PITCH + (j << 3) + r3];
    uint64_t tw[7];
    b_round_twiddles(w64, r3, tw);
    __syncthreads();  // everybody has read the regrouping layout
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::b_round<INVERSE>(z[g2], tw);
        else
#endif
            b_round<INVERSE>(z[g2], tw);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[((blk0 + 64 * g2) * 8 + j) * B_PITCH + r3] = z[g2][j];  // b_lds_index(blk, (j << 3) | r3)
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        uint64_t* row = tile + (t + 512 * g2) * B_PITCH;  // row = blk * 8 + (r >> 3): eight consecutive local indices
        uint64_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[j];
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::radix8_dif_const<INVERSE>(v);
        else
#endif
            radix8_dif_const<INVERSE>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t o = v[j];
            if (a.out_scale) o = gl::mul(o, a.out_scale);
            if (a.canon_out && (!CANON || a.out_scale)) o = gl::canon(o);  // CANON: already < p
            row[j] = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[((blk0 + 64 * g2) << 6) + (j << 3) + r3] = tile[((blk0 + 64 * g2) * 8 + j) * B_PITCH + r3];
}

// The same sweep with a workgroup that walks over tiles (round 6; opt-in, GLP_NTT_AB_PERSIST — measured slower, see its launch in
// glprover.hip): two workgroups of 75 KB fit a CU, so a tile's loads and stores have only the other workgroup's arithmetic to hide
// behind; here the next tile's sixteen words per thread are requested while pass B of the current tile runs (32 more registers:
// the 75 KB of LDS allow four waves per SIMD anyway).  Grid = the workgroups the chip holds; tile ids are flattened over the
// columns of the launch.
template <bool INVERSE, bool CANON>
__global__ __launch_bounds__(AB_THREADS, 4) void ntt_pass_ab_persist_kernel(PassArgs a, uint32_t log_tiles_per_col, uint32_t n_tiles) {
    extern __shared__ uint64_t ab_lds[];
    uint64_t* tile = ab_lds;                    // 128 * AB_PITCH = 9216 words
    uint64_t* w128 = ab_lds + 128 * AB_PITCH;   // w_128^i
    uint64_t* w64 = w128 + 128;                 // w_64^i, i < 32
    const uint32_t t = threadIdx.x;
    if (t < 128) {
        uint64_t e = (uint64_t)t << (a.K - A_R_BITS);
        if (INVERSE) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        w128[t] = root_pow(a, e);
    } else if (t < 128 + (uint32_t)B_W64_ENTRIES) {
        uint64_t e = (uint64_t)(t - 128) << (a.K - B_R_BITS);
        if (INVERSE) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        w64[t - 128] = root_pow(a, e);
    }
    // ---- pass A, s_lo = 6: p = (r : 7 | b : 6);  thread (t_hi : 3 | b : 6) holds r = (q << 3) | t_hi ----
    const uint32_t tile_mask = (1u << log_tiles_per_col) - 1;
    uint32_t id = blockIdx.x;  // flattened tile: column id >> log_tiles_per_col, tile id & tile_mask
    uint64_t x[16];
    {
        const uint32_t b = t & 63, t_hi = t >> 6;
        const uint64_t* src = a.src + (size_t)(id >> log_tiles_per_col) * a.src_col_stride + ((size_t)(id & tile_mask) << AB_LOG_TILE);
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = src[((((uint32_t)q << 3) | t_hi) << 6) | b];  // 512-byte runs
    }
  for (;;) {
    // the thread index behind an opaque copy, once per tile: every address of the tile body is a function of it, and taken out of
    // the loop as invariants they would be ~80 registers (the one-tile kernel computes them where it uses them)
    uint32_t tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    const uint32_t t = tl, b = t & 63, t_hi = t >> 6;
    uint64_t* dst = a.dst + (size_t)(id >> log_tiles_per_col) * a.dst_col_stride + ((size_t)(id & tile_mask) << AB_LOG_TILE);
    // the eight inter-pass twiddles of each radix-8 half are fetched now, ahead of everything (table of 2^13 words, L2); behind an
    // opaque pointer: they are the same for every tile, and hoisted out of the loop they would hold 32 registers for good
    const uint64_t* twp = a.tw_pass;
    asm volatile("" : "+s"(twp));
    uint64_t twa[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r = ((t_hi + 8 * g2) << 3) | j;
            twa[g2][j] = twp[(r << 6) | b];
        }
    const uint32_t next = id + gridDim.x;
    if (CANON && !a.src_canonical) {
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = gl::canon(x[q]);
    }
    __syncthreads();  // w128, w64
#if defined(__HIP_DEVICE_COMPILE__)
    if (CANON)
        nttc::radix16_dif_ct<INVERSE>(x, w128, t_hi);
    else
#endif
        radix16_dif_ct<INVERSE>(x, w128, t_hi);
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[((((uint32_t)q << 3) | t_hi) << 6) | b] = x[q];
    __syncthreads();
    uint64_t y[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) y[g2][j] = tile[((((t_hi + 8 * g2) << 3) | j) << 6) | b];
    __syncthreads();  // everybody has read layout 1
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::radix8_dif_const<INVERSE>(y[g2]);
        else
#endif
            radix8_dif_const<INVERSE>(y[g2]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r = ((t_hi + 8 * g2) << 3) | j;
            uint64_t o;
#if defined(__HIP_DEVICE_COMPILE__)
            if (CANON)
                o = glc::mulc(y[g2][j], twa[g2][j]);
            else
#endif
                o = gl::mul(y[g2][j], twa[g2][j]);
            tile[r * AB_PITCH + b] = o;  // position (r << 6) | b of the block, regrouping layout
        }
    }
    __syncthreads();
    // ---- pass B on the 128 contiguous 64-point blocks: thread (blk0 : 6 | r3 : 3) holds local indices (j << 3) | r3 of blocks blk0, blk0 + 64 ----
    const uint32_t r3 = t & 7, blk0 = t >> 3;
    uint64_t z[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) z[g2][j] = tile[(blk0 + 64 * g2) * AB_PITCH + (j << 3) + r3];
    uint64_t tw[7];
    b_round_twiddles(w64, r3, tw);
    __syncthreads();  // everybody has read the regrouping layout
    // the NEXT tile of this workgroup starts its way from HBM now — pass A's registers are free — and arrives under pass B
    uint64_t xn[16];
    if (next < n_tiles) {
        const uint64_t* srcn = a.src + (size_t)(next >> log_tiles_per_col) * a.src_col_stride + ((size_t)(next & tile_mask) << AB_LOG_TILE);
#pragma unroll
        for (int q = 0; q < 16; ++q) xn[q] = srcn[((((uint32_t)q << 3) | t_hi) << 6) | b];
    }
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::b_round<INVERSE>(z[g2], tw);
        else
#endif
            b_round<INVERSE>(z[g2], tw);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[((blk0 + 64 * g2) * 8 + j) * B_PITCH + r3] = z[g2][j];  // b_lds_index(blk, (j << 3) | r3)
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        uint64_t* row = tile + (t + 512 * g2) * B_PITCH;  // row = blk * 8 + (r >> 3): eight consecutive local indices
        uint64_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[j];
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::radix8_dif_const<INVERSE>(v);
        else
#endif
            radix8_dif_const<INVERSE>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t o = v[j];
            if (a.out_scale) o = gl::mul(o, a.out_scale);
            if (a.canon_out && (!CANON || a.out_scale)) o = gl::canon(o);  // CANON: already < p
            row[j] = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[((blk0 + 64 * g2) << 6) + (j << 3) + r3] = tile[((blk0 + 64 * g2) * 8 + j) * B_PITCH + r3];
    if (next >= n_tiles) break;
    id = next;
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = xn[q];
  }
}

// The last NINE bits of a transform in one sweep (2^16-point transforms — the reference's block-circuit size — are 7 + 9 bits, and
// the nine used to go through the generic kernel: four LDS round trips of register-blocked radix-8 rounds with general twiddles,
// 2.3 times the time per bit of the specialised passes; 7.7 % of the 64-block-proof workload).  Same construction as the fused A+B
// pass with a 3-bit head: a workgroup of 128 threads owns 2048 contiguous elements = four 512-point blocks; thread (half, b) runs the
// shift-twiddle radix-8 over r = 0..7 (stride 64) of blocks half and half + 2 at column b, multiplies by w_512^(b rev3(r')) (table of
// the pass geometry (6, 3)), regroups through LDS into the thirty-two 64-point blocks, and pass B's two radix-8 rounds finish.
// HEAD = 1, 2, 3: the last 7, 8 or 9 bits (2^14-, 2^15-, 2^16-point transforms: the 2^(6 + HEAD)-point blocks of a tile take a
// radix-2 / radix-4 / radix-8 shift-twiddle round before pass B's rounds)
constexpr int C9_R_BITS = 9;
template <bool INVERSE, bool CANON, int HEAD>
__device__ __forceinline__ void c_head_round(uint64_t (&y)[1 << HEAD]) {
    if constexpr (HEAD == 3) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::radix8_dif_const<INVERSE>(y);
        else
#endif
            radix8_dif_const<INVERSE>(y);
    } else if constexpr (HEAD == 2) {  // w_4 = w_8^2
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON) {
            gl::bfly(y[0], y[2]);
            nttc::bfly_w8<INVERSE, 2>(y[1], y[3]);
            gl::bfly(y[0], y[1]);
            gl::bfly(y[2], y[3]);
        } else
#endif
        {
            uint64_t u = y[0], v = y[2];
            y[0] = gl::add(u, v), y[2] = gl::sub(u, v);
            u = y[1], v = y[3];
            y[1] = gl::add(u, v), y[3] = diff_times_w8<INVERSE, 2>(u, v);
            u = y[0], v = y[1];
            y[0] = gl::add(u, v), y[1] = gl::sub(u, v);
            u = y[2], v = y[3];
            y[2] = gl::add(u, v), y[3] = gl::sub(u, v);
        }
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            gl::bfly(y[0], y[1]);
        else
#endif
        {
            const uint64_t u = y[0], v = y[1];
            y[0] = gl::add(u, v), y[1] = gl::sub(u, v);
        }
    }
}
template <bool INVERSE, bool CANON, int HEAD = 3>
__global__ __launch_bounds__(A_THREADS, 4) void ntt_pass_c9_kernel(PassArgs a) {
    constexpr int G = 1 << HEAD, NG = 16 / G;  // elements per head group, groups per thread
    __shared__ uint64_t tile[32 * AB_PITCH];  // 2304 words: [block of 64][72] for the regrouping, then pass B's rows of 9
    __shared__ uint64_t w64[B_W64_ENTRIES];
    const uint32_t t = threadIdx.x;
    const size_t col = blockIdx.y;
    const uint64_t* src = a.src + col * a.src_col_stride + ((size_t)blockIdx.x << 11);
    uint64_t* dst = a.dst + col * a.dst_col_stride + ((size_t)blockIdx.x << 11);
    if (t < (uint32_t)B_W64_ENTRIES) {
        uint64_t e = (uint64_t)t << (a.K - B_R_BITS);
        if (INVERSE) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        w64[t] = root_pow(a, e);
    }
    const uint32_t b = t & 63, half = t >> 6;
    uint64_t y[NG][G], twa[G];
#pragma unroll
    for (int g2 = 0; g2 < NG; ++g2)
#pragma unroll
        for (int j = 0; j < G; ++j) y[g2][j] = src[((half + 2 * g2) << (6 + HEAD)) + (j << 6) + b];  // 512-byte runs
#pragma unroll
    for (int j = 0; j < G; ++j) twa[j] = a.tw_pass[(j << 6) | b];  // w_(64 G)^(b rev(j)): the same for every block
    if (CANON && !a.src_canonical) {
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2)
#pragma unroll
            for (int j = 0; j < G; ++j) y[g2][j] = gl::canon(y[g2][j]);
    }
#pragma unroll
    for (int g2 = 0; g2 < NG; ++g2) {
        c_head_round<INVERSE, CANON, HEAD>(y[g2]);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            uint64_t o;
#if defined(__HIP_DEVICE_COMPILE__)
            if (CANON)
                o = glc::mulc(y[g2][j], twa[j]);
            else
#endif
                o = gl::mul(y[g2][j], twa[j]);
            tile[(((half + 2 * g2) << HEAD) | j) * AB_PITCH + b] = o;  // 64-point block (blk << HEAD) | r', position b
        }
    }
    __syncthreads();
    // ---- pass B on the 32 contiguous 64-point blocks (ntt_pass_b_kernel's body, fed from LDS) ----
    const uint32_t r3 = t & 7, blk0 = t >> 3;
    uint64_t z[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) z[g2][j] = tile[(blk0 + 16 * g2) * AB_PITCH + (j << 3) + r3];
    uint64_t tw[7];
    b_round_twiddles(w64, r3, tw);
    __syncthreads();  // everybody has read the regrouping layout
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::b_round<INVERSE>(z[g2], tw);
        else
#endif
            b_round<INVERSE>(z[g2], tw);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[((blk0 + 16 * g2) * 8 + j) * B_PITCH + r3] = z[g2][j];
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        uint64_t* row = tile + (t + 128 * g2) * B_PITCH;
        uint64_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[j];
#if defined(__HIP_DEVICE_COMPILE__)
        if (CANON)
            nttc::radix8_dif_const<INVERSE>(v);
        else
#endif
            radix8_dif_const<INVERSE>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t o = v[j];
            if (a.out_scale) o = gl::mul(o, a.out_scale);
            if (a.canon_out && (!CANON || a.out_scale)) o = gl::canon(o);
            row[j] = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[((blk0 + 16 * g2) << 6) + (j << 3) + r3] = tile[((blk0 + 16 * g2) * 8 + j) * B_PITCH + r3];
}

// Pass B with the bit reversal folded into its store: the LAST pass of a forward / inverse transform that must deliver NATURAL
// order (fft / ifft of field/src/fft.rs as glp_ntt_dev exposes them; BASELINE configs[1] `ntt_fwd`), out of place.  After the DIF
// passes position p = (h : 4 | m : log_n - 10 | l : 6) holds X[rev(p)], rev(p) = rev6(l) << (log_n - 6) | rev(m) << 4 | rev4(h).
// A workgroup takes the 64-point transforms of ALL sixteen h for two values of m: it reads sixteen 512-byte runs per m and, for
// every l, writes the sixteen values of h as one 128-byte run — coalesced on both sides, so the separate bit-reversal sweep of
// the in-place form (16 n bytes of traffic, 0.6 of 2.1 ms for 135 columns of 2^20 points) disappears.  Not usable in place:
// the positions written are not the positions read.
template <bool INVERSE>
__global__ __launch_bounds__(A_THREADS, 4) void ntt_pass_b_natural_kernel(PassArgs a) {
    constexpr int BLK_PITCH = 8 * B_PITCH + 1;  // 73 words per 64-point block: the final read walks the blocks (h) across lanes
    __shared__ uint64_t tile[32 * BLK_PITCH];
    __shared__ uint64_t ltw[B_W64_ENTRIES];
    const uint32_t t = threadIdx.x, r3 = t & 7, blk0 = t >> 3;  // blk0 = h
    const size_t col = blockIdx.y;
    const int hb = 4, mb = a.log_n - B_R_BITS - hb;
    const size_t m0 = (size_t)blockIdx.x * 2;
    const uint64_t* src = a.src + col * a.src_col_stride;
    uint64_t* dst = a.dst + col * a.dst_col_stride;
    if (t < (uint32_t)B_W64_ENTRIES) {
        uint64_t e = (uint64_t)t << (a.K - B_R_BITS);
        if (INVERSE) e = (((uint64_t)1 << a.K) - e) & (((uint64_t)1 << a.K) - 1);
        ltw[t] = root_pow(a, e);
    }
    uint64_t x[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[g2][j] = src[((size_t)blk0 << (a.log_n - hb)) + ((m0 + g2) << 6) + (j << 3) + r3];
    __syncthreads();
    uint64_t tw[7];
    b_round_twiddles(ltw, r3, tw);
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        b_round<INVERSE>(x[g2], tw);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[(blk0 + 16 * g2) * BLK_PITCH + j * B_PITCH + r3] = x[g2][j];  // local index (j << 3) | r3
    }
    __syncthreads();
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        const uint32_t rowid = t + 128 * g2;  // block rowid >> 3, eight consecutive local indices (rowid & 7) << 3 ..
        uint64_t* row = tile + (rowid >> 3) * BLK_PITCH + (rowid & 7) * B_PITCH;
        uint64_t y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = row[j];
        radix8_dif_const<INVERSE>(y);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t v = y[j];
            if (a.out_scale) v = gl::mul(v, a.out_scale);
            if (a.canon_out) v = gl::canon(v);
            row[j] = v;
        }
    }
    __syncthreads();
    // store: lane = (h' = t & 15, lq = t >> 4); sixteen lanes write one 128-byte run of h' at frequency index rev6(l)
    const uint32_t hq = t & 15, lq = t >> 4, h = __brev(hq) >> 28;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t g2 = i >> 3, l = ((i & 7) << 3) | lq;  // local position l of block (g2, h)
        const uint64_t v = tile[(h + 16 * g2) * BLK_PITCH + (i & 7) * B_PITCH + lq];
        const size_t mr = (size_t)(__brev((uint32_t)(m0 + g2)) >> (32 - mb));
        dst[((size_t)(__brev(l) >> 26) << (a.log_n - B_R_BITS)) + (mr << hb) + hq] = v;
    }
}

// out[col][rev(p)] = in[col][p] * scale  (natural <-> bit-reversed reorder, out of place)
__global__ __launch_bounds__(THREADS) void bitrev_scale_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                               size_t in_col_stride, size_t out_col_stride, int log_n,
                                                               uint64_t scale) {
    size_t p = (size_t)blockIdx.x * THREADS + threadIdx.x;
    size_t col = blockIdx.y;
    if (p >= ((size_t)1 << log_n)) return;
    uint64_t v = in[col * in_col_stride + p];
    if (scale) v = gl::mul(v, scale);
    out[col * out_col_stride + rev_bits((uint32_t)p, log_n)] = gl::canon(v);
}

// the same map for log_n >= 12, tiled through LDS: with p = (hi:5 | mid | lo:5) the 32 x 32 tile "mid" of the input lands in
// tile rev(mid) of the output with rows and columns exchanged (hi' = rev5(lo), lo' = rev5(hi)): 256-byte runs on both sides
__global__ __launch_bounds__(THREADS) void bitrev_scale_tiled_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                     size_t in_col_stride, size_t out_col_stride, int log_n,
                                                                     uint64_t scale) {
    __shared__ uint64_t tile[32][33];
    const int mb = log_n - 10;
    const uint32_t mid = blockIdx.x, midr = rev_bits(mid, mb);
    const uint64_t* src = in + (size_t)blockIdx.y * in_col_stride;
    uint64_t* dst = out + (size_t)blockIdx.y * out_col_stride;
    const uint32_t lo = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hi = r0 + 8 * k;
        tile[hi][lo] = src[((size_t)hi << (mb + 5)) | ((size_t)mid << 5) | lo];
    }
    __syncthreads();
    const uint32_t rl = __brev(lo) >> 27;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hi = r0 + 8 * k, rh = __brev(hi) >> 27;
        uint64_t v = tile[rl][rh];
        if (scale) v = gl::mul(v, scale);
        dst[((size_t)hi << (mb + 5)) | ((size_t)midr << 5) | lo] = gl::canon(v);
    }
}

}  // namespace ntt
