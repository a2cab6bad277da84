// fri.hip — host-side Fiat-Shamir challenger, polynomial openings and the FRI prover on the GPU.
//
// Mirrors plonky2 @ f99ed9c src/iop/challenger.rs, src/plonk/proof.rs::OpeningSet::new, src/fri/oracle.rs::
// PolynomialBatch::prove_openings and src/fri/prover.rs::{fri_committed_trees, fri_proof_of_work,
// fri_prover_query_rounds}; the reference reaches them only through CircuitData::prove
// (reference src/rollup/circuits/mod.rs:1247, src/zkdsa/circuits/mod.rs:326).
//
// GPU formulation of prove_openings' front half.  plonky2 forms, per batch, comp(X) = sum_k alpha^k f_k(X) in
// coefficient form and divides by (X - z) with a sequential Horner scan.  Here the same polynomial
//     final(X) = X * sum_b alpha^(..) (comp_b(X) - comp_b(z_b)) / (X - z_b)        (degree < n)
// is produced from its n values on the coset 7*H — which are exactly chunk 0 of every committed LDE column, so no
// coefficient is re-read and no scan is needed — followed by one inverse coset NTT.  Interpolation is unique, so the
// coefficients (and everything derived from them) are identical to plonky2's.
#include "fri_kernels.h"
#include "internal.h"

using namespace glpi;

namespace {

void host_permute(uint64_t (&st)[12]) {
    poseidon::permute<poseidon::MDS_INT>(st);
    for (int i = 0; i < 12; ++i) st[i] = gl::canon(st[i]);
}

void duplexing(glp_challenger* ch) {
    uint64_t st[12];
    for (int i = 0; i < 12; ++i) st[i] = ch->sponge_state[i];
    for (uint32_t i = 0; i < ch->n_input; ++i) st[i] = ch->input_buffer[i];
    ch->n_input = 0;
    host_permute(st);
    for (int i = 0; i < 12; ++i) ch->sponge_state[i] = st[i];
    for (int i = 0; i < 8; ++i) ch->output_buffer[i] = st[i];
    ch->n_output = 8;
}

gl::ext2 ext_pow(gl::ext2 a, uint64_t e) {
    gl::ext2 r = {1, 0};
    while (e) {
        if (e & 1) r = gl::ext_mul(r, a);
        a = gl::ext_mul(a, a);
        e >>= 1;
    }
    return r;
}

unsigned total_bits(const glp_fri_params* p, unsigned upto) {
    unsigned s = 0;
    for (unsigned i = 0; i < upto; ++i) s += p->reduction_arity_bits[i];
    return s;
}

// zpow table for the evaluation point z (cached in the context)
int32_t ensure_zpow(glp_ctx* c, int log_n, const uint64_t z[2]) {
    const size_t n = (size_t)1 << log_n;
    if (c->zpow_dev && c->zpow_log_n == log_n && c->zpow_z[0] == z[0] && c->zpow_z[1] == z[1]) return GLP_OK;
    if (c->zpow_dev && c->zpow_log_n != log_n) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(c->zpow_dev));
        c->zpow_dev = nullptr;
    }
    if (!c->zpow_dev) HIP_TRY(hipMalloc((void**)&c->zpow_dev, 2 * n * 8));
    fri::PowArgs a;
    gl::ext2 p = {gl::canon(z[0]), gl::canon(z[1])};
    for (int k = 0; k < 32; ++k) {
        a.z2k[k][0] = gl::canon(p.a);
        a.z2k[k][1] = gl::canon(p.b);
        p = gl::ext_mul(p, p);
    }
    Timed t(c, GLP_K_OPENINGS);
    size_t threads = (n + 15) / 16;
    hipLaunchKernelGGL(fri::ext_powers_kernel, dim3((unsigned)((threads + fri::THREADS - 1) / fri::THREADS)),
                       dim3(fri::THREADS), 0, c->stream, a, c->zpow_dev, c->zpow_dev + n, n);
    HIP_TRY(hipGetLastError());
    c->zpow_log_n = log_n;
    c->zpow_z[0] = z[0];
    c->zpow_z[1] = z[1];
    return GLP_OK;
}

int32_t eval_columns(glp_ctx* c, const uint64_t* coeffs, size_t col_stride, size_t n_cols, int log_n, const uint64_t z[2],
                     uint64_t* out_host) {
    const size_t n = (size_t)1 << log_n;
    GLP_TRY(ensure_zpow(c, log_n, z));
    const size_t chunk = (size_t)fri::THREADS * fri::EVAL_PER;
    const int n_blocks = (int)((n + chunk - 1) / chunk);
    uint64_t *partial = nullptr, *out_dev = nullptr;
    GLP_TRY(pool_alloc_t(c, n_cols * n_blocks * 2, &partial));
    int32_t s = pool_alloc_t(c, n_cols * 2, &out_dev);
    if (s == GLP_OK) {
        Timed t(c, GLP_K_OPENINGS);
        for (size_t c0 = 0; c0 < n_cols && s == GLP_OK; c0 += 65535) {
            size_t cnt = n_cols - c0 < 65535 ? n_cols - c0 : 65535;
            hipLaunchKernelGGL(fri::eval_partial_kernel, dim3((unsigned)n_blocks, (unsigned)cnt), dim3(fri::THREADS), 0,
                               c->stream, coeffs + c0 * col_stride, col_stride, n, c->zpow_dev, c->zpow_dev + n,
                               partial + c0 * n_blocks * 2);
            hipLaunchKernelGGL(fri::eval_final_kernel, dim3((unsigned)cnt), dim3(64), 0, c->stream,
                               partial + c0 * n_blocks * 2, n_blocks, out_dev + c0 * 2);
            if (hipGetLastError() != hipSuccess) s = GLP_ERR_HIP;
        }
    }
    if (s == GLP_OK) {
        hipError_t e = hipMemcpyAsync(out_host, out_dev, n_cols * 16, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            g_last_hip = (int32_t)e;
            s = GLP_ERR_HIP;
        }
    }
    pool_free(c, partial);
    pool_free(c, out_dev);
    return s;
}

struct DevBuf {  // pool-backed scratch released on scope exit
    glp_ctx* c;
    std::vector<void*> ptrs;
    explicit DevBuf(glp_ctx* ctx) : c(ctx) {}
    ~DevBuf() {
        for (void* p : ptrs) pool_free(c, p);
    }
    int32_t alloc(size_t words, uint64_t** out) {
        GLP_TRY(pool_alloc_t(c, words ? words : 1, out));
        ptrs.push_back(*out);
        return GLP_OK;
    }
};

}  // namespace

extern "C" {

void glp_challenger_init(glp_challenger* ch) {
    if (ch) memset(ch, 0, sizeof *ch);
}

void glp_challenger_observe(glp_challenger* ch, const uint64_t* elements, size_t n) {
    if (!ch || !elements) return;
    for (size_t k = 0; k < n; ++k) {
        ch->n_output = 0;  // any observation invalidates buffered challenges
        ch->input_buffer[ch->n_input++] = gl::canon(elements[k]);
        if (ch->n_input == 8) duplexing(ch);
    }
}

uint64_t glp_challenger_get(glp_challenger* ch) {
    if (ch->n_input != 0 || ch->n_output == 0) duplexing(ch);
    return ch->output_buffer[--ch->n_output];  // Vec::pop
}

void glp_challenger_get_n(glp_challenger* ch, uint64_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = glp_challenger_get(ch);
}

int32_t glp_batch_eval_ext(glp_ctx* c, const glp_batch* b, size_t col_begin, size_t n_cols, const uint64_t z[2],
                           uint64_t* out) {
    if (!c || !b || !z || !out || n_cols == 0 || col_begin + n_cols > b->n_cols) return GLP_ERR_INVALID;
    const size_t n = (size_t)1 << b->log_n;
    return eval_columns(c, b->coeffs + col_begin * n, n, n_cols, (int)b->log_n, z, out);
}

size_t glp_fri_proof_len(const glp_fri_params* p, const size_t* oracle_n_cols, size_t n_oracles) {
    if (!p || !oracle_n_cols) return 0;
    const unsigned lde_bits = p->degree_bits + p->rate_bits;
    const size_t cap = (size_t)4 << p->cap_height;
    size_t len = p->n_reduction_rounds * cap, per_query = 0;
    for (size_t o = 0; o < n_oracles; ++o) per_query += oracle_n_cols[o] + (size_t)(lde_bits - p->cap_height) * 4;
    for (unsigned r = 0; r < p->n_reduction_rounds; ++r) {
        unsigned tree_bits = lde_bits - total_bits(p, r + 1);
        per_query += ((size_t)2 << p->reduction_arity_bits[r]) + (size_t)(tree_bits - p->cap_height) * 4;
    }
    len += per_query * p->num_query_rounds;
    len += (size_t)2 << (p->degree_bits - total_bits(p, p->n_reduction_rounds));
    return len + 1;
}

int32_t glp_fri_prove(glp_ctx* c, const glp_batch* const* oracles, size_t n_oracles, const glp_fri_batch* batches,
                      size_t n_batches, const glp_fri_params* p, glp_challenger* ch, uint64_t* proof_out) {
    if (!c || !oracles || !batches || !p || !ch || !proof_out || n_oracles == 0 || n_oracles > 8 || n_batches == 0)
        return GLP_ERR_INVALID;
    const int log_n = (int)p->degree_bits, rb = (int)p->rate_bits;
    const unsigned lde_bits = p->degree_bits + p->rate_bits;
    if (p->n_reduction_rounds > 16 || lde_bits > (unsigned)MASTER_K) return GLP_ERR_INVALID;
    if (total_bits(p, p->n_reduction_rounds) > p->degree_bits) return GLP_ERR_INVALID;
    for (size_t o = 0; o < n_oracles; ++o)
        if (!oracles[o] || oracles[o]->log_n != p->degree_bits || oracles[o]->rate_bits != p->rate_bits ||
            oracles[o]->cap_height != p->cap_height)
            return GLP_ERR_INVALID;
    for (unsigned r = 0; r < p->n_reduction_rounds; ++r) {
        unsigned tree_bits = lde_bits - total_bits(p, r + 1);
        if (tree_bits < p->cap_height || p->reduction_arity_bits[r] == 0 || p->reduction_arity_bits[r] > 6)
            return GLP_ERR_INVALID;
    }
    const size_t n = (size_t)1 << log_n, N = n << rb;
    const size_t cap_words = (size_t)4 << p->cap_height;
    DevBuf scratch(c);

    // ---- alpha and the final polynomial (values on 7*H -> coefficients) ----
    uint64_t alpha_w[2];
    glp_challenger_get_n(ch, alpha_w, 2);
    const gl::ext2 alpha = {alpha_w[0], alpha_w[1]};
    uint64_t* fin = nullptr;  // [2][n] limbs: values, then coefficients of final_poly
    GLP_TRY(scratch.alloc(2 * n, &fin));
    for (size_t b = 0; b < n_batches; ++b) {
        const glp_fri_batch& bt = batches[b];
        if (!bt.oracle_index || !bt.poly_index || bt.n_polys == 0) return GLP_ERR_INVALID;
        std::vector<const uint64_t*> cols(bt.n_polys);
        std::vector<uint64_t> apow(2 * bt.n_polys);
        gl::ext2 ap = {1, 0}, opened = {0, 0};
        std::vector<uint64_t> ev;
        for (size_t k = 0; k < bt.n_polys; ++k) {
            if (bt.oracle_index[k] >= n_oracles || bt.poly_index[k] >= oracles[bt.oracle_index[k]]->n_cols)
                return GLP_ERR_INVALID;
            const glp_batch* ob = oracles[bt.oracle_index[k]];
            cols[k] = ob->lde + (size_t)bt.poly_index[k] * N;  // chunk 0 = values on 7*H, bit-reversed
            apow[2 * k] = gl::canon(ap.a);
            apow[2 * k + 1] = gl::canon(ap.b);
            gl::ext2 v;
            if (bt.opened_values) {
                v = {bt.opened_values[2 * k], bt.opened_values[2 * k + 1]};
            } else {
                uint64_t tmp[2];
                GLP_TRY(eval_columns(c, ob->coeffs + (size_t)bt.poly_index[k] * n, n, 1, log_n, bt.point, tmp));
                v = {tmp[0], tmp[1]};
            }
            opened = gl::ext_add(opened, gl::ext_mul(ap, v));
            ap = gl::ext_mul(ap, alpha);
        }
        uint64_t *cols_dev = nullptr, *apow_dev = nullptr;
        GLP_TRY(scratch.alloc(bt.n_polys, &cols_dev));
        GLP_TRY(scratch.alloc(2 * bt.n_polys, &apow_dev));
        HIP_TRY(hipMemcpyAsync(cols_dev, cols.data(), bt.n_polys * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(apow_dev, apow.data(), 2 * bt.n_polys * 8, hipMemcpyHostToDevice, c->stream));
        fri::CombineArgs a;
        a.cols = reinterpret_cast<const uint64_t* const*>(cols_dev);
        a.apow = apow_dev;
        a.n_polys = (int)bt.n_polys;
        a.log_n = log_n;
        a.shift = gl::COSET_SHIFT;
        a.opened[0] = gl::canon(opened.a);
        a.opened[1] = gl::canon(opened.b);
        a.z[0] = gl::canon(bt.point[0]);
        a.z[1] = gl::canon(bt.point[1]);
        gl::ext2 as = ap;  // alpha^(n_polys)
        a.ashift[0] = gl::canon(as.a);
        a.ashift[1] = gl::canon(as.b);
        a.first = b == 0;
        a.last = b + 1 == n_batches && !(p->flags & GLP_FLAG_FRI_NO_MUL_X);
        a.K = MASTER_K;
        a.w_lo = c->w_lo_dev;
        a.w_hi = c->w_hi_dev;
        {
            Timed t(c, GLP_K_FRI);
            hipLaunchKernelGGL(fri::fri_combine_kernel, dim3((unsigned)((n + fri::THREADS - 1) / fri::THREADS)),
                               dim3(fri::THREADS), 0, c->stream, a, fin, fin + n);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(c->stream));  // host vectors cols/apow go out of scope
    }
    // values on 7*H (bit-reversed) -> coefficients: inverse DIT, 1/n, then un-shift by 7^-j
    const uint64_t n_inv = gl::canon(gl::inv((uint64_t)n));
    if (log_n < 9) {
        uint64_t* tmp = nullptr;
        GLP_TRY(scratch.alloc(2 * n, &tmp));
        HIP_TRY(hipMemcpyAsync(tmp, fin, 2 * n * 8, hipMemcpyDeviceToDevice, c->stream));
        GLP_TRY(launch_small(c, tmp, n, fin, n, 2, log_n, log_n, 1, 1, gl::canon(gl::inv(gl::COSET_SHIFT)), n_inv, 0, 1));
    } else {
        GLP_TRY(run_dit(c, fin, n, 2, log_n, 1, n_inv));
        GLP_TRY(scale_by_powers(c, fin, n, 2, log_n, gl::canon(gl::inv(gl::COSET_SHIFT)), 1));
    }

    // ---- commit phase ----
    struct Tree {
        uint64_t* leaves;   // [n_leaves][2*arity] row-major
        uint64_t* digests;
        size_t n_leaves;
        int arity_bits;
    };
    std::vector<Tree> trees;
    uint64_t* coeffs = fin;  // [2][cur_n] limbs, stride cur_n
    int cur_log = log_n;
    uint64_t shift = gl::COSET_SHIFT;
    uint64_t* w = proof_out;
    for (unsigned r = 0; r < p->n_reduction_rounds; ++r) {
        const int ab = (int)p->reduction_arity_bits[r];
        const size_t cur_n = (size_t)1 << cur_log, cur_N = cur_n << rb;
        // values = coset_fft(lde(coeffs), shift), bit-reversed
        uint64_t* vals = nullptr;
        GLP_TRY(scratch.alloc(2 * cur_N, &vals));
        GLP_TRY(lde_from_coeffs(c, coeffs, vals, 2, cur_log, rb, shift));
        Tree t;
        t.arity_bits = ab;
        t.n_leaves = cur_N >> ab;
        GLP_TRY(scratch.alloc(2 * cur_N, &t.leaves));
        GLP_TRY(scratch.alloc(digests_len(t.n_leaves, p->cap_height), &t.digests));
        {
            Timed tt(c, GLP_K_FRI);
            hipLaunchKernelGGL(fri::fri_interleave_kernel, dim3((unsigned)((cur_N + fri::THREADS - 1) / fri::THREADS)),
                               dim3(fri::THREADS), 0, c->stream, vals, vals + cur_N, cur_N, t.leaves);
            HIP_TRY(hipGetLastError());
        }
        const size_t leaf_len = (size_t)2 << ab;
        GLP_TRY(merkle_build(c, t.leaves, t.n_leaves, leaf_len, leaf_len, 1, p->cap_height, t.digests, w));
        trees.push_back(t);
        glp_challenger_observe(ch, w, cap_words);
        w += cap_words;
        uint64_t beta[2];
        glp_challenger_get_n(ch, beta, 2);
        // fold the coefficients
        const size_t out_n = cur_n >> ab;
        uint64_t* folded = nullptr;
        GLP_TRY(scratch.alloc(2 * (out_n ? out_n : 1), &folded));
        {
            Timed tt(c, GLP_K_FRI);
            hipLaunchKernelGGL(fri::fri_fold_kernel, dim3((unsigned)((out_n + fri::THREADS - 1) / fri::THREADS)),
                               dim3(fri::THREADS), 0, c->stream, coeffs, coeffs + cur_n, out_n, 1 << ab, beta[0], beta[1],
                               folded, folded + out_n);
            HIP_TRY(hipGetLastError());
        }
        coeffs = folded;
        cur_log -= ab;
        shift = gl::canon(gl::pow(shift, (uint64_t)1 << ab));
    }
    // final polynomial
    const size_t n_final = (size_t)1 << cur_log;
    std::vector<uint64_t> fp(2 * n_final), final_poly(2 * n_final);
    HIP_TRY(hipMemcpyAsync(fp.data(), coeffs, 2 * n_final * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n_final; ++i) {
        final_poly[2 * i] = gl::canon(fp[i]);
        final_poly[2 * i + 1] = gl::canon(fp[n_final + i]);
    }
    glp_challenger_observe(ch, final_poly.data(), 2 * n_final);

    // ---- proof of work ----
    fri::PowArgs2 pa;
    memset(&pa, 0, sizeof pa);
    pa.bits = (int)p->proof_of_work_bits;
    if (p->flags & GLP_FLAG_POW_DUPLEX) {  // response = the challenge squeezed right after observing the witness
        for (int i = 0; i < 12; ++i) pa.state[i] = ch->sponge_state[i];
        for (uint32_t i = 0; i < ch->n_input; ++i) pa.state[i] = ch->input_buffer[i];
        pa.pos = (int)ch->n_input;
        pa.out_word = 7;
    } else {  // response = hash_no_pad(challenger.get_hash() || witness).elements[0]
        glp_challenger_get_n(ch, pa.state, 4);
        pa.pos = 4;
        pa.out_word = 0;
    }
    uint64_t pow_witness = 0;
    {
        unsigned long long* best = nullptr;
        uint64_t* best_w = nullptr;
        GLP_TRY(scratch.alloc(1, &best_w));
        best = reinterpret_cast<unsigned long long*>(best_w);
        // expected work is 2^proof_of_work_bits candidates: first launch 2x that (hit rate 86 %), then double
        uint64_t batch = (uint64_t)2 << (p->proof_of_work_bits < 19 ? p->proof_of_work_bits : 19);
        if (batch < 4096) batch = 4096;
        uint64_t start = 0;
        for (;;) {
            HIP_TRY(hipMemsetAsync(best, 0xFF, 8, c->stream));
            {
                Timed t(c, GLP_K_FRI);
                hipLaunchKernelGGL(fri::fri_pow_kernel, dim3((unsigned)((batch + fri::THREADS - 1) / fri::THREADS)), dim3(fri::THREADS), 0,
                                   c->stream, pa, start, batch, best);
                HIP_TRY(hipGetLastError());
            }
            unsigned long long h = 0;
            HIP_TRY(hipMemcpyAsync(&h, best, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (h != ~0ull) {
                pow_witness = h;
                break;
            }
            start += batch;
            if (batch < ((uint64_t)1 << 22)) batch <<= 1;
            if (start >= gl::P - batch) return GLP_ERR_INVALID;  // "Proof of work failed. This is highly unlikely!"
        }
    }

    if (p->flags & GLP_FLAG_POW_DUPLEX) {
        glp_challenger_observe(ch, &pow_witness, 1);
        const uint64_t resp = glp_challenger_get(ch);
        if (p->proof_of_work_bits && (resp >> (64 - p->proof_of_work_bits)) != 0) return GLP_ERR_INVALID;  // same computation
    }

    // ---- query phase ----
    const uint32_t nq = p->num_query_rounds;
    std::vector<uint64_t> idx(nq);
    for (uint32_t q = 0; q < nq; ++q) idx[q] = glp_challenger_get(ch) % N;
    const int plen0 = (int)(lde_bits - p->cap_height);
    std::vector<std::vector<uint64_t>> rows(n_oracles), paths(n_oracles);
    for (size_t o = 0; o < n_oracles; ++o) {
        rows[o].resize((size_t)nq * oracles[o]->n_cols);
        paths[o].resize((size_t)nq * plen0 * 4 + 1);
        GLP_TRY(gather_rows(c, oracles[o]->lde, 1, N, oracles[o]->n_cols, oracles[o]->digests, N, plen0, idx.data(), nq,
                            rows[o].data(), plen0 ? paths[o].data() : nullptr));
    }
    std::vector<std::vector<uint64_t>> trows(trees.size()), tpaths(trees.size());
    std::vector<uint64_t> tidx = idx;
    unsigned tb = lde_bits;
    for (size_t r = 0; r < trees.size(); ++r) {
        const int ab = trees[r].arity_bits;
        const size_t leaf_len = (size_t)2 << ab;
        for (uint32_t q = 0; q < nq; ++q) tidx[q] >>= ab;
        const int pl = (int)(tb - ab - p->cap_height);
        trows[r].resize((size_t)nq * leaf_len);
        tpaths[r].resize((size_t)nq * pl * 4 + 1);
        GLP_TRY(gather_rows(c, trees[r].leaves, leaf_len, 1, leaf_len, trees[r].digests, trees[r].n_leaves, pl, tidx.data(),
                            nq, trows[r].data(), pl ? tpaths[r].data() : nullptr));
        tb -= ab;
    }
    for (uint32_t q = 0; q < nq; ++q) {
        for (size_t o = 0; o < n_oracles; ++o) {
            const size_t nc = oracles[o]->n_cols;
            memcpy(w, rows[o].data() + (size_t)q * nc, nc * 8);
            w += nc;
            memcpy(w, paths[o].data() + (size_t)q * plen0 * 4, (size_t)plen0 * 32);
            w += (size_t)plen0 * 4;
        }
        unsigned bits = lde_bits;
        for (size_t r = 0; r < trees.size(); ++r) {
            const int ab = trees[r].arity_bits;
            const size_t leaf_len = (size_t)2 << ab;
            const int pl = (int)(bits - ab - p->cap_height);
            memcpy(w, trows[r].data() + (size_t)q * leaf_len, leaf_len * 8);
            w += leaf_len;
            memcpy(w, tpaths[r].data() + (size_t)q * pl * 4, (size_t)pl * 32);
            w += (size_t)pl * 4;
            bits -= ab;
        }
    }
    memcpy(w, final_poly.data(), 2 * n_final * 8);
    w += 2 * n_final;
    *w++ = pow_witness;
    return GLP_OK;
}

}  // extern "C"
