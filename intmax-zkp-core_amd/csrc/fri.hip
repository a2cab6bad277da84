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

}  // namespace

namespace glpi {
// power tables z_k^j, j < n, for K evaluation points: zpow [K][2][n] limb columns (a pool block owned by the caller)
int32_t build_zpow(glp_ctx* c, int log_n, size_t K, const uint64_t* zs /* host [K][2] */, uint64_t** zpow_out) {
    const size_t n = (size_t)1 << log_n;
    if (K > 65535) return GLP_ERR_UNSUPPORTED;
    fri::PowArgs* args = arena_alloc_t<fri::PowArgs>(c, K);  // pinned staging: no wait for the upload
    if (!args) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
    parallel_for(K, [&](size_t k) {
        gl::ext2 p = {gl::canon(zs[2 * k]), gl::canon(zs[2 * k + 1])};
        for (int i = 0; i < 32; ++i) {
            args[k].z2k[i][0] = gl::canon(p.a);
            args[k].z2k[i][1] = gl::canon(p.b);
            p = gl::ext_mul(p, p);
        }
    });
    uint64_t* zpow = nullptr;
    fri::PowArgs* args_dev = nullptr;
    GLP_TRY(pool_alloc_t(c, K * 2 * n, &zpow));
    int32_t s = pool_alloc_t(c, K, &args_dev);
    hipError_t e = hipSuccess;
    if (s == GLP_OK) {
        e = hipMemcpyAsync(args_dev, args, K * sizeof(fri::PowArgs), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            Timed t(c, GLP_K_OPENINGS);
            const size_t threads = (n + 15) / 16;
            hipLaunchKernelGGL(fri::ext_powers_kernel, dim3((unsigned)((threads + fri::THREADS - 1) / fri::THREADS), (unsigned)K),
                               dim3(fri::THREADS), 0, c->stream, args_dev, zpow, n);
            e = hipGetLastError();
        }
    }
    pool_free(c, args_dev);  // stream-ordered pool: the kernel above reads it before any later allocation can write it
    if (s == GLP_OK && e != hipSuccess) {
        note_hip_error(e);
        s = GLP_ERR_HIP;
    }
    if (s != GLP_OK) {
        pool_free(c, zpow);
        return s;
    }
    *zpow_out = zpow;
    return GLP_OK;
}

size_t eval_partial_blocks(int log_n) {
    const size_t n = (size_t)1 << log_n, chunk = (size_t)fri::THREADS * fri::EVAL_PER;
    return (n + chunk - 1) / chunk;
}

// launches only (no wait, no read-back): out_dev [K][n_cols][2]
int32_t eval_launch(glp_ctx* c, const uint64_t* coeffs, size_t col_stride, size_t n_cols, size_t tree_cols, int log_n, size_t K,
                    const uint64_t* zpow, uint64_t* partial, uint64_t* out_dev) {
    const size_t n = (size_t)1 << log_n;
    const int n_blocks = (int)eval_partial_blocks(log_n);
    if (n_cols > 65535 || K > 65535) return GLP_ERR_UNSUPPORTED;
    Timed t(c, GLP_K_OPENINGS);
    if (n_blocks == 1) {  // one block per column IS the result (same layout, already canonical): no second launch
        hipLaunchKernelGGL(fri::eval_partial_kernel, dim3(1u, (unsigned)n_cols, (unsigned)K), dim3(fri::THREADS), 0, c->stream, coeffs,
                           col_stride, n, tree_cols, zpow, out_dev);
        HIP_TRY(hipGetLastError());
        return GLP_OK;
    }
    hipLaunchKernelGGL(fri::eval_partial_kernel, dim3((unsigned)n_blocks, (unsigned)n_cols, (unsigned)K), dim3(fri::THREADS), 0,
                       c->stream, coeffs, col_stride, n, tree_cols, zpow, partial);
    hipLaunchKernelGGL(fri::eval_final_kernel, dim3((unsigned)(n_cols * K)), dim3(64), 0, c->stream, partial, n_blocks, out_dev);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

// out_host [K][n_cols][2] = polynomials [k * tree_cols + col] (tree_cols = 0: the same n_cols columns for every point) at
// the K points whose power tables are in zpow
int32_t eval_with_zpow(glp_ctx* c, const uint64_t* coeffs, size_t col_stride, size_t n_cols, size_t tree_cols, int log_n,
                       size_t K, const uint64_t* zpow, uint64_t* out_host) {
    uint64_t *partial = nullptr, *out_dev = nullptr;
    GLP_TRY(pool_alloc_t(c, K * n_cols * eval_partial_blocks(log_n) * 2, &partial));
    int32_t s = pool_alloc_t(c, K * n_cols * 2, &out_dev);
    if (s == GLP_OK) s = eval_launch(c, coeffs, col_stride, n_cols, tree_cols, log_n, K, zpow, partial, out_dev);
    hipError_t e = hipSuccess;
    if (s == GLP_OK) e = hipMemcpyAsync(out_host, out_dev, K * n_cols * 16, hipMemcpyDeviceToHost, c->stream);
    if (s == GLP_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    pool_free(c, partial);
    pool_free(c, out_dev);
    if (s == GLP_OK && e != hipSuccess) {
        note_hip_error(e);
        s = GLP_ERR_HIP;
    }
    return s;
}

}  // namespace glpi

namespace {
// single point, with the context's cache of the last point's power table (OpeningSet::new evaluates four batches at zeta)
int32_t eval_columns(glp_ctx* c, const uint64_t* coeffs, size_t col_stride, size_t n_cols, int log_n, const uint64_t z[2],
                     uint64_t* out_host) {
    if (!(c->zpow_dev && c->zpow_log_n == log_n && c->zpow_z[0] == z[0] && c->zpow_z[1] == z[1])) {
        if (c->zpow_dev) {
            pool_free(c, c->zpow_dev);
            c->zpow_dev = nullptr;
        }
        GLP_TRY(build_zpow(c, log_n, 1, z, &c->zpow_dev));
        c->zpow_log_n = log_n;
        c->zpow_z[0] = z[0];
        c->zpow_z[1] = z[1];
    }
    for (size_t c0 = 0; c0 < n_cols; c0 += 65535) {
        const size_t cnt = n_cols - c0 < 65535 ? n_cols - c0 : 65535;
        GLP_TRY(eval_with_zpow(c, coeffs + c0 * col_stride, col_stride, cnt, 0, log_n, 1, c->zpow_dev, out_host + 2 * c0));
    }
    return GLP_OK;
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

// ---- host-side Poseidon utilities (known-answer tests, the reference's native one-off hashes) ----
void glp_poseidon_permute(uint64_t state[12]) {
    if (!state) return;
    uint64_t st[12];
    for (int i = 0; i < 12; ++i) st[i] = state[i];
    host_permute(st);
    for (int i = 0; i < 12; ++i) state[i] = st[i];
}

void glp_poseidon_round_constants(uint64_t out[360]) {
    static const uint64_t rc[360] = {GL_POSEIDON_RC_LIST};
    if (!out) return;
    for (int i = 0; i < 360; ++i) out[i] = rc[i];
}

void glp_hash_no_pad(const uint64_t* in, size_t n, uint64_t out[4]) {
    if (!out || (n && !in)) return;
    uint64_t st[12] = {0};
    for (size_t off = 0; off < n; off += 8) {  // overwrite-mode sponge; no chunk, no permutation: hash of nothing = 0
        size_t k = n - off < 8 ? n - off : 8;
        for (size_t i = 0; i < k; ++i) st[i] = gl::canon(in[off + i]);
        host_permute(st);
    }
    for (int i = 0; i < 4; ++i) out[i] = st[i];
}

void glp_hash_pad(const uint64_t* in, size_t n, uint64_t out[4]) {
    if (!out || (n && !in)) return;
    std::vector<uint64_t> v(in, in + n);  // push 1; push 0 while (len + 1) % 12 != 0; push 1 (a multiple of the WIDTH)
    v.push_back(1);
    while ((v.size() + 1) % 12 != 0) v.push_back(0);
    v.push_back(1);
    glp_hash_no_pad(v.data(), v.size(), out);
}

void glp_hash_or_noop(const uint64_t* in, size_t n, uint64_t out[4]) {
    if (!out || (n && !in)) return;
    if (n <= 4) {
        for (size_t i = 0; i < 4; ++i) out[i] = i < n ? gl::canon(in[i]) : 0;
        return;
    }
    glp_hash_no_pad(in, n, out);
}

void glp_two_to_one(const uint64_t left[4], const uint64_t right[4], uint64_t out[4]) {
    if (!left || !right || !out) return;
    uint64_t st[12] = {0};
    for (int i = 0; i < 4; ++i) {
        st[i] = gl::canon(left[i]);
        st[4 + i] = gl::canon(right[i]);
    }
    host_permute(st);
    for (int i = 0; i < 4; ++i) out[i] = st[i];
}

int32_t glp_batch_eval_ext(glp_ctx* c, const glp_batch* b, size_t col_begin, size_t n_cols, const uint64_t z[2],
                           uint64_t* out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !b || !z || !out || n_cols == 0 || col_begin + n_cols > b->n_cols) return GLP_ERR_INVALID;
    if (b->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    const size_t n = (size_t)1 << b->log_n;
    GLP_TRY(arena_reset(c));
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

}  // extern "C"

namespace glpi {

// PolynomialBatch::prove_openings for K proofs side by side.  Oracle o holds either one tree shared by all proofs (the
// circuit's constants_sigmas) or K trees (n_trees == K, proof k's polynomials in tree k).  batches: [K][n_batches] — the
// points and opened values differ per proof, the polynomial lists (oracle_index / poly_index) are those of batches[0..n_batches).
// chs [K], proof_out [K].
// what the eleven fixed words of the proof-of-work permutation contribute to the state entering round 1 (fri::PowArgs2)
static void pow_precompute(const uint64_t state[12], int pos, uint64_t base[12]) {
    uint64_t sb[12];
    for (int i = 0; i < 12; ++i) sb[i] = i == pos ? 0 : poseidon::sbox7(gl::add(state[i], poseidon::RC[i]));
    for (int r = 0; r < 12; ++r) {
        uint64_t acc = poseidon::RC[12 + r];
        for (int i = 0; i < 12; ++i) acc = gl::add(acc, gl::mul(sb[(i + r) % 12], poseidon::MDS_CIRC_TAB[i]));
        if (r == 0) acc = gl::add(acc, gl::mul(sb[0], 8));
        base[r] = gl::canon(acc);
    }
}

int32_t fri_prove_forest(glp_ctx* c, const glp_batch* const* oracles, size_t n_oracles, size_t K, const glp_fri_batch* batches,
                         size_t n_batches, const glp_fri_params* p, glp_challenger* chs, uint64_t* const* proof_out) {
    if (!c || !oracles || !batches || !p || !chs || !proof_out || n_oracles == 0 || n_oracles > 8 || n_batches == 0 || K == 0 ||
        K > 65535)
        return GLP_ERR_INVALID;
    const int log_n = (int)p->degree_bits, rb = (int)p->rate_bits;
    const unsigned lde_bits = p->degree_bits + p->rate_bits;
    if (p->n_reduction_rounds > 16 || p->rate_bits > 8 || p->degree_bits > (unsigned)MASTER_K || lde_bits > (unsigned)MASTER_K ||
        p->cap_height > lde_bits || p->proof_of_work_bits > 32 || p->num_query_rounds == 0 || p->num_query_rounds > 4096)
        return GLP_ERR_INVALID;
    if (total_bits(p, p->n_reduction_rounds) > p->degree_bits) return GLP_ERR_INVALID;
    for (size_t o = 0; o < n_oracles; ++o)
        if (!oracles[o] || oracles[o]->log_n != p->degree_bits || oracles[o]->rate_bits != p->rate_bits ||
            oracles[o]->cap_height != p->cap_height || (oracles[o]->n_trees != 1 && oracles[o]->n_trees != K))
            return GLP_ERR_INVALID;
    for (unsigned r = 0; r < p->n_reduction_rounds; ++r) {
        unsigned tree_bits = lde_bits - total_bits(p, r + 1);
        if (tree_bits < p->cap_height || p->reduction_arity_bits[r] == 0 || p->reduction_arity_bits[r] > 6)
            return GLP_ERR_INVALID;
    }
    const size_t n = (size_t)1 << log_n, N = n << rb;
    const size_t cap_words = (size_t)4 << p->cap_height;
    DevBuf scratch(c);
    auto tree_of = [&](size_t o, size_t k) { return oracles[o]->n_trees == 1 ? (size_t)0 : k; };

    StepTrace trace("fri_prove_forest");
    // ---- alpha and the final polynomial (values on 7*H -> coefficients) ----
    std::vector<gl::ext2> alpha(K);
    parallel_for(K, [&](size_t k) {
        uint64_t aw[2];
        glp_challenger_get_n(&chs[k], aw, 2);
        alpha[k] = {aw[0], aw[1]};
    });
    uint64_t* fin = nullptr;  // [K][2][n] limbs: values, then coefficients of final_poly
    GLP_TRY(scratch.alloc(K * 2 * n, &fin));
    // the tables of ALL batches (column pointers, alpha powers, per-proof parameters) are made first and go up in one copy
    struct BatchTabs {
        const uint64_t** cols;
        uint64_t* apow;
        fri::CombineProof* prs;
        size_t np;
    };
    std::vector<BatchTabs> tabs(n_batches);
    size_t tab_bytes = 0;
    for (size_t b = 0; b < n_batches; ++b) {
        const glp_fri_batch& b0 = batches[b];
        if (!b0.oracle_index || !b0.poly_index || b0.n_polys == 0) return GLP_ERR_INVALID;
        const size_t np = b0.n_polys;
        for (size_t j = 0; j < np; ++j)
            if (b0.oracle_index[j] >= n_oracles || b0.poly_index[j] >= oracles[b0.oracle_index[j]]->n_cols) return GLP_ERR_INVALID;
        tab_bytes += StagedUpload::need(K * np * 8) + StagedUpload::need(K * 2 * np * 8) + StagedUpload::need(K * sizeof(fri::CombineProof));
    }
    StagedUpload up(c, tab_bytes);
    for (size_t b = 0; b < n_batches; ++b) {
        const glp_fri_batch& b0 = batches[b];
        const size_t np = b0.n_polys;
        const uint64_t** cols = up.take<const uint64_t*>(K * np);
        uint64_t* apow = up.take<uint64_t>(K * 2 * np);
        fri::CombineProof* prs = up.take<fri::CombineProof>(K);
        if (!cols || !apow || !prs) return up.host ? GLP_ERR_INVALID : GLP_ERR_OOM;
        tabs[b] = {cols, apow, prs, np};
        // openings the caller did not supply are evaluated here (device work: one proof after the other)
        std::vector<uint64_t> own;
        for (size_t k = 0; k < K; ++k) {
            const glp_fri_batch& bt = batches[k * n_batches + b];
            if (bt.opened_values) continue;
            if (own.empty()) own.resize(K * 2 * np);
            for (size_t j = 0; j < np; ++j) {
                const glp_batch* ob = oracles[b0.oracle_index[j]];
                const size_t col = tree_of(b0.oracle_index[j], k) * ob->n_cols + b0.poly_index[j];
                GLP_TRY(eval_columns(c, ob->coeffs + col * n, n, 1, log_n, bt.point, &own[(k * np + j) * 2]));
            }
        }
        parallel_for(K, [&](size_t k) {
            const glp_fri_batch& bt = batches[k * n_batches + b];
            const uint64_t* ov = bt.opened_values ? bt.opened_values : &own[k * np * 2];
            gl::ext2 ap = {1, 0}, opened = {0, 0};
            for (size_t j = 0; j < np; ++j) {
                const glp_batch* ob = oracles[b0.oracle_index[j]];
                const size_t col = tree_of(b0.oracle_index[j], k) * ob->n_cols + b0.poly_index[j];
                cols[k * np + j] = ob->lde + col * N;  // chunk 0 = values on 7*H, bit-reversed
                apow[(k * np + j) * 2] = gl::canon(ap.a);
                apow[(k * np + j) * 2 + 1] = gl::canon(ap.b);
                opened = gl::ext_add(opened, gl::ext_mul(ap, gl::ext2{ov[2 * j], ov[2 * j + 1]}));
                ap = gl::ext_mul(ap, alpha[k]);
            }
            prs[k].opened[0] = gl::canon(opened.a);
            prs[k].opened[1] = gl::canon(opened.b);
            prs[k].z[0] = gl::canon(bt.point[0]);
            prs[k].z[1] = gl::canon(bt.point[1]);
            prs[k].ashift[0] = gl::canon(ap.a);  // alpha^(n_polys)
            prs[k].ashift[1] = gl::canon(ap.b);
        }, 2);
    }
    GLP_TRY(up.flush(scratch));
    for (size_t b = 0; b < n_batches; ++b) {
        const size_t np = tabs[b].np;
        fri::CombineArgs a;
        a.cols = reinterpret_cast<const uint64_t* const*>(up.dev(tabs[b].cols));
        a.apow = up.dev(tabs[b].apow);
        a.proofs = up.dev(tabs[b].prs);
        a.n_polys = (int)np;
        a.log_n = log_n;
        a.shift = gl::COSET_SHIFT;
        a.first = b == 0;
        a.last = b + 1 == n_batches && !(p->flags & GLP_FLAG_FRI_NO_MUL_X);
        a.K = MASTER_K;
        a.w_lo = c->w_lo_dev;
        a.w_hi = c->w_hi_dev;
        {
            Timed t(c, GLP_K_FRI);
            if (n * K <= ((size_t)1 << 14))  // few points: a wave per point (fri_kernels.h)
                hipLaunchKernelGGL(fri::fri_combine_kernel<true>, dim3((unsigned)((n + fri::THREADS / 64 - 1) / (fri::THREADS / 64)), (unsigned)K),
                                   dim3(fri::THREADS), 0, c->stream, a, fin);
            else
                hipLaunchKernelGGL(fri::fri_combine_kernel<false>, dim3((unsigned)((n + fri::THREADS - 1) / fri::THREADS), (unsigned)K),
                                   dim3(fri::THREADS), 0, c->stream, a, fin);
            HIP_TRY(hipGetLastError());
        }
    }
    // values on 7*H (bit-reversed) -> coefficients: inverse DIT, 1/n, then un-shift by 7^-j
    const uint64_t n_inv = gl::canon(gl::inv((uint64_t)n));
    if (log_n < 9) {
        uint64_t* tmp = nullptr;
        GLP_TRY(scratch.alloc(K * 2 * n, &tmp));
        HIP_TRY(hipMemcpyAsync(tmp, fin, K * 2 * n * 8, hipMemcpyDeviceToDevice, c->stream));
        GLP_TRY(launch_small(c, tmp, n, fin, n, 2 * K, log_n, log_n, 1, 1, gl::canon(gl::inv(gl::COSET_SHIFT)), n_inv, 0, 1));
    } else {
        GLP_TRY(run_dit(c, fin, n, 2 * K, log_n, 1, n_inv));
        GLP_TRY(scale_by_powers(c, fin, n, 2 * K, log_n, gl::canon(gl::inv(gl::COSET_SHIFT)), 1));
    }

    trace.mark("combine launched");
    // ---- commit phase ----
    struct Tree {
        uint64_t* leaves;   // [K][n_leaves][2*arity] row-major
        uint64_t* digests;  // forest of K trees
        size_t n_leaves;
        int arity_bits;
    };
    std::vector<Tree> trees;
    uint64_t* coeffs = fin;  // [K][2][cur_n] limb columns
    int cur_log = log_n;
    uint64_t shift = gl::COSET_SHIFT;
    std::vector<uint64_t*> w(K);
    for (size_t k = 0; k < K; ++k) w[k] = proof_out[k];
    std::vector<uint64_t> caps(K * cap_words);
    uint64_t* betas_dev = nullptr;
    GLP_TRY(scratch.alloc(K * 2 * (p->n_reduction_rounds ? p->n_reduction_rounds : 1), &betas_dev));
    uint64_t* final_mapped = nullptr;  // the final polynomial in pinned host memory, stored there by the last fold
    for (unsigned r = 0; r < p->n_reduction_rounds; ++r) {
        const int ab = (int)p->reduction_arity_bits[r];
        const size_t cur_n = (size_t)1 << cur_log, cur_N = cur_n << rb;
        // values = coset_fft(lde(coeffs), shift), bit-reversed
        uint64_t* vals = nullptr;
        GLP_TRY(scratch.alloc(K * 2 * cur_N, &vals));
        GLP_TRY(lde_from_coeffs(c, coeffs, vals, 2 * K, cur_log, rb, shift));
        Tree t;
        t.arity_bits = ab;
        t.n_leaves = cur_N >> ab;
        GLP_TRY(scratch.alloc(K * 2 * cur_N, &t.leaves));
        GLP_TRY(scratch.alloc(K * digests_len(t.n_leaves, p->cap_height), &t.digests));
        {
            Timed tt(c, GLP_K_FRI);
            hipLaunchKernelGGL(fri::fri_interleave_kernel, dim3((unsigned)((cur_N + fri::THREADS - 1) / fri::THREADS), (unsigned)K),
                               dim3(fri::THREADS), 0, c->stream, vals, cur_N, t.leaves);
            HIP_TRY(hipGetLastError());
        }
        const size_t leaf_len = (size_t)2 << ab;
        GLP_TRY(merkle_build(c, t.leaves, t.n_leaves, leaf_len, leaf_len, 1, p->cap_height, t.digests, caps.data(), K, 2 * cur_N));
        trees.push_back(t);
        uint64_t* betas = arena_alloc_t<uint64_t>(c, K * 2);  // one block per round: the upload is not waited for
        if (!betas) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
        parallel_for(K, [&](size_t k) {
            memcpy(w[k], caps.data() + k * cap_words, cap_words * 8);
            glp_challenger_observe(&chs[k], w[k], cap_words);
            w[k] += cap_words;
            glp_challenger_get_n(&chs[k], betas + 2 * k, 2);
        }, 2);
        // fold the coefficients
        const size_t out_n = cur_n >> ab;
        uint64_t* folded = nullptr;
        // the LAST fold's output is the final polynomial, which only the host reads: the kernel stores it (canonical values) straight
        // into a pinned arena block and the host spins on it (internal.h, mapped_*) instead of a copy + stream synchronisation
        if (r + 1 == p->n_reduction_rounds && mapped_readbacks_enabled() && K * 2 * out_n * 8 <= ((size_t)256 << 10) && out_n > 0) {
            uint64_t* hostp = arena_alloc_t<uint64_t>(c, K * 2 * out_n);
            if (hostp && arena_is_pinned(c, hostp, K * 2 * out_n * 8)) {
                mapped_prefill(hostp, K * 2 * out_n);
                final_mapped = folded = hostp;
            }
        }
        if (!folded) GLP_TRY(scratch.alloc(K * 2 * (out_n ? out_n : 1), &folded));
        uint64_t* bd = betas_dev + (size_t)r * K * 2;
        HIP_TRY(hipMemcpyAsync(bd, betas, K * 16, hipMemcpyHostToDevice, c->stream));
        {
            Timed tt(c, GLP_K_FRI);
            hipLaunchKernelGGL(fri::fri_fold_kernel, dim3((unsigned)((out_n + fri::THREADS - 1) / fri::THREADS), (unsigned)K),
                               dim3(fri::THREADS), 0, c->stream, coeffs, out_n, 1 << ab, bd, folded);
            HIP_TRY(hipGetLastError());
        }
        coeffs = folded;
        cur_log -= ab;
        shift = gl::canon(gl::pow(shift, (uint64_t)1 << ab));
    }
    // final polynomial
    const size_t n_final = (size_t)1 << cur_log;
    std::vector<uint64_t> fp(K * 2 * n_final), final_poly(K * 2 * n_final);
    if (final_mapped) {
        GLP_TRY(mapped_wait(c, final_mapped, K * 2 * n_final));
        memcpy(fp.data(), final_mapped, K * 2 * n_final * 8);
    } else {
        GLP_TRY(read_small(c, fp.data(), coeffs, K * 2 * n_final * 8));
    }
    parallel_for(K, [&](size_t k) {
        uint64_t* dst = final_poly.data() + k * 2 * n_final;
        for (size_t i = 0; i < n_final; ++i) {
            dst[2 * i] = gl::canon(fp[k * 2 * n_final + i]);
            dst[2 * i + 1] = gl::canon(fp[k * 2 * n_final + n_final + i]);
        }
        glp_challenger_observe(&chs[k], dst, 2 * n_final);
    }, 2);

    trace.mark("commit phase + final poly");
    // ---- proof of work ----
    fri::PowArgs2* pa = arena_alloc_t<fri::PowArgs2>(c, K);
    if (!pa) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
    parallel_for(K, [&](size_t k) {
        memset(&pa[k], 0, sizeof pa[k]);
        pa[k].bits = (int)p->proof_of_work_bits;
        uint64_t state[12] = {0};
        if (p->flags & GLP_FLAG_POW_DUPLEX) {  // response = the challenge squeezed right after observing the witness
            for (int i = 0; i < 12; ++i) state[i] = chs[k].sponge_state[i];
            for (uint32_t i = 0; i < chs[k].n_input; ++i) state[i] = chs[k].input_buffer[i];
            pa[k].pos = (int)chs[k].n_input;
            pa[k].out_word = 7;
        } else {  // response = hash_no_pad(challenger.get_hash() || witness).elements[0]
            glp_challenger_get_n(&chs[k], state, 4);
            pa[k].pos = 4;
            pa[k].out_word = 0;
        }
        pow_precompute(state, pa[k].pos, pa[k].base);
    });
    std::vector<unsigned long long> pow_witness(K, ~0ull);
    {
        uint64_t *best_w = nullptr, *pa_dev = nullptr;
        GLP_TRY(scratch.alloc(K, &best_w));
        GLP_TRY(scratch.alloc(K * sizeof(fri::PowArgs2) / 8 + 1, &pa_dev));
        unsigned long long* best = reinterpret_cast<unsigned long long*>(best_w);
        HIP_TRY(hipMemcpyAsync(pa_dev, pa, K * sizeof(fri::PowArgs2), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(best, 0xFF, 8 * K, c->stream));
        // The candidates of every unsettled proof are scanned in short launches issued back to back (fri_pow_kernel skips a
        // proof once an earlier launch has settled it), about 2^21 lanes per launch (eight waves deep: the tail of a launch, where
        // the last waves run alone, is an eighth of it) split evenly over the unsettled proofs; the
        // host looks at the result at intervals and deals the lanes again to the proofs that are left, so the unlucky tail of a
        // large batch is ground by the whole chip and not by one workgroup.
        const uint32_t pb = p->proof_of_work_bits;
        const uint64_t expect = (uint64_t)1 << (pb < 40 ? pb : 40);
        std::vector<uint32_t> todo(K);
        for (size_t k = 0; k < K; ++k) todo[k] = (uint32_t)k;
        uint32_t* ids_dev = nullptr;
        {
            uint64_t* raw = nullptr;
            GLP_TRY(scratch.alloc(K / 2 + 1, &raw));
            ids_dev = reinterpret_cast<uint32_t*>(raw);
        }
        unsigned long long* got = arena_alloc_t<unsigned long long>(c, K);
        if (!got) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
        uint64_t start = 0;
        for (int phase = 0; !todo.empty(); ++phase) {
            const size_t Kp = todo.size();
            uint64_t lanes = ((uint64_t)1 << 21) / Kp;
            if (lanes > 2 * expect) lanes = 2 * expect;
            // A FEW unsettled proofs (2 x the expected tries of all of them fit one 2^21-lane launch): each gets what a single proof
            // gets — one launch over 2 x expect candidates (86 % settle in it) — instead of looks every expect / 2: a batch of eight
            // signature-sized proofs spent 0.51 of its 1.59 ms in five or six launch + look rounds (round 6).  Many proofs: throughput
            // matters, a settled proof's lanes finish their launch, so the looks stay frequent and that waste small.
            const bool few = K > 1 && Kp * 2 * expect <= ((uint64_t)1 << 21);
            if (K > 1 && !few && lanes > (expect + 1) / 2) lanes = (expect + 1) / 2;
            if (lanes > ((uint64_t)1 << 22)) lanes = (uint64_t)1 << 22;
            lanes -= lanes % fri::THREADS;  // whole workgroups: a launch scans exactly `lanes` consecutive candidates
            if (lanes < fri::THREADS) lanes = fri::THREADS;
            // K = 1: one look after 2x the expected tries (hit rate 86 %); a batch is looked at every expect / 2 candidates, so
            // that the lanes of the proofs settled meanwhile go to the others
            uint64_t span = (K == 1 || few) ? (phase ? 4 : 2) * expect : (expect + 1) / 2;
            if (span < lanes) span = lanes;
            if (span > ((uint64_t)1 << 26)) span = (uint64_t)1 << 26;  // at most 2^26 / lanes launches between two looks
            // "Proof of work failed. This is highly unlikely!": 1024 x the expected tries without a witness (e^-1024), or the
            // whole field scanned
            if (start >= gl::P - span || (pb <= 40 && start >= (expect << 10))) return GLP_ERR_INVALID;
            const uint32_t* ids = nullptr;
            if (phase) {
                uint32_t* ids_h = arena_alloc_t<uint32_t>(c, Kp);
                if (!ids_h) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
                memcpy(ids_h, todo.data(), Kp * 4);
                HIP_TRY(hipMemcpyAsync(ids_dev, ids_h, Kp * 4, hipMemcpyHostToDevice, c->stream));
                ids = ids_dev;
            }
            {
                Timed t(c, GLP_K_FRI);
                uint64_t off = 0;
                for (; off < span; off += lanes)
                    hipLaunchKernelGGL(fri::fri_pow_kernel, dim3((unsigned)(lanes / fri::THREADS), (unsigned)Kp), dim3(fri::THREADS), 0,
                                       c->stream, reinterpret_cast<const fri::PowArgs2*>(pa_dev), ids, start + off, lanes, best);
                HIP_TRY(hipGetLastError());
                span = off;  // whole launches: what was actually scanned
            }
            start += span;
            HIP_TRY(hipMemcpyAsync(got, best, 8 * K, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            todo.clear();
            for (size_t k = 0; k < K; ++k)
                if (got[k] == ~0ull)
                    todo.push_back((uint32_t)k);
                else
                    pow_witness[k] = got[k];
        }
    }
    if (p->flags & GLP_FLAG_POW_DUPLEX)
        for (size_t k = 0; k < K; ++k) {
            const uint64_t wv = pow_witness[k];
            glp_challenger_observe(&chs[k], &wv, 1);
            const uint64_t resp = glp_challenger_get(&chs[k]);
            if (p->proof_of_work_bits && (resp >> (64 - p->proof_of_work_bits)) != 0) return GLP_ERR_INVALID;  // same computation
        }

    trace.mark("proof of work");
    // ---- query phase ----
    const uint32_t nq = p->num_query_rounds;
    std::vector<uint64_t> idx(K * nq);
    std::vector<uint32_t> tr(K * nq);
    parallel_for(K, [&](size_t k) {
        for (uint32_t q = 0; q < nq; ++q) {
            idx[k * nq + q] = glp_challenger_get(&chs[k]) % N;
            tr[k * nq + q] = (uint32_t)k;
        }
    }, 2);
    // every oracle's and every round tree's rows + paths: all gathers are launched, then ONE wait for the read-backs
    const int plen0 = (int)(lde_bits - p->cap_height);
    const size_t nqk = K * nq, n_sets = n_oracles + trees.size();
    uint64_t* idx_dev = nullptr;  // [n_sets][nqk] packed (tree << 40 | leaf)
    GLP_TRY(scratch.alloc(n_sets * nqk, &idx_dev));
    uint64_t* packed = arena_alloc_t<uint64_t>(c, n_sets * nqk);
    if (!packed) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
    std::vector<uint64_t*> rows(n_oracles), paths(n_oracles), trows(trees.size()), tpaths(trees.size());
    std::vector<uint64_t> tidx = idx;
    for (size_t o = 0; o < n_oracles; ++o)
        for (size_t i = 0; i < nqk; ++i) packed[o * nqk + i] = idx[i] | ((uint64_t)(oracles[o]->n_trees == 1 ? 0 : tr[i]) << 40);
    for (size_t r = 0; r < trees.size(); ++r) {
        for (auto& x : tidx) x >>= trees[r].arity_bits;
        for (size_t i = 0; i < nqk; ++i) packed[(n_oracles + r) * nqk + i] = tidx[i] | ((uint64_t)tr[i] << 40);
    }
    HIP_TRY(hipMemcpyAsync(idx_dev, packed, n_sets * nqk * 8, hipMemcpyHostToDevice, c->stream));
    // all rows and paths land in ONE device block and come back with ONE copy (a copy per set was twelve blits and their
    // launch gaps per proof: ~80 us of a 2^4-row proof's 1.1 ms)
    struct Set {
        const uint64_t *leaves, *digests;
        size_t row_stride, col_stride, n_cols, n_leaves, n_trees, tree_stride, rows_off, paths_off;
        int pl;
    };
    std::vector<Set> sets;
    size_t q_words = 0;
    auto add_set = [&](const uint64_t* leaves, size_t row_stride, size_t col_stride, size_t n_cols, const uint64_t* digests,
                       size_t n_leaves, int pl, size_t n_trees, size_t tree_stride) {
        Set st{leaves, digests, row_stride, col_stride, n_cols, n_leaves, n_trees, tree_stride, q_words, 0, pl};
        q_words += nqk * n_cols;
        st.paths_off = q_words;
        q_words += nqk * (pl > 0 ? (size_t)pl : 1) * 4;
        sets.push_back(st);
    };
    for (size_t o = 0; o < n_oracles; ++o) {
        const glp_batch* ob = oracles[o];
        add_set(ob->lde, 1, N, leaf_width(ob), ob->digests, N, plen0, ob->n_trees, ob->n_cols * N);  // the whole leaf: a blinded oracle's salts too
    }
    {
        unsigned tb = lde_bits;
        for (size_t r = 0; r < trees.size(); ++r) {
            const int ab = trees[r].arity_bits;
            const size_t leaf_len = (size_t)2 << ab;
            add_set(trees[r].leaves, leaf_len, 1, leaf_len, trees[r].digests, trees[r].n_leaves, (int)(tb - ab - p->cap_height), K,
                    trees[r].n_leaves * leaf_len);
            tb -= ab;
        }
    }
    uint64_t* q_dev = nullptr;
    GLP_TRY(scratch.alloc(q_words, &q_dev));
    uint64_t* q_host = arena_alloc_t<uint64_t>(c, q_words);
    if (!q_host) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
    for (size_t si = 0; si < sets.size(); ++si) {
        const Set& st = sets[si];
        GLP_TRY(gather_rows_launch(c, st.leaves, st.row_stride, st.col_stride, st.n_cols, st.digests, st.n_leaves, st.pl,
                                   idx_dev + si * nqk, nqk, q_dev + st.rows_off, st.pl > 0 ? q_dev + st.paths_off : nullptr, st.n_trees,
                                   st.tree_stride));
        (si < n_oracles ? rows[si] : trows[si - n_oracles]) = q_host + st.rows_off;
        (si < n_oracles ? paths[si] : tpaths[si - n_oracles]) = q_host + st.paths_off;
    }
    HIP_TRY(hipMemcpyAsync(q_host, q_dev, q_words * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    trace.mark("query gathers");
    parallel_for(K, [&](size_t k) {
        uint64_t* wp = w[k];
        for (uint32_t q = 0; q < nq; ++q) {
            const size_t qi = k * nq + q;
            for (size_t o = 0; o < n_oracles; ++o) {
                const size_t nc = leaf_width(oracles[o]);
                memcpy(wp, rows[o] + qi * nc, nc * 8);
                wp += nc;
                memcpy(wp, paths[o] + qi * plen0 * 4, (size_t)plen0 * 32);
                wp += (size_t)plen0 * 4;
            }
            unsigned bits = lde_bits;
            for (size_t r = 0; r < trees.size(); ++r) {
                const int ab = trees[r].arity_bits;
                const size_t leaf_len = (size_t)2 << ab;
                const int pl = (int)(bits - ab - p->cap_height);
                memcpy(wp, trows[r] + qi * leaf_len, leaf_len * 8);
                wp += leaf_len;
                memcpy(wp, tpaths[r] + qi * pl * 4, (size_t)pl * 32);
                wp += (size_t)pl * 4;
                bits -= ab;
            }
        }
        memcpy(wp, final_poly.data() + k * 2 * n_final, 2 * n_final * 8);
        wp += 2 * n_final;
        *wp++ = pow_witness[k];
    }, 2);
    trace.mark("proof assembly");
    return GLP_OK;
}

}  // namespace glpi

extern "C" {

int32_t glp_fri_prove(glp_ctx* c, const glp_batch* const* oracles, size_t n_oracles, const glp_fri_batch* batches,
                      size_t n_batches, const glp_fri_params* p, glp_challenger* ch, uint64_t* proof_out) {
    GLP_ON_DEVICE_OF(c);
    if (!ch || !proof_out) return GLP_ERR_INVALID;
    if (!c) return GLP_ERR_INVALID;
    for (size_t o = 0; oracles && o < n_oracles; ++o)
        if (oracles[o] && (oracles[o]->n_trees != 1 || oracles[o]->ctx->device != c->device)) return GLP_ERR_INVALID;
    uint64_t* outs[1] = {proof_out};
    GLP_TRY(arena_reset(c));
    return fri_prove_forest(c, oracles, n_oracles, 1, batches, n_batches, p, ch, outs);
}

}  // extern "C"
