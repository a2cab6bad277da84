// plonk.hip — the PLONK layer of the prover on the GPU: build-time commitment, partial products / Z, quotient
// polynomials, openings, and the whole `prove` sequence with its Fiat-Shamir transcript.
//
// Mirrors plonky2 @ f99ed9c src/plonk/prover.rs::prove (step order, transcript order, polynomial order inside each
// oracle) and the commitment step of src/plonk/circuit_builder.rs::build; the reference calls them as
// CircuitData::prove / CircuitBuilder::build (reference src/rollup/circuits/mod.rs:605,1247,
// src/zkdsa/circuits/mod.rs:37,326, src/transaction/circuits/mod.rs:453).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

// This translation unit keeps gl::mul's select form of the final correction (gl_field.h): the fused quotient kernel lives at 128
// registers and four waves per SIMD, and the signed-count form — faster everywhere else — costs it a spill and 0.1 ms per 2^20-row
// launch (profiles/r06_mul_tail_ab.txt).  Same values either way.
#define GL_MUL_SELECT_TAIL 1

#include "internal.h"
#include "plonk_kernels.h"
#include "quotient_fused.h"

using namespace glpi;

struct glp_circuit_data {
    glp_ctx* ctx;
    glp_circuit common;
    std::vector<glp_gate> gates;
    std::vector<uint64_t> k_is;
    glp_batch* cs;            // constants_sigmas commitment
    uint64_t* sigma_values;   // [num_routed][n] on the device (values on H)
    // the fused quotient kernel's task list (round-6 plan: a function of the circuit only), made by the first proof and kept:
    // planning is a few hundred simulated deals, ~0.2 ms — a tenth of a 2^11-row proof
    mutable std::mutex fq_mu;
    mutable bool fq_planned = false;
    mutable std::vector<plonk::FqTask> fq_plan;  // in deal order, roles assigned; empty: more tasks than the kernel's table
    mutable int fq_poseidon_gate = -1;
};

namespace {

void host_hash_no_pad(const uint64_t* in, size_t len, uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {  // no chunk, no permutation: the hash of nothing is zero
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; ++i) st[i] = gl::canon(in[off + i]);
        poseidon::permute<poseidon::MDS_INT>(st);
    }
    for (int i = 0; i < 4; ++i) out[i] = gl::canon(st[i]);
}

struct BatchGuard {  // frees a batch on scope exit unless released
    glp_batch* b = nullptr;
    ~BatchGuard() {
        if (b) glp_batch_free(b);
    }
};

int32_t check_common(const glp_circuit* cc) {
    if (!cc || !cc->gates || !cc->k_is) return GLP_ERR_INVALID;
    if (cc->num_gates == 0 || cc->num_gates > (uint32_t)plonk::MAX_GATES) return GLP_ERR_UNSUPPORTED;
    if (cc->num_challenges == 0 || cc->num_challenges > 2) return GLP_ERR_UNSUPPORTED;
    // sizes first: everything below shifts by them, and the drivers size host arrays from them before any device work
    if (cc->fri.rate_bits > 8 || cc->degree_bits > (uint32_t)MASTER_K || cc->degree_bits + cc->fri.rate_bits > (uint32_t)MASTER_K ||
        cc->fri.cap_height > cc->degree_bits + cc->fri.rate_bits)
        return GLP_ERR_INVALID;
    if (cc->fri.rate_bits > 3) return GLP_ERR_UNSUPPORTED;  // the quotient kernels hold Z_H on the coset as 2^rate_bits <= 8 values
    if ((1u << cc->fri.rate_bits) != cc->quotient_degree_factor) return GLP_ERR_UNSUPPORTED;
    if (cc->fri.degree_bits != cc->degree_bits) return GLP_ERR_INVALID;
    if (cc->num_wires > 4096 || cc->num_constants > 4096 || cc->num_gate_constraints > (1u << 20)) return GLP_ERR_INVALID;
    if (cc->fri.num_query_rounds == 0 || cc->fri.num_query_rounds > 4096 || cc->fri.proof_of_work_bits > 32 ||
        cc->fri.n_reduction_rounds > 16)
        return GLP_ERR_INVALID;
    {
        uint32_t total = 0;
        for (uint32_t r = 0; r < cc->fri.n_reduction_rounds; ++r) {
            const uint32_t ab = cc->fri.reduction_arity_bits[r];
            if (ab == 0 || ab > 6) return GLP_ERR_INVALID;
            total += ab;
            // every round's tree keeps at least a cap's worth of leaves (plonky2 asserts the same when it builds the tree)
            if (total > cc->degree_bits || cc->degree_bits + cc->fri.rate_bits - total < cc->fri.cap_height) return GLP_ERR_INVALID;
        }
    }
    if (cc->num_routed_wires == 0 || cc->num_routed_wires > cc->num_wires) return GLP_ERR_INVALID;
    if (cc->num_selectors == 0 || cc->num_selectors > cc->num_constants) return GLP_ERR_INVALID;
    if (cc->quotient_degree_factor == 0) return GLP_ERR_INVALID;
    const uint32_t n_chunks = (cc->num_routed_wires + cc->quotient_degree_factor - 1) / cc->quotient_degree_factor;
    if (cc->num_partial_products + 1 != n_chunks) return GLP_ERR_INVALID;
    if (n_chunks > (uint32_t)plonk::MAX_PP_CHUNKS) return GLP_ERR_UNSUPPORTED;  // chunk quotients a lane inverts together
    const uint32_t gate_constants = cc->num_constants - cc->num_selectors;  // config.num_constants
    for (uint32_t g = 0; g < cc->num_gates; ++g) {
        const glp_gate& gt = cc->gates[g];
        if (gt.id > GLP_GATE_COSET_INTERPOLATION) return GLP_ERR_UNSUPPORTED;
        if (gt.selector_index >= cc->num_selectors) return GLP_ERR_INVALID;
        if (gt.group_start > g || gt.group_end <= g || gt.group_end > cc->num_gates) return GLP_ERR_INVALID;
        // parameter ranges the quotient kernel is written for; every wire and every constant a gate touches must exist
        // (the kernels index the wires / constants_sigmas device arrays with these, unchecked)
        uint32_t last_wire = 0, last_constant = 0, n_constraints = 0;
        if (gt.id == GLP_GATE_CONSTANT) {
            last_wire = last_constant = n_constraints = gt.param;
        } else if (gt.id == GLP_GATE_PUBLIC_INPUT) {
            last_wire = n_constraints = 4;
        } else if (gt.id == GLP_GATE_ARITHMETIC) {
            if (gt.param > (1u << 20)) return GLP_ERR_INVALID;
            last_wire = 4 * gt.param;
            last_constant = 2;
            n_constraints = gt.param;
        } else if (gt.id == GLP_GATE_ARITHMETIC_EXT || gt.id == GLP_GATE_MUL_EXT) {
            if (gt.param > (1u << 20)) return GLP_ERR_INVALID;
            last_wire = (gt.id == GLP_GATE_ARITHMETIC_EXT ? 8 : 6) * gt.param;
            last_constant = 2;
            n_constraints = 2 * gt.param;
        } else if (gt.id == GLP_GATE_POSEIDON) {
            last_wire = 135;
            n_constraints = 123;
        } else if (gt.id == GLP_GATE_BASE_SUM) {
            if (gt.param == 0 || gt.param > 64) return GLP_ERR_INVALID;
            last_wire = 1 + gt.param;
            n_constraints = 1 + gt.param;
        } else if (gt.id == GLP_GATE_RANDOM_ACCESS) {
            const uint32_t bits = gt.param & 0xFF, copies = (gt.param >> 8) & 0xFF, extra = (gt.param >> 16) & 0xFF;
            if (bits == 0 || copies == 0) return GLP_ERR_INVALID;
            if (bits > 4) return GLP_ERR_UNSUPPORTED;
            last_constant = extra;
            last_wire = (2 + (1u << bits)) * copies + extra + copies * bits;
            n_constraints = copies * (bits + 2) + extra;
        } else if (gt.id == GLP_GATE_REDUCING) {
            if (gt.param == 0 || gt.param > (1u << 20)) return GLP_ERR_INVALID;
            last_wire = 6 + gt.param + 2 * (gt.param - 1);
            n_constraints = 2 * gt.param;
        } else if (gt.id == GLP_GATE_REDUCING_EXT) {
            if (gt.param == 0 || gt.param > (1u << 20)) return GLP_ERR_INVALID;
            last_wire = 6 + 2 * gt.param + 2 * (gt.param - 1);
            n_constraints = 2 * gt.param;
        } else if (gt.id == GLP_GATE_EXPONENTIATION) {
            if (gt.param == 0 || gt.param > (1u << 20)) return GLP_ERR_INVALID;
            last_wire = 2 + 2 * gt.param;
            n_constraints = gt.param + 1;
        } else if (gt.id == GLP_GATE_POSEIDON_MDS) {
            last_wire = 48;
            n_constraints = 24;
        } else if (gt.id == GLP_GATE_COSET_INTERPOLATION) {
            const uint32_t bits = gt.param & 0xFF, degree = (gt.param >> 8) & 0xFF;
            if (bits == 0 || degree < 2) return GLP_ERR_INVALID;
            if (bits > 4) return GLP_ERR_UNSUPPORTED;
            const uint32_t np = 1u << bits, ni = (np - 2) / (degree - 1);
            last_wire = 1 + 2 * np + 4 + 2 * (2 * ni + 1);
            n_constraints = 2 + 4 * ni + 2;
        }
        if (last_wire > cc->num_wires || last_constant > gate_constants) return GLP_ERR_INVALID;
        // the alpha-power table holds num_gate_constraints entries per challenge for the gate terms
        if (n_constraints > cc->num_gate_constraints) return GLP_ERR_INVALID;
    }
    return GLP_OK;
}

size_t n_terms_of(const glp_circuit& cc) {
    return cc.num_challenges + cc.num_challenges * (cc.num_partial_products + 1) + cc.num_gate_constraints;
}

// the per-proof Fiat-Shamir values of the PLONK layer (host side)
struct ProofChal {
    uint64_t betas[2], gammas[2], alphas[2];
    uint64_t pi_hash[4];
};

// beta_ch * k_j tables of K proofs: [K][nch][num_routed] (host side; uploaded with the step's other tables)
void fill_bk(const glp_circuit_data* d, size_t K, const ProofChal* ch, uint64_t* bk) {
    const glp_circuit& cc = d->common;
    const size_t per = (size_t)cc.num_challenges * cc.num_routed_wires;
    parallel_for(K, [&](size_t k) {
        for (uint32_t cidx = 0; cidx < cc.num_challenges; ++cidx)
            for (uint32_t j = 0; j < cc.num_routed_wires; ++j)
                bk[k * per + cidx * cc.num_routed_wires + j] = gl::canon(gl::mul(gl::canon(ch[k].betas[cidx]), d->k_is[j]));
    });
}

// all_wires_permutation_partial_products for K proofs: proof k's witness at wires_dev + k * proof_stride, out [K][nzs][n]
int32_t partial_products(glp_ctx* c, const glp_circuit_data* d, const uint64_t* wires_dev, size_t stride, size_t proof_stride,
                         size_t K, const ProofChal* ch, uint64_t* out_dev) {
    const glp_circuit& cc = d->common;
    const size_t n = (size_t)1 << cc.degree_bits;
    if (stride != n) return GLP_ERR_UNSUPPORTED;
    if (K == 0 || K > 65535) return GLP_ERR_INVALID;
    const int n_chunks = (int)cc.num_partial_products + 1, nch = (int)cc.num_challenges;
    Scratch sc(c);
    uint64_t *bk = nullptr, *quot = nullptr, *tot = nullptr, *bp = nullptr, *chal = nullptr;
    const size_t bk_words = K * (size_t)nch * cc.num_routed_wires;
    StagedUpload up(c, StagedUpload::need(K * 32) + StagedUpload::need(bk_words * 8));
    uint64_t* chal_h = up.take<uint64_t>(K * 4);
    uint64_t* bk_h = up.take<uint64_t>(bk_words);
    if (!chal_h || !bk_h) return up.host ? GLP_ERR_INVALID : GLP_ERR_OOM;
    memset(chal_h, 0, K * 32);
    for (size_t k = 0; k < K; ++k)
        for (int i = 0; i < nch; ++i) {
            chal_h[k * 4 + i] = gl::canon(ch[k].betas[i]);
            chal_h[k * 4 + 2 + i] = gl::canon(ch[k].gammas[i]);
        }
    fill_bk(d, K, ch, bk_h);
    GLP_TRY(up.flush(sc));
    chal = up.dev(chal_h);
    bk = up.dev(bk_h);
    GLP_TRY(sc.alloc(K * nch * n_chunks * n, &quot));
    GLP_TRY(sc.alloc(K * nch * n, &tot));
    const size_t rows_per_block = (size_t)plonk::THREADS * plonk::SCAN_PER;
    const int n_blocks = (int)((n + rows_per_block - 1) / rows_per_block);
    GLP_TRY(sc.alloc(K * nch * n_blocks, &bp));
    plonk::PPArgs a;
    a.wires = wires_dev;
    a.wires_proof_stride = proof_stride;
    a.sigmas = d->sigma_values;
    a.bk = bk;
    a.chal = chal;
    a.quot = quot;
    a.tot = tot;
    a.log_n = (int)cc.degree_bits;
    a.num_routed = (int)cc.num_routed_wires;
    a.chunk = (int)cc.quotient_degree_factor;
    a.n_chunks = n_chunks;
    a.nch = nch;
    a.K = MASTER_K;
    a.w_lo = c->w_lo_dev;
    a.w_hi = c->w_hi_dev;
    Timed t(c, GLP_K_PLONK);
    const size_t pp_rows = plonk::THREADS / nch;  // rows per workgroup: both challenges of a row side by side (check_common: nch is 1 or 2)
    if (a.n_chunks <= 10)
        hipLaunchKernelGGL((plonk::pp_quotients_kernel<10>), dim3((unsigned)((n + pp_rows - 1) / pp_rows), 1u, (unsigned)K),
                       dim3(plonk::THREADS), 0, c->stream, a);
    else
        hipLaunchKernelGGL((plonk::pp_quotients_kernel<plonk::MAX_PP_CHUNKS>), dim3((unsigned)((n + pp_rows - 1) / pp_rows), 1u, (unsigned)K),
                       dim3(plonk::THREADS), 0, c->stream, a);
    // the scans treat (proof, challenge) as one flat index: tot is [K][nch][n], the block products [K][nch][n_blocks]
    hipLaunchKernelGGL(plonk::pp_block_products_kernel, dim3((unsigned)n_blocks, (unsigned)(nch * K)), dim3(plonk::THREADS), 0,
                       c->stream, tot, n, bp);
    hipLaunchKernelGGL(plonk::pp_scan_blocks_kernel, dim3((unsigned)(nch * K)), dim3(plonk::THREADS), 0, c->stream, bp, n_blocks);
    hipLaunchKernelGGL(plonk::pp_finalize_kernel, dim3((unsigned)n_blocks, (unsigned)nch, (unsigned)K), dim3(plonk::THREADS), 0,
                       c->stream, tot, quot, bp, n, n_chunks, nch, out_dev);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

// compute_quotient_polys for K proofs: wires / zs are forests of K trees, out_dev [K][nch * quotient_degree_factor][n]
int32_t quotient(glp_ctx* c, const glp_circuit_data* d, const glp_batch* wires, const glp_batch* zs, size_t K,
                 const ProofChal* ch, uint64_t* out_dev) {
    const glp_circuit& cc = d->common;
    if (K == 0 || K > 65535 || wires->n_trees != K || zs->n_trees != K) return GLP_ERR_INVALID;
    const int nch = (int)cc.num_challenges, rb = (int)cc.fri.rate_bits;
    const int log_n = (int)cc.degree_bits, lde_bits = log_n + rb;
    const size_t n = (size_t)1 << log_n, N = n << rb;
    if (wires->log_n != cc.degree_bits || zs->log_n != cc.degree_bits || wires->rate_bits != cc.fri.rate_bits ||
        zs->rate_bits != cc.fri.rate_bits || wires->n_cols != cc.num_wires ||
        zs->n_cols != cc.num_challenges * (1 + cc.num_partial_products))
        return GLP_ERR_INVALID;
    Scratch sc(c);
    const size_t n_terms = n_terms_of(cc);
    // per proof: two rows of alpha powers (row 1 = 0 if nch = 1: the fused kernel always combines both) and its challenges
    const size_t bk_words = K * (size_t)nch * cc.num_routed_wires;
    StagedUpload up(c, StagedUpload::need(K * 2 * n_terms * 8) + StagedUpload::need(K * sizeof(plonk::QuotProof)) +
                           StagedUpload::need(bk_words * 8));
    uint64_t* apow = up.take<uint64_t>(K * 2 * n_terms);
    plonk::QuotProof* prs = up.take<plonk::QuotProof>(K);
    uint64_t* bk_h = up.take<uint64_t>(bk_words);
    if (!apow || !prs || !bk_h) return up.host ? GLP_ERR_INVALID : GLP_ERR_OOM;
    memset(apow, 0, K * 2 * n_terms * 8);
    for (size_t k = 0; k < K; ++k) {
        memset(&prs[k], 0, sizeof prs[k]);
        for (int cidx = 0; cidx < nch; ++cidx) {
            uint64_t x = 1, al = gl::canon(ch[k].alphas[cidx]);
            for (size_t t = 0; t < n_terms; ++t) {
                apow[(k * 2 + cidx) * n_terms + t] = gl::canon(x);
                x = gl::mul(x, al);
            }
            prs[k].betas[cidx] = gl::canon(ch[k].betas[cidx]);
            prs[k].gammas[cidx] = gl::canon(ch[k].gammas[cidx]);
        }
        for (int i = 0; i < 4; ++i) prs[k].pi_hash[i] = gl::canon(ch[k].pi_hash[i]);
    }
    fill_bk(d, K, ch, bk_h);
    GLP_TRY(up.flush(sc));
    uint64_t *apow_dev = up.dev(apow), *bk = up.dev(bk_h), *vals = nullptr;
    plonk::QuotProof* prs_dev = up.dev(prs);
    GLP_TRY(sc.alloc(K * nch * N, &vals));
    plonk::QuotArgs a;
    memset(&a, 0, sizeof a);
    a.cs = d->cs->lde;
    a.w = wires->lde;
    a.z = zs->lde;
    a.out = vals;
    a.apow = apow_dev;
    a.bk = bk;
    a.proofs = reinterpret_cast<const plonk::QuotProof*>(prs_dev);
    a.n_terms = (int)n_terms;
    a.log_n = log_n;
    a.lde_bits = lde_bits;
    a.num_constants = (int)cc.num_constants;
    a.num_selectors = (int)cc.num_selectors;
    a.num_routed = (int)cc.num_routed_wires;
    a.num_wires = (int)cc.num_wires;
    a.nch = nch;
    a.npp = (int)cc.num_partial_products;
    a.chunk = (int)cc.quotient_degree_factor;
    a.num_gates = (int)cc.num_gates;
    for (uint32_t g = 0; g < cc.num_gates; ++g) {
        const glp_gate& s = d->gates[g];
        a.gates[g] = plonk::Gate{s.id, s.selector_index, s.group_start, s.group_end, s.param};
    }
    // Z_H(x) = x^n - 1 on the coset 7 * <w_N>: x^n = 7^n * w_{2^rb}^(i mod 2^rb)
    const uint64_t g_n = gl::pow(gl::COSET_SHIFT, n);
    for (int r = 0; r < (1 << rb) && r < 8; ++r) {
        uint64_t wr = host_root_pow(c, (uint64_t)r << (MASTER_K - rb));
        a.zh[r] = gl::canon(gl::sub(gl::mul(g_n, wr), 1));
        a.zh_inv[r] = gl::canon(gl::inv(a.zh[r]));
    }
    a.n_field = (uint64_t)n;
    for (uint32_t g = 0; g < cc.num_gates; ++g)
        if (d->gates[g].id == GLP_GATE_COSET_INTERPOLATION) {  // subgroup points g^i and their barycentric weights
            const int bits = (int)(d->gates[g].param & 0xFF), np = 1 << bits;
            for (int i = 0; i < np; ++i) a.interp_dom[i] = gl::canon(host_root_pow(c, (uint64_t)i << (MASTER_K - bits)));
            for (int i = 0; i < np; ++i) {
                uint64_t pr = 1;
                for (int j = 0; j < np; ++j)
                    if (j != i) pr = gl::mul(pr, gl::sub(a.interp_dom[i], a.interp_dom[j]));
                a.interp_wts[i] = gl::canon(gl::inv(pr));
            }
        }
    a.K = MASTER_K;
    a.w_lo = c->w_lo_dev;
    a.w_hi = c->w_hi_dev;
    // Which evaluator: the LDS-tiled fused kernel reads every column once (16 GB instead of 91 GB per 2^20-row launch of the
    // 14-gate circuit) and is the faster one for circuits with the recursive gate set (23.1 vs 25.3 ms); for circuits with few
    // gates the per-point kernels re-read little and keep more waves in flight (10.6 vs 12.6 ms for the 4-gate circuit) — once
    // there are enough points to fill the chip with one lane per point: below ~2^19 points in the launch their time is the
    // latency of ONE lane's constraint stream (≈ 200 µs whatever the size), while the fused kernel spreads a point's constraints
    // over eight waves (2^4 rows: 1.32 -> 1.16 ms per proof, 2^12: 3.46 -> 3.28, 2^16: even).
    // GLP_QUOT_FUSED = 0 / 1 forces one or the other (both are parity-tested).
    const int want_fused = [] {  // read per call: the parity tests run both evaluators in one process
        const char* e = getenv("GLP_QUOT_FUSED");
        return e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    const size_t fused_lds = plonk::fq_lds_bytes((int)cc.num_wires, (int)cc.num_constants);
    size_t n_poseidon_gates = 0;
    for (uint32_t g = 0; g < cc.num_gates; ++g) n_poseidon_gates += d->gates[g].id == GLP_GATE_POSEIDON;
    // the fused kernel holds one PoseidonGate (plonky2 dedups gates, so real circuits have one); anything else takes the
    // per-point evaluator, which has no such limit
    bool fused = (want_fused == 1 || (want_fused == -1 && (cc.num_gates >= 9 || K * N <= ((size_t)1 << 19)))) && lde_bits >= 6 &&
                 fused_lds <= 80 * 1024 && n_poseidon_gates <= 1;
    // One LDS-tiled launch: every wire column is read once per point (quotient_fused.h).  The constraint tasks are dealt
    // to the waves of a workgroup by estimated cost (VALU instructions per wave, longest first onto the least loaded).
    plonk::FusedArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.q = a;
    fa.poseidon_gate = -1;
    // tasks: (kind, gate, iteration range) with estimated VALU instructions per wave; long loops are cut into ranges of at
    // most ~`piece` instructions so that the longest-first deal onto the least loaded wave ends up level
    struct Cand { double cost; plonk::FqTask t; };
    std::vector<Cand> cand;
    // The cost model and the uniform cut of rounds 3-5 are the default.  GLP_FQ_PLAN=1: the round-6 plan below — a CLOSED EXPERIMENT
    // (profiles/r06_quotient_plan.md): measured per-task cycles, the pieces chosen by simulating the deal.  The simulation promised a
    // makespan of 167 k cycles against 175 k; the kernel ran 18.78-18.86 ms against 18.63-18.65.  A task's duration is not a property of
    // the task: it depends on what the other three waves of its SIMD are issuing (the ReducingGate loop took 134 k cycles beside one
    // set of neighbours and 110 k beside another), so a deal planned on fixed costs plans on noise, and the four extra tasks cost their
    // prologues (+3.7 % summed task cycles).
    static const bool legacy_plan = !(getenv("GLP_FQ_PLAN") && atoi(getenv("GLP_FQ_PLAN")) == 1) || getenv("GLP_FQ_PIECE") ||
                                    getenv("GLP_FQ_CAL") || getenv("GLP_FQ_PIECE_SMALL");
    bool have_plan = false;
    if (fused && !legacy_plan) {
        std::lock_guard<std::mutex> lk(d->fq_mu);
        if (d->fq_planned) {
            have_plan = true;
            if (d->fq_plan.empty()) fused = false;
            fa.poseidon_gate = d->fq_poseidon_gate;
            for (const plonk::FqTask& t : d->fq_plan) cand.push_back({0.0, t});
        }
    }
    if (fused && !legacy_plan && !have_plan) {
        // (opt-in, see above)  The deal planned on MEASURED costs.  GLP_FQ_DEBUG's per-task cycle counts (2^20 rows, 14 gates, four waves per
        // SIMD: profiles/r06_quotient_plan.md) give a cost per iteration of every loop and ~5 k cycles per task for its prologue and
        // epilogue; the round-3 estimates in "instructions" were off by up to 2.3 x between gates (the ReducingGate loop: 3.0 k cycles per
        // coefficient, the ExponentiationGate loop 1.66 k, both estimated at 105), which the dynamic deal absorbs only as far as the
        // ORDER is right: with the tasks listed by estimated cost the last wave of a workgroup finished at 175 k cycles where the
        // mean is 163 k.  Now the tasks are listed by measured cost, and how many pieces each divisible loop is cut into is chosen
        // by simulating the deal (longest first onto the first free wave) and descending on the makespan.
        struct Loop { int kind, gate, n_iter; double unit, first; };
        std::vector<Loop> loops;
        loops.push_back({plonk::FQ_KIND_PERM, 0, (int)cc.num_partial_products + 1, cc.quotient_degree_factor * nch * 1750.0, 16000.0});
        for (uint32_t g = 0; g < cc.num_gates; ++g) {
            const glp_gate& gt = d->gates[g];
            const int pr = (int)gt.param;
            switch (gt.id) {
                case GLP_GATE_POSEIDON:
                    fa.poseidon_gate = (int)g;
                    loops.push_back({plonk::FQ_KIND_POS_FULL, (int)g, 7, 16500.0, 0});
                    loops.push_back({plonk::FQ_KIND_POS_PARTIAL, (int)g, 1, 114800.0, 0});
                    break;
                case GLP_GATE_CONSTANT: loops.push_back({plonk::FQ_KIND_GATE, (int)g, pr, 500.0, 0}); break;
                case GLP_GATE_PUBLIC_INPUT: loops.push_back({plonk::FQ_KIND_GATE, (int)g, 4, 450.0, 0}); break;
                case GLP_GATE_ARITHMETIC: loops.push_back({plonk::FQ_KIND_GATE, (int)g, pr, 1500.0, 0}); break;
                case GLP_GATE_BASE_SUM: loops.push_back({plonk::FQ_KIND_GATE, (int)g, pr, 1450.0, 150.0 * pr}); break;
                case GLP_GATE_ARITHMETIC_EXT: loops.push_back({plonk::FQ_KIND_GATE, (int)g, pr, 3200.0, 0}); break;
                case GLP_GATE_MUL_EXT: loops.push_back({plonk::FQ_KIND_GATE, (int)g, pr, 2700.0, 0}); break;
                case GLP_GATE_RANDOM_ACCESS: {
                    const int bits = gt.param & 0xFF, copies = (gt.param >> 8) & 0xFF;
                    loops.push_back({plonk::FQ_KIND_GATE, (int)g, copies, 21.5 * (65.0 * bits + 42.0 * ((1 << bits) - 1) + 70), 0});
                    break;
                }
                case GLP_GATE_REDUCING: loops.push_back({plonk::FQ_KIND_GATE, (int)g, pr, 3000.0, 0}); break;
                case GLP_GATE_REDUCING_EXT: loops.push_back({plonk::FQ_KIND_GATE, (int)g, pr, 2260.0, 0}); break;
                case GLP_GATE_EXPONENTIATION: loops.push_back({plonk::FQ_KIND_GATE, (int)g, pr, 1660.0, 0}); break;
                case GLP_GATE_POSEIDON_MDS: loops.push_back({plonk::FQ_KIND_GATE, (int)g, 24, 2300.0, 0}); break;
                case GLP_GATE_COSET_INTERPOLATION:  // its chunks chain through eval / prod: one task
                    loops.push_back({plonk::FQ_KIND_GATE, (int)g, 1, 3200.0 * (1 << (gt.param & 0xFF)), 0});
                    break;
                default: break;  // NoopGate: no constraints
            }
        }
        constexpr double TASK_FIXED = 5000.0;
        std::vector<int> parts(loops.size(), 1);
        auto build = [&](std::vector<Cand>& out) {
            out.clear();
            for (size_t i = 0; i < loops.size(); ++i) {
                const Loop& lp = loops[i];
                if (lp.n_iter <= 0) continue;
                const int np = parts[i] > lp.n_iter ? lp.n_iter : parts[i];
                for (int k = 0; k < np; ++k) {
                    const int lo = (int)((long)lp.n_iter * k / np), hi = (int)((long)lp.n_iter * (k + 1) / np);
                    out.push_back({(hi - lo) * lp.unit + (k == 0 ? lp.first : 0) + TASK_FIXED,
                                   plonk::FqTask{0, (uint8_t)lp.kind, (uint8_t)lp.gate, 0, (uint16_t)lo, (uint16_t)hi}});
                }
            }
            std::stable_sort(out.begin(), out.end(), [](const Cand& x, const Cand& y) { return x.cost > y.cost; });
        };
        auto makespan = [&](const std::vector<Cand>& tl) {
            double fin[plonk::FQ_ROLES] = {0};
            for (const Cand& cd2 : tl) {
                int best = 0;
                for (int r = 1; r < plonk::FQ_ROLES; ++r)
                    if (fin[r] < fin[best]) best = r;
                fin[best] += cd2.cost;
            }
            double m = 0;
            for (double f : fin) m = f > m ? f : m;
            return m;
        };
        // the permutation argument alone is longer than a wave's share: it starts in two pieces
        if (!loops.empty() && loops[0].n_iter >= 2) parts[0] = 2;
        std::vector<Cand> trial;
        build(cand);
        double best = makespan(cand);
        for (int sweep = 0; sweep < 4; ++sweep) {  // coordinate descent on the piece counts (a few hundred simulated deals)
            bool improved = false;
            for (size_t i = 0; i < loops.size(); ++i) {
                if (loops[i].n_iter < 2 || loops[i].kind == plonk::FQ_KIND_POS_PARTIAL) continue;
                const int keep = parts[i];
                int best_p = keep;
                for (int pcs = 1; pcs <= 5 && pcs <= loops[i].n_iter; ++pcs) {
                    if (pcs == keep) continue;
                    parts[i] = pcs;
                    build(trial);
                    if (trial.size() > (size_t)plonk::FQ_MAX_TASKS) continue;
                    const double m = makespan(trial);
                    if (m < best - 1.0) {
                        best = m;
                        best_p = pcs;
                        improved = true;
                    }
                }
                parts[i] = best_p;
            }
            if (!improved) break;
        }
        build(cand);
        if (cand.size() > (size_t)plonk::FQ_MAX_TASKS) {  // more tasks than the kernel's table: the per-point evaluator, remembered
            fused = false;
            std::lock_guard<std::mutex> lk(d->fq_mu);
            d->fq_plan.clear();
            d->fq_planned = true;
        }
    }
    if (fused && legacy_plan) {
        static const double piece = [] {
            const char* e = getenv("GLP_FQ_PIECE");
            // swept with the dynamic deal (2^20 rows, 14 gates).  Round 3: 1800 21.7 ms, 2500 21.1, 3500 20.5, 5000 20.8.  Round 5, after
            // the kernel's later changes (tools/quotient_probe.py, two runs each): 3500 18.91 / 18.90 ms, 8000 18.69 / 18.68, 11000 18.57 /
            // 18.50, 14000 18.51 / 18.48, 18000 18.55 / 18.56, 25000 22.06 / 22.10 (the permutation argument in one task: the tail).  A
            // task costs ~5 k cycles of prologue / epilogue (x, filter, the alpha-sum reductions: GLP_FQ_DEBUG shows 5.4 k cycles for the
            // two-constraint ConstantGate task), so fewer, longer tasks win until the slowest wave decides
            return e ? atof(e) : 14000.0;
        }();
        // GLP_FQ_CAL=1: unit costs re-calibrated against the measured per-task cycles of round 5 (GLP_FQ_DEBUG: the reducing-extension,
        // exponentiation, MDS and interpolation loops and the partial-round chain cost 1.25-2.3 x their round-3 estimates)
        static const bool cal = [] {
            const char* e = getenv("GLP_FQ_CAL");
            return e ? atoi(e) != 0 : false;
        }();
        auto add_range = [&](int kind, int gate, int n_iter, double unit, double fixed) {
            if (n_iter <= 0) return;
            // GLP_FQ_PIECE_SMALL (closed experiment, off): loops shorter than one piece cut by a finer measure so that the tasks handed out
            // LAST are short — 1500: 19.6 ms, 2500: 19.3, 4000: 18.9 against 18.58 without: a task's prologue costs more than its tail saves
            static const double piece_small = [] {
                const char* e = getenv("GLP_FQ_PIECE_SMALL");
                return e ? atof(e) : 0.0;
            }();
            const double pc = (piece_small > 0 && n_iter * unit < piece) ? piece_small : piece;
            int parts = (int)((n_iter * unit + pc - 1) / pc);
            parts = parts < 1 ? 1 : (parts > n_iter ? n_iter : parts);
            for (int k = 0; k < parts; ++k) {
                const int lo = (int)((long)n_iter * k / parts), hi = (int)((long)n_iter * (k + 1) / parts);
                plonk::FqTask t = {0, (uint8_t)kind, (uint8_t)gate, 0, (uint16_t)lo, (uint16_t)hi};
                cand.push_back({(hi - lo) * unit + (k == 0 ? fixed : 0) + 150, t});
            }
        };
        add_range(plonk::FQ_KIND_PERM, 0, (int)cc.num_partial_products + 1, cc.quotient_degree_factor * 135.0 * nch, 900);
        for (uint32_t g = 0; g < cc.num_gates; ++g) {
            const glp_gate& gt = d->gates[g];
            const int pr = (int)gt.param;
            switch (gt.id) {
                case GLP_GATE_POSEIDON:
                    fa.poseidon_gate = (int)g;
                    add_range(plonk::FQ_KIND_POS_FULL, (int)g, 7, 1400, 100);
                    cand.push_back({cal ? 6900.0 : 5500.0, plonk::FqTask{0, (uint8_t)plonk::FQ_KIND_POS_PARTIAL, (uint8_t)g, 0, 0, 1}});
                    break;
                case GLP_GATE_CONSTANT: add_range(plonk::FQ_KIND_GATE, (int)g, pr, 25, 0); break;
                case GLP_GATE_PUBLIC_INPUT: add_range(plonk::FQ_KIND_GATE, (int)g, 4, 25, 0); break;
                case GLP_GATE_ARITHMETIC: add_range(plonk::FQ_KIND_GATE, (int)g, pr, 95, 0); break;
                case GLP_GATE_BASE_SUM: add_range(plonk::FQ_KIND_GATE, (int)g, pr, 60, 12.0 * pr); break;
                case GLP_GATE_ARITHMETIC_EXT: add_range(plonk::FQ_KIND_GATE, (int)g, pr, 200, 0); break;
                case GLP_GATE_MUL_EXT: add_range(plonk::FQ_KIND_GATE, (int)g, pr, 190, 0); break;
                case GLP_GATE_RANDOM_ACCESS: {
                    const int bits = gt.param & 0xFF, copies = (gt.param >> 8) & 0xFF;
                    add_range(plonk::FQ_KIND_GATE, (int)g, copies, 65.0 * bits + 42.0 * ((1 << bits) - 1) + 70, 50);
                    break;
                }
                case GLP_GATE_REDUCING:
                case GLP_GATE_REDUCING_EXT:
                    add_range(plonk::FQ_KIND_GATE, (int)g, pr, !cal ? 105 : (gt.id == GLP_GATE_REDUCING_EXT ? 240 : 120), 0);
                    break;
                case GLP_GATE_EXPONENTIATION: add_range(plonk::FQ_KIND_GATE, (int)g, pr, cal ? 165 : 105, 0); break;
                case GLP_GATE_POSEIDON_MDS: add_range(plonk::FQ_KIND_GATE, (int)g, 24, cal ? 160 : 116, 0); break;
                case GLP_GATE_COSET_INTERPOLATION:  // its chunks chain through eval / prod: one task
                    cand.push_back({(cal ? 235.0 : 190.0) * (1 << (gt.param & 0xFF)) + 200,
                                    plonk::FqTask{0, (uint8_t)plonk::FQ_KIND_GATE, (uint8_t)g, 0, 0, 1}});
                    break;
                default: break;  // NoopGate: no constraints
            }
        }
        // Measurement aid (profiles/r04_quotient_split.md): GLP_FQ_SUBSET = 1 keeps only the permutation + PoseidonGate tasks, 2 only
        // the other gates' tasks — the two halves of a two-launch split, timed with the same kernel.  The quotient is then
        // INCOMPLETE (wrong); never set outside tools/quotient_probe.py.
        // Compiled only into probe builds (make GL_EXTRA_FLAGS=-DGLP_PROBES): a release library cannot be made to emit an incomplete
        // quotient through its environment.
#ifdef GLP_PROBES
        if (const char* e = getenv("GLP_FQ_SUBSET")) {
            const int which = atoi(e);
            if (which == 1 || which == 2) {
                std::vector<Cand> keep;
                for (const Cand& cd2 : cand)
                    if ((cd2.t.kind == plonk::FQ_KIND_GATE) == (which == 2)) keep.push_back(cd2);
                cand.swap(keep);
            }
        }
#endif
        if (cand.size() > (size_t)plonk::FQ_MAX_TASKS) fused = false;  // more tasks than the kernel's table: per-point evaluator
    }
    if (fused) {
        static const bool dynamic_deal = [] {
            const char* e = getenv("GLP_FQ_DYNAMIC");
            return e ? atoi(e) != 0 : true;
        }();
        fa.dynamic = dynamic_deal ? 1 : 0;
        if (have_plan) {  // dealt once, when the plan was made
            for (const Cand& cd2 : cand) fa.tasks[fa.n_tasks++] = cd2.t;
        } else {
            std::stable_sort(cand.begin(), cand.end(), [](const Cand& x, const Cand& y) { return x.cost > y.cost; });
            double load[plonk::FQ_ROLES] = {0};
            for (auto& cd2 : cand) {
                int best = 0;
                for (int r = 1; r < plonk::FQ_ROLES; ++r)
                    if (load[r] < load[best]) best = r;
                cd2.t.role = (uint8_t)best;
                load[best] += cd2.cost;
                fa.tasks[fa.n_tasks++] = cd2.t;
            }
            if (!legacy_plan) {
                std::lock_guard<std::mutex> lk(d->fq_mu);
                if (!d->fq_planned) {
                    d->fq_plan.assign(fa.tasks, fa.tasks + fa.n_tasks);
                    d->fq_poseidon_gate = fa.poseidon_gate;
                    d->fq_planned = true;
                }
            }
        }
        if (!c->fq_attr_set) {  // per context (= per device; a context is single-threaded): the attribute is a per-device setting
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(plonk::quotient_fused_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            c->fq_attr_set = true;
        }
        static const bool debug_cycles = getenv("GLP_FQ_DEBUG") != nullptr;
        if (debug_cycles) {
            uint64_t* dbg = nullptr;
            GLP_TRY(sc.alloc(plonk::FQ_MAX_TASKS + 8, &dbg));
            HIP_TRY(hipMemsetAsync(dbg, 0, (plonk::FQ_MAX_TASKS + 8) * 8, c->stream));
            fa.task_cycles = reinterpret_cast<unsigned long long*>(dbg);
        }
        // GLP_FQ_PERSISTENT=1: the persistent form (quotient_fused.h: one 16-wave workgroup per CU walks the tiles, next tile in
        // flight by LDS-DMA, no barrier per tile).  Bit-identical (parity-tested), and MEASURED SLOWER: 19.85 ms against 19.11 for a
        // 2^20-row 14-gate launch, at the same 84.5 % VALUBusy — the idle 15 % of the one-tile kernel is not its fill or its
        // tail (the persistent form has neither) but four waves per SIMD each needing the VALU a quarter of the time with
        // ~500 short waits (LDS, scalar loads, wait states) in their instruction streams.  Off by default.
        const int want_persistent = [] {
            const char* e = getenv("GLP_FQ_PERSISTENT");
            return e ? (atoi(e) != 0 ? 1 : 0) : 0;
        }();
        if (c->n_cus == 0) {  // of THIS context's device (a process may hold contexts on several GPUs)
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || v <= 0) v = 256;
            c->n_cus = v;
        }
        const int n_cus = c->n_cus;
        const size_t n_tiles_all = N / plonk::FQ_POINTS * K;
        const size_t persistent_lds = plonk::fqp_lds_bytes((int)cc.num_wires, (int)cc.num_constants);
        const bool persistent = (want_persistent == 1 || (want_persistent == -1 && n_tiles_all >= (size_t)16 * n_cus)) &&
                                persistent_lds <= 160 * 1024 && n_tiles_all < ((size_t)1 << 31);
        if (persistent && !c->fqp_attr_set) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(plonk::quotient_fused_persistent_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            c->fqp_attr_set = true;
        }
        {
            Timed t(c, GLP_K_PLONK);
            if (persistent) {
                fa.n_proofs = (unsigned)K;
                const unsigned grid = (unsigned)(n_tiles_all < (size_t)n_cus ? n_tiles_all : (size_t)n_cus);
                hipLaunchKernelGGL(plonk::quotient_fused_persistent_kernel, dim3(grid), dim3(plonk::FQP_THREADS), persistent_lds,
                                   c->stream, fa);
            } else {
                hipLaunchKernelGGL(plonk::quotient_fused_kernel, dim3((unsigned)(N / plonk::FQ_POINTS), (unsigned)K),
                                   dim3(plonk::FQ_THREADS), fused_lds, c->stream, fa);
            }
            HIP_TRY(hipGetLastError());
        }
        if (debug_cycles) {  // measured cycles per task next to the cost model's estimate (calibration of the table above)
            std::vector<unsigned long long> cyc(plonk::FQ_MAX_TASKS + 8);
            HIP_TRY(hipMemcpyAsync(cyc.data(), fa.task_cycles, (plonk::FQ_MAX_TASKS + 8) * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (const double ns = (double)cyc[plonk::FQ_MAX_TASKS + 4]; ns > 0) {
                const unsigned long long* tlv = &cyc[plonk::FQ_MAX_TASKS];
                fprintf(stderr, "fq timeline per workgroup (%d waves, %.0f samples): fill %.0f cycles, in the task loop %.0f per wave, waiting "
                                "behind it %.0f per wave, lifetime %.0f  =>  fill %.1f %%, tail %.1f %% of a wave's life\n",
                        plonk::FQ_ROLES, ns, tlv[0] / ns, tlv[1] / ns / plonk::FQ_ROLES, tlv[2] / ns / plonk::FQ_ROLES, tlv[3] / ns,
                        100.0 * tlv[0] / tlv[3], 100.0 * tlv[2] / plonk::FQ_ROLES / tlv[3]);
            }
            const double wgs = N / plonk::FQ_POINTS < 1024 ? (double)(N / plonk::FQ_POINTS) : 1024.0;
            double per_role[plonk::FQ_ROLES] = {0};
            for (int t2 = 0; t2 < fa.n_tasks; ++t2) {
                const plonk::FqTask& tk = fa.tasks[t2];
                fprintf(stderr, "fq task %2d role %d kind %d gate %2d (id %2u) [%3u,%3u) est %7.0f measured %8.0f cycles\n", t2, tk.role,
                        tk.kind, tk.gate, tk.kind == plonk::FQ_KIND_PERM ? 99u : d->gates[tk.gate].id, tk.lo, tk.hi, cand[t2].cost,
                        cyc[t2] / wgs);
                per_role[tk.role] += cyc[t2] / wgs;
            }
            for (int r = 0; r < plonk::FQ_ROLES; ++r) fprintf(stderr, "fq role %d: %8.0f cycles\n", r, per_role[r]);
        }
    }
    for (uint32_t g = 0; g < cc.num_gates && !fused; ++g)
        if (d->gates[g].id == GLP_GATE_POSEIDON && !a.ga_poseidon) {  // its constraints first, in a kernel of their own
            uint64_t* ga = nullptr;
            GLP_TRY(sc.alloc(K * nch * N, &ga));
            Timed t(c, GLP_K_PLONK);
            hipLaunchKernelGGL(plonk::poseidon_gate_kernel, dim3((unsigned)((N + plonk::THREADS - 1) / plonk::THREADS), (unsigned)K),
                               dim3(plonk::THREADS), 0, c->stream, a, nch + nch * ((int)cc.num_partial_products + 1), ga);
            HIP_TRY(hipGetLastError());
            a.ga_poseidon = ga;
        }
    if (!fused) {
        Timed t(c, GLP_K_PLONK);
        hipLaunchKernelGGL(plonk::quotient_kernel, dim3((unsigned)((N + plonk::THREADS - 1) / plonk::THREADS), (unsigned)K),
                           dim3(plonk::THREADS), 0, c->stream, a);
        HIP_TRY(hipGetLastError());
    }
    // coset_ifft: values at bit-reversed positions -> coefficients (natural order), then chunks of n
    const uint64_t N_inv = gl::canon(gl::inv((uint64_t)N));
    const uint64_t shift_inv = gl::canon(gl::inv(gl::COSET_SHIFT));
    if (lde_bits < 9) {
        uint64_t* tmp = nullptr;
        GLP_TRY(sc.alloc(K * nch * N, &tmp));
        GLP_TRY(launch_small(c, vals, N, tmp, N, K * nch, lde_bits, lde_bits, 1, 1, shift_inv, N_inv, 0, 1));
        HIP_TRY(hipMemcpyAsync(out_dev, tmp, K * nch * N * 8, hipMemcpyDeviceToDevice, c->stream));
    } else {
        GLP_TRY(run_dit(c, vals, N, K * nch, lde_bits, 1, N_inv));
        GLP_TRY(scale_by_powers(c, vals, N, K * nch, lde_bits, shift_inv, 1));
        HIP_TRY(hipMemcpyAsync(out_dev, vals, K * nch * N * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    return GLP_OK;  // no wait: the scratch goes back to a stream-ordered pool, and out_dev is read on the same stream
}

}  // namespace

extern "C" {

int32_t glp_circuit_validate(const glp_circuit* common) { return check_common(common); }

int32_t glp_circuit_build(glp_ctx* c, const glp_circuit* common, const uint64_t* cs_values, glp_circuit_data** data_out,
                          uint64_t* cap_out, uint64_t digest_out[4]) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !cs_values || !data_out) return GLP_ERR_INVALID;
    GLP_TRY(check_common(common));
    glp_circuit_data* d = new (std::nothrow) glp_circuit_data();
    if (!d) return GLP_ERR_OOM;
    d->ctx = c;
    ctx_retain(c);
    d->common = *common;
    d->gates.assign(common->gates, common->gates + common->num_gates);
    d->k_is.resize(common->num_routed_wires);
    for (uint32_t j = 0; j < common->num_routed_wires; ++j) d->k_is[j] = gl::canon(common->k_is[j]);
    d->common.gates = d->gates.data();
    d->common.k_is = d->k_is.data();
    d->cs = nullptr;
    d->sigma_values = nullptr;
    const size_t n = (size_t)1 << common->degree_bits;
    const size_t ncs = (size_t)common->num_constants + common->num_routed_wires;
    std::vector<uint64_t> cap((size_t)4 << common->fri.cap_height);
    int32_t s = glp_batch_from_values(c, cs_values, ncs, common->degree_bits, common->fri.rate_bits, common->fri.cap_height,
                                      0, &d->cs, cap.data());
    if (s == GLP_OK) s = pool_alloc_t(c, (size_t)common->num_routed_wires * n, &d->sigma_values);
    if (s == GLP_OK)
        s = upload_staged(c, d->sigma_values, cs_values + (size_t)common->num_constants * n, (size_t)common->num_routed_wires * n * 8);
    if (s != GLP_OK) {
        glp_circuit_free(d);
        return s;
    }
    // circuit_digest = hash_no_pad(constants_sigmas_cap.flatten() ++ []) at f99ed9c; later revisions append a domain-separator
    // digest (hash_pad of the empty separator: [1, 0 x 10, 1]) and / or degree_bits
    {
        std::vector<uint64_t> parts(cap);
        const uint32_t mode = common->fri.flags & GLP_FLAG_DIGEST_MASK;
        if (mode == GLP_FLAG_DIGEST_CAP_DOM_DEG) {
            uint64_t pad[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1}, dom[4];
            host_hash_no_pad(pad, 12, dom);
            parts.insert(parts.end(), dom, dom + 4);
        }
        if (mode != GLP_FLAG_DIGEST_CAP) parts.push_back(common->degree_bits);
        host_hash_no_pad(parts.data(), parts.size(), d->common.circuit_digest);
    }
    if (cap_out) memcpy(cap_out, cap.data(), cap.size() * 8);
    if (digest_out) memcpy(digest_out, d->common.circuit_digest, 32);
    *data_out = d;
    return GLP_OK;
}

int32_t glp_circuit_set_digest(glp_circuit_data* d, const uint64_t digest[4]) {
    if (!d || !digest) return GLP_ERR_INVALID;
    for (int i = 0; i < 4; ++i) d->common.circuit_digest[i] = gl::canon(digest[i]);
    return GLP_OK;
}

void glp_circuit_free(glp_circuit_data* d) {
    GLP_ON_DEVICE_OF(d ? d->ctx : nullptr);
    if (!d) return;
    if (d->cs) glp_batch_free(d->cs);
    if (d->sigma_values) pool_free(d->ctx, d->sigma_values);
    glp_ctx* c = d->ctx;
    delete d;
    ctx_release(c);
}

const glp_batch* glp_circuit_constants_sigmas(const glp_circuit_data* d) { return d ? d->cs : nullptr; }

size_t glp_proof_len(const glp_circuit* cc) {
    if (!cc) return 0;
    const size_t nch = cc->num_challenges;
    size_t cols[4] = {(size_t)cc->num_constants + cc->num_routed_wires, cc->num_wires, nch * (1 + cc->num_partial_products),
                      nch * cc->quotient_degree_factor};
    size_t openings = cols[0] + cols[1] + cols[2] + cols[3] + nch;
    if (cc->fri.flags & GLP_FLAG_ZERO_KNOWLEDGE)  // the query openings of the three blinded oracles carry their salt words
        for (int o = 1; o < 4; ++o) cols[o] += GLP_SALT_SIZE;
    return 3 * ((size_t)4 << cc->fri.cap_height) + 2 * openings + glp_fri_proof_len(&cc->fri, cols, 4);
}

int32_t glp_partial_products_dev(glp_ctx* c, const glp_circuit_data* d, const uint64_t* wires_dev, size_t col_stride,
                                 const uint64_t* betas, const uint64_t* gammas, uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !wires_dev || !betas || !gammas || !out_dev) return GLP_ERR_INVALID;
    if (d->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    ProofChal ch;
    memset(&ch, 0, sizeof ch);
    for (uint32_t i = 0; i < d->common.num_challenges; ++i) ch.betas[i] = betas[i], ch.gammas[i] = gammas[i];
    GLP_TRY(arena_reset(c));
    return partial_products(c, d, wires_dev, col_stride, 0, 1, &ch, out_dev);
}

int32_t glp_quotient_dev(glp_ctx* c, const glp_circuit_data* d, const glp_batch* wires, const glp_batch* zs_pp,
                         const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas, const uint64_t pi_hash[4],
                         uint64_t* out_dev) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !wires || !zs_pp || !betas || !gammas || !alphas || !pi_hash || !out_dev) return GLP_ERR_INVALID;
    if (d->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    ProofChal ch;
    memset(&ch, 0, sizeof ch);
    for (uint32_t i = 0; i < d->common.num_challenges; ++i) ch.betas[i] = betas[i], ch.gammas[i] = gammas[i], ch.alphas[i] = alphas[i];
    for (int i = 0; i < 4; ++i) ch.pi_hash[i] = pi_hash[i];
    GLP_TRY(arena_reset(c));
    return quotient(c, d, wires, zs_pp, 1, &ch, out_dev);
}

int32_t glp_poseidon_gate_witness_dev(glp_ctx* c, const uint64_t* inputs_dev, size_t n_rows, uint64_t* rows_dev,
                                      size_t row_stride, size_t col_stride) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !inputs_dev || !rows_dev || n_rows == 0) return GLP_ERR_INVALID;
    Timed t(c, GLP_K_PLONK);
    hipLaunchKernelGGL(plonk::poseidon_gate_witness_kernel, dim3((unsigned)((n_rows + plonk::THREADS - 1) / plonk::THREADS)),
                       dim3(plonk::THREADS), 0, c->stream, inputs_dev, n_rows, rows_dev, row_stride, col_stride);
    HIP_TRY(hipGetLastError());
    return GLP_OK;
}

// plonk::prover::prove after witness generation, for K witnesses of one circuit side by side: every step below is ONE launch
// sequence over the K proofs (commitments over K x C columns as a forest of K Merkle trees; kernels with a proof dimension),
// the K Fiat-Shamir transcripts advance together on the host between the steps.
int32_t glp_prove_batch_dev(glp_ctx* c, const glp_circuit_data* d, size_t K, const uint64_t* wires_dev, size_t col_stride,
                            size_t proof_stride, const uint64_t* public_inputs, size_t n_pi, uint64_t* proofs_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !wires_dev || !proofs_out || (n_pi && !public_inputs) || K == 0 || K > 65535) return GLP_ERR_INVALID;
    if (d->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    const glp_circuit& cc = d->common;
    const size_t n = (size_t)1 << cc.degree_bits;
    const uint32_t rb = cc.fri.rate_bits, caph = cc.fri.cap_height, nch = cc.num_challenges;
    const size_t cap_words = (size_t)4 << caph;
    const size_t ncs = (size_t)cc.num_constants + cc.num_routed_wires, nw = cc.num_wires;
    const size_t nzs = (size_t)nch * (1 + cc.num_partial_products), nq = (size_t)nch * cc.quotient_degree_factor;
    const size_t proof_len = glp_proof_len(&cc);
    if (col_stride != n) return GLP_ERR_UNSUPPORTED;  // the partial-product kernels take the witness columns back to back
    if (K > 1 && proof_stride < nw * col_stride) return GLP_ERR_INVALID;
    // config.zero_knowledge: the wires, Z / partial-product and quotient commitments are blinded (PlonkOracle::*.blinding); a
    // blinded commitment is one tree, so K such proofs are proven one after the other.  Their salts: c->zk_salts_dev when the
    // caller supplied them (glp_prove_salted*: [3][GLP_SALT_SIZE][N], consumed by this call), else the context's ChaCha12 stream.
    const uint32_t zk = (cc.fri.flags & GLP_FLAG_ZERO_KNOWLEDGE) ? 1u : 0u;
    const uint64_t* zk_salts = c->zk_salts_dev;
    c->zk_salts_dev = nullptr;
    if (zk && K > 1) {
        if (zk_salts) return GLP_ERR_INVALID;
        for (size_t k = 0; k < K; ++k)
            GLP_TRY(glp_prove_batch_dev(c, d, 1, wires_dev + k * proof_stride, col_stride, 0, public_inputs + k * n_pi, n_pi,
                                        proofs_out + k * proof_len));
        return GLP_OK;
    }
    if (!zk && zk_salts) return GLP_ERR_INVALID;  // salts for a circuit that is not configured for zero knowledge
    const size_t salt_stride = (size_t)GLP_SALT_SIZE << (cc.degree_bits + rb);
    if (c->arrival_base && (c->arrival_base != wires_dev || K != 1)) {  // an announced chunked upload that is not this witness:
        const size_t chunks = c->arrival_bounds.size() - 1;                 // enqueue and wait for all of it, plainly
        GLP_TRY(arrival_feed(c, chunks - 1));
        for (size_t k = 0; k < chunks; ++k) HIP_TRY(hipStreamWaitEvent(c->stream, c->arrival_ev[k], 0));
        c->arrival_base = nullptr;
    }
    GLP_TRY(arena_reset(c));
    StepTrace trace("glp_prove_batch_dev");
    std::vector<ProofChal> ch(K);
    std::vector<glp_challenger> chal(K);
    std::vector<uint64_t*> w(K);
    std::vector<uint64_t> caps(K * cap_words);
    parallel_for(K, [&](size_t k) {
        memset(&ch[k], 0, sizeof ch[k]);
        host_hash_no_pad(public_inputs + k * n_pi, n_pi, ch[k].pi_hash);
        w[k] = proofs_out + k * proof_len;
    });
    Scratch sc(c);
    auto take_caps = [&]() {  // append commitment k's cap to proof k and to its transcript
        parallel_for(K, [&](size_t k) {
            memcpy(w[k], caps.data() + k * cap_words, cap_words * 8);
            glp_challenger_observe(&chal[k], w[k], cap_words);
            w[k] += cap_words;
        }, 2);
    };

    // wires commitment
    BatchGuard wires, zs, quot;
    c->next_salts_dev = zk_salts;
    GLP_TRY(batch_from_values_forest(c, wires_dev, col_stride, proof_stride, nw, K, cc.degree_bits, rb, caph, &wires.b, caps.data(), zk));
    parallel_for(K, [&](size_t k) {
        glp_challenger_init(&chal[k]);
        glp_challenger_observe(&chal[k], cc.circuit_digest, 4);
        glp_challenger_observe(&chal[k], ch[k].pi_hash, 4);
    });
    take_caps();
    parallel_for(K, [&](size_t k) {
        glp_challenger_get_n(&chal[k], ch[k].betas, nch);
        glp_challenger_get_n(&chal[k], ch[k].gammas, nch);
    });

    trace.mark("wires commitment + b,g");
    // Z and partial products
    uint64_t* zs_vals = nullptr;
    GLP_TRY(sc.alloc(K * nzs * n, &zs_vals));
    GLP_TRY(partial_products(c, d, wires_dev, col_stride, proof_stride, K, ch.data(), zs_vals));
    c->next_salts_dev = zk_salts ? zk_salts + salt_stride : nullptr;
    GLP_TRY(batch_from_values_forest(c, zs_vals, n, nzs * n, nzs, K, cc.degree_bits, rb, caph, &zs.b, caps.data(), zk));
    take_caps();
    parallel_for(K, [&](size_t k) { glp_challenger_get_n(&chal[k], ch[k].alphas, nch); });

    trace.mark("Z commitment + alphas");
    // quotient polynomials
    uint64_t* q_coeffs = nullptr;
    GLP_TRY(sc.alloc(K * nq * n, &q_coeffs));
    GLP_TRY(quotient(c, d, wires.b, zs.b, K, ch.data(), q_coeffs));
    c->next_salts_dev = zk_salts ? zk_salts + 2 * salt_stride : nullptr;
    GLP_TRY(batch_from_coeffs_forest(c, q_coeffs, nq, K, cc.degree_bits, rb, caph, &quot.b, caps.data(), zk));
    take_caps();
    std::vector<uint64_t> zeta(2 * K), gz(2 * K);
    const uint64_t g = host_root_pow(c, (uint64_t)1 << (MASTER_K - cc.degree_bits));
    std::vector<uint8_t> in_subgroup(K, 0);
    parallel_for(K, [&](size_t k) {
        glp_challenger_get_n(&chal[k], &zeta[2 * k], 2);
        // "Opening point is in the subgroup." guard of plonky2's prover
        gl::ext2 zn = {zeta[2 * k], zeta[2 * k + 1]};
        for (uint32_t i = 0; i < cc.degree_bits; ++i) zn = gl::ext_mul(zn, zn);
        in_subgroup[k] = gl::canon(zn.a) == 1 && gl::canon(zn.b) == 0;
        gz[2 * k] = gl::canon(gl::mul(zeta[2 * k], g));
        gz[2 * k + 1] = gl::canon(gl::mul(zeta[2 * k + 1], g));
    });
    for (size_t k = 0; k < K; ++k)
        if (in_subgroup[k]) return GLP_ERR_INVALID;

    trace.mark("quotient commitment + zeta");
    // OpeningSet::new — proof order: constants, sigmas, wires, zs, zs_next, partial products, quotient
    const size_t total = ncs + nw + nzs + nq;
    std::vector<uint64_t> opened0(K * 2 * total);  // per proof, FRI batch 0 order: constants_sigmas, wires, zs ++ pp, quotient
    std::vector<uint64_t*> o_zn(K);
    {
        // the power tables of zeta and g * zeta of all proofs: one launch over 2K points
        std::vector<uint64_t> pts(4 * K);
        memcpy(pts.data(), zeta.data(), 2 * K * 8);
        memcpy(pts.data() + 2 * K, gz.data(), 2 * K * 8);
        uint64_t* zp = nullptr;
        GLP_TRY(build_zpow(c, (int)cc.degree_bits, 2 * K, pts.data(), &zp));
        sc.ptrs.push_back(zp);
        const uint64_t* zpg = zp + K * 2 * n;
        // five evaluation launches into one device buffer, ONE read-back: [cs | wires | zs | quotient | Z at g zeta], each [K][cols][2]
        const int lg = (int)cc.degree_bits;
        const size_t all_cols = ncs + nw + nzs + nq + nch;
        uint64_t *ev_dev = nullptr, *ev_part = nullptr;
        GLP_TRY(sc.alloc(K * all_cols * 2, &ev_dev));
        GLP_TRY(sc.alloc(K * all_cols * 2 * eval_partial_blocks(lg), &ev_part));
        uint64_t* ev = arena_alloc_t<uint64_t>(c, K * all_cols * 2);
        if (!ev) return GLP_ERR_OOM;  // neither pinned nor pageable host memory left
        const size_t off_w = K * 2 * ncs, off_z = off_w + K * 2 * nw, off_q = off_z + K * 2 * nzs, off_zn = off_q + K * 2 * nq;
        const size_t nb = eval_partial_blocks(lg);
        // small proofs: the evaluation kernels store their results (canonical values) straight into the pinned arena block and the
        // host spins on them (internal.h, mapped_*) — no copy kernel, no stream synchronisation on this round trip
        const bool mapped = mapped_readbacks_enabled() && K * all_cols * 16 <= ((size_t)256 << 10) && arena_is_pinned(c, ev, K * all_cols * 16);
        uint64_t* ev_to = mapped ? ev : ev_dev;
        if (mapped) mapped_prefill(ev, K * all_cols * 2);
        GLP_TRY(eval_launch(c, d->cs->coeffs, n, ncs, 0, lg, K, zp, ev_part, ev_to));                                  // shared by all proofs
        GLP_TRY(eval_launch(c, wires.b->coeffs, n, nw, nw, lg, K, zp, ev_part + off_w * nb, ev_to + off_w));
        GLP_TRY(eval_launch(c, zs.b->coeffs, n, nzs, nzs, lg, K, zp, ev_part + off_z * nb, ev_to + off_z));
        GLP_TRY(eval_launch(c, quot.b->coeffs, n, nq, nq, lg, K, zp, ev_part + off_q * nb, ev_to + off_q));
        GLP_TRY(eval_launch(c, zs.b->coeffs, n, nch, nzs, lg, K, zpg, ev_part + off_zn * nb, ev_to + off_zn));         // Z at g * zeta
        if (mapped) {
            GLP_TRY(mapped_wait(c, ev, K * all_cols * 2));
        } else {
            HIP_TRY(hipMemcpyAsync(ev, ev_dev, K * all_cols * 16, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        const uint64_t *e_cs = ev, *e_w = ev + off_w, *e_z = ev + off_z, *e_q = ev + off_q, *e_zn = ev + off_zn;
        parallel_for(K, [&](size_t k) {
            uint64_t* o_cs = w[k];
            uint64_t* o_w = o_cs + 2 * ncs;
            uint64_t* o_z = o_w + 2 * nw;
            o_zn[k] = o_z + 2 * nch;
            uint64_t* o_pp = o_zn[k] + 2 * nch;
            uint64_t* o_q = o_pp + 2 * (nzs - nch);
            memcpy(o_cs, &e_cs[k * 2 * ncs], 2 * ncs * 8);
            memcpy(o_w, &e_w[k * 2 * nw], 2 * nw * 8);
            memcpy(o_z, &e_z[k * 2 * nzs], 2 * nch * 8);
            memcpy(o_pp, &e_z[k * 2 * nzs + 2 * nch], 2 * (nzs - nch) * 8);
            memcpy(o_q, &e_q[k * 2 * nq], 2 * nq * 8);
            memcpy(o_zn[k], &e_zn[k * 2 * nch], 2 * nch * 8);
            w[k] = o_q + 2 * nq;
            // challenger.observe_openings(to_fri_openings): zeta batch, then zeta_next batch
            glp_challenger_observe(&chal[k], o_cs, 2 * ncs);
            glp_challenger_observe(&chal[k], o_w, 2 * nw);
            glp_challenger_observe(&chal[k], o_z, 2 * nch);
            glp_challenger_observe(&chal[k], o_pp, 2 * (nzs - nch));
            glp_challenger_observe(&chal[k], o_q, 2 * nq);
            glp_challenger_observe(&chal[k], o_zn[k], 2 * nch);
            uint64_t* dst = &opened0[k * 2 * total];
            memcpy(dst, o_cs, 2 * ncs * 8); dst += 2 * ncs;
            memcpy(dst, o_w, 2 * nw * 8); dst += 2 * nw;
            memcpy(dst, o_z, 2 * nch * 8); dst += 2 * nch;
            memcpy(dst, o_pp, 2 * (nzs - nch) * 8); dst += 2 * (nzs - nch);
            memcpy(dst, o_q, 2 * nq * 8);
        }, 1);  // (the heaviest step: 65 permutations per proof — worth a second thread from two proofs on)
    }

    trace.mark("openings");
    // FRI over the four oracles (common_data.get_fri_instance(zeta))
    std::vector<uint32_t> oi(total + nch), pi(total + nch);
    {
        size_t j = 0;
        const size_t cols[4] = {ncs, nw, nzs, nq};
        for (uint32_t o = 0; o < 4; ++o)
            for (uint32_t i = 0; i < cols[o]; ++i) {
                oi[j] = o;
                pi[j++] = i;
            }
        for (uint32_t i = 0; i < nch; ++i) {
            oi[j] = 2;
            pi[j++] = i;
        }
    }
    std::vector<glp_fri_batch> fb(2 * K);
    for (size_t k = 0; k < K; ++k) {
        glp_fri_batch* f0 = &fb[2 * k];
        f0[0].point[0] = zeta[2 * k];
        f0[0].point[1] = zeta[2 * k + 1];
        f0[0].n_polys = total;
        f0[0].oracle_index = oi.data();
        f0[0].poly_index = pi.data();
        f0[0].opened_values = &opened0[k * 2 * total];
        f0[1].point[0] = gz[2 * k];
        f0[1].point[1] = gz[2 * k + 1];
        f0[1].n_polys = nch;
        f0[1].oracle_index = oi.data() + total;
        f0[1].poly_index = pi.data() + total;
        f0[1].opened_values = o_zn[k];
    }
    const glp_batch* oracles[4] = {d->cs, wires.b, zs.b, quot.b};
    return fri_prove_forest(c, oracles, 4, K, fb.data(), 2, &cc.fri, chal.data(), w.data());
}

int32_t glp_prove_dev(glp_ctx* c, const glp_circuit_data* d, const uint64_t* wires_dev, size_t col_stride,
                      const uint64_t* public_inputs, size_t n_pi, uint64_t* proof_out) {
    GLP_ON_DEVICE_OF(c);
    return glp_prove_batch_dev(c, d, 1, wires_dev, col_stride, 0, public_inputs, n_pi, proof_out);
}

int32_t glp_prove_stream(glp_ctx* c, const glp_circuit_data* d, size_t n_proofs, const uint64_t* const* wire_values,
                         const uint64_t* public_inputs, size_t n_pi, uint64_t* proofs_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !wire_values || !proofs_out || (n_pi && !public_inputs)) return GLP_ERR_INVALID;
    if (d->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    if (n_proofs == 0) return GLP_OK;
    for (size_t k = 0; k < n_proofs; ++k)
        if (!wire_values[k]) return GLP_ERR_INVALID;
    const size_t n = (size_t)1 << d->common.degree_bits, words = (size_t)d->common.num_wires * n;
    const size_t proof_len = glp_proof_len(&d->common);
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    // Small circuits are proven in GROUPS (glp_prove_batch_dev): enough witnesses per group to fill the chip (~2^19 rows), the
    // next group's witnesses uploaded while this group proves.  A 2^20-row circuit is a group of one.  GLP_STREAM_BATCH forces
    // the group size (tests).
    size_t group = ((size_t)1 << 19) / n;
    if (const char* e = getenv("GLP_STREAM_BATCH")) group = (size_t)atol(e);
    group = group < 1 ? 1 : (group > 1024 ? 1024 : group);
    if (group > n_proofs) group = n_proofs;
    const size_t n_groups = (n_proofs + group - 1) / group;
    Scratch sc(c);
    uint64_t* stage[2] = {nullptr, nullptr};
    GLP_TRY(sc.alloc(group * words, &stage[0]));
    if (n_groups > 1) GLP_TRY(sc.alloc(group * words, &stage[1]));
    hipEvent_t ev[2] = {nullptr, nullptr};
    int32_t s = GLP_OK;
    auto hip = [&](hipError_t e) {
        if (e != hipSuccess && s == GLP_OK) {
            note_hip_error(e);
            s = e == hipErrorOutOfMemory ? GLP_ERR_OOM : GLP_ERR_HIP;
        }
    };
    hip(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    hip(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    auto upload = [&](size_t g) {  // the witnesses of group g -> staging buffer g mod 2, on the copy stream
        const size_t k0 = g * group, k1 = k0 + group < n_proofs ? k0 + group : n_proofs;
        for (size_t k = k0; k < k1; ++k)
            hip(hipMemcpyAsync(stage[g & 1] + (k - k0) * words, wire_values[k], words * 8, hipMemcpyHostToDevice, c->copy_stream));
        hip(hipEventRecord(ev[g & 1], c->copy_stream));
    };
    // The staging buffers come from the context's pool, which orders the reuse of a block on c->stream only: an earlier call
    // may have returned with kernels still queued there and their scratch already back in the pool (glp_quotient_dev,
    // glp_partial_products_dev, glp_dev_free).  The copy stream therefore starts after everything queued on c->stream so far.
    hip(hipEventRecord(ev[1], c->stream));
    hip(hipStreamWaitEvent(c->copy_stream, ev[1], 0));
    // the first witness of a stream of large proofs has nothing to hide behind: it goes up in column chunks and its wires
    // commitment starts on the first of them (arrival_announce); every later upload runs under the previous proof
    const bool first_chunked = group == 1;
    if (s == GLP_OK) {
        if (first_chunked)
            s = arrival_announce(c, stage[0], nullptr, wire_values[0], d->common.num_wires, n, true);
        else
            upload(0);
    }
    for (size_t g = 0; g < n_groups && s == GLP_OK; ++g) {
        // buffer (g + 1) mod 2 held group g - 1, whose proofs are complete (the prover returns after its last read-back)
        if (g + 1 < n_groups) upload(g + 1);
        if (!(first_chunked && g == 0)) hip(hipStreamWaitEvent(c->stream, ev[g & 1], 0));
        const size_t k0 = g * group, cnt = k0 + group < n_proofs ? group : n_proofs - k0;
        if (s == GLP_OK)
            s = glp_prove_batch_dev(c, d, cnt, stage[g & 1], n, words, public_inputs + k0 * n_pi, n_pi, proofs_out + k0 * proof_len);
    }
    arrival_finish(c);  // the borrowed host buffers are not read after return
    if (ev[0]) (void)hipEventDestroy(ev[0]);
    if (ev[1]) (void)hipEventDestroy(ev[1]);
    return s;
}

// CircuitData::prove under config.zero_knowledge (common.fri.flags & GLP_FLAG_ZERO_KNOWLEDGE) with the CALLER's salts:
// salts = [3][GLP_SALT_SIZE][2^(degree_bits + rate_bits)] for the wires, Z / partial-product and quotient commitments
int32_t glp_prove_salted_dev(glp_ctx* c, const glp_circuit_data* d, const uint64_t* wires_dev, size_t col_stride,
                             const uint64_t* public_inputs, size_t n_pi, const uint64_t* salts_dev, uint64_t* proof_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !salts_dev) return GLP_ERR_INVALID;
    if (!(d->common.fri.flags & GLP_FLAG_ZERO_KNOWLEDGE)) return GLP_ERR_INVALID;
    c->zk_salts_dev = salts_dev;
    const int32_t s = glp_prove_batch_dev(c, d, 1, wires_dev, col_stride, 0, public_inputs, n_pi, proof_out);
    c->zk_salts_dev = nullptr;
    c->next_salts_dev = nullptr;
    return s;
}

int32_t glp_prove_salted(glp_ctx* c, const glp_circuit_data* d, const uint64_t* wire_values, const uint64_t* public_inputs,
                         size_t n_pi, const uint64_t* salts, uint64_t* proof_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !wire_values || !salts) return GLP_ERR_INVALID;
    if (d->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    if (!(d->common.fri.flags & GLP_FLAG_ZERO_KNOWLEDGE)) return GLP_ERR_INVALID;
    const size_t n = (size_t)1 << d->common.degree_bits, nw = d->common.num_wires;
    const size_t salt_words = (size_t)3 * GLP_SALT_SIZE * (n << d->common.fri.rate_bits);
    uint64_t *staged = nullptr, *salts_dev = nullptr;
    GLP_TRY(pool_alloc_t(c, nw * n, &staged));
    int32_t s = pool_alloc_t(c, salt_words, &salts_dev);
    if (s == GLP_OK) s = upload_staged(c, salts_dev, salts, salt_words * 8);
    if (s == GLP_OK) s = upload_staged(c, staged, wire_values, nw * n * 8);
    if (s == GLP_OK) s = glp_prove_salted_dev(c, d, staged, n, public_inputs, n_pi, salts_dev, proof_out);
    pool_free(c, salts_dev);
    pool_free(c, staged);
    return s;
}

// K witnesses in the caller's host memory: as one array (flat, K = 1) or as one pointer per column (cols: K x num_wires of them)
static int32_t prove_from_host(glp_ctx* c, const glp_circuit_data* d, size_t K, const uint64_t* const* cols, const uint64_t* flat,
                               const uint64_t* public_inputs, size_t n_pi, uint64_t* proofs_out) {
    const size_t n = (size_t)1 << d->common.degree_bits, nw = d->common.num_wires;
    uint64_t* staged = nullptr;
    GLP_TRY(pool_alloc_t(c, K * nw * n, &staged));
    // a large witness travels in column chunks on the copy stream; the wires commitment starts on the first of them and follows
    // the rest (inverse NTT, LDE and the leaf sponge per chunk: batch_from_values_forest).  K > 1: the witnesses go up as K x
    // num_wires columns of one array and the batched prover waits for all of them.
    int32_t s = arrival_announce(c, staged, cols, flat, K * nw, n, true);
    if (s == GLP_OK) s = glp_prove_batch_dev(c, d, K, staged, n, nw * n, public_inputs, n_pi, proofs_out);
    arrival_finish(c);  // the borrowed host buffers are not read after return (error paths too)
    pool_free(c, staged);
    return s;
}

int32_t glp_prove(glp_ctx* c, const glp_circuit_data* d, const uint64_t* wire_values, const uint64_t* public_inputs,
                  size_t n_pi, uint64_t* proof_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !wire_values) return GLP_ERR_INVALID;
    if (d->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    return prove_from_host(c, d, 1, nullptr, wire_values, public_inputs, n_pi, proof_out);
}

// CircuitData::prove with the witness as plonky2 holds it when witness generation ends: `witness.wire_values: Vec<Vec<F>>`, one
// allocation per wire column, pageable (reference src/rollup/circuits/mod.rs:1247, src/zkdsa/circuits/mod.rs:326)
int32_t glp_prove_cols(glp_ctx* c, const glp_circuit_data* d, const uint64_t* const* wire_columns, size_t n_columns,
                       const uint64_t* public_inputs, size_t n_pi, uint64_t* proof_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !wire_columns || !proof_out || (n_pi && !public_inputs)) return GLP_ERR_INVALID;
    if (d->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    if (n_columns != d->common.num_wires) return GLP_ERR_INVALID;
    for (size_t j = 0; j < n_columns; ++j)
        if (!wire_columns[j]) return GLP_ERR_INVALID;
    return prove_from_host(c, d, 1, wire_columns, nullptr, public_inputs, n_pi, proof_out);
}

// K witnesses of one circuit, each as plonky2 holds it (K x num_wires column pointers, witness k's columns at wire_columns + k *
// num_wires), proven in one launch sequence: glp_prove_batch_dev without a flattened, uploaded copy made by the caller
int32_t glp_prove_batch_cols(glp_ctx* c, const glp_circuit_data* d, size_t K, const uint64_t* const* wire_columns, size_t n_columns,
                             const uint64_t* public_inputs, size_t n_pi, uint64_t* proofs_out) {
    GLP_ON_DEVICE_OF(c);
    if (!c || !d || !wire_columns || !proofs_out || (n_pi && !public_inputs) || K == 0 || K > 65535) return GLP_ERR_INVALID;
    if (d->ctx->device != c->device) return GLP_ERR_INVALID;  // an object of another GPU's context
    if (n_columns != d->common.num_wires) return GLP_ERR_INVALID;
    for (size_t j = 0; j < K * n_columns; ++j)
        if (!wire_columns[j]) return GLP_ERR_INVALID;
    return prove_from_host(c, d, K, wire_columns, nullptr, public_inputs, n_pi, proofs_out);
}

}  // extern "C"
