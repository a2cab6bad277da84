// quotient_fused.h — the constraint-quotient evaluation as ONE LDS-tiled kernel (replaces quotient_kernel +
// poseidon_gate_kernel for circuits whose wire tile fits the LDS; those two remain as the generic fallback and as the A/B
// reference, GLP_QUOT_FUSED=0).
//
// plonky2 @ f99ed9c src/plonk/prover.rs::compute_quotient_polys -> vanishing_poly.rs::eval_vanishing_poly_base_batch and every
// gate's eval_unfiltered_base_batch, reached from the reference through CircuitData::prove (src/rollup/circuits/mod.rs:1247);
// every circuit that contains builder.verify_proof (src/recursion/gadgets/mod.rs:103) carries all 14 gate types, and all of
// them are evaluated at every one of the 8n points.
//
// Why a tile: with one lane per point and the wires streamed from HBM, every gate re-reads the wire columns it needs — the
// 14-gate circuit read 90.6 GB per launch for 16.2 GB of input (profiles/r02a_summary.md) and ran HBM- and VALU-bound at once.
// Here a workgroup owns 64 consecutive points: its waves first copy the 64-point slice of ALL wire columns and of the
// selector / constant columns into LDS (each column exactly once: 512-byte coalesced loads, conflict-free ds_write_b64 /
// ds_read_b64 with one lane per point), then every wave evaluates a different set of constraint "tasks" on the same 64 points
// — the permutation argument in two halves, the PoseidonGate in three independent pieces (its rounds restart from wire values),
// one task per other gate; which task runs on which wave comes from a host-side cost model (QuotArgs::task_role, so the split
// is tuned without recompiling).  The per-wave partial sums meet in LDS and one wave divides by Z_H and stores.
//
// Alpha combination without a modular reduction per constraint: sum_k c_k alpha^k is accumulated as three 96-bit integers
//     A += lo(c) lo(a),  B += lo(c) hi(a) + hi(c) lo(a),  C += hi(c) hi(a)        (a = alpha^k, 32-bit halves)
// = 4 v_mad_u64_u32 + 4 v_addc per constraint and challenge (the reduced form costs a 14-instruction multiply and a modular
// add), and reduced once per task.  (Measured and rejected in round 3: alpha^k cut into three 22-bit limbs — six products below
// 2^54 per challenge into six plain 64-bit accumulators, no carry words: 12 instead of 16 instructions per constraint, but a
// v_mad_u64_u32 issues in ~5 cycles and a v_addc in ~2.7, so 12 mads cost what 8 mads + 8 addc do: 21.2 -> 21.9 ms.)
#pragma once
#include <hip/hip_runtime.h>

#include "plonk_kernels.h"

namespace plonk {

// unroll factor of the short per-gate loops (arithmetic, base sum, extension, reducing, exponentiation).  Measured in round 3: 2
// (so that the LDS reads of two iterations are in flight together) 19.15-19.28 ms against 19.11 rolled: these loops do not wait.
#ifndef GL_FQ_UNROLL
#define GL_FQ_UNROLL 1
#endif
constexpr int FQ_POINTS = 64;   // points per workgroup = lanes of one wave
constexpr int FQ_ROLES = 8;     // waves per workgroup: two workgroups per CU (LDS) = 4 waves per SIMD at <= 128 VGPRs
                                // (measured, round 3: ten roles at five waves per SIMD — 96 VGPRs, 28 B of scratch — 19.6 ms: no gain)
constexpr int FQ_THREADS = FQ_POINTS * FQ_ROLES;
// A task = a range [lo, hi) of the iteration space of one constraint group, run by one wave ("role") of the workgroup:
//   PERM: permutation chunks (the L_0 terms ride with chunk 0);  POS_FULL: PoseidonGate full-round pieces 0..6;
//   POS_PARTIAL: its partial-round chain (indivisible);  GATE: gate `gate` of the circuit — ops / limbs / copies / coefficients /
//   bits / outputs lo..hi-1 (every gate's constraints restart from wire values, so any split is valid)
constexpr int FQ_KIND_PERM = 0, FQ_KIND_POS_FULL = 1, FQ_KIND_POS_PARTIAL = 2, FQ_KIND_GATE = 3;
constexpr int FQ_MAX_TASKS = 48;
struct FqTask {
    uint8_t role, kind, gate, pad;
    uint16_t lo, hi;
};

struct FusedArgs {
    QuotArgs q;                    // apow must hold 2 rows of n_terms (row 1 = zeros when nch == 1)
    int n_tasks;
    FqTask tasks[FQ_MAX_TASKS];    // dealt to the roles by a host-side cost model (plonk.hip)
    int poseidon_gate;             // index of the PoseidonGate in q.gates, or -1
    unsigned long long* task_cycles;  // tuning aid (GLP_FQ_DEBUG): [n_tasks] cycles summed over the first 1024 workgroups, or null
    unsigned n_proofs;             // persistent form: the launch covers this many proofs (tile index = proof * tiles + tile)
    int dynamic;                   // 1: the waves of a workgroup take the tasks (longest first) from a counter in LDS as they become
                                   // free; 0: task t runs on wave tasks[t].role (the host's static deal)
};

inline size_t fq_lds_bytes(int num_wires, int num_constants) {
    // the tile (the roles' partial sums take its place once every wave is done with it) + the task counter
    const size_t tile = (size_t)(num_wires + num_constants) * FQ_POINTS, part = (size_t)FQ_ROLES * 2 * FQ_POINTS;
    return ((tile > part ? tile : part) + 1) * sizeof(uint64_t);
}

// ---- sum_k c_k alpha_c^k for both challenges, unreduced ----
struct AlphaAcc {
    uint64_t a[2], b[2], c[2];     // low 64 bits of the three partial sums, per challenge
    uint32_t ah[2], bh[2], ch[2];  // overflow words
    const uint64_t* ap0;           // alpha_0^t table (uniform)
    const uint64_t* ap1;

    __device__ __forceinline__ void init(const uint64_t* apow, int n_terms) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = b[i] = c[i] = 0;
            ah[i] = bh[i] = ch[i] = 0;
        }
        ap0 = apow;
        ap1 = apow + n_terms;
    }
    // term t of the vanishing-polynomial list (uniform index): acc_c += cv * alpha_c^t
    __device__ __forceinline__ void mac(uint64_t cv, int t) {
        const uint64_t w0 = ap0[t], w1 = ap1[t];  // uniform: scalar loads
        const uint32_t c0 = (uint32_t)cv, c1 = (uint32_t)(cv >> 32);
        const uint32_t p0 = (uint32_t)w0, p1 = (uint32_t)(w0 >> 32), q0 = (uint32_t)w1, q1 = (uint32_t)(w1 >> 32);
        uint64_t y0, y1, y2, y3, y4, y5, y6, y7;  // carries (SGPR pairs); each is read no sooner than two issue slots after its write
        asm("v_mad_u64_u32 %[a0], %[y0], %[c0], %[p0], %[a0]\n\t"
            "v_mad_u64_u32 %[b0], %[y1], %[c0], %[p1], %[b0]\n\t"
            "v_mad_u64_u32 %[k0], %[y2], %[c1], %[p1], %[k0]\n\t"
            "v_mad_u64_u32 %[a1], %[y3], %[c0], %[q0], %[a1]\n\t"
            "v_mad_u64_u32 %[b1], %[y4], %[c0], %[q1], %[b1]\n\t"
            "v_mad_u64_u32 %[k1], %[y5], %[c1], %[q1], %[k1]\n\t"
            "v_mad_u64_u32 %[b0], %[y6], %[c1], %[p0], %[b0]\n\t"
            "v_mad_u64_u32 %[b1], %[y7], %[c1], %[q0], %[b1]\n\t"
            "v_addc_co_u32_e64 %[ah0], vcc, %[ah0], 0, %[y0]\n\t"
            "v_addc_co_u32_e64 %[bh0], vcc, %[bh0], 0, %[y1]\n\t"
            "v_addc_co_u32_e64 %[kh0], vcc, %[kh0], 0, %[y2]\n\t"
            "v_addc_co_u32_e64 %[ah1], vcc, %[ah1], 0, %[y3]\n\t"
            "v_addc_co_u32_e64 %[bh1], vcc, %[bh1], 0, %[y4]\n\t"
            "v_addc_co_u32_e64 %[kh1], vcc, %[kh1], 0, %[y5]\n\t"
            "v_addc_co_u32_e64 %[bh0], vcc, %[bh0], 0, %[y6]\n\t"
            "v_addc_co_u32_e64 %[bh1], vcc, %[bh1], 0, %[y7]"
            : [a0] "+v"(a[0]), [b0] "+v"(b[0]), [k0] "+v"(c[0]), [a1] "+v"(a[1]), [b1] "+v"(b[1]), [k1] "+v"(c[1]),
              [ah0] "+v"(ah[0]), [bh0] "+v"(bh[0]), [kh0] "+v"(ch[0]), [ah1] "+v"(ah[1]), [bh1] "+v"(bh[1]), [kh1] "+v"(ch[1]),
              [y0] "=&s"(y0), [y1] "=&s"(y1), [y2] "=&s"(y2), [y3] "=&s"(y3), [y4] "=&s"(y4), [y5] "=&s"(y5), [y6] "=&s"(y6),
              [y7] "=&s"(y7)
            : [c0] "v"(c0), [c1] "v"(c1), [p0] "s"(p0), [p1] "s"(p1), [q0] "s"(q0), [q1] "s"(q1)
            : "vcc");
    }
    // value mod p of challenge i:  A + 2^32 B + 2^64 C  with A = a + 2^64 ah, B = b + 2^64 bh, C = c + 2^64 ch and
    // 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32 (mod p):   a + 2^32 (b + u) - u - w,   u = c + ah,  w = bh + 2^32 ch  (w < 2^64, exact)
    __device__ __forceinline__ uint64_t value(int i) const {
        const uint64_t u = gl::add(c[i], (uint64_t)ah[i]);
        const uint64_t w = ((uint64_t)ch[i] << 32) | bh[i];
        uint64_t r = gl::add(a[i], gl::mul_2exp<32>(gl::add(b[i], u)));
        r = gl::sub(r, u);
        return gl::sub(r, w);
    }
};

// The kernel's arguments as the tasks read them: through a constant-address-space pointer to the kernarg segment that is made
// opaque once per task.  Read as members of the by-value parameter, every scalar a task needs (gate parameters, table
// pointers, Z_H values, interpolation points) is loop-invariant in the task loop, so the compiler loaded ALL of them for ALL
// tasks in the kernel's prologue and kept what did not fit the SGPR file in VGPR lanes (277 v_writelane up front, ~5 % of the
// inner loops' VALU instructions v_readlane).  Behind the laundered pointer a task loads what it needs, when it runs.
typedef const QuotArgs __attribute__((address_space(4))) * QArgsP;
__device__ __forceinline__ Gate load_gate(QArgsP a, int g) {
    Gate r;
    r.id = a->gates[g].id;
    r.selector_index = a->gates[g].selector_index;
    r.group_start = a->gates[g].group_start;
    r.group_end = a->gates[g].group_end;
    r.param = a->gates[g].param;
    return r;
}
__device__ __forceinline__ QuotView quot_view(QArgsP a, size_t k) {
    const size_t N = (size_t)1 << a->lde_bits;
    QuotView v;
    v.w = a->w + k * (size_t)a->num_wires * N;
    v.z = a->z + k * (size_t)a->nch * (1 + a->npp) * N;
    v.apow = a->apow + k * 2 * (size_t)a->n_terms;
    v.bk = a->bk + k * (size_t)a->nch * a->num_routed;
    v.ga = nullptr;
    v.out = a->out + k * (size_t)a->nch * N;
    v.pr = a->proofs + k;
    return v;
}

struct FQ {
    QArgsP a;
    const QuotView& v;     // this workgroup's proof
    const uint64_t* tile;  // [num_wires + num_constants][64]
    int lane;
    size_t p, N;
    int zs;  // a run-time zero, opaque per task: added to constant-table indices so that their reads are not hoisted (poseidon.h zoff)
    __device__ __forceinline__ uint64_t W(int col) const { return tile[col * FQ_POINTS + lane]; }
    __device__ __forceinline__ uint64_t K(int col) const { return tile[(a->num_wires + col) * FQ_POINTS + lane]; }  // cs column
    __device__ __forceinline__ uint64_t filter(int g) const {
        const Gate gt = load_gate(a, g);
        const uint64_t s = K((int)gt.selector_index);
        uint64_t f = 1;
        for (uint32_t r = gt.group_start; r < gt.group_end; ++r)
            if ((int)r != g) f = gl::mul(f, gl::sub((uint64_t)r, s));
        if (a->num_selectors > 1) f = gl::mul(f, gl::sub(0xFFFFFFFFull, s));
        return f;
    }
};

__device__ __forceinline__ void fq_finish(const AlphaAcc& acc, uint64_t filter, uint64_t (&res)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) res[c] = gl::add(res[c], gl::mul(filter, acc.value(c)));
}

// ---- permutation argument, chunks [ck0, ck1) (+ the L_0(x)(Z(x) - 1) terms with the first half) ----
// (rolled loops throughout this file: the six waves of a workgroup run six different code regions at once, and the
// instruction cache two CUs share holds 64 KB — the first, fully unrolled version of this kernel was 150 KB and slower than
// the kernels it replaced)
__device__ __forceinline__ void fq_task_perm(const FQ& f, int ck0, int ck1, bool with_l0, uint64_t (&res)[2]) {
    const QArgsP a = f.a;
    const size_t N = f.N, p = f.p;
    const uint32_t i = __brev((uint32_t)p) >> (32 - a->lde_bits);  // natural index of the point
    const uint64_t ex = (uint64_t)i << (a->K - a->lde_bits);
    const uint64_t x = gl::mul(gl::COSET_SHIFT, gl::mul(a->w_lo[ex & 4095], a->w_hi[ex >> 12]));
    const int n_chunks = a->npp + 1;
    AlphaAcc acc;
    acc.init(f.v.apow, a->n_terms);
    if (with_l0) {
        const int rem = i & ((1 << (a->lde_bits - a->log_n)) - 1);
        const uint64_t l0 = gl::mul(a->zh[rem], gl::inv_compact(gl::mul(a->n_field, gl::sub(x, 1))));
#pragma unroll 1
        for (int c2 = 0; c2 < a->nch; ++c2) acc.mac(gl::mul(l0, gl::sub(f.v.z[(size_t)c2 * N + p], 1)), c2);
    }
    const size_t p_next = __brev((uint32_t)((i + (1u << (a->lde_bits - a->log_n))) & (uint32_t)(N - 1))) >> (32 - a->lde_bits);
#pragma unroll 1
    for (int ck = ck0; ck < ck1; ++ck) {
        uint64_t nump[2] = {1, 1}, denp[2] = {1, 1};
        const int j0 = ck * a->chunk;
        const int j1 = j0 + a->chunk < a->num_routed ? j0 + a->chunk : a->num_routed;
        // the sigma values of the chunk are the only HBM reads of this loop: all of them are issued before the arithmetic
        // (with 4 waves per SIMD a load per iteration left the wave waiting on each one)
        // ... and so are the chunk's four Z / partial-product values, which used to be fetched one challenge at a time right
        // before the constraint that needs them: two more HBM round trips per chunk in a row
        uint64_t zprev[2], znext[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int cc = c2 < a->nch ? c2 : 0;
            zprev[c2] = ck == 0 ? f.v.z[(size_t)cc * N + p] : f.v.z[(size_t)(a->nch + cc * a->npp + ck - 1) * N + p];
            znext[c2] = ck < a->npp ? f.v.z[(size_t)(a->nch + cc * a->npp + ck) * N + p] : f.v.z[(size_t)cc * N + p_next];
        }
        constexpr int PF = 8;
        uint64_t sgv[PF], wvv[PF];
#pragma unroll 1
        for (int jb = j0; jb < j1; jb += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k)
                sgv[k] = jb + k < j1 ? a->cs[(size_t)(a->num_constants + jb + k) * N + p] : 0;
#pragma unroll
            for (int k = 0; k < PF; ++k) wvv[k] = jb + k < j1 ? f.W(jb + k) : 0;  // the wires of the batch: LDS reads in flight together
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int j = jb + k;
                if (j < j1) {
                    const uint64_t wv = wvv[k];
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {  // nch = 1: the second challenge runs on zero parameters and is never read
                        const uint64_t wg = gl::add(wv, f.v.gamma(c2));
                        nump[c2] = gl::mul(nump[c2], gl::add(wg, gl::mul(f.v.bk[c2 * a->num_routed + j], x)));
                        denp[c2] = gl::mul(denp[c2], gl::add(wg, gl::mul(f.v.beta(c2), sgv[k])));
                    }
                }
            }
        }
#pragma unroll 1
        for (int c2 = 0; c2 < a->nch; ++c2) {
            const uint64_t prev = c2 ? zprev[1] : zprev[0], next = c2 ? znext[1] : znext[0];
            acc.mac(gl::sub(gl::mul(prev, c2 ? nump[1] : nump[0]), gl::mul(next, c2 ? denp[1] : denp[0])),
                    a->nch + c2 * n_chunks + ck);
        }
    }
    fq_finish(acc, 1, res);
}

// ---- PoseidonGate (gates/poseidon.rs::eval_unfiltered_base): every round restarts from its S-box-input wires, so the
// gate splits into independent pieces.  Constraint order: 0 swap bit, 1..4 deltas, 5..40 full rounds 1..3, 41..62 partial
// rounds, 63..110 second full rounds, 111..122 outputs.
// Full-round pieces 0..6: piece q takes the state entering the S-boxes of round R(q) = q (q < 3) or 26 + q - 3 from wires
// (piece 0: from the input wires through the swap), applies S-boxes and the MDS layer (+ next round's constants) and compares
// with the wires the next round starts from (the output wires for the last one).  ONE copy of the round code serves all seven.
__device__ __forceinline__ void fq_poseidon_full_pieces(const FQ& f, AlphaAcc& acc, int q0, int q1, int t_gates) {
#pragma unroll 1
    for (int q = q0; q < q1; ++q) {
        uint64_t st[12];
        const int round = q < 3 ? q : 23 + q;                      // 0,1,2, 26,27,28,29
        const int src = q < 3 ? 29 + 12 * (q - 1) : 87 + 12 * (q - 3);   // S-box-input wires of this round (q = 0: see below)
        const int cmp = q < 3 ? 29 + 12 * q : (q < 6 ? 87 + 12 * (q - 2) : 12);
        const int k0 = t_gates + (q < 3 ? 5 + 12 * q : 63 + 12 * (q - 2));
        if (q == 0) {
            const uint64_t swap = f.W(24);
            acc.mac(gl::mul(swap, gl::sub(swap, 1)), t_gates);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t lhs = f.W(i), rhs = f.W(i + 4), delta = f.W(25 + i);
                acc.mac(gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta), t_gates + 1 + i);
                st[i] = gl::add(lhs, delta);
                st[i + 4] = gl::sub(rhs, delta);
            }
#pragma unroll
            for (int i = 8; i < 12; ++i) st[i] = f.W(i);
#pragma unroll
            for (int i = 0; i < 12; ++i) st[i] = gl::add(st[i], poseidon::RC[i + f.zs]);
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) st[i] = f.W(src + i);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = poseidon::sbox7(st[i]);
        poseidon::mds_layer_add(st, &poseidon::RC[(round + 1) * 12 + f.zs], f.zs);
#pragma unroll
        for (int i = 0; i < 12; ++i) acc.mac(gl::sub(st[i], f.W(cmp + i)), k0 + i);
    }
}

__device__ __forceinline__ void fq_task_poseidon(const FQ& f, bool partial, int lo, int hi, int g, int t_gates, uint64_t (&res)[2]) {
    AlphaAcc acc;
    acc.init(f.v.apow, f.a->n_terms);
    if (partial) {  // round 3's S-boxes, its MDS layer fused with partial round 4, partial rounds 5..25
        uint64_t st[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = poseidon::sbox7(f.W(53 + i));
        int pr = 0;
        auto wire_in = [&](uint64_t computed) {
            const uint64_t sin = f.W(65 + pr);
            acc.mac(gl::sub(computed, sin), t_gates + 41 + pr);
            ++pr;
            return sin;
        };
        poseidon::mds_partial1_fused<decltype(wire_in)&, true>(st, wire_in, f.zs);
#pragma unroll 1
        for (int grp = 0; grp < 7; ++grp)  // the last group is followed by a full constant vector, the others by one scalar
            poseidon::partial_rounds3_rt(st, poseidon::RC_PARTIAL_S0[3 * grp + 1 + f.zs], poseidon::RC_PARTIAL_S0[3 * grp + 2 + f.zs],
                                         grp < 6 ? &poseidon::RC_PARTIAL_S0[3 * grp + 3 + f.zs] : &poseidon::RC_AFTER_PARTIAL[f.zs], grp < 6,
                                         wire_in, f.zs);
#pragma unroll
        for (int i = 0; i < 12; ++i) acc.mac(gl::sub(st[i], f.W(87 + i)), t_gates + 63 + i);
    } else {
        fq_poseidon_full_pieces(f, acc, lo, hi, t_gates);
    }
    fq_finish(acc, f.filter(g), res);
}

// ---- every other gate: the constraints of quotient_kernel, wires and constants from the LDS tile ----
__device__ __forceinline__ void fq_task_gate(const FQ& f, int g, int lo, int hi, int t_gates, uint64_t (&res)[2]) {
    const QArgsP a = f.a;
    const Gate gt = load_gate(a, g);
    AlphaAcc acc;
    acc.init(f.v.apow, a->n_terms);
    const int nsel = a->num_selectors;
    auto emit = [&](uint64_t cons, int k) { acc.mac(cons, t_gates + k); };
    if (gt.id == GATE_CONSTANT) {
#pragma unroll 1
        for (int k = lo; k < hi; ++k) emit(gl::sub(f.K(nsel + k), f.W(k)), k);
    } else if (gt.id == GATE_PUBLIC_INPUT) {
#pragma unroll 1
        for (int k = lo; k < hi; ++k) emit(gl::sub(f.W(k), f.v.pi_hash(k)), k);
    } else if (gt.id == GATE_ARITHMETIC) {
        const uint64_t c0 = f.K(nsel), c1 = f.K(nsel + 1);
#pragma unroll GL_FQ_UNROLL
        for (int k = lo; k < hi; ++k) {
            const uint64_t comp = gl::add(gl::mul(gl::mul(f.W(4 * k), f.W(4 * k + 1)), c0), gl::mul(f.W(4 * k + 2), c1));
            emit(gl::sub(f.W(4 * k + 3), comp), k);
        }
    } else if (gt.id == GATE_BASE_SUM) {
        if (lo == 0) {  // the sum constraint rides with the first range
            uint64_t sum = 0;
#pragma unroll 1
            for (int k = (int)gt.param - 1; k >= 0; --k) sum = gl::add(gl::add(sum, sum), f.W(1 + k));
            emit(gl::sub(sum, f.W(0)), 0);
        }
#pragma unroll GL_FQ_UNROLL
        for (int k = lo; k < hi; ++k) {
            const uint64_t limb = f.W(1 + k);
            emit(gl::mul(limb, gl::sub(limb, 1)), 1 + k);
        }
    } else if (gt.id == GATE_ARITHMETIC_EXT || gt.id == GATE_MUL_EXT) {
        const bool arith = gt.id == GATE_ARITHMETIC_EXT;
        const int per = arith ? 8 : 6;
        const uint64_t c0 = f.K(nsel), c1 = f.K(nsel + 1);
#pragma unroll GL_FQ_UNROLL
        for (int k = lo; k < hi; ++k) {
            const int q = per * k;
            const gl::ext2 m0 = {f.W(q), f.W(q + 1)}, m1 = {f.W(q + 2), f.W(q + 3)};
            gl::ext2 pr = gl::ext_mul_base(gl::ext_mul(m0, m1), c0);
            gl::ext2 out = {f.W(q + 4), f.W(q + 5)};
            if (arith) {
                pr = gl::ext_add(pr, gl::ext_mul_base(out, c1));  // wires q + 4, q + 5 are the addend here
                out = gl::ext2{f.W(q + 6), f.W(q + 7)};
            }
            emit(gl::sub(out.a, pr.a), 2 * k);
            emit(gl::sub(out.b, pr.b), 2 * k + 1);
        }
    } else if (gt.id == GATE_RANDOM_ACCESS) {
        const int bits = gt.param & 0xFF, copies = (gt.param >> 8) & 0xFF, extra = (gt.param >> 16) & 0xFF;
        const int vec = 1 << bits, routed = (2 + vec) * copies + extra;
        int k = lo * (bits + 2);
#pragma unroll 1
        for (int cp = lo; cp < hi; ++cp) {
            const int q = (2 + vec) * cp, b = routed + cp * bits;
            uint64_t rec = 0;
#pragma unroll 1
            for (int i = bits - 1; i >= 0; --i) rec = gl::add(gl::add(rec, rec), f.W(b + i));
#pragma unroll 1
            for (int i = 0; i < bits; ++i) {
                const uint64_t bi = f.W(b + i);
                emit(gl::mul(bi, gl::sub(bi, 1)), k++);
            }
            emit(gl::sub(rec, f.W(q)), k++);
            uint64_t items[MAX_INTERP_POINTS];
#pragma unroll
            for (int j = 0; j < MAX_INTERP_POINTS; ++j) items[j] = j < vec ? f.W(q + 2 + j) : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < bits) {
                    const uint64_t bi = f.W(b + i);
#pragma unroll
                    for (int j = 0; j < (MAX_INTERP_POINTS / 2) >> i; ++j)
                        items[j] = gl::add(items[2 * j], gl::mul(bi, gl::sub(items[2 * j + 1], items[2 * j])));
                }
            emit(gl::sub(items[0], f.W(q + 1)), k++);
        }
        if (hi == copies) {  // the extra constants ride with the last range
#pragma unroll 1
            for (int i = 0; i < extra; ++i) emit(gl::sub(f.K(nsel + i), f.W((2 + vec) * copies + i)), k++);
        }
    } else if (gt.id == GATE_REDUCING || gt.id == GATE_REDUCING_EXT) {
        const bool is_ext = gt.id == GATE_REDUCING_EXT;
        const int nc = (int)gt.param, start_accs = 6 + (is_ext ? 2 * nc : nc);
        const gl::ext2 alpha = {f.W(2), f.W(3)};
        const int pa = lo == 0 ? 4 : start_accs + 2 * (lo - 1);  // the accumulator before coefficient lo: old_acc or a wire
        gl::ext2 acc2 = {f.W(pa), f.W(pa + 1)};
#pragma unroll GL_FQ_UNROLL
        for (int i = lo; i < hi; ++i) {
            const gl::ext2 coeff = is_ext ? gl::ext2{f.W(6 + 2 * i), f.W(7 + 2 * i)} : gl::ext2{f.W(6 + i), 0};
            const int nx = i == nc - 1 ? 0 : start_accs + 2 * i;
            const gl::ext2 next = {f.W(nx), f.W(nx + 1)};
            const gl::ext2 d = gl::ext_sub(gl::ext_add(gl::ext_mul(acc2, alpha), coeff), next);
            emit(d.a, 2 * i);
            emit(d.b, 2 * i + 1);
            acc2 = next;
        }
    } else if (gt.id == GATE_EXPONENTIATION) {
        const int nb = (int)gt.param;
        const uint64_t base = f.W(0);
        uint64_t last = lo == 0 ? 0 : f.W(2 + nb + lo - 1);
#pragma unroll GL_FQ_UNROLL
        for (int i = lo; i < hi; ++i) {
            const uint64_t prev = i == 0 ? 1 : gl::mul(last, last);
            const uint64_t bit = f.W(1 + nb - 1 - i);
            const uint64_t factor = gl::add(gl::mul(bit, base), gl::sub(1, bit));
            last = f.W(2 + nb + i);
            emit(gl::sub(gl::mul(prev, factor), last), i);
        }
        if (hi == nb) emit(gl::sub(f.W(1 + nb), last), nb);
    } else if (gt.id == GATE_POSEIDON_MDS) {
        // the MDS layer on 12 extension elements = on 24 base-field columns: 32-bit halves x small constants in u64
        // accumulators, as the permutation's own layer; rolled over the 24 outputs, the inputs re-read from the LDS tile
#pragma unroll 1
        for (int o = lo; o < hi; ++o) {
            const int r = o >> 1, comp = o & 1;
            uint64_t alo = 0, ahi = 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                int src = i + r;
                src = src >= 12 ? src - 12 : src;
                const uint64_t v = f.W(2 * src + comp);
                alo += (uint64_t)(uint32_t)v * poseidon::MDS_CIRC_TAB[i + f.zs];
                ahi += (v >> 32) * poseidon::MDS_CIRC_TAB[i + f.zs];
            }
            if (r == 0) {
                const uint64_t v = f.W(comp);
                alo += (uint64_t)(uint32_t)v * 8u;
                ahi += (v >> 32) * 8u;
            }
            emit(gl::sub(f.W(24 + o), poseidon::mds_combine(alo, ahi)), o);
        }
    } else if (gt.id == GATE_COSET_INTERPOLATION) {
        const int bits = gt.param & 0xFF, degree = (gt.param >> 8) & 0xFF, npts = 1 << bits;
        const int ni = (npts - 2) / (degree - 1);
        const int sp = 1 + 2 * npts, sv = sp + 2, si = sv + 2;
        const uint64_t shift = f.W(0);
        const gl::ext2 point = {f.W(sp), f.W(sp + 1)};
        const gl::ext2 shifted = {f.W(si + 4 * ni), f.W(si + 4 * ni + 1)};
        int k = 0;
        gl::ext2 d = gl::ext_sub(point, gl::ext_mul_base(shifted, shift));
        emit(d.a, k++);
        emit(d.b, k++);
        gl::ext2 eval = {0, 0}, prod = {1, 0};
        auto run = [&](int from, int to) {
#pragma unroll 1
            for (int j = from; j < to; ++j) {
                const gl::ext2 term = {gl::sub(shifted.a, a->interp_dom[j]), shifted.b};
                const gl::ext2 val = gl::ext_mul_base(gl::ext2{f.W(1 + 2 * j), f.W(2 + 2 * j)}, a->interp_wts[j]);
                const gl::ext2 ne = gl::ext_add(gl::ext_mul(eval, term), gl::ext_mul(val, prod));
                prod = gl::ext_mul(prod, term);
                eval = ne;
            }
        };
#pragma unroll 1
        for (int i = -1; i < ni; ++i) {  // i = -1: the first `degree` points from (0, 1); then one chunk per intermediate pair
            int from = 0, to = degree;
            if (i >= 0) {
                const gl::ext2 ie = {f.W(si + 2 * i), f.W(si + 2 * i + 1)};
                const gl::ext2 ip = {f.W(si + 2 * (ni + i)), f.W(si + 2 * (ni + i) + 1)};
                d = gl::ext_sub(ie, eval);
                emit(d.a, k++);
                emit(d.b, k++);
                d = gl::ext_sub(ip, prod);
                emit(d.a, k++);
                emit(d.b, k++);
                from = 1 + (degree - 1) * (i + 1);
                to = from + degree - 1 < npts ? from + degree - 1 : npts;
                eval = ie;
                prod = ip;
            }
            run(from, to);
        }
        d = gl::ext_sub(gl::ext2{f.W(sv), f.W(sv + 1)}, eval);
        emit(d.a, k++);
        emit(d.b, k++);
    }
    fq_finish(acc, f.filter(g), res);
}

__global__ __launch_bounds__(FQ_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void quotient_fused_kernel(FusedArgs fa) {
    extern __shared__ uint64_t fq_lds[];
#ifdef GL_FQ_TIMELINE  // build-time debug option (GL_EXTRA_FLAGS=-DGL_FQ_TIMELINE): even unused, the three time stamps cost the
    // kernel a register and 8 B of scratch (19.3 -> 19.7 ms), so they are not in the default build
    const unsigned long long tl_entry = fa.task_cycles ? __builtin_readcyclecounter() : 0;
#endif
    const QuotArgs& a = fa.q;
    const size_t N = (size_t)1 << a.lde_bits;
    const int lane = threadIdx.x & (FQ_POINTS - 1);
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / FQ_POINTS));  // wave-uniform: scalar branches below
    const size_t p = (size_t)blockIdx.x * FQ_POINTS + lane;  // N is a multiple of 64 (the host checks lde_bits >= 6)
    const QuotView v = quot_view(a, blockIdx.y);                // blockIdx.y = proof of the batch
    uint64_t* tile = fq_lds;
    uint64_t* partial = fq_lds;  // [role][2][64], over the tile once all waves have left the task loop
    const size_t tile_words = (size_t)(a.num_wires + a.num_constants) * FQ_POINTS, part_words = (size_t)FQ_ROLES * 2 * FQ_POINTS;
    // every wire column and the selector / constant columns, once
    {   // loads in batches of 10 columns per wave, all issued before the first LDS write (one load per iteration left the
        // whole workgroup waiting ~18 memory latencies in a row before its first constraint).  Measured in round 3 (GLP_FQ_DEBUG
        // timeline, mid-launch workgroups): the fill is 6.4-7 % of a workgroup's life (~13 k cycles), the wait behind the task
        // loop 5.1 %; all 18 columns of a wave in ONE batch at raised priority (s_setprio 3) left the fill at 12.7 k cycles and
        // cost a register (128 VGPRs, 8 B of scratch): 19.7 ms instead of 19.3 — the fill is not the loads' round trips.
        constexpr int LB = 10;
        const int n_cols = a.num_wires + a.num_constants;
#pragma unroll 1
        for (int c0 = role; c0 < n_cols; c0 += LB * FQ_ROLES) {
            uint64_t ld[LB];
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const int col = c0 + k * FQ_ROLES;
                if (col < n_cols)
                    ld[k] = col < a.num_wires ? v.w[(size_t)col * N + p] : a.cs[(size_t)(col - a.num_wires) * N + p];
            }
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const int col = c0 + k * FQ_ROLES;
                if (col < n_cols) tile[col * FQ_POINTS + lane] = ld[k];
            }
        }
    }
    unsigned* next_task = reinterpret_cast<unsigned*>(fq_lds + (tile_words > part_words ? tile_words : part_words));
    if (threadIdx.x == 0) *next_task = 0;
    __syncthreads();
#ifdef GL_FQ_TIMELINE
    // GLP_FQ_DEBUG timeline of a workgroup (slots behind the per-task cycles, summed over the sampled workgroups):
    //   +0 fill (entry -> tile ready), +1 sum over the waves of their time in the task loop, +2 sum over the waves of their wait
    //   at the barrier behind it (the tail), +3 workgroup lifetime, +4 samples
    const bool tl = fa.task_cycles && (gridDim.x >= 131072 ? (blockIdx.x >> 10) == 64 : blockIdx.x < 1024) && lane == 0;  // mid-launch workgroups at full size: steady state, not the first burst
    const unsigned long long tl_fill = fa.task_cycles ? __builtin_readcyclecounter() : 0;
#endif
    const int t_gates = a.nch + a.nch * (a.npp + 1);
    uint64_t res[2] = {0, 0};
    int t = -1;
#pragma unroll 1
    for (;;) {
        // Which task next.  Dynamic: the tasks are listed longest first and a wave takes the next one when it becomes free, so
        // the eight waves finish within one short task of each other whatever the cost model got wrong (the static deal left
        // the SIMDs idle behind the slowest role: VALUBusy 83 %).
        if (fa.dynamic) {
            unsigned got = 0;
            if (lane == 0) got = atomicAdd(next_task, 1u);
            t = __builtin_amdgcn_readfirstlane((int)got);
        } else {
            ++t;
        }
        if (t >= fa.n_tasks) break;
        const FqTask tk = fa.tasks[t];
        if (!fa.dynamic && tk.role != role) continue;
        // The lane index is made opaque once per task: everything a task derives from it (LDS addresses, the point's x, filter
        // values, first wire reads) would otherwise be hoisted out of this loop by the compiler — for ALL tasks at once, ten
        // VGPRs of which ended up in scratch (2.8 GB of HBM writes per 2^20-row launch) and 277 SGPRs in VGPR lanes.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        QArgsP ap = (QArgsP)__builtin_amdgcn_kernarg_segment_ptr();  // FusedArgs is the only parameter and `q` its first member
        asm volatile("" : "+s"(ap));
        const QuotView tv = quot_view(ap, blockIdx.y);
        int zs = 0;
        asm volatile("" : "+s"(zs));
        const FQ f{ap, tv, tile, ln, (size_t)blockIdx.x * FQ_POINTS + ln, N, zs};
        const unsigned long long t_begin = fa.task_cycles ? __builtin_readcyclecounter() : 0;
        if (tk.kind == FQ_KIND_PERM)
            fq_task_perm(f, tk.lo, tk.hi, tk.lo == 0, res);
        else if (tk.kind == FQ_KIND_GATE)
            fq_task_gate(f, tk.gate, tk.lo, tk.hi, t_gates, res);
        else
            fq_task_poseidon(f, tk.kind == FQ_KIND_POS_PARTIAL, tk.lo, tk.hi, tk.gate, t_gates, res);
        if (fa.task_cycles && blockIdx.x < 1024 && lane == 0)
            atomicAdd(fa.task_cycles + t, __builtin_readcyclecounter() - t_begin + (res[0] & 1 ? 0 : 0));
    }
#ifdef GL_FQ_TIMELINE
    const unsigned long long tl_loop = fa.task_cycles ? __builtin_readcyclecounter() : 0;
#endif
    __syncthreads();  // nobody reads the tile any more
#ifdef GL_FQ_TIMELINE
    if (tl) {
        const unsigned long long now = __builtin_readcyclecounter();
        atomicAdd(fa.task_cycles + FQ_MAX_TASKS + 1, tl_loop - tl_fill);
        atomicAdd(fa.task_cycles + FQ_MAX_TASKS + 2, now - tl_loop);
        if (role == 0) {
            atomicAdd(fa.task_cycles + FQ_MAX_TASKS + 0, tl_fill - tl_entry);
            atomicAdd(fa.task_cycles + FQ_MAX_TASKS + 3, now - tl_entry);
            atomicAdd(fa.task_cycles + FQ_MAX_TASKS + 4, 1ull);
        }
    }
#endif
    partial[(role * 2 + 0) * FQ_POINTS + lane] = res[0];
    partial[(role * 2 + 1) * FQ_POINTS + lane] = res[1];
    __syncthreads();
    if (role < a.nch) {  // wave c finishes challenge c
        uint64_t s = 0;
#pragma unroll
        for (int r = 0; r < FQ_ROLES; ++r) s = gl::add(s, partial[(r * 2 + role) * FQ_POINTS + lane]);
        const uint32_t i = __brev((uint32_t)p) >> (32 - a.lde_bits);
        const int rem = i & ((1 << (a.lde_bits - a.log_n)) - 1);
        v.out[(size_t)role * N + p] = gl::canon(gl::mul(s, a.zh_inv[rem]));
    }
}

// ---- the persistent form for large evaluations (round 3) ----
// ONE workgroup of sixteen waves per CU walks the tiles t = blockIdx.x, blockIdx.x + gridDim.x, ...  Two tile buffers: while the
// constraints of tile k run out of buffer k & 1, tile k + 1 is already in the other one, and tile k + 2 is brought into buffer
// k & 1 by LDS-DMA (global_load_lds_dwordx4: no register staging) as soon as the last wave has left tile k.  There is NO barrier
// per tile: a wave that finds no task left in tile k deposits its partial sums (LDS atomics on 32-bit halves: exact, order-free),
// and goes on to the tasks of tile k + 1; the wave that happens to finish a tile last reduces it, stores the quotient values, and
// issues the DMA.  So no wave waits for a fill (the one-tile kernel: every wave, 6-7 % of its life) or behind the slowest wave
// of its workgroup (5 %) — the two idle shares that kept the one-tile kernel at 85 % VALUBusy.
constexpr int FQP_ROLES = 16;
constexpr int FQP_THREADS = FQ_POINTS * FQP_ROLES;
// LDS: [2][tile]  +  acc[2 buffers][2 challenges][2 halves][64]  +  control words
constexpr int FQP_CTL_WORDS = 8;  // u64 words holding the unsigned counters: next_task[2], done[2], ready[2]
inline size_t fqp_lds_bytes(int num_wires, int num_constants) {
    return ((size_t)2 * (num_wires + num_constants) * FQ_POINTS + 2 * 2 * 2 * FQ_POINTS + FQP_CTL_WORDS) * sizeof(uint64_t);
}

__global__ __launch_bounds__(FQP_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void quotient_fused_persistent_kernel(FusedArgs fa) {
    extern __shared__ uint64_t fq_lds[];
    const QuotArgs& a = fa.q;
    const size_t N = (size_t)1 << a.lde_bits;
    const int lane = threadIdx.x & (FQ_POINTS - 1);
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / FQ_POINTS));
    const int n_cols = a.num_wires + a.num_constants;
    const unsigned tile_words = (unsigned)n_cols * FQ_POINTS;
    const unsigned tiles_per_proof = (unsigned)(N / FQ_POINTS), n_total = tiles_per_proof * fa.n_proofs;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(fq_lds + 2 * (size_t)tile_words);  // [buf][challenge][half][64]
    unsigned* ctl = reinterpret_cast<unsigned*>(acc + 2 * 2 * 2 * FQ_POINTS);
    unsigned* next_task = ctl;      // [2]
    unsigned* done = ctl + 2;       // [2]
    volatile unsigned* ready = ctl + 4;  // [2]: 1 + the local index k of the tile that is complete in the buffer

    // a tile = 64 consecutive points of every wire / constant column: one wave-instruction moves 1 KB = two columns (lanes 0..31
    // bring column c, two points each, lanes 32..63 column c + 1); the LDS destination is the wave-uniform base + 16 * lane, i.e.
    // exactly rows c and c + 1 of the [column][64] tile
    auto fetch = [&](unsigned T, unsigned buf, int first_pair, int pair_step) {
        const unsigned proof = T / tiles_per_proof;
        const size_t p0 = (size_t)(T - proof * tiles_per_proof) * FQ_POINTS;
        const uint64_t* w = a.w + (size_t)proof * a.num_wires * N;
        const int half = lane >> 5, sub = lane & 31;
#pragma unroll 1
        for (int pair = first_pair; 2 * pair < n_cols; pair += pair_step) {
            const int col = 2 * pair + half;
            if (col < n_cols) {
                const uint64_t* src = (col < a.num_wires ? w + (size_t)col * N : a.cs + (size_t)(col - a.num_wires) * N) + p0 + 2 * sub;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(fq_lds + (size_t)buf * tile_words + (size_t)2 * pair * FQ_POINTS),
                                                 16, 0, 0);
            }
        }
    };
    if (threadIdx.x < 2 * 2 * 2 * FQ_POINTS) acc[threadIdx.x] = 0;  // 512 words, 1024 threads
    if (threadIdx.x < 4) ctl[threadIdx.x] = 0;
    const unsigned G = gridDim.x, T0 = blockIdx.x;
    if (T0 < n_total) fetch(T0, 0, role, FQP_ROLES);
    if (T0 + G < n_total) fetch(T0 + G, 1, role, FQP_ROLES);
    __builtin_amdgcn_s_waitcnt(0);  // this wave's pieces have landed
    if (threadIdx.x == 0) {
        ready[0] = 1;
        ready[1] = 2;
    }
    __syncthreads();

#pragma unroll 1
    for (unsigned k = 0;; ++k) {
        const unsigned T = T0 + k * G;
        if (T >= n_total) break;
        const unsigned par = k & 1;
        while (ready[par] != k + 1) __builtin_amdgcn_s_sleep(4);  // in steady state the tile has been there for a tile's time
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned proof = T / tiles_per_proof;
        const unsigned tile_idx = T - proof * tiles_per_proof;
        uint64_t res[2] = {0, 0};
        const int t_gates = a.nch + a.nch * (a.npp + 1);
#pragma unroll 1
        for (;;) {
            unsigned got = 0;
            if (lane == 0) got = atomicAdd(next_task + par, 1u);
            const int t = __builtin_amdgcn_readfirstlane((int)got);
            if (t >= fa.n_tasks) break;
            const FqTask tk = fa.tasks[t];
            // everything a task derives from these is made opaque once per task (see the one-tile kernel)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            QArgsP ap = (QArgsP)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ap));
            unsigned toff = par * tile_words;
            asm volatile("" : "+s"(toff));
            const QuotView tv = quot_view(ap, proof);
            int zs = 0;
            asm volatile("" : "+s"(zs));
            const FQ f{ap, tv, fq_lds + toff, ln, (size_t)tile_idx * FQ_POINTS + ln, N, zs};
            const unsigned long long t_begin = fa.task_cycles ? __builtin_readcyclecounter() : 0;
            if (tk.kind == FQ_KIND_PERM)
                fq_task_perm(f, tk.lo, tk.hi, tk.lo == 0, res);
            else if (tk.kind == FQ_KIND_GATE)
                fq_task_gate(f, tk.gate, tk.lo, tk.hi, t_gates, res);
            else
                fq_task_poseidon(f, tk.kind == FQ_KIND_POS_PARTIAL, tk.lo, tk.hi, tk.gate, t_gates, res);
            if (fa.task_cycles && k == 1 && lane == 0)
                atomicAdd(fa.task_cycles + t, __builtin_readcyclecounter() - t_begin + (res[0] & 1 ? 0 : 0));
        }
        // deposit: sums of 32-bit halves stay far below 2^64 for sixteen waves, so plain integer atomics are exact
        unsigned long long* ac = acc + (size_t)par * 2 * 2 * FQ_POINTS;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            atomicAdd(ac + (c2 * 2 + 0) * FQ_POINTS + lane, (unsigned long long)(uint32_t)res[c2]);
            atomicAdd(ac + (c2 * 2 + 1) * FQ_POINTS + lane, (unsigned long long)(res[c2] >> 32));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the deposits are in LDS before this wave counts as done
        unsigned d = 0;
        if (lane == 0) d = atomicAdd(done + par, 1u);
        if (__builtin_amdgcn_readfirstlane((int)d) != FQP_ROLES - 1) continue;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the last wave out of tile k: reduce, divide by Z_H, store; recycle the buffer
        {
            const size_t p = (size_t)tile_idx * FQ_POINTS + lane;
            const uint32_t i = __brev((uint32_t)p) >> (32 - a.lde_bits);
            const int rem = i & ((1 << (a.lde_bits - a.log_n)) - 1);
            const uint64_t zi = a.zh_inv[rem];
            uint64_t* out = a.out + (size_t)proof * a.nch * N;
#pragma unroll 1
            for (int c2 = 0; c2 < a.nch; ++c2) {
                const uint64_t lo = ac[(c2 * 2 + 0) * FQ_POINTS + lane], hi = ac[(c2 * 2 + 1) * FQ_POINTS + lane];
                const uint64_t sum = gl::add(lo, gl::mul(hi, (uint64_t)1 << 32));
                out[(size_t)c2 * N + p] = gl::canon(gl::mul(sum, zi));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) ac[j * FQ_POINTS + lane] = 0;
            if (lane == 0) {
                next_task[par] = 0;
                done[par] = 0;
            }
            if (T + 2 * G < n_total) {
                fetch(T + 2 * G, par, 0, 1);
                __builtin_amdgcn_s_waitcnt(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) ready[par] = k + 3;
        }
    }
}

}  // namespace plonk
