#define _POSIX_C_SOURCE 200809L
/*
 * plonk.c — CPU ORACLE (test infrastructure): the PLONK layer of plonky2's prover and its verifier.
 *
 * Restates plonky2 @ f99ed9c src/plonk/prover.rs (prove, wires_permutation_partial_products_and_zs,
 * compute_quotient_polys), src/plonk/vanishing_poly.rs (eval_vanishing_poly[_base_batch], evaluate_gate_constraints),
 * src/plonk/plonk_common.rs (eval_l_0, reduce_with_powers), src/plonk/permutation_argument / check_partial_products,
 * src/plonk/verifier.rs (verify_with_challenges), src/plonk/get_challenges.rs and the gates
 * src/gates/{noop,constant,public_input,arithmetic_base,poseidon}.rs.  The dependency is absent from /root/reference
 * (Cargo.toml:12); the reference drives it through CircuitData::prove / verify
 * (src/rollup/circuits/mod.rs:1247,1321; src/zkdsa/circuits/mod.rs:326).
 *
 * Gate set: NoopGate, ConstantGate, PublicInputGate, ArithmeticGate, PoseidonGate (the reference's traces are dominated
 * by PoseidonGate rows), BaseSumGate<2>, ArithmeticExtensionGate, MulExtensionGate, and the six gates
 * builder.verify_proof adds (reference src/recursion/gadgets/mod.rs:114-127): RandomAccessGate, ReducingGate,
 * ReducingExtensionGate, ExponentiationGate, PoseidonMdsGate, CosetInterpolationGate (src/gates/{random_access,reducing,
 * reducing_extension,exponentiation,poseidon_mds,coset_interpolation}.rs).  PoseidonGate's partial
 * rounds are evaluated in the textbook round form; with the S-box inputs substituted from the wires this is the same
 * polynomial as plonky2's "fast" parameterisation (both are exact linear re-arrangements between substitutions).
 *
 * PARITY STATUS: "parity unpinned" (no vectors in the reference for this layer).  Checked by self-consistency:
 * glo_plonk_verify accepts proofs of glo_plonk_prove and of the GPU prover, rejects corrupted ones, and rejects proofs
 * of unsatisfied witnesses.
 */
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#include "fast.h"
#include "gl_oracle.h"

typedef struct {
    uint64_t a, b;
} ext;

static ext E(uint64_t a) {
    ext r = {a, 0};
    return r;
}
static ext eadd(ext x, ext y) {
    ext r = {glo_add(x.a, y.a), glo_add(x.b, y.b)};
    return r;
}
static ext esub(ext x, ext y) {
    ext r = {glo_sub(x.a, y.a), glo_sub(x.b, y.b)};
    return r;
}
static ext emul(ext x, ext y) {
    if ((x.b | y.b) == 0) { /* base-field operands (every product of the prover's quotient evaluation; values are canonical) */
        ext r0 = {glo_mul(x.a, y.a), 0};
        return r0;
    }
    uint64_t xx[2] = {x.a, x.b}, yy[2] = {y.a, y.b}, o[2];
    glo_ext_mul(xx, yy, o);
    ext r = {o[0], o[1]};
    return r;
}
static ext emulb(ext x, uint64_t s) {
    ext r = {glo_mul(x.a, s), x.b ? glo_mul(x.b, s) : 0};
    return r;
}
static ext einv(ext x) {
    uint64_t xx[2] = {x.a, x.b}, o[2];
    glo_ext_inv(xx, o);
    ext r = {o[0], o[1]};
    return r;
}
static int eeq(ext x, ext y) { return glo_canon(x.a) == glo_canon(y.a) && glo_canon(x.b) == glo_canon(y.b); }
static ext epow2k(ext x, unsigned k) {
    for (unsigned i = 0; i < k; i++) x = emul(x, x);
    return x;
}

/* ------------------------------------------------ gates ------------------------------------------------ */
static const uint64_t MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

static ext esbox(ext x) {
    ext x2 = emul(x, x), x4 = emul(x2, x2), x3 = emul(x, x2);
    return emul(x3, x4);
}
static void emds(ext st[12]) {
    ext out[12];
    for (int r = 0; r < 12; r++) {
        ext acc = {0, 0};
        for (int i = 0; i < 12; i++) acc = eadd(acc, emulb(st[(i + r) % 12], MDS_CIRC[i]));
        if (r == 0) acc = eadd(acc, emulb(st[0], 8));
        out[r] = acc;
    }
    memcpy(st, out, sizeof out);
}

/* PoseidonGate wire layout (gates/poseidon.rs) */
#define PG_IN(i) (i)
#define PG_OUT(i) (12 + (i))
#define PG_SWAP 24
#define PG_DELTA(i) (25 + (i))
#define PG_FULL0(r, i) (29 + 12 * ((r)-1) + (i))
#define PG_PARTIAL(r) (65 + (r))
#define PG_FULL1(r, i) (87 + 12 * (r) + (i))

static unsigned poseidon_gate_eval(const ext *w, ext *c) {
    const uint64_t *rc = glo_poseidon_round_constants();
    unsigned k = 0;
    ext swap = w[PG_SWAP];
    c[k++] = emul(swap, esub(swap, E(1)));
    for (int i = 0; i < 4; i++) c[k++] = esub(emul(swap, esub(w[PG_IN(i + 4)], w[PG_IN(i)])), w[PG_DELTA(i)]);
    ext st[12];
    for (int i = 0; i < 4; i++) {
        st[i] = eadd(w[PG_IN(i)], w[PG_DELTA(i)]);
        st[i + 4] = esub(w[PG_IN(i + 4)], w[PG_DELTA(i)]);
    }
    for (int i = 8; i < 12; i++) st[i] = w[PG_IN(i)];
    unsigned round = 0;
    for (int r = 0; r < 4; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = eadd(st[i], E(rc[round * 12 + i]));
        if (r != 0)
            for (int i = 0; i < 12; i++) {
                c[k++] = esub(st[i], w[PG_FULL0(r, i)]);
                st[i] = w[PG_FULL0(r, i)];
            }
        for (int i = 0; i < 12; i++) st[i] = esbox(st[i]);
        emds(st);
    }
    for (int r = 0; r < 22; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = eadd(st[i], E(rc[round * 12 + i]));
        c[k++] = esub(st[0], w[PG_PARTIAL(r)]);
        st[0] = esbox(w[PG_PARTIAL(r)]);
        emds(st);
    }
    for (int r = 0; r < 4; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = eadd(st[i], E(rc[round * 12 + i]));
        for (int i = 0; i < 12; i++) {
            c[k++] = esub(st[i], w[PG_FULL1(r, i)]);
            st[i] = w[PG_FULL1(r, i)];
        }
        for (int i = 0; i < 12; i++) st[i] = esbox(st[i]);
        emds(st);
    }
    for (int i = 0; i < 12; i++) c[k++] = esub(st[i], w[PG_OUT(i)]);
    return k; /* 123 */
}

/* circuit_builder.rs::build — circuit_digest under the composition selected by the transcript-variant flags (gl_oracle.h):
 * f99ed9c hashes the constants_sigmas cap alone ("TODO: This should also include an encoding of gate constraints"); later
 * revisions append degree_bits, then a domain-separator digest (hash_pad of an empty separator by default) as well. */
void glo_circuit_digest(const uint64_t *cap, size_t cap_words, unsigned degree_bits, unsigned flags, uint64_t out[4]) {
    uint64_t *buf = (uint64_t *)malloc((cap_words + 5) * 8);
    size_t len = cap_words;
    memcpy(buf, cap, cap_words * 8);
    const unsigned mode = flags & GLO_FLAG_DIGEST_MASK;
    if (mode == GLO_FLAG_DIGEST_CAP_DOM_DEG) {
        glo_hash_pad(NULL, 0, buf + len);
        len += 4;
    }
    if (mode != GLO_FLAG_DIGEST_CAP) buf[len++] = degree_bits;
    glo_hash_no_pad(buf, len, out);
    free(buf);
}

unsigned glo_gate_num_constraints(unsigned id, unsigned param) {
    switch (id) {
        case GLO_GATE_NOOP: return 0;
        case GLO_GATE_CONSTANT: return param;
        case GLO_GATE_PUBLIC_INPUT: return 4;
        case GLO_GATE_ARITHMETIC: return param;
        case GLO_GATE_POSEIDON: return 123;
        case GLO_GATE_BASE_SUM: return 1 + param;
        case GLO_GATE_ARITHMETIC_EXT: return 2 * param;
        case GLO_GATE_MUL_EXT: return 2 * param;
        case GLO_GATE_RANDOM_ACCESS: return ((param >> 8) & 0xFF) * ((param & 0xFF) + 2) + ((param >> 16) & 0xFF);
        case GLO_GATE_REDUCING: return 2 * param;
        case GLO_GATE_REDUCING_EXT: return 2 * param;
        case GLO_GATE_EXPONENTIATION: return param + 1;
        case GLO_GATE_POSEIDON_MDS: return 24;
        case GLO_GATE_COSET_INTERPOLATION: {
            unsigned n_points = 1u << (param & 0xFF), degree = (param >> 8) & 0xFF;
            return 2 + 4 * ((n_points - 2) / (degree - 1)) + 2;
        }
        default: return 0;
    }
}

/* ExtensionAlgebra over the wire field: pairs (a0, a1) = a0 + a1 Y, Y^2 = 7, whose components are themselves `ext`
 * (F::Extension at the verifier's point, embedded base values on the prover's coset) */
typedef struct {
    ext c0, c1;
} alg;
static alg amul(alg x, alg y) {
    alg r = {eadd(emul(x.c0, y.c0), emulb(emul(x.c1, y.c1), 7)), eadd(emul(x.c0, y.c1), emul(x.c1, y.c0))};
    return r;
}
static alg aadd(alg x, alg y) {
    alg r = {eadd(x.c0, y.c0), eadd(x.c1, y.c1)};
    return r;
}
static alg asub(alg x, alg y) {
    alg r = {esub(x.c0, y.c0), esub(x.c1, y.c1)};
    return r;
}
static alg ascal(alg x, ext s) { /* scalar_mul by a wire-field element */
    alg r = {emul(x.c0, s), emul(x.c1, s)};
    return r;
}
static alg aw(const ext *w, unsigned i) { /* get_local_ext_algebra(i..i+2) */
    alg r = {w[i], w[i + 1]};
    return r;
}

/* interpolation.rs::barycentric_weights over the two-adic subgroup of size 2^bits, natural order g^i */
void glo_coset_interpolation_tables(unsigned bits, uint64_t *domain, uint64_t *weights) {
    const unsigned n = 1u << bits;
    uint64_t g = glo_root_of_unity(bits), x = 1;
    for (unsigned i = 0; i < n; i++, x = glo_mul(x, g)) domain[i] = x;
    for (unsigned i = 0; i < n; i++) {
        uint64_t prod = 1;
        for (unsigned j = 0; j < n; j++)
            if (j != i) prod = glo_mul(prod, glo_sub(domain[i], domain[j]));
        weights[i] = glo_inv(prod);
    }
}

/* coset_interpolation.rs::partial_interpolate_ext_algebra over points [from, to) */
static void partial_interpolate(const uint64_t *domain, const uint64_t *weights, const alg *values, unsigned from, unsigned to,
                                alg x, alg *eval, alg *prod) {
    for (unsigned k = from; k < to; k++) {
        alg term = x;
        term.c0 = esub(term.c0, E(domain[k]));
        alg val = {emulb(values[k].c0, weights[k]), emulb(values[k].c1, weights[k])};
        alg next_eval = aadd(amul(*eval, term), amul(val, *prod));
        *prod = amul(*prod, term);
        *eval = next_eval;
    }
}

/* eval_unfiltered: consts = the gate's own constants (selector prefix removed) */
static unsigned gate_eval(const glo_gate *g, const ext *consts, const ext *w, const uint64_t pi_hash[4], ext *c) {
    switch (g->id) {
        case GLO_GATE_NOOP: return 0;
        case GLO_GATE_CONSTANT:
            for (unsigned i = 0; i < g->param; i++) c[i] = esub(consts[i], w[i]);
            return g->param;
        case GLO_GATE_PUBLIC_INPUT:
            for (unsigned i = 0; i < 4; i++) c[i] = esub(w[i], E(pi_hash[i]));
            return 4;
        case GLO_GATE_ARITHMETIC:
            for (unsigned i = 0; i < g->param; i++) {
                ext m0 = w[4 * i], m1 = w[4 * i + 1], ad = w[4 * i + 2], out = w[4 * i + 3];
                ext comp = eadd(emul(emul(m0, m1), consts[0]), emul(ad, consts[1]));
                c[i] = esub(out, comp);
            }
            return g->param;
        case GLO_GATE_POSEIDON: return poseidon_gate_eval(w, c);
        case GLO_GATE_BASE_SUM: { /* gates/base_sum.rs, B = 2: sum wire 0, limbs 1.. */
            ext acc = {0, 0};
            for (unsigned i = g->param; i-- > 0;) acc = eadd(emulb(acc, 2), w[1 + i]);
            c[0] = esub(acc, w[0]);
            for (unsigned i = 0; i < g->param; i++) c[1 + i] = emul(w[1 + i], esub(w[1 + i], E(1)));
            return 1 + g->param;
        }
        case GLO_GATE_ARITHMETIC_EXT: /* gates/arithmetic_extension.rs: (m0 * m1) * c0 + addend * c1 in the algebra */
        case GLO_GATE_MUL_EXT: {      /* gates/multiplication_extension.rs: (m0 * m1) * c0 */
            const unsigned per = g->id == GLO_GATE_ARITHMETIC_EXT ? 8 : 6;
            for (unsigned i = 0; i < g->param; i++) {
                const ext *q = w + per * i;
                /* (a0 + a1 X)(b0 + b1 X) mod X^2 - 7, coefficients themselves generic field elements */
                ext p0 = eadd(emul(q[0], q[2]), emulb(emul(q[1], q[3]), 7));
                ext p1 = eadd(emul(q[0], q[3]), emul(q[1], q[2]));
                p0 = emul(p0, consts[0]);
                p1 = emul(p1, consts[0]);
                const ext *out = q + 4;
                if (g->id == GLO_GATE_ARITHMETIC_EXT) {
                    p0 = eadd(p0, emul(q[4], consts[1]));
                    p1 = eadd(p1, emul(q[5], consts[1]));
                    out = q + 6;
                }
                c[2 * i] = esub(out[0], p0);
                c[2 * i + 1] = esub(out[1], p1);
            }
            return 2 * g->param;
        }
        case GLO_GATE_RANDOM_ACCESS: { /* gates/random_access.rs */
            const unsigned bits = g->param & 0xFF, copies = (g->param >> 8) & 0xFF, extra = (g->param >> 16) & 0xFF;
            const unsigned vec = 1u << bits, routed = (2 + vec) * copies + extra;
            unsigned k = 0;
            for (unsigned cp = 0; cp < copies; cp++) {
                const ext *q = w + (2 + vec) * cp; /* access index, claimed element, list items */
                const ext *b = w + routed + cp * bits;
                for (unsigned i = 0; i < bits; i++) c[k++] = emul(b[i], esub(b[i], E(1)));
                ext rec = {0, 0};
                for (unsigned i = bits; i-- > 0;) rec = eadd(eadd(rec, rec), b[i]);
                c[k++] = esub(rec, q[0]);
                ext items[64];
                for (unsigned j = 0; j < vec; j++) items[j] = q[2 + j];
                for (unsigned i = 0, len = vec; i < bits; i++, len >>= 1)
                    for (unsigned j = 0; j < len / 2; j++)
                        items[j] = eadd(items[2 * j], emul(b[i], esub(items[2 * j + 1], items[2 * j])));
                c[k++] = esub(items[0], q[1]);
            }
            for (unsigned i = 0; i < extra; i++) c[k++] = esub(consts[i], w[(2 + vec) * copies + i]);
            return k;
        }
        case GLO_GATE_REDUCING:       /* gates/reducing.rs: acc' = acc * alpha + coeff (base-field coefficients) */
        case GLO_GATE_REDUCING_EXT: { /* gates/reducing_extension.rs: the same with extension coefficients */
            const unsigned nc = g->param, is_ext = g->id == GLO_GATE_REDUCING_EXT;
            const unsigned start_accs = 6 + (is_ext ? 2 * nc : nc);
            alg alpha = aw(w, 2), acc = aw(w, 4);
            for (unsigned i = 0; i < nc; i++) {
                alg coeff;
                if (is_ext) {
                    coeff = aw(w, 6 + 2 * i);
                } else {
                    coeff.c0 = w[6 + i];
                    coeff.c1 = E(0);
                }
                alg next = i == nc - 1 ? aw(w, 0) : aw(w, start_accs + 2 * i);
                alg d = asub(aadd(amul(acc, alpha), coeff), next);
                c[2 * i] = d.c0;
                c[2 * i + 1] = d.c1;
                acc = next;
            }
            return 2 * nc;
        }
        case GLO_GATE_EXPONENTIATION: { /* gates/exponentiation.rs: wire 0 base, 1.. bits (LE), then output, intermediates */
            const unsigned nb = g->param;
            ext base = w[0], output = w[1 + nb];
            for (unsigned i = 0; i < nb; i++) {
                ext prev = i == 0 ? E(1) : emul(w[2 + nb + i - 1], w[2 + nb + i - 1]);
                ext bit = w[1 + (nb - 1 - i)];
                ext factor = eadd(emul(bit, base), esub(E(1), bit));
                c[i] = esub(emul(prev, factor), w[2 + nb + i]);
            }
            c[nb] = esub(output, w[2 + nb + nb - 1]);
            return nb + 1;
        }
        case GLO_GATE_POSEIDON_MDS: { /* gates/poseidon_mds.rs: the MDS layer on 12 algebra elements, component-wise */
            for (unsigned r = 0; r < 12; r++) {
                alg acc = {{0, 0}, {0, 0}};
                for (unsigned i = 0; i < 12; i++) {
                    alg in = aw(w, 2 * ((i + r) % 12));
                    acc.c0 = eadd(acc.c0, emulb(in.c0, MDS_CIRC[i]));
                    acc.c1 = eadd(acc.c1, emulb(in.c1, MDS_CIRC[i]));
                }
                if (r == 0) {
                    acc.c0 = eadd(acc.c0, emulb(w[0], 8));
                    acc.c1 = eadd(acc.c1, emulb(w[1], 8));
                }
                alg out = aw(w, 24 + 2 * r);
                c[2 * r] = esub(out.c0, acc.c0);
                c[2 * r + 1] = esub(out.c1, acc.c1);
            }
            return 24;
        }
        case GLO_GATE_COSET_INTERPOLATION: { /* gates/coset_interpolation.rs */
            const unsigned bits = g->param & 0xFF, degree = (g->param >> 8) & 0xFF, np = 1u << bits;
            const unsigned ni = (np - 2) / (degree - 1);
            const unsigned start_point = 1 + 2 * np, start_value = start_point + 2, start_inter = start_value + 2;
            uint64_t domain[64], weights[64];
            glo_coset_interpolation_tables(bits, domain, weights);
            ext shift = w[0];
            alg point = aw(w, start_point), shifted = aw(w, start_inter + 4 * ni);
            alg values[64];
            for (unsigned i = 0; i < np; i++) values[i] = aw(w, 1 + 2 * i);
            unsigned k = 0;
            alg d = asub(point, ascal(shifted, shift));
            c[k++] = d.c0;
            c[k++] = d.c1;
            alg eval = {{0, 0}, {0, 0}}, prod = {{1, 0}, {0, 0}};
            partial_interpolate(domain, weights, values, 0, degree, shifted, &eval, &prod);
            for (unsigned i = 0; i < ni; i++) {
                alg ie = aw(w, start_inter + 2 * i), ip = aw(w, start_inter + 2 * (ni + i));
                d = asub(ie, eval);
                c[k++] = d.c0;
                c[k++] = d.c1;
                d = asub(ip, prod);
                c[k++] = d.c0;
                c[k++] = d.c1;
                unsigned from = 1 + (degree - 1) * (i + 1), to = from + degree - 1;
                if (to > np) to = np;
                eval = ie;
                prod = ip;
                partial_interpolate(domain, weights, values, from, to, shifted, &eval, &prod);
            }
            d = asub(aw(w, start_value), eval);
            c[k++] = d.c0;
            c[k++] = d.c1;
            return k;
        }
        default: return 0;
    }
}

#define UNUSED_SELECTOR 0xFFFFFFFFull

/* vanishing_poly.rs::eval_vanishing_poly — generic point x in the extension field */
/* the vanishing terms before the alpha reduction: terms[nch + nch * (npp + 1) + num_gate_constraints] (zeroed by the caller);
 * diag_filters[g] / diag_unf[g * 128 + j] / diag_nc[g] (optional): every gate's filter, unfiltered constraints and their
 * count — the first-contact kit compares them with plonky2's gate by gate */
static void vanishing_terms(const glo_circuit *cc, ext x, ext x_pow_n, const ext *consts, const ext *wires, const ext *zs,
                            const ext *next_zs, const ext *pps, const ext *sigmas, const uint64_t *betas,
                            const uint64_t *gammas, const uint64_t pi_hash[4], ext *terms, ext *diag_filters, ext *diag_unf,
                            uint32_t *diag_nc) {
    const unsigned nch = cc->num_challenges, npp = cc->num_partial_products, nr = cc->num_routed_wires;
    const unsigned chunk = cc->quotient_degree_factor;
    /* l_0(x) = (x^n - 1) / (n (x - 1)) */
    ext l0;
    if (eeq(x, E(1))) {
        l0 = E(1);
    } else {
        ext den = emulb(esub(x, E(1)), (uint64_t)1 << cc->degree_bits);
        l0 = emul(esub(x_pow_n, E(1)), einv(den));
    }
    unsigned t = 0;
    for (unsigned i = 0; i < nch; i++) terms[t++] = emul(l0, esub(zs[i], E(1)));
    for (unsigned i = 0; i < nch; i++) {
        /* check_partial_products over chunks of `chunk` routed wires; accumulators z_x, pp_0.., z_gx */
        for (unsigned ck = 0; ck * chunk < nr; ck++) {
            ext nump = E(1), denp = E(1);
            for (unsigned j = ck * chunk; j < (ck + 1) * chunk && j < nr; j++) {
                ext s_id = emulb(x, cc->k_is[j]);
                nump = emul(nump, eadd(eadd(wires[j], emulb(s_id, betas[i])), E(gammas[i])));
                denp = emul(denp, eadd(eadd(wires[j], emulb(sigmas[j], betas[i])), E(gammas[i])));
            }
            ext prev = ck == 0 ? zs[i] : pps[i * npp + ck - 1];
            ext next = ck == npp ? next_zs[i] : pps[i * npp + ck];
            terms[t++] = esub(emul(prev, nump), emul(next, denp));
        }
    }
    /* evaluate_gate_constraints: constraints[j] += filter_g * c_{g,j} */
    ext *gc = terms + t;
    ext tmp[128];
    for (unsigned g = 0; g < cc->num_gates; g++) {
        const glo_gate *gt = &cc->gates[g];
        ext s = consts[gt->selector_index];
        ext filter = E(1);
        for (unsigned r = gt->group_start; r < gt->group_end; r++)
            if (r != g) filter = emul(filter, esub(E(r), s));
        if (cc->num_selectors > 1) filter = emul(filter, esub(E(UNUSED_SELECTOR), s));
        unsigned nc = gate_eval(gt, consts + cc->num_selectors, wires, pi_hash, tmp);
        for (unsigned j = 0; j < nc; j++) gc[j] = eadd(gc[j], emul(filter, tmp[j]));
        if (diag_filters) diag_filters[g] = filter;
        if (diag_nc) diag_nc[g] = nc;
        if (diag_unf)
            for (unsigned j = 0; j < nc; j++) diag_unf[(size_t)g * 128 + j] = tmp[j];
    }
}

static void eval_vanishing(const glo_circuit *cc, ext x, ext x_pow_n, const ext *consts, const ext *wires, const ext *zs,
                           const ext *next_zs, const ext *pps, const ext *sigmas, const uint64_t *betas,
                           const uint64_t *gammas, const uint64_t *alphas, const uint64_t pi_hash[4], ext *out) {
    const unsigned nch = cc->num_challenges, npp = cc->num_partial_products;
    unsigned n_terms = nch + nch * (npp + 1) + cc->num_gate_constraints;
    ext *terms = (ext *)calloc(n_terms, sizeof(ext));
    vanishing_terms(cc, x, x_pow_n, consts, wires, zs, next_zs, pps, sigmas, betas, gammas, pi_hash, terms, NULL, NULL, NULL);
    for (unsigned i = 0; i < nch; i++) {
        ext acc = {0, 0};
        for (unsigned j = n_terms; j-- > 0;) acc = eadd(emulb(acc, alphas[i]), terms[j]);
        out[i] = acc;
    }
    free(terms);
}

/* First-contact diagnostics (tests/first_contact/compare_plonky2_dump.py): from an OpeningSet (flat extension values in the
 * proof's order: constants ++ sigmas, wires, zs, zs_next, partial products) and the challenges plonky2 itself reports, the
 * vanishing-term vector at zeta before the alpha reduction and every gate's filter and eval_unfiltered output.
 * terms_out [n_terms][2] (returns n_terms), filters_out [num_gates][2], unfiltered_out [num_gates][128][2], n_cons_out [num_gates] */
size_t glo_vanishing_diagnose(const glo_circuit *cc, const uint64_t zeta[2], const uint64_t *o_cs, const uint64_t *o_w,
                              const uint64_t *o_z, const uint64_t *o_zn, const uint64_t *o_pp, const uint64_t *betas,
                              const uint64_t *gammas, const uint64_t pi_hash[4], uint64_t *terms_out, uint64_t *filters_out,
                              uint64_t *unfiltered_out, uint32_t *n_cons_out) {
    const unsigned nch = cc->num_challenges;
    const size_t nw = cc->num_wires, npp = (size_t)nch * cc->num_partial_products;
    const unsigned n_terms = nch + nch * (cc->num_partial_products + 1) + cc->num_gate_constraints;
    ext consts[16], wires[160], zs[4], nzs[4], pps[32], sig[96];
    /* a dump from a foreign circuit (more constants, wires, partial products, or a gate with more than 128 constraints) must be
     * refused, not written past these arrays: 0 terms = "dimensions not supported by the diagnostic" */
    if (cc->num_constants > 16 || nw > 160 || nch > 4 || npp > 32 || cc->num_routed_wires > 96) return 0;
    for (unsigned g = 0; g < cc->num_gates; g++)
        if (glo_gate_num_constraints(cc->gates[g].id, cc->gates[g].param) > 128) return 0;
    for (size_t k = 0; k < cc->num_constants; k++) consts[k] = (ext){o_cs[2 * k], o_cs[2 * k + 1]};
    for (size_t k = 0; k < cc->num_routed_wires; k++)
        sig[k] = (ext){o_cs[2 * (cc->num_constants + k)], o_cs[2 * (cc->num_constants + k) + 1]};
    for (size_t k = 0; k < nw; k++) wires[k] = (ext){o_w[2 * k], o_w[2 * k + 1]};
    for (size_t k = 0; k < nch; k++) {
        zs[k] = (ext){o_z[2 * k], o_z[2 * k + 1]};
        nzs[k] = (ext){o_zn[2 * k], o_zn[2 * k + 1]};
    }
    for (size_t k = 0; k < npp; k++) pps[k] = (ext){o_pp[2 * k], o_pp[2 * k + 1]};
    ext z = {zeta[0], zeta[1]};
    ext *terms = (ext *)calloc(n_terms, sizeof(ext));
    ext *filt = (ext *)calloc(cc->num_gates, sizeof(ext));
    ext *unf = (ext *)calloc((size_t)cc->num_gates * 128, sizeof(ext));
    vanishing_terms(cc, z, epow2k(z, cc->degree_bits), consts, wires, zs, nzs, pps, sig, betas, gammas, pi_hash, terms, filt, unf,
                    n_cons_out);
    for (unsigned j = 0; j < n_terms; j++) {
        terms_out[2 * j] = glo_canon(terms[j].a);
        terms_out[2 * j + 1] = glo_canon(terms[j].b);
    }
    for (unsigned g = 0; g < cc->num_gates; g++) {
        filters_out[2 * g] = glo_canon(filt[g].a);
        filters_out[2 * g + 1] = glo_canon(filt[g].b);
        for (unsigned j = 0; j < 128; j++) {
            unfiltered_out[2 * ((size_t)g * 128 + j)] = glo_canon(unf[(size_t)g * 128 + j].a);
            unfiltered_out[2 * ((size_t)g * 128 + j) + 1] = glo_canon(unf[(size_t)g * 128 + j].b);
        }
    }
    free(terms);
    free(filt);
    free(unf);
    return n_terms;
}

/* ------------------------------------------------ prover pieces ------------------------------------------------ */
/* wires_permutation_partial_products_and_zs: out = [Z_0, Z_1, pp(ch 0) x npp, pp(ch 1) x npp] columns of n values */
void glo_partial_products_and_zs(const glo_circuit *cc, const uint64_t *wires /* [num_wires][n] */,
                                 const uint64_t *sigmas /* [num_routed][n] values on H */, const uint64_t *betas,
                                 const uint64_t *gammas, uint64_t *out) {
    const size_t n = (size_t)1 << cc->degree_bits;
    const unsigned nr = cc->num_routed_wires, npp = cc->num_partial_products, nch = cc->num_challenges;
    const unsigned chunk = cc->quotient_degree_factor;
    /* as plonky2 does it: the chunk quotients prod(num) / prod(den) of every row in parallel (one inversion per chunk: a zero
     * denominator makes the chunk quotient zero, exactly what multiplying the per-factor quotients num * inv(den) with
     * inv(0) = 0 gives), then one cheap running product down the rows */
    const uint64_t w = glo_root_of_unity(cc->degree_bits);
    const unsigned n_chunks = (nr + chunk - 1) / chunk;
    uint64_t *xs = (uint64_t *)malloc(n * 8);
    uint64_t *q = (uint64_t *)glo_big_alloc((size_t)nch * n_chunks * n * 8);
    {
        uint64_t x = 1;
        for (size_t i = 0; i < n; i++) {
            xs[i] = x;
            x = glo_mul(x, w);
        }
    }
    if (glo_get_fast()) /* the timed leg: VL rows per step, one inversion per step (fast_impl.h), the same quotients */
        glo_fast_chunk_quotients(cc, wires, sigmas, betas, gammas, q, glo_get_threads());
    else {
#pragma omp parallel for num_threads(glo_get_threads()) schedule(static)
    for (long i = 0; i < (long)n; i++)
        for (unsigned ch = 0; ch < nch; ch++)
            for (unsigned ck = 0; ck < n_chunks; ck++) {
                uint64_t num = 1, den = 1;
                for (unsigned j = ck * chunk; j < (ck + 1) * chunk && j < nr; j++) {
                    uint64_t wv = wires[(size_t)j * n + i];
                    num = glo_mul(num, glo_add(glo_add(wv, glo_mul(betas[ch], glo_mul(cc->k_is[j], xs[i]))), gammas[ch]));
                    den = glo_mul(den, glo_add(glo_add(wv, glo_mul(betas[ch], sigmas[(size_t)j * n + i])), gammas[ch]));
                }
                q[((size_t)ch * n_chunks + ck) * n + i] = glo_mul(num, glo_inv(den));
            }
    } /* (scalar path) */
    for (unsigned ch = 0; ch < nch; ch++) {
        uint64_t z = 1;
        uint64_t *zcol = out + (size_t)ch * n;
        uint64_t *ppcols = out + ((size_t)nch + (size_t)ch * npp) * n;
        for (size_t i = 0; i < n; i++) {
            zcol[i] = z;
            uint64_t acc = z;
            for (unsigned ck = 0; ck < n_chunks; ck++) {
                acc = glo_mul(acc, q[((size_t)ch * n_chunks + ck) * n + i]);
                if (ck < npp)
                    ppcols[(size_t)ck * n + i] = acc;
                else
                    z = acc; /* Z(g x) */
            }
        }
    }
    glo_big_free(q);
    free(xs);
}

/* compute_quotient_polys + chunking: out = [num_challenges * quotient_degree_factor][n] coefficient columns */
void glo_quotient_chunks(const glo_circuit *cc, const uint64_t *cs_leaves, const uint64_t *wires_leaves,
                         const uint64_t *zs_leaves, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                         const uint64_t pi_hash[4], uint64_t *out) {
    const unsigned qb = 3; /* log2(quotient_degree_factor = 8) = rate_bits */
    const unsigned lde_bits = cc->degree_bits + qb;
    const size_t n = (size_t)1 << cc->degree_bits, N = (size_t)1 << lde_bits;
    const unsigned nch = cc->num_challenges, npp = cc->num_partial_products;
    const unsigned ncs = cc->num_constants + cc->num_routed_wires, nzs = nch * (1 + npp);
    /* words per leaf row: blinded oracles carry GLO_SALT_SIZE salt words behind their polynomial values (get_lde_values slices
     * them off: `&leaves[index][..self.polynomials.len()]`) */
    const unsigned salt = (cc->fri.flags & GLO_FLAG_ZERO_KNOWLEDGE) ? GLO_SALT_SIZE : 0;
    const size_t ww = cc->num_wires + salt, zw = nzs + salt;
    uint64_t wN = glo_root_of_unity(lde_bits);
    uint64_t *vals = (uint64_t *)glo_big_alloc((size_t)nch * N * 8);
    /* Z_H on the coset: x^n = 7^n * w_8^(i mod 8) */
    uint64_t zh_inv[8], g_n = glo_pow(7, n), w8 = glo_root_of_unity(qb);
    for (unsigned r = 0; r < 8; r++) zh_inv[r] = glo_inv(glo_sub(glo_mul(g_n, glo_pow(w8, r)), 1));
    (void)glo_poseidon_round_constants();
    if (glo_get_fast()) /* the timed leg: VL points per step in base-field arithmetic (fast_impl.h), the same values */
        glo_fast_quotient_values(cc, cs_leaves, wires_leaves, zs_leaves, betas, gammas, alphas, pi_hash, vals, glo_get_threads());
    else {
#pragma omp parallel for num_threads(glo_get_threads()) schedule(dynamic, 64)
    for (long ii = 0; ii < (long)N; ii++) {
        size_t i = (size_t)ii;
        uint64_t x = glo_mul(7, glo_pow(wN, i));
        size_t row = glo_reverse_bits(i, lde_bits), row_next = glo_reverse_bits((i + 8) % N, lde_bits);
        ext consts[16], wires[160], zs[4], nzsv[4], pps[32], sig[96], res[4];
        for (unsigned k = 0; k < cc->num_constants; k++) consts[k] = E(cs_leaves[row * ncs + k]);
        for (unsigned k = 0; k < cc->num_routed_wires; k++) sig[k] = E(cs_leaves[row * ncs + cc->num_constants + k]);
        for (unsigned k = 0; k < cc->num_wires; k++) wires[k] = E(wires_leaves[row * ww + k]);
        for (unsigned k = 0; k < nch; k++) {
            zs[k] = E(zs_leaves[row * zw + k]);
            nzsv[k] = E(zs_leaves[row_next * zw + k]);
        }
        for (unsigned k = 0; k < nch * npp; k++) pps[k] = E(zs_leaves[row * zw + nch + k]);
        ext xn = E(glo_mul(g_n, glo_pow(w8, i % 8)));
        eval_vanishing(cc, E(x), xn, consts, wires, zs, nzsv, pps, sig, betas, gammas, alphas, pi_hash, res);
        for (unsigned k = 0; k < nch; k++) vals[(size_t)k * N + i] = glo_mul(res[k].a, zh_inv[i % 8]);
    }
    } /* (scalar path) */
    for (unsigned k = 0; k < nch; k++) {
        if (glo_get_fast()) { /* the timed leg: one column at a time, the threads inside the transform */
            glo_fast_fft_mt(vals + (size_t)k * N, lde_bits, 1, glo_get_threads());
            glo_fast_scale_powers_mt(vals + (size_t)k * N, N, glo_inv(7), glo_get_threads());
        } else
        glo_coset_ifft(vals + (size_t)k * N, lde_bits, 7);
        /* trim_to_len(8n) is a no-op here; chunks(degree) */
        memcpy(out + (size_t)k * 8 * n, vals + (size_t)k * N, N * 8);
    }
    glo_big_free(vals);
}

/* ------------------------------------------------ proof layout ------------------------------------------------ */
static void fri_params_of(const glo_circuit *cc, glo_fri_params *p) { *p = cc->fri; }

static unsigned n_openings(const glo_circuit *cc) {
    unsigned nch = cc->num_challenges;
    return cc->num_constants + cc->num_routed_wires + cc->num_wires + nch + nch + nch * cc->num_partial_products +
           nch * cc->quotient_degree_factor;
}

size_t glo_plonk_proof_len(const glo_circuit *cc) {
    glo_fri_params p;
    fri_params_of(cc, &p);
    const size_t salt = (cc->fri.flags & GLO_FLAG_ZERO_KNOWLEDGE) ? GLO_SALT_SIZE : 0; /* leaf widths of the blinded oracles */
    size_t cols[4] = {cc->num_constants + cc->num_routed_wires, cc->num_wires + salt,
                      cc->num_challenges * (1 + cc->num_partial_products) + salt, cc->num_challenges * cc->quotient_degree_factor + salt};
    return 3 * ((size_t)4 << p.cap_height) + 2 * (size_t)n_openings(cc) + glo_fri_proof_len(&p, cols, 4);
}

/* FRI instance of common_data.get_fri_instance(zeta): batch 0 = every polynomial at zeta, batch 1 = Zs at g*zeta */
typedef struct {
    glo_fri_batch b[2];
    glo_fri_instance inst;
    uint32_t *oi, *pi;
} fri_inst;

static void make_fri_instance(const glo_circuit *cc, const uint64_t zeta[2], fri_inst *fi) {
    size_t cols[4] = {cc->num_constants + cc->num_routed_wires, cc->num_wires,
                      cc->num_challenges * (1 + cc->num_partial_products), cc->num_challenges * cc->quotient_degree_factor};
    size_t total = cols[0] + cols[1] + cols[2] + cols[3];
    fi->oi = (uint32_t *)malloc((total + cc->num_challenges) * 4);
    fi->pi = (uint32_t *)malloc((total + cc->num_challenges) * 4);
    size_t k = 0;
    for (uint32_t o = 0; o < 4; o++)
        for (uint32_t i = 0; i < cols[o]; i++) {
            fi->oi[k] = o;
            fi->pi[k++] = i;
        }
    for (uint32_t i = 0; i < cc->num_challenges; i++) {
        fi->oi[k] = 2;
        fi->pi[k++] = i;
    }
    uint64_t g = glo_root_of_unity(cc->degree_bits);
    fi->b[0].point[0] = zeta[0];
    fi->b[0].point[1] = zeta[1];
    fi->b[0].n_polys = total;
    fi->b[0].oracle_index = fi->oi;
    fi->b[0].poly_index = fi->pi;
    fi->b[1].point[0] = glo_mul(zeta[0], g);
    fi->b[1].point[1] = glo_mul(zeta[1], g);
    fi->b[1].n_polys = cc->num_challenges;
    fi->b[1].oracle_index = fi->oi + total;
    fi->b[1].poly_index = fi->pi + total;
    fi->inst.n_batches = 2;
    fi->inst.batches = fi->b;
}

/* ------------------------------------------------ prove ------------------------------------------------ */
/* GLO_TIMING=1: phase times of the CPU proof on stderr (where the bounded cpu_baseline sample of bench.py spends its time) */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static double g_phase_s[8];
static int g_phase_n = 0;
#define PHASE(name)                                                                  \
    do {                                                                             \
        double t_ = now_s();                                                         \
        if (g_phase_n < 8) g_phase_s[g_phase_n++] = t_ - t_phase;                    \
        if (timing) fprintf(stderr, "  oracle prove: %-22s %8.3f s\n", name, t_ - t_phase); \
        t_phase = t_;                                                                \
    } while (0)
/* seconds of the last glo_plonk_prove call per phase: wires commitment, partial products, Z commitment, quotient evaluation,
 * quotient commitment, openings, FRI (bench.py's cpu_baseline record) */
int glo_last_phase_seconds(double out[8]) {
    for (int i = 0; i < g_phase_n; i++) out[i] = g_phase_s[i];
    return g_phase_n;
}

int glo_plonk_prove(const glo_circuit *cc, const uint64_t *cs_coeffs, const uint64_t *cs_leaves, const uint64_t *cs_digests,
                    const uint64_t *sigma_values, const uint64_t *wire_values, const uint64_t *public_inputs,
                    size_t n_public_inputs, uint64_t *proof_out) {
    return glo_plonk_prove_salted(cc, cs_coeffs, cs_leaves, cs_digests, sigma_values, wire_values, public_inputs, n_public_inputs, NULL,
                                  proof_out);
}

int glo_plonk_prove_salted(const glo_circuit *cc, const uint64_t *cs_coeffs, const uint64_t *cs_leaves, const uint64_t *cs_digests,
                           const uint64_t *sigma_values, const uint64_t *wire_values, const uint64_t *public_inputs,
                           size_t n_public_inputs, const uint64_t *salts, uint64_t *proof_out) {
    const unsigned zk = (cc->fri.flags & GLO_FLAG_ZERO_KNOWLEDGE) != 0;
    if (zk != (salts != NULL)) return -30; /* the salts and the zero-knowledge flag come together */
    const size_t salt = zk ? GLO_SALT_SIZE : 0;
    const int timing = getenv("GLO_TIMING") != NULL;
    double t_phase = now_s();
    g_phase_n = 0;
    const size_t n = (size_t)1 << cc->degree_bits;
    const unsigned rb = cc->fri.rate_bits, ch_ = cc->fri.cap_height, nch = cc->num_challenges;
    const size_t N = n << rb, cap_words = (size_t)4 << ch_;
    const size_t ncs = cc->num_constants + cc->num_routed_wires, nw = cc->num_wires;
    const size_t nzs = nch * (1 + cc->num_partial_products), nq = nch * cc->quotient_degree_factor;
    uint64_t pi_hash[4];
    glo_hash_no_pad(public_inputs, n_public_inputs, pi_hash);

    uint64_t *w = proof_out;
    /* wires commitment */
    const uint64_t *salts_w = zk ? salts : NULL, *salts_z = zk ? salts + GLO_SALT_SIZE * N : NULL,
                   *salts_q = zk ? salts + 2 * GLO_SALT_SIZE * N : NULL;
    uint64_t *w_coeffs = (uint64_t *)glo_big_alloc(nw * n * 8), *w_leaves = (uint64_t *)glo_big_alloc((nw + salt) * N * 8);
    uint64_t *w_dig = (uint64_t *)glo_big_alloc(glo_merkle_levels_len(N, ch_) * 8);
    glo_batch_from_values_salted(wire_values, nw, cc->degree_bits, rb, ch_, salts_w, w_coeffs, w_leaves, w_dig, w);
    PHASE("wires commitment");
    glo_challenger chal;
    glo_challenger_init(&chal);
    glo_challenger_observe(&chal, cc->circuit_digest, 4);
    glo_challenger_observe(&chal, pi_hash, 4);
    glo_challenger_observe(&chal, w, cap_words);
    w += cap_words;
    uint64_t betas[4], gammas[4], alphas[4];
    glo_challenger_get_n(&chal, betas, nch);
    glo_challenger_get_n(&chal, gammas, nch);
    /* Z and partial products */
    uint64_t *zs_vals = (uint64_t *)glo_big_alloc(nzs * n * 8);
    glo_partial_products_and_zs(cc, wire_values, sigma_values, betas, gammas, zs_vals);
    PHASE("partial products");
    uint64_t *z_coeffs = (uint64_t *)glo_big_alloc(nzs * n * 8), *z_leaves = (uint64_t *)glo_big_alloc((nzs + salt) * N * 8);
    uint64_t *z_dig = (uint64_t *)glo_big_alloc(glo_merkle_levels_len(N, ch_) * 8);
    glo_batch_from_values_salted(zs_vals, nzs, cc->degree_bits, rb, ch_, salts_z, z_coeffs, z_leaves, z_dig, w);
    PHASE("Z commitment");
    glo_challenger_observe(&chal, w, cap_words);
    w += cap_words;
    glo_challenger_get_n(&chal, alphas, nch);
    /* quotient */
    uint64_t *q_coeffs = (uint64_t *)glo_big_alloc(nq * n * 8), *q_leaves = (uint64_t *)glo_big_alloc((nq + salt) * N * 8);
    uint64_t *q_dig = (uint64_t *)glo_big_alloc(glo_merkle_levels_len(N, ch_) * 8);
    glo_quotient_chunks(cc, cs_leaves, w_leaves, z_leaves, betas, gammas, alphas, pi_hash, q_coeffs);
    PHASE("quotient evaluation");
    glo_batch_from_coeffs_salted(q_coeffs, nq, cc->degree_bits, rb, ch_, salts_q, q_leaves, q_dig, w);
    PHASE("quotient commitment");
    glo_challenger_observe(&chal, w, cap_words);
    w += cap_words;
    uint64_t zeta[2];
    glo_challenger_get_n(&chal, zeta, 2);
    /* openings: constants, sigmas, wires, zs, zs_next, partial products, quotient chunks */
    uint64_t gz[2] = {glo_mul(zeta[0], glo_root_of_unity(cc->degree_bits)), glo_mul(zeta[1], glo_root_of_unity(cc->degree_bits))};
    uint64_t *op = w;
    { /* one polynomial per task (plonky2: eval_polys is a rayon map) */
        const size_t o_w = ncs, o_z = o_w + nw, o_zn = o_z + nch, o_pp = o_zn + nch, o_q = o_pp + (nzs - nch), total = o_q + nq;
        if (glo_get_fast()) { /* the timed leg: tables of the powers of zeta and g zeta, then one dot product per polynomial */
            uint64_t *zp = (uint64_t *)malloc(4 * n * 8); /* planes: zeta^j limb 0, limb 1, (g zeta)^j limb 0, limb 1 */
            for (int wch = 0; wch < 2; wch++) {
                const uint64_t *pt = wch ? gz : zeta;
                uint64_t cur[2] = {1, 0}, nx[2];
                for (size_t j = 0; j < n; j++) {
                    zp[(size_t)(2 * wch) * n + j] = cur[0];
                    zp[(size_t)(2 * wch + 1) * n + j] = cur[1];
                    glo_ext_mul(cur, pt, nx);
                    cur[0] = nx[0];
                    cur[1] = nx[1];
                }
            }
#pragma omp parallel for num_threads(glo_get_threads()) schedule(dynamic, 1)
            for (long t = 0; t < (long)total; t++) {
                const size_t k = (size_t)t;
                uint64_t *dst = op + 2 * k;
                const uint64_t *z0 = zp, *z1 = zp + n, *src;
                if (k < o_w)
                    src = cs_coeffs + k * n;
                else if (k < o_z)
                    src = w_coeffs + (k - o_w) * n;
                else if (k < o_zn)
                    src = z_coeffs + (k - o_z) * n;
                else if (k < o_pp) {
                    src = z_coeffs + (k - o_zn) * n;
                    z0 = zp + 2 * n;
                    z1 = zp + 3 * n;
                } else if (k < o_q)
                    src = z_coeffs + (nch + (k - o_pp)) * n;
                else
                    src = q_coeffs + (k - o_q) * n;
                glo_fast_eval_with_powers(src, n, z0, z1, dst);
            }
            free(zp);
        } else {
#pragma omp parallel for num_threads(glo_get_threads()) schedule(dynamic, 1)
        for (long t = 0; t < (long)total; t++) {
            const size_t k = (size_t)t;
            uint64_t *dst = op + 2 * k;
            if (k < o_w)
                glo_poly_eval_ext(cs_coeffs + k * n, n, zeta, dst);
            else if (k < o_z)
                glo_poly_eval_ext(w_coeffs + (k - o_w) * n, n, zeta, dst);
            else if (k < o_zn)
                glo_poly_eval_ext(z_coeffs + (k - o_z) * n, n, zeta, dst);
            else if (k < o_pp)
                glo_poly_eval_ext(z_coeffs + (k - o_zn) * n, n, gz, dst);
            else if (k < o_q)
                glo_poly_eval_ext(z_coeffs + (nch + (k - o_pp)) * n, n, zeta, dst);
            else
                glo_poly_eval_ext(q_coeffs + (k - o_q) * n, n, zeta, dst);
        }
    } /* (scalar path) */
        w += 2 * total;
    }
    PHASE("openings");
    /* observe_openings(to_fri_openings): zeta batch = [constants, sigmas, wires, zs, partial products, quotient]; then zs_next */
    {
        const uint64_t *o_cs = op, *o_w = op + 2 * ncs, *o_z = o_w + 2 * nw, *o_zn = o_z + 2 * nch, *o_pp = o_zn + 2 * nch;
        const uint64_t *o_q = o_pp + 2 * (nzs - nch);
        glo_challenger_observe(&chal, o_cs, 2 * ncs);
        glo_challenger_observe(&chal, o_w, 2 * nw);
        glo_challenger_observe(&chal, o_z, 2 * nch);
        glo_challenger_observe(&chal, o_pp, 2 * (nzs - nch));
        glo_challenger_observe(&chal, o_q, 2 * nq);
        glo_challenger_observe(&chal, o_zn, 2 * nch);
    }
    /* FRI */
    fri_inst fi;
    make_fri_instance(cc, zeta, &fi);
    const uint64_t *oc[4] = {cs_coeffs, w_coeffs, z_coeffs, q_coeffs};
    const uint64_t *ol[4] = {cs_leaves, w_leaves, z_leaves, q_leaves};
    const uint64_t *od[4] = {cs_digests, w_dig, z_dig, q_dig};
    size_t cols[4] = {ncs, nw + salt, nzs + salt, nq + salt}; /* words per leaf row */
    glo_fri_params fp;
    fri_params_of(cc, &fp);
    int rc = glo_fri_prove(&fi.inst, oc, ol, od, cols, 4, &fp, &chal, w);
    PHASE("FRI");
    free(fi.oi);
    free(fi.pi);
    glo_big_free(w_coeffs); glo_big_free(w_leaves); glo_big_free(w_dig);
    glo_big_free(z_coeffs); glo_big_free(z_leaves); glo_big_free(z_dig); glo_big_free(zs_vals);
    glo_big_free(q_coeffs); glo_big_free(q_leaves); glo_big_free(q_dig);
    return rc;
}

/* ------------------------------------------------ verify ------------------------------------------------ */
int glo_plonk_verify(const glo_circuit *cc, const uint64_t *cs_cap, const uint64_t *public_inputs, size_t n_public_inputs,
                     const uint64_t *proof) {
    const unsigned nch = cc->num_challenges, ch_ = cc->fri.cap_height;
    const size_t cap_words = (size_t)4 << ch_;
    const size_t ncs = cc->num_constants + cc->num_routed_wires, nw = cc->num_wires;
    const size_t npp = nch * cc->num_partial_products, nq = nch * cc->quotient_degree_factor;
    uint64_t pi_hash[4];
    glo_hash_no_pad(public_inputs, n_public_inputs, pi_hash);
    const uint64_t *wires_cap = proof, *zs_cap = proof + cap_words, *q_cap = proof + 2 * cap_words;
    const uint64_t *op = proof + 3 * cap_words;
    const uint64_t *o_cs = op, *o_w = op + 2 * ncs, *o_z = o_w + 2 * nw, *o_zn = o_z + 2 * nch, *o_pp = o_zn + 2 * nch;
    const uint64_t *o_q = o_pp + 2 * npp;
    const uint64_t *fri = o_q + 2 * nq;
    /* get_challenges */
    glo_challenger chal;
    glo_challenger_init(&chal);
    glo_challenger_observe(&chal, cc->circuit_digest, 4);
    glo_challenger_observe(&chal, pi_hash, 4);
    glo_challenger_observe(&chal, wires_cap, cap_words);
    uint64_t betas[4], gammas[4], alphas[4], zeta[2];
    glo_challenger_get_n(&chal, betas, nch);
    glo_challenger_get_n(&chal, gammas, nch);
    glo_challenger_observe(&chal, zs_cap, cap_words);
    glo_challenger_get_n(&chal, alphas, nch);
    glo_challenger_observe(&chal, q_cap, cap_words);
    glo_challenger_get_n(&chal, zeta, 2);
    glo_challenger_observe(&chal, o_cs, 2 * ncs);
    glo_challenger_observe(&chal, o_w, 2 * nw);
    glo_challenger_observe(&chal, o_z, 2 * nch);
    glo_challenger_observe(&chal, o_pp, 2 * npp);
    glo_challenger_observe(&chal, o_q, 2 * nq);
    glo_challenger_observe(&chal, o_zn, 2 * nch);
    /* vanishing(zeta) == Z_H(zeta) * t(zeta) */
    ext consts[16], wires[160], zs[4], nzs[4], pps[32], sig[96], van[4];
    for (size_t k = 0; k < cc->num_constants; k++) consts[k] = (ext){o_cs[2 * k], o_cs[2 * k + 1]};
    for (size_t k = 0; k < cc->num_routed_wires; k++)
        sig[k] = (ext){o_cs[2 * (cc->num_constants + k)], o_cs[2 * (cc->num_constants + k) + 1]};
    for (size_t k = 0; k < nw; k++) wires[k] = (ext){o_w[2 * k], o_w[2 * k + 1]};
    for (size_t k = 0; k < nch; k++) {
        zs[k] = (ext){o_z[2 * k], o_z[2 * k + 1]};
        nzs[k] = (ext){o_zn[2 * k], o_zn[2 * k + 1]};
    }
    for (size_t k = 0; k < npp; k++) pps[k] = (ext){o_pp[2 * k], o_pp[2 * k + 1]};
    ext z = {zeta[0], zeta[1]};
    ext zeta_pow_n = epow2k(z, cc->degree_bits);
    eval_vanishing(cc, z, zeta_pow_n, consts, wires, zs, nzs, pps, sig, betas, gammas, alphas, pi_hash, van);
    ext z_h = esub(zeta_pow_n, E(1));
    for (unsigned i = 0; i < nch; i++) {
        ext acc = {0, 0};
        for (unsigned j = cc->quotient_degree_factor; j-- > 0;) {
            ext q = {o_q[2 * (i * cc->quotient_degree_factor + j)], o_q[2 * (i * cc->quotient_degree_factor + j) + 1]};
            acc = eadd(emul(acc, zeta_pow_n), q);
        }
        if (!eeq(van[i], emul(z_h, acc))) return -20 - (int)i;
    }
    /* FRI */
    fri_inst fi;
    make_fri_instance(cc, zeta, &fi);
    size_t n0 = fi.b[0].n_polys;
    uint64_t *b0 = (uint64_t *)malloc(2 * n0 * 8);
    {
        uint64_t *d = b0;
        memcpy(d, o_cs, 2 * ncs * 8); d += 2 * ncs;
        memcpy(d, o_w, 2 * nw * 8); d += 2 * nw;
        memcpy(d, o_z, 2 * nch * 8); d += 2 * nch;
        memcpy(d, o_pp, 2 * npp * 8); d += 2 * npp;
        memcpy(d, o_q, 2 * nq * 8);
    }
    const uint64_t *bo[2] = {b0, o_zn};
    const uint64_t *caps[4] = {cs_cap, wires_cap, zs_cap, q_cap};
    const size_t salt = (cc->fri.flags & GLO_FLAG_ZERO_KNOWLEDGE) ? GLO_SALT_SIZE : 0; /* unsalted_eval: the batch sums never reach them */
    size_t cols[4] = {ncs, nw + salt, nch * (1 + cc->num_partial_products) + salt, nq + salt};
    glo_fri_params fp;
    fri_params_of(cc, &fp);
    int rc = glo_fri_verify(&fi.inst, bo, caps, cols, 4, &fp, &chal, fri);
    free(b0);
    free(fi.oi);
    free(fi.pi);
    return rc;
}

/* witness of one PoseidonGate row (gates/poseidon.rs::PoseidonGenerator): 12 inputs, swap = 0 -> all 135 wires */
void glo_poseidon_gate_witness(const uint64_t in[12], uint64_t wires[135]) {
    const uint64_t *rc = glo_poseidon_round_constants();
    memset(wires, 0, 135 * sizeof(uint64_t));
    uint64_t st[12];
    for (int i = 0; i < 12; i++) wires[PG_IN(i)] = st[i] = glo_canon(in[i]);
    unsigned round = 0;
    for (int r = 0; r < 4; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = glo_add(st[i], rc[round * 12 + i]);
        if (r != 0)
            for (int i = 0; i < 12; i++) wires[PG_FULL0(r, i)] = st[i];
        ext e[12];
        for (int i = 0; i < 12; i++) e[i] = esbox(E(st[i]));
        emds(e);
        for (int i = 0; i < 12; i++) st[i] = e[i].a;
    }
    for (int r = 0; r < 22; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = glo_add(st[i], rc[round * 12 + i]);
        wires[PG_PARTIAL(r)] = st[0];
        ext e[12];
        for (int i = 0; i < 12; i++) e[i] = E(st[i]);
        e[0] = esbox(e[0]);
        emds(e);
        for (int i = 0; i < 12; i++) st[i] = e[i].a;
    }
    for (int r = 0; r < 4; r++, round++) {
        for (int i = 0; i < 12; i++) st[i] = glo_add(st[i], rc[round * 12 + i]);
        for (int i = 0; i < 12; i++) wires[PG_FULL1(r, i)] = st[i];
        ext e[12];
        for (int i = 0; i < 12; i++) e[i] = esbox(E(st[i]));
        emds(e);
        for (int i = 0; i < 12; i++) st[i] = e[i].a;
    }
    for (int i = 0; i < 12; i++) wires[PG_OUT(i)] = glo_canon(st[i]);
}

/* k rows at once (the Python harness fills whole dependency levels of a circuit's PoseidonGate rows per call) */
void glo_poseidon_gate_witness_many(const uint64_t *in /* [k][12] */, size_t k, uint64_t *wires /* [k][135] */) {
#pragma omp parallel for schedule(static) if (k >= 256)
    for (size_t i = 0; i < k; i++) glo_poseidon_gate_witness(in + 12 * i, wires + 135 * i);
}
