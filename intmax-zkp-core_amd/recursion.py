"""Host-side mirror of plonky2's RECURSIVE VERIFIER on this repository's CircuitBuilder, behind the reference's interface for it:
`RecursiveProofTarget::add_virtual_to(builder, circuit_data)` / `set_witness(pw, proof, enabled)` (reference
src/recursion/gadgets/mod.rs:26-127 = `builder.add_virtual_proof_with_pis` + `builder.constant_hash` of the inner verifier data +
`builder.verify_proof` + `pw.set_proof_with_pis_target`).  This is what every aggregation circuit of the reference is made of: the
block circuit verifies one user-transaction proof and one signature proof per transaction
(src/rollup/circuits/mod.rs:389-624), the batch circuit verifies block proofs (src/rollup/gadgets/batch/mod.rs:21-55).

What is restated (plonky2 @ f99ed9c, [dep], absent from /root/reference; the native algorithm it mirrors is the oracle's verifier,
the verifier of oracle/plonk.c and oracle/fri.c, which the tests hold every GPU proof to):
  plonk/recursive_verifier.rs::verify_proof    public-input hash, challenges, vanishing(zeta) == Z_H(zeta) t(zeta), FRI
  iop/challenger.rs::RecursiveChallenger       the duplex sponge on PoseidonGate rows
  plonk/vanishing_poly.rs::eval_vanishing_poly_circuit + every gate's eval_unfiltered_circuit (all 14 gate types of this backend),
                                               in extension-field targets (ArithmeticExtensionGate / MulExtensionGate / PoseidonMdsGate rows)
  fri/recursive_verifier.rs::verify_fri_proof  proof of work, reduced openings (ReducingExtensionGate), per query: the four initial
                                               Merkle proofs (PoseidonGate rows with the swap wire, cap entry by RandomAccessGate),
                                               fri_combine_initial (ReducingGate), per reduction step: RandomAccessGate on the coset's
                                               evaluations, CosetInterpolationGate, ExponentiationGate for the coset start, Merkle proof;
                                               the final polynomial at the last point
The row count is this builder's (one reducing / exponentiation / interpolation op per row, no constant folding): an upper estimate
of plonky2's.  Transcript variant: the default flags (0) of include/glprover.h.  No arithmetic happens outside the builder: the
witness is whatever CircuitData.generate_witness derives from the proof's words."""
from . import codec
from .circuit import (GATE_ARITHMETIC, GATE_ARITHMETIC_EXT, GATE_BASE_SUM, GATE_CONSTANT, GATE_COSET_INTERPOLATION, GATE_EXPONENTIATION,
                      GATE_MUL_EXT, GATE_NOOP, GATE_POSEIDON, GATE_POSEIDON_MDS, GATE_PUBLIC_INPUT, GATE_RANDOM_ACCESS, GATE_REDUCING,
                      GATE_REDUCING_EXT, MDS_CIRC, UNUSED_SELECTOR, P, coset_interpolation_tables, ext_inv, ext_mul, root_of_unity)


def fri_reduction_arity_bits(degree_bits, rate_bits, cap_height, arity_bits=4, final_poly_bits=5):
    """FriReductionStrategy::ConstantArityBits(4, 5) of standard_recursion_config (fri/reduction_strategies.rs)"""
    out, d = [], degree_bits
    while d > final_poly_bits and d + rate_bits - arity_bits >= cap_height:
        out.append(arity_bits)
        d -= arity_bits
    return out


class Ext:
    """extension-field targets ([c0, c1] target pairs) on a CircuitBuilder: gadgets/arithmetic_extension.rs"""

    def __init__(self, b):
        self.b = b
        z, o = b.zero(), b.one()
        self.zero, self.one = [z, z], [o, z]

    def const(self, c0, c1=0):
        return [self.b.constant(c0), self.b.constant(c1)]

    def base(self, t):  # convert_to_ext
        return [t, self.b.zero()]

    def add(self, x, y):
        return self.b.arithmetic_extension(1, 1, x, self.one, y)

    def sub(self, x, y):
        return self.b.arithmetic_extension(1, P - 1, x, self.one, y)

    def mul(self, x, y):
        return self.b.mul_extension(x, y)

    def mul_add(self, x, y, z):  # x y + z
        return self.b.arithmetic_extension(1, 1, x, y, z)

    def mul_sub(self, x, y, z):  # x y - z
        return self.b.arithmetic_extension(1, P - 1, x, y, z)

    def scale_add(self, c, x, z):  # c x + z for a small constant c
        return self.b.arithmetic_extension(c, 1, x, self.one, z)

    def sq(self, x):
        return self.mul(x, x)

    def pow_const(self, x, e):
        acc, base = None, x
        while e:
            if e & 1:
                acc = base if acc is None else self.mul(acc, base)
            e >>= 1
            if e:
                base = self.sq(base)
        return self.one if acc is None else acc

    def div(self, x, y):
        """div_extension: q from a generator (QuotientGeneratorExtension), q * y == x"""
        b = self.b
        q = [b.virtual(), b.virtual()]

        def fn(get, put, raw):
            r = ext_mul((get(x[0]), get(x[1])), ext_inv((get(y[0]), get(y[1]))))
            put(q[0], r[0])
            put(q[1], r[1])

        b._gen(list(x) + list(y), fn)
        self.eq(self.mul(q, y), x)
        return q

    def eq(self, x, y):  # connect_extension
        self.b.connect(x[0], y[0])
        self.b.connect(x[1], y[1])


class Alg:
    """ExtensionAlgebraTarget: pairs of extension targets, (a0 + a1 X)(b0 + b1 X) mod X^2 - 7 (gadgets/arithmetic_extension.rs)"""

    def __init__(self, E):
        self.E = E

    def mul(self, x, y):
        E = self.E
        c0 = E.b.arithmetic_extension(7, 1, x[1], y[1], E.mul(x[0], y[0]))
        c1 = E.mul_add(x[0], y[1], E.mul(x[1], y[0]))
        return [c0, c1]

    def add(self, x, y):
        return [self.E.add(x[0], y[0]), self.E.add(x[1], y[1])]

    def sub(self, x, y):
        return [self.E.sub(x[0], y[0]), self.E.sub(x[1], y[1])]

    def scal(self, x, s):
        return [self.E.mul(x[0], s), self.E.mul(x[1], s)]


class RecursiveChallenger:
    """iop/challenger.rs::RecursiveChallenger: overwrite-mode duplex sponge, rate 8; outputs are popped from the END of the buffer"""

    def __init__(self, b):
        self.b = b
        self.state = [b.zero()] * 12
        self.inbuf, self.outbuf = [], []

    def observe(self, targets):
        for t in targets:
            self.outbuf = []
            self.inbuf.append(t)

    def _absorb(self):
        b = self.b
        if not self.inbuf:
            self.state = b.poseidon(self.state)
        for off in range(0, len(self.inbuf), 8):
            chunk = self.inbuf[off:off + 8]
            self.state = b.poseidon(list(chunk) + self.state[len(chunk):])
        self.inbuf = []
        self.outbuf = self.state[:8]

    def get(self):
        if self.inbuf or not self.outbuf:
            self._absorb()
        return self.outbuf.pop()

    def get_n(self, n):
        return [self.get() for _ in range(n)]


def _le_sum(b, bits):
    acc = bits[-1]
    for t in reversed(bits[:-1]):
        acc = b.arithmetic(2, 1, acc, b.one(), t)
    return acc


def _reduce(b, E, terms_ascending, alpha, ext):
    """ReducingFactorTarget::reduce / reduce_base: sum_k terms[k] alpha^k through Reducing(Extension)Gate rows — the terms reversed,
    zeros in FRONT of the highest power (reducing.rs)"""
    nc = b.cfg.reducing_ext_coeffs if ext else b.cfg.reducing_coeffs
    zero = E.zero if ext else b.zero()
    if len(terms_ascending) <= 4:  # a handful of terms: Horner in arithmetic ops instead of a whole gate row
        acc = E.zero
        for t in reversed(terms_ascending):
            acc = E.mul_add(acc, alpha, t if ext else E.base(t))
        return acc
    rev = list(terms_ascending)
    while len(rev) % nc:
        rev.append(zero)
    rev.reverse()
    acc = E.zero
    for off in range(0, len(rev), nc):
        chunk = rev[off:off + nc]
        acc = b.reduce_extension(chunk, alpha, acc) if ext else b.reduce_base(chunk, alpha, acc)
    return acc


def verify_merkle_proof(b, leaf, index_bits, siblings, cap):
    """verify_merkle_proof_to_cap: hash_or_noop of the leaf, one swapped PoseidonGate row per sibling, the cap entry by random access"""
    zero = b.zero()
    state = list(leaf) + [zero] * (4 - len(leaf)) if len(leaf) <= 4 else b.hash_n_to_hash_no_pad(list(leaf))
    for bit, sib in zip(index_bits, siblings):
        state = b.poseidon(state + list(sib) + [zero] * 4, swap=bit)[:4]
    cap_bits = index_bits[len(siblings):]
    if len(cap) == 1:
        for k in range(4):
            b.connect(state[k], cap[0][k])
        return
    assert len(cap) == 1 << len(cap_bits) == 1 << b.cfg.random_access_bits
    idx = _le_sum(b, cap_bits)
    for k in range(4):
        b.connect(b.random_access(idx, [h[k] for h in cap]), state[k])


# ---------------------------------------------------------------------------------------------------------------------------
# the gates' eval_unfiltered_circuit: w = the wire openings, consts = the gate's constants (selector prefix removed), all
# extension targets; returns the constraint list.  Statement for statement the CPU restatement's gate evaluator (oracle/plonk.c:237-392), in targets.
# ---------------------------------------------------------------------------------------------------------------------------
def _poseidon_gate_constraints(b, E, w, rc):
    """gates/poseidon.rs.  Round constants are carried as known offsets through the linear layers (MDS(s + k) = MDS(s) + MDS(k)) and
    added only where an S-box or a constraint needs the value: 4 x 12 + 22 + 4 x 12 + 12 constants instead of 30 x 12"""
    cons = []
    swap = w[24]
    cons.append(E.mul_sub(swap, swap, swap))
    for i in range(4):
        cons.append(E.mul_sub(swap, E.sub(w[4 + i], w[i]), w[25 + i]))
    st = [E.add(w[i], w[25 + i]) for i in range(4)] + [E.sub(w[4 + i], w[25 + i]) for i in range(4)] + [w[i] for i in range(8, 12)]
    off = [0] * 12  # the known part of each state word

    def real(i):
        v = st[i] if off[i] == 0 else E.add(st[i], E.const(off[i]))
        return v

    def sbox(x):
        x2 = E.sq(x)
        return E.mul(E.mul(x, x2), E.sq(x2))

    def mds():
        nonlocal st, off
        st = b.poseidon_mds(st)
        off = [(sum(MDS_CIRC[i] * off[(i + r) % 12] for i in range(12)) + (8 * off[0] if r == 0 else 0)) % P for r in range(12)]

    rnd = 0
    for r in range(4):
        off = [(o + rc[rnd * 12 + i]) % P for i, o in enumerate(off)]
        vals = [real(i) for i in range(12)]
        if r != 0:
            for i in range(12):
                cons.append(E.sub(vals[i], w[29 + 12 * (r - 1) + i]))
                vals[i] = w[29 + 12 * (r - 1) + i]
        st, off = [sbox(v) for v in vals], [0] * 12
        mds()
        rnd += 1
    for r in range(22):
        off = [(o + rc[rnd * 12 + i]) % P for i, o in enumerate(off)]
        cons.append(E.sub(real(0), w[65 + r]))
        st[0], off[0] = sbox(w[65 + r]), 0
        mds()
        rnd += 1
    for r in range(4):
        off = [(o + rc[rnd * 12 + i]) % P for i, o in enumerate(off)]
        vals = [real(i) for i in range(12)]
        for i in range(12):
            cons.append(E.sub(vals[i], w[87 + 12 * r + i]))
        st, off = [sbox(w[87 + 12 * r + i]) for i in range(12)], [0] * 12
        mds()
        rnd += 1
    for i in range(12):
        cons.append(E.sub(real(i), w[12 + i]))
    return cons


def gate_constraints(b, E, gate, consts, w, pi_hash, round_constants):
    gid, param = gate["id"], gate["param"]
    A = Alg(E)
    aw = lambda i: [w[i], w[i + 1]]  # noqa: E731  (get_local_ext_algebra)
    if gid == GATE_NOOP:
        return []
    if gid == GATE_CONSTANT:
        return [E.sub(consts[i], w[i]) for i in range(param)]
    if gid == GATE_PUBLIC_INPUT:
        return [E.sub(w[i], E.base(pi_hash[i])) for i in range(4)]
    if gid == GATE_ARITHMETIC:
        out = []
        for i in range(param):
            m0, m1, ad, o = w[4 * i:4 * i + 4]
            comp = E.mul_add(ad, consts[1], E.mul(E.mul(m0, m1), consts[0]))
            out.append(E.sub(o, comp))
        return out
    if gid == GATE_POSEIDON:
        return _poseidon_gate_constraints(b, E, w, round_constants)
    if gid == GATE_BASE_SUM:
        acc = E.zero
        for i in reversed(range(param)):
            acc = E.scale_add(2, acc, w[1 + i])
        return [E.sub(acc, w[0])] + [E.mul_sub(w[1 + i], w[1 + i], w[1 + i]) for i in range(param)]
    if gid in (GATE_ARITHMETIC_EXT, GATE_MUL_EXT):
        per = 8 if gid == GATE_ARITHMETIC_EXT else 6
        out = []
        for i in range(param):
            q = w[per * i:per * i + per]
            p = A.mul([q[0], q[1]], [q[2], q[3]])
            p = [E.mul(p[0], consts[0]), E.mul(p[1], consts[0])]
            o = q[4:6]
            if gid == GATE_ARITHMETIC_EXT:
                p = [E.mul_add(q[4], consts[1], p[0]), E.mul_add(q[5], consts[1], p[1])]
                o = q[6:8]
            out += [E.sub(o[0], p[0]), E.sub(o[1], p[1])]
        return out
    if gid == GATE_RANDOM_ACCESS:
        bits, copies, extra = param & 0xFF, (param >> 8) & 0xFF, (param >> 16) & 0xFF
        vec = 1 << bits
        routed = (2 + vec) * copies + extra
        out = []
        for cp in range(copies):
            q = w[(2 + vec) * cp:(2 + vec) * (cp + 1)]
            bt = w[routed + cp * bits:routed + (cp + 1) * bits]
            out += [E.mul_sub(t, t, t) for t in bt]
            rec = E.zero
            for i in reversed(range(bits)):
                rec = E.scale_add(2, rec, bt[i])
            out.append(E.sub(rec, q[0]))
            items = list(q[2:])
            for i in range(bits):
                items = [E.mul_add(bt[i], E.sub(items[2 * j + 1], items[2 * j]), items[2 * j]) for j in range(len(items) // 2)]
            out.append(E.sub(items[0], q[1]))
        out += [E.sub(consts[i], w[(2 + vec) * copies + i]) for i in range(extra)]
        return out
    if gid in (GATE_REDUCING, GATE_REDUCING_EXT):
        nc, is_ext = param, gid == GATE_REDUCING_EXT
        start_accs = 6 + (2 * nc if is_ext else nc)
        alpha, acc = aw(2), aw(4)
        out = []
        for i in range(nc):
            coeff = aw(6 + 2 * i) if is_ext else [w[6 + i], E.zero]
            nxt = aw(0) if i == nc - 1 else aw(start_accs + 2 * i)
            d = A.sub(A.add(A.mul(acc, alpha), coeff), nxt)
            out += d
            acc = nxt
        return out
    if gid == GATE_EXPONENTIATION:
        nb = param
        base, output = w[0], w[1 + nb]
        out = []
        for i in range(nb):
            prev = E.one if i == 0 else E.sq(w[2 + nb + i - 1])
            bit = w[1 + (nb - 1 - i)]
            factor = E.add(E.mul_sub(bit, base, bit), E.one)  # bit * base + 1 - bit
            out.append(E.mul_sub(prev, factor, w[2 + nb + i]))
        out.append(E.sub(output, w[2 + nb + nb - 1]))
        return out
    if gid == GATE_POSEIDON_MDS:
        ins = [aw(2 * i) for i in range(12)]
        out = []
        for r in range(12):
            acc = [E.zero, E.zero]
            for i in range(12):
                src = ins[(i + r) % 12]
                acc = [E.scale_add(MDS_CIRC[i], src[0], acc[0]), E.scale_add(MDS_CIRC[i], src[1], acc[1])]
            if r == 0:
                acc = [E.scale_add(8, ins[0][0], acc[0]), E.scale_add(8, ins[0][1], acc[1])]
            o = aw(24 + 2 * r)
            out += [E.sub(o[0], acc[0]), E.sub(o[1], acc[1])]
        return out
    if gid == GATE_COSET_INTERPOLATION:
        bits, degree = param & 0xFF, (param >> 8) & 0xFF
        npts = 1 << bits
        ni = (npts - 2) // (degree - 1)
        sp = 1 + 2 * npts
        sv, si = sp + 2, sp + 4
        dom, wts = coset_interpolation_tables(bits)
        shift = w[0]
        point, shifted = aw(sp), aw(si + 4 * ni)
        values = [aw(1 + 2 * i) for i in range(npts)]
        out = list(A.sub(point, A.scal(shifted, shift)))

        def partial(lo, hi, ev, pr):
            for k in range(lo, hi):
                term = [E.sub(shifted[0], E.const(dom[k])), shifted[1]]
                wk = E.const(wts[k])
                val = [E.mul(values[k][0], wk), E.mul(values[k][1], wk)]
                ev = A.add(A.mul(ev, term), A.mul(val, pr))
                pr = A.mul(pr, term)
            return ev, pr

        ev, pr = partial(0, degree, [E.zero, E.zero], [E.one, E.zero])
        for i in range(ni):
            ie, ip = aw(si + 2 * i), aw(si + 2 * (ni + i))
            out += A.sub(ie, ev) + A.sub(ip, pr)
            lo = 1 + (degree - 1) * (i + 1)
            ev, pr = partial(lo, min(lo + degree - 1, npts), ie, ip)
        out += A.sub(aw(sv), ev)
        return out
    raise ValueError("gate %d has no in-circuit evaluator" % gid)


class RecursiveProofTarget:
    """`RecursiveProofTarget::add_virtual_to(builder, circuit_data)` (src/recursion/gadgets/mod.rs:70-108): the targets of one inner
    proof with public inputs, the inner circuit's constants_sigmas cap and circuit digest as constants, `builder.verify_proof`, and
    the `enabled` bit the callers combine with the inner public inputs.  `inner`: the inner circuit's CircuitData with its commitment
    attached (constants_sigmas_cap [16][4] words and circuit_digest, as ProverCircuit / the oracle attach them);
    `round_constants`: Poseidon's 360 round constants (the package's table)."""

    def __init__(self, b, inner, cs_cap, circuit_digest, round_constants, n_public_inputs=None, zero_knowledge=False):
        """zero_knowledge: the inner proofs were made under config.zero_knowledge (GLP_FLAG_ZERO_KNOWLEDGE): the leaves of the three
        blinded oracles carry SALT_SIZE = 4 more words, which the Merkle proofs hash and the batch sums never reach (`unsalted_eval`)"""
        cfg = inner.cfg
        assert cfg.cap_height == 4 and cfg.num_challenges == b.cfg.num_challenges
        self.inner = inner
        self.arity_bits = fri_reduction_arity_bits(inner.degree_bits, cfg.rate_bits, cfg.cap_height)
        self.layout = codec.ProofLayout(inner, self.arity_bits, zero_knowledge=zero_knowledge)
        n_pis = n_public_inputs if n_public_inputs is not None else len(inner.builder.public_inputs)  # (circuits made without the builder say so)
        self.public_inputs = [b.virtual() for _ in range(n_pis)]
        probe = _index_view(self.layout)
        self.proof = [b.virtual() for _ in range(probe["len"])]
        self.view = _map_view(probe["proof"], self.proof)
        self.constants_sigmas_cap = [[b.constant(x) for x in h] for h in cs_cap]
        self.circuit_digest = [b.constant(x) for x in circuit_digest]
        self.round_constants = [int(x) for x in round_constants]
        self._verify(b)
        self.enabled = b.virtual_bool_safe()

    def set_witness(self, inputs, proof_flat, public_inputs, enabled=True):
        """pw.set_proof_with_pis_target + set_bool_target(enabled) into the `inputs` dict of generate_witness"""
        assert len(proof_flat) == len(self.proof) and len(public_inputs) == len(self.public_inputs)
        for t, v in zip(self.proof, proof_flat):
            inputs[t] = int(v)
        for t, v in zip(self.public_inputs, public_inputs):
            inputs[t] = int(v)
        inputs[self.enabled] = 1 if enabled else 0

    # -- plonk/recursive_verifier.rs::verify_proof --
    def _verify(self, b):
        inner, cfg, v = self.inner, self.inner.cfg, self.view
        E = Ext(b)
        nch = cfg.num_challenges
        flat = lambda hs: [t for h in hs for t in h]  # noqa: E731
        op = v["openings"]
        pi_hash = b.hash_n_to_hash_no_pad(self.public_inputs)
        ch = RecursiveChallenger(b)
        ch.observe(self.circuit_digest)
        ch.observe(pi_hash)
        ch.observe(flat(v["wires_cap"]))
        betas, gammas = ch.get_n(nch), ch.get_n(nch)
        ch.observe(flat(v["plonk_zs_partial_products_cap"]))
        alphas = ch.get_n(nch)
        ch.observe(flat(v["quotient_polys_cap"]))
        zeta = ch.get_n(2)
        for name in ("constants", "plonk_sigmas", "wires", "plonk_zs", "partial_products", "quotient_polys", "plonk_zs_next"):
            ch.observe(flat(op[name]))
        self.challenges = dict(betas=betas, gammas=gammas, alphas=alphas, zeta=zeta)
        # ---- vanishing(zeta) == Z_H(zeta) * t(zeta) ----
        zeta_pow_n = zeta
        for _ in range(inner.degree_bits):
            zeta_pow_n = E.sq(zeta_pow_n)
        z_h = E.sub(zeta_pow_n, E.one)
        l0 = E.div(z_h, E.b.arithmetic_extension(1 << inner.degree_bits, 0, E.sub(zeta, E.one), E.one, E.zero))
        terms = [E.mul(l0, E.sub(op["plonk_zs"][i], E.one)) for i in range(nch)]
        nr, chunk, npp = cfg.num_routed_wires, inner.quotient_degree_factor, inner.num_partial_products
        wires, sigmas = op["wires"], op["plonk_sigmas"]
        assert all(k == pow(7, j, P) for j, k in enumerate(inner.k_is[:nr]))  # k_i = g^i: s_id_j = 7 * s_id_(j-1), no constants needed
        s_ids = [zeta]
        for _ in range(1, nr):
            s_ids.append(b.arithmetic_extension(7, 0, s_ids[-1], E.one, E.zero))
        for i in range(nch):
            beta, gamma = E.base(betas[i]), E.base(gammas[i])
            for ck in range(-(-nr // chunk)):
                nump, denp = None, None
                for j in range(ck * chunk, min((ck + 1) * chunk, nr)):
                    wg = E.add(wires[j], gamma)
                    nt, dt = E.mul_add(s_ids[j], beta, wg), E.mul_add(sigmas[j], beta, wg)
                    nump = nt if nump is None else E.mul(nump, nt)
                    denp = dt if denp is None else E.mul(denp, dt)
                prev = op["plonk_zs"][i] if ck == 0 else op["partial_products"][i * npp + ck - 1]
                nxt = op["plonk_zs_next"][i] if ck == npp else op["partial_products"][i * npp + ck]
                terms.append(E.mul_sub(prev, nump, E.mul(nxt, denp)))
        gc = [None] * inner.num_gate_constraints
        consts = op["constants"]
        for g, gate in enumerate(inner.gates):
            if gate["id"] == GATE_NOOP:
                continue
            s = consts[gate["selector_index"]]
            filt = None
            for r in list(range(gate["group_start"], gate["group_end"])) + ([UNUSED_SELECTOR] if inner.num_selectors > 1 else []):
                if r != g:
                    f = E.sub(E.const(r), s)
                    filt = f if filt is None else E.mul(filt, f)
            cons = gate_constraints(b, E, gate, consts[inner.num_selectors:], wires, pi_hash, self.round_constants)
            for j, c in enumerate(cons):
                if filt is not None:
                    gc[j] = E.mul(filt, c) if gc[j] is None else E.mul_add(filt, c, gc[j])
                else:
                    gc[j] = c if gc[j] is None else E.add(gc[j], c)
        terms += [E.zero if c is None else c for c in gc]
        for i in range(nch):
            alpha = E.base(alphas[i])
            acc = E.zero
            for t in reversed(terms):
                acc = E.mul_add(acc, alpha, t)
            q = E.zero
            for j in reversed(range(chunk)):
                q = E.mul_add(q, zeta_pow_n, op["quotient_polys"][i * chunk + j])
            E.eq(acc, E.mul(z_h, q))
        self._verify_fri(b, E, ch, zeta)

    # -- fri/recursive_verifier.rs::verify_fri_proof --
    def _verify_fri(self, b, E, ch, zeta):
        inner, cfg, v, L = self.inner, self.inner.cfg, self.view, self.layout
        op, fri = v["openings"], v["opening_proof"]
        flat = lambda hs: [t for h in hs for t in h]  # noqa: E731
        lde_bits = inner.degree_bits + cfg.rate_bits
        alpha = ch.get_n(2)
        betas = []
        for cap in fri["commit_phase_merkle_caps"]:
            ch.observe(flat(cap))
            betas.append(ch.get_n(2))
        ch.observe(flat(fri["final_poly"]))
        # proof of work: hash_no_pad(4 challenges ++ witness)[0] has proof_of_work_bits leading zeros
        h = b.hash_n_to_hash_no_pad(ch.get_n(4) + [fri["pow_witness"]])
        b.split_le(h[0], 64 - cfg.proof_of_work_bits)
        # the FRI instance of a plonk proof: every polynomial at zeta, the Zs at g * zeta (oracle/plonk.c::make_fri_instance)
        nch = cfg.num_challenges
        cols = inner.oracle_cols()  # the polynomials of each oracle; a blinded oracle's leaf is wider (L.oracle_cols) by its salt words
        batch0_polys = [(o, i) for o in range(4) for i in range(cols[o])]
        batch1_polys = [(2, i) for i in range(nch)]
        batch0_open = op["constants"] + op["plonk_sigmas"] + op["wires"] + op["plonk_zs"] + op["partial_products"] + op["quotient_polys"]
        zeta_next = E.mul(zeta, E.base(b.constant(root_of_unity(inner.degree_bits))))
        batches = [(batch0_polys, batch0_open, zeta), (batch1_polys, op["plonk_zs_next"], zeta_next)]
        reduced = [_reduce(b, E, opens, alpha, True) for _, opens, _ in batches]  # PrecomputedReducedOpenings
        shifts = [E.pow_const(alpha, len(polys)) for polys, _, _ in batches]
        caps = [self.constants_sigmas_cap, v["wires_cap"], v["plonk_zs_partial_products_cap"], v["quotient_polys_cap"]]
        w_n = b.constant(root_of_unity(lde_bits))
        for qr in fri["query_round_proofs"]:
            x = ch.get()
            bits = b.split_le_wide(x, 64)[:lde_bits]
            rows = []
            for o, tree in enumerate(qr["initial_trees_proof"]):
                rows.append(tree["evals"])
                verify_merkle_proof(b, tree["evals"], bits, tree["siblings"], caps[o])
            # subgroup_x = 7 * w_N^reverse_bits(x_index)
            sub_x = b.arithmetic(7, 0, b.exp_from_bits(w_n, bits[::-1]), b.one(), b.zero())
            total = E.zero
            for k, (polys, _, point) in enumerate(batches):
                acc = _reduce(b, E, [rows[o][i] for o, i in polys], alpha, False)
                numer = E.sub(acc, reduced[k])
                denom = E.sub(E.base(sub_x), point)
                total = E.mul_add(total, shifts[k], E.div(numer, denom))
            old_eval = E.mul(total, E.base(sub_x))  # the "multiply by X" of prove_openings
            for r, step in enumerate(qr["steps"]):
                ab = self.arity_bits[r]
                assert ab == b.cfg.interpolation_bits == b.cfg.random_access_bits
                evals = step["evals"]
                within, coset = bits[:ab], bits[ab:]
                idx = _le_sum(b, within)
                for k in range(2):
                    b.connect(b.random_access(idx, [e[k] for e in evals]), old_eval[k])
                # compute_evaluation: the coset starts at x * g^-reverse_bits(within); its points in natural order
                g_inv = pow(root_of_unity(ab), P - 2, P)
                start = b.mul(sub_x, b.exp_from_bits(b.constant(g_inv), within[::-1]))
                natural = [evals[int(format(i, "0%db" % ab)[::-1], 2)] for i in range(1 << ab)]
                old_eval = b.interpolate_coset(start, natural, betas[r])
                verify_merkle_proof(b, flat(evals), coset, step["siblings"], fri["commit_phase_merkle_caps"][r])
                for _ in range(ab):
                    sub_x = b.mul(sub_x, sub_x)
                bits = coset
            acc = E.zero
            xe = E.base(sub_x)
            for c in reversed(fri["final_poly"]):
                acc = E.mul_add(acc, xe, c)
            E.eq(acc, old_eval)


def _index_view(layout):
    """the proof's structure (codec.proof_from_flat) over word INDICES, and the number of words"""
    n = _flat_len(layout)
    return {"len": n, "proof": codec.proof_from_flat(list(range(n)), layout)}


def _flat_len(L):
    cap = 4 << L.cap_height
    n = 3 * cap + 2 * sum(c for _, c in L.openings) + len(L.arity_bits) * cap
    per_query = sum(c + 4 * L.path_len(L.lde_bits) for c in L.oracle_cols)
    bits = L.lde_bits
    for a in L.arity_bits:
        bits -= a
        per_query += (2 << a) + 4 * L.path_len(bits)
    return n + L.num_query_rounds * per_query + 2 * L.final_poly_len + 1


def _map_view(x, targets):
    if isinstance(x, dict):
        return {k: _map_view(v, targets) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_map_view(v, targets) for v in x]
    return targets[x]


class RecursionCircuit:
    """The circuit of the reference's own recursion tests (src/recursion/gadgets/mod.rs:129-260): one RecursiveProofTarget and
    nothing else; the inner proof's public inputs are forwarded as this circuit's public inputs (so that chains of such circuits
    carry their statement along — the reference's test registers none)."""

    def __init__(self, inner, cs_cap, circuit_digest, round_constants, config=None, zero_knowledge=False, n_public_inputs=None):
        from .circuit import CircuitBuilder, CircuitConfig

        b = CircuitBuilder(CircuitConfig if config is None else config)
        self.target = RecursiveProofTarget(b, inner, cs_cap, circuit_digest, round_constants, n_public_inputs, zero_knowledge)
        b.register_public_inputs(self.target.public_inputs)
        self.data = b.build()

    def witness(self, proof_flat, public_inputs, poseidon_rows_fn, enabled=True):
        inputs = {}
        self.target.set_witness(inputs, proof_flat, public_inputs, enabled)
        return self.data.generate_witness(inputs, poseidon_rows_fn)


class BlockBatchCircuit:
    """`BlockBatchTarget::add_virtual_to(builder, block_circuit_data, n_blocks)` (src/rollup/gadgets/batch/mod.rs:15-55) as a circuit:
    n_blocks RecursiveProofTargets of the same inner circuit.  `set_witness`: the given proofs with enabled = true; the unused slots
    get the LAST given proof again with enabled = false.  Public inputs: every slot's inner public inputs, then the enabled bits."""

    def __init__(self, inner, cs_cap, circuit_digest, round_constants, n_blocks=2, config=None):
        from .circuit import CircuitBuilder, CircuitConfig

        b = CircuitBuilder(CircuitConfig if config is None else config)
        self.block_proofs = [RecursiveProofTarget(b, inner, cs_cap, circuit_digest, round_constants) for _ in range(n_blocks)]
        for t in self.block_proofs:
            b.register_public_inputs(t.public_inputs)
        b.register_public_inputs([t.enabled for t in self.block_proofs])
        self.data = b.build()

    def witness(self, proofs, poseidon_rows_fn):
        """proofs: [(flat proof, public inputs)]"""
        assert 0 < len(proofs) <= len(self.block_proofs)
        inputs = {}
        for t, (pf, pis) in zip(self.block_proofs, proofs):
            t.set_witness(inputs, pf, pis, True)
        for t in self.block_proofs[len(proofs):]:
            t.set_witness(inputs, proofs[-1][0], proofs[-1][1], False)
        return self.data.generate_witness(inputs, poseidon_rows_fn)
