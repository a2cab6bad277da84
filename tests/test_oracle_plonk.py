"""End-to-end self-consistency of the oracle's PLONK layer on circuits of the reference's shape: the zkdsa
"simple signature" circuit (reference src/zkdsa/circuits/mod.rs:24-43, src/zkdsa/gadgets/signature/mod.rs:50-63:
pk = two_to_one(sk, sk), sig = two_to_one(sk, msg), public inputs [msg, pk, sig]) and a Merkle-path circuit
(reference src/merkle_tree/gadgets/mod.rs:70-111).  prove -> verify accepts; corrupted proofs, wrong public inputs and
unsatisfied witnesses are rejected.  ('parity unpinned' for this layer: the reference has no vectors)"""
import importlib

import numpy as np
import pytest

P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def circ():
    import __graft_entry__ as g

    g.load_package()
    return importlib.import_module("intmax_zkp_core_amd.circuit")


def zkdsa_circuit(circ):
    b = circ.CircuitBuilder()
    sk = [b.virtual() for _ in range(4)]
    msg = [b.virtual() for _ in range(4)]
    pk = b.two_to_one(sk, sk)
    sig = b.two_to_one(sk, msg)
    b.register_public_inputs(msg + pk + sig)
    return b.build(), sk, msg


def prove(oracle, cd, wires, pis, fast=True):
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"])
    fp = None
    if fast:  # fewer queries / PoW bits keep the CPU test quick; the standard parameters are used in one test below
        fp = (cd.degree_bits, 3, 4, 6, 4, oracle.fri_reduction_arity_bits(cd.degree_bits, 3, 4))
    return cs, fp, oracle.plonk_prove(cd, cs, wires, pis, fp)


def test_zkdsa_shape_circuit_roundtrip(oracle, circ, kat):
    cd, sk, msg = zkdsa_circuit(circ)
    assert cd.degree_bits == 3 and cd.num_selectors == 2 and cd.num_constants == 4
    assert [g["id"] for g in cd.gates] == [circ.GATE_NOOP, circ.GATE_CONSTANT, circ.GATE_PUBLIC_INPUT, circ.GATE_POSEIDON]
    skv = kat["accounts"]["private_keys"][1]
    msgv = [11, 22, 33, 44]
    inputs = dict(zip(sk, skv))
    inputs.update(zip(msg, msgv))
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    # public inputs = [msg, H(sk||sk), H(sk||msg)]; the address golden of block1_info.json re-appears as pk
    assert pis[:4] == msgv
    assert oracle.hashout_to_hex(pis[4:8]) == kat["accounts"]["sender_addresses"][1]
    assert pis[8:12] == [int(x) for x in oracle.two_to_one(skv, msgv)]
    cs, fp, proof = prove(oracle, cd, wires, pis)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp) == 0
    # wrong public input
    bad_pis = list(pis)
    bad_pis[0] += 1
    assert oracle.plonk_verify(cd, cs["cap"], bad_pis, proof, fp) != 0
    # corrupted opening (first wire opening) -> vanishing identity or FRI fails
    cap_words = 4 << 4
    off = 3 * cap_words + 2 * (cd.num_constants + 80)
    bad = proof.copy()
    bad[off] ^= np.uint64(1)
    assert oracle.plonk_verify(cd, cs["cap"], pis, bad, fp) != 0
    # corrupted cap
    bad = proof.copy()
    bad[0] ^= np.uint64(1)
    assert oracle.plonk_verify(cd, cs["cap"], pis, bad, fp) != 0


def test_unsatisfied_witness_is_rejected(oracle, circ):
    cd, sk, msg = zkdsa_circuit(circ)
    inputs = dict(zip(sk, [1, 2, 3, 4]))
    inputs.update(zip(msg, [5, 6, 7, 8]))
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    row = [i for i, (g, _) in enumerate(cd.builder.rows) if g == circ.GATE_POSEIDON][1]
    for col in (70, 13):  # an S-box witness wire (gate constraint), an output wire (gate + copy constraint)
        w = wires.copy()
        w[col, row] = (int(w[col, row]) + 1) % P
        cs, fp, proof = prove(oracle, cd, w, pis)
        assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp) != 0, col
    # breaking only a copy constraint (both gate rows stay internally consistent) must fail through the permutation
    rows = oracle.poseidon_gate_rows(np.array([[9] * 12], dtype=np.uint64))
    w = wires.copy()
    w[:, row] = rows[0]
    cs, fp, proof = prove(oracle, cd, w, pis)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp) != 0


def test_merkle_path_circuit_with_arithmetic(oracle, circ):
    """depth-6 Merkle path (two_to_one per level, reference src/merkle_tree/gadgets/mod.rs:70-111) + arithmetic gates;
    standard FRI parameters (16 PoW bits, 28 queries)."""
    b = circ.CircuitBuilder()
    leaf = [b.virtual() for _ in range(4)]
    sibs = [[b.virtual() for _ in range(4)] for _ in range(6)]
    cur = leaf
    for s in sibs:
        cur = b.two_to_one(cur, s)
    acc = b.mul(leaf[0], leaf[1])
    acc = b.add(acc, leaf[2])
    b.register_public_inputs(cur + [acc])
    cd = b.build()
    assert cd.degree_bits == 4 and len(cd.gates) == 5
    rng = np.random.default_rng(3)
    inputs = {t: int(v) for t, v in zip(leaf, rng.integers(0, P, 4, dtype=np.uint64))}
    for s in sibs:
        inputs.update({t: int(v) for t, v in zip(s, rng.integers(0, P, 4, dtype=np.uint64))})
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    want = [inputs[t] for t in leaf]
    for s in sibs:
        want = [int(x) for x in oracle.two_to_one(want, [inputs[t] for t in s])]
    assert pis[:4] == want
    assert pis[4] == (inputs[leaf[0]] * inputs[leaf[1]] + inputs[leaf[2]]) % P
    cs, fp, proof = prove(oracle, cd, wires, pis, fast=False)
    assert len(proof) == oracle.plonk_proof_len(cd)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof) == 0


def all_gates_circuit(circ):
    """every supported gate type in one circuit: a range-checked value split into bits (BaseSumGate<2>), recombined with
    arithmetic gates, an extension-field product (MulExtensionGate) and multiply-add (ArithmeticExtensionGate) as the
    reference's is_non_zero gadget uses (src/transaction/gadgets/utils/mod.rs:54-67), and Poseidon hashing."""
    b = circ.CircuitBuilder()
    x = b.virtual()
    bits = b.split_le(x, 40)
    acc = b.constant(0)
    for k in range(39, -1, -1):  # recombine: acc = 2 acc + bit
        acc = b.arithmetic(2, 1, acc, b.constant(1), bits[k])
    b.connect(acc, x)
    u = [b.virtual(), b.virtual()]
    v = [b.virtual(), b.virtual()]
    uv = b.mul_extension(u, v)
    w = b.arithmetic_extension(3, 5, uv, u, v)  # 3 * uv * u + 5 * v
    h = b.hash_n_to_hash_no_pad([x] + uv + w)
    b.register_public_inputs(h + w)
    return b.build(), x, u, v


def test_all_gate_types_roundtrip(oracle, circ):
    cd, x, u, v = all_gates_circuit(circ)
    ids = [g["id"] for g in cd.gates]
    assert ids == [circ.GATE_NOOP, circ.GATE_CONSTANT, circ.GATE_PUBLIC_INPUT, circ.GATE_BASE_SUM, circ.GATE_ARITHMETIC_EXT,
                   circ.GATE_ARITHMETIC, circ.GATE_MUL_EXT, circ.GATE_POSEIDON]
    assert cd.num_selectors == 2 and cd.gates[6]["selector_index"] == 1  # groups [0..6) and [6..8)
    xv, uvv, vv = 0xABCDE12345, [5, 7], [11, 13]
    inputs = {x: xv, u[0]: uvv[0], u[1]: uvv[1], v[0]: vv[0], v[1]: vv[1]}
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    # expected extension arithmetic: uv = u * v ; w = 3 uv u + 5 v  in F[X]/(X^2 - 7)
    def emul(a, b2):
        return [(a[0] * b2[0] + 7 * a[1] * b2[1]) % P, (a[0] * b2[1] + a[1] * b2[0]) % P]
    uv = emul(uvv, vv)
    t = emul(uv, uvv)
    w = [(3 * t[0] + 5 * vv[0]) % P, (3 * t[1] + 5 * vv[1]) % P]
    assert pis[4:6] == w
    cs, fp, proof = prove(oracle, cd, wires, pis)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp) == 0
    # a non-boolean limb, a wrong extension product, and an out-of-range value are all rejected
    rows = {g: [i for i, (gid, _) in enumerate(cd.builder.rows) if gid == g] for g in (circ.GATE_BASE_SUM, circ.GATE_MUL_EXT)}
    for (col, row) in ((3, rows[circ.GATE_BASE_SUM][0]), (4, rows[circ.GATE_MUL_EXT][0])):
        wbad = wires.copy()
        wbad[col, row] = (int(wbad[col, row]) + 2) % P
        cs2, fp2, proof2 = prove(oracle, cd, wbad, pis)
        assert oracle.plonk_verify(cd, cs2["cap"], pis, proof2, fp2) != 0, (col, row)
    with pytest.raises(ValueError):  # 2^41 does not fit the 40-bit range check: the witness itself is inconsistent
        cd.generate_witness({**inputs, x: 1 << 41}, oracle.poseidon_gate_rows)


def recursion_gates_circuit(circ):
    """the six gates builder.verify_proof adds (reference src/recursion/gadgets/mod.rs:114-127), wired the way the
    recursive FRI verifier uses them: a Merkle-cap style random access, reduce_with_powers over base and extension
    coefficients, x = g^bits by ExponentiationGate, a coset interpolation of 16 evaluations, and a PoseidonMds layer;
    everything flows into the public inputs."""
    b = circ.CircuitBuilder()
    idx = b.virtual()
    items = [b.virtual() for _ in range(16)]
    picked = b.random_access(idx, items)
    bits = b.split_le(idx, 4)
    alpha = [b.virtual(), b.virtual()]
    zero = b.constant(0)
    red = b.reduce_base(items + [picked], alpha, [zero, zero])
    vals = [[b.virtual(), b.virtual()] for _ in range(16)]
    red2 = b.reduce_extension(vals[:5] + [red], alpha, red)
    g = b.constant(circ.root_of_unity(4))
    xpow = b.exp_from_bits(g, bits)          # g^idx: a point of the 16-subgroup
    shift = b.virtual()
    ev = b.interpolate_coset(shift, vals, red2)
    mds = b.poseidon_mds(vals[:11] + [ev])
    b.register_public_inputs([picked, xpow] + red + red2 + ev + mds[0] + mds[11])
    return b.build(), dict(idx=idx, items=items, alpha=alpha, vals=vals, shift=shift)


def recursion_gates_inputs(t, seed=5):
    rng = np.random.default_rng(seed)

    def r():
        return int(rng.integers(1, P, dtype=np.uint64))

    inputs = {t["idx"]: 11, t["shift"]: r(), t["alpha"][0]: r(), t["alpha"][1]: r()}
    for it in t["items"]:
        inputs[it] = r()
    for pair in t["vals"]:
        for x in pair:
            inputs[x] = r()
    return inputs


def test_recursion_gates_roundtrip(oracle, circ):
    cd, t = recursion_gates_circuit(circ)
    ids = {g["id"] for g in cd.gates}
    assert {circ.GATE_RANDOM_ACCESS, circ.GATE_REDUCING, circ.GATE_REDUCING_EXT, circ.GATE_EXPONENTIATION,
            circ.GATE_POSEIDON_MDS, circ.GATE_COSET_INTERPOLATION} <= ids
    inputs = recursion_gates_inputs(t)
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    # spot values: the random access picked item 11, x = g^11, and the interpolant agrees with direct Lagrange evaluation
    assert pis[0] == inputs[t["items"][11]]
    assert pis[1] == pow(circ.root_of_unity(4), 11, P)
    red2 = (pis[4], pis[5])
    sh = inputs[t["shift"]]
    xs = [sh * pow(circ.root_of_unity(4), i, P) % P for i in range(16)]
    ys = [(inputs[p[0]], inputs[p[1]]) for p in t["vals"]]
    acc = (0, 0)
    for i in range(16):
        num, den = (1, 0), 1
        for j in range(16):
            if j != i:
                num = circ.ext_mul(num, ((red2[0] - xs[j]) % P, red2[1]))
                den = den * (xs[i] - xs[j]) % P
        term = circ.ext_mul(circ.ext_mul(num, ys[i]), (pow(den, P - 2, P), 0))
        acc = ((acc[0] + term[0]) % P, (acc[1] + term[1]) % P)
    assert (pis[6], pis[7]) == acc
    cs, fp, proof = prove(oracle, cd, wires, pis)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp) == 0
    # one broken wire per new gate type -> rejected
    rows = {g: [i for i, (gid, _) in enumerate(cd.builder.rows) if gid == g][0] for g in ids}
    breaks = [(circ.GATE_RANDOM_ACCESS, 74 + 1),      # a selector bit of copy 0
              (circ.GATE_REDUCING, 6 + 43 + 4),       # an intermediate accumulator
              (circ.GATE_REDUCING_EXT, 6 + 64 + 3),
              (circ.GATE_EXPONENTIATION, 2 + 66 + 40),  # an intermediate power
              (circ.GATE_POSEIDON_MDS, 24 + 5),       # an output word
              (circ.GATE_COSET_INTERPOLATION, 37 + 1)]  # an intermediate evaluation
    for gid, col in breaks:
        wbad = wires.copy()
        wbad[col, rows[gid]] = (int(wbad[col, rows[gid]]) + 1) % P
        cs2, fp2, proof2 = prove(oracle, cd, wbad, pis)
        assert oracle.plonk_verify(cd, cs2["cap"], pis, proof2, fp2) != 0, (gid, col)


def test_mixed_gate_bulk_circuit_roundtrip(oracle, circ):
    """the 14-gate bulk circuit of bench.py (recursive-circuit gate set, array-built witness): the oracle's prover output is
    accepted, and one broken wire in a row of each gate type is rejected"""
    mc = circ.MixedGateCircuit(7, 4)
    cd = mc.data
    assert len(cd.gates) == 14 and cd.num_selectors == 4
    wires, pis = mc.witness_host(oracle.poseidon_gate_rows, seed=3)
    cs, fp, proof = prove(oracle, cd, wires, pis)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp) == 0
    cols = {circ.GATE_ARITHMETIC: 3, circ.GATE_BASE_SUM: 0, circ.GATE_ARITHMETIC_EXT: 6, circ.GATE_MUL_EXT: 4,
            circ.GATE_RANDOM_ACCESS: 1, circ.GATE_REDUCING: 0, circ.GATE_REDUCING_EXT: 0, circ.GATE_EXPONENTIATION: 67,
            circ.GATE_POSEIDON_MDS: 24, circ.GATE_COSET_INTERPOLATION: 35}
    for gid, (r0, cnt) in mc.blocks.items():
        w = wires.copy()
        w[cols[gid], r0 + cnt // 2] = (int(w[cols[gid], r0 + cnt // 2]) + 1) % P
        bad = oracle.plonk_prove(cd, cs, w, pis, fp)
        assert oracle.plonk_verify(cd, cs["cap"], pis, bad, fp) != 0, gid


VARIANT_FLAGS = [0, 1, 2, 3, 0x100, 0x200, 0x203, 0x101]


@pytest.mark.parametrize("flags", VARIANT_FLAGS)
def test_transcript_variants_roundtrip(oracle, circ, flags):
    """the three version-sensitive choices (multiply the FRI final polynomial by X, the proof-of-work rule, the circuit
    digest composition; gl_oracle.h GLO_FLAG_*) are switchable; under every setting the oracle's prover and verifier
    agree, and a proof made under one setting is rejected under another"""
    cd, sk, msg = zkdsa_circuit(circ)
    inputs = dict(zip(sk, [1, 2, 3, 4]))
    inputs.update(zip(msg, [5, 6, 7, 8]))
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)

    def fp(fl):
        return (cd.degree_bits, 3, 4, 6, 4, oracle.fri_reduction_arity_bits(cd.degree_bits, 3, 4), fl)

    cd.circuit_digest = oracle.circuit_digest(cs["cap"], cd.degree_bits, flags)
    proof = oracle.plonk_prove(cd, cs, wires, pis, fp(flags))
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp(flags)) == 0
    base_digest = oracle.circuit_digest(cs["cap"])
    assert (cd.circuit_digest == base_digest).all() == ((flags & 0x300) == 0)
    for other in (flags ^ 1, flags ^ 2):  # same digest, different FRI rule
        assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp(other)) != 0
    if flags & 0x300:  # the verifier of the f99ed9c composition starts from another digest
        cd.circuit_digest = base_digest
        assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp(flags)) != 0
