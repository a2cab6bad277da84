"""GPU parity for the PLONK layer (SURVEY.md §8a rows a1, a6, a7 + N3) through the C-ABI: partial products / Z,
quotient chunks and the whole proof must equal the oracle's word for word; the oracle's restated verifier must accept
the GPU proof and reject it for wrong public inputs."""
import importlib

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g

    return g.load_package()


@pytest.fixture(scope="module")
def circ(pkg):
    return importlib.import_module("intmax_zkp_core_amd.circuit")


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def hash_chain_circuit(circ, n_hashes, with_arith=True):
    """public inputs: a 4-limb seed, the end of a two_to_one chain over `n_hashes` levels with fresh siblings, and an
    arithmetic expression of the seed (Merkle-path shape, reference src/merkle_tree/gadgets/mod.rs:70-111)."""
    b = circ.CircuitBuilder()
    seed = [b.virtual() for _ in range(4)]
    sibs = [[b.virtual() for _ in range(4)] for _ in range(n_hashes)]
    cur = seed
    for s in sibs:
        cur = b.two_to_one(cur, s)
    pis = seed + cur
    if with_arith:
        acc = b.mul(seed[0], seed[1])
        for k in range(25):
            acc = b.add(b.mul(acc, seed[k % 4]), seed[(k + 1) % 4])
        pis = pis + [acc]
    b.register_public_inputs(pis)
    return b, seed, sibs


def make(pkg, circ, ctx, oracle, n_hashes, seed_val, with_arith=True, min_bits=0):
    b, seed, sibs = hash_chain_circuit(circ, n_hashes, with_arith)
    cd = b.build(min_bits)
    rng = np.random.default_rng(seed_val)
    inputs = {t: int(v) for t, v in zip(seed, rng.integers(0, P, 4, dtype=np.uint64))}
    for s in sibs:
        inputs.update({t: int(v) for t, v in zip(s, rng.integers(0, P, 4, dtype=np.uint64))})
    wires, pis = cd.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
    return cd, wires, pis


def test_poseidon_gate_rows_match_oracle(pkg, ctx, oracle):
    rng = np.random.default_rng(1)
    ins = rng.integers(0, 2**64, (300, 12), dtype=np.uint64)
    got = pkg.poseidon_gate_rows(ctx, ins)
    want = oracle.poseidon_gate_rows(ins)
    assert (got == want).all()
    assert (got[:, 12:24] == np.array([oracle.permute(r) for r in ins])).all()


@pytest.mark.parametrize("n_hashes,min_bits", [(3, 0), (20, 0), (100, 0), (40, 9), (300, 0)])
def test_partial_products_and_quotient_match_oracle(pkg, circ, ctx, oracle, n_hashes, min_bits):
    cd, wires, pis = make(pkg, circ, ctx, oracle, n_hashes, seed_val=n_hashes, min_bits=min_bits)
    pc = pkg.ProverCircuit(ctx, cd)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
    assert (pc.circuit_digest == oracle.circuit_digest(cs["cap"])).all()
    rng = np.random.default_rng(5)
    betas, gammas, alphas = (rng.integers(0, P, 2, dtype=np.uint64) for _ in range(3))
    wd = ctx.to_device(wires)
    got_zs = ctx.to_host(pc.partial_products(wd, betas, gammas))
    want_zs = oracle.partial_products_and_zs(cd, wires, betas, gammas)
    assert (got_zs == want_zs).all()
    # Z wraps around to 1: the permutation argument holds for a satisfied witness
    wb = pkg.PolynomialBatch.from_values(ctx, wd, 3, False, 4)
    zb = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(want_zs), 3, False, 4)
    pi_hash = oracle.hash_no_pad(pis)
    got_q = ctx.to_host(pc.quotient_chunks(wb, zb, betas, gammas, alphas, pi_hash))
    if cd.degree_bits <= 9:
        wo = oracle.batch_from_values(wires, 3, 4)
        zo = oracle.batch_from_values(want_zs, 3, 4)
        want_q = oracle.quotient_chunks(cd, cs, wo, zo, betas, gammas, alphas, pi_hash)
        assert (got_q == want_q).all()
    # the quotient has degree < 8n: nothing to check beyond equality; but a broken witness must NOT divide:
    wb.free()
    zb.free()
    pc.free()


@pytest.mark.parametrize("n_hashes,min_bits,with_arith", [(2, 0, False), (20, 0, True), (60, 0, True), (30, 9, True),
                                                          (500, 0, True)])
def test_proof_matches_oracle_and_verifies(pkg, circ, ctx, oracle, n_hashes, min_bits, with_arith):
    cd, wires, pis = make(pkg, circ, ctx, oracle, n_hashes, seed_val=100 + n_hashes, with_arith=with_arith, min_bits=min_bits)
    pc = pkg.ProverCircuit(ctx, cd)
    proof = pc.prove(ctx.to_device(wires), pis)
    assert len(proof) == oracle.plonk_proof_len(cd)
    cap = pc.constants_sigmas_cap.elements
    assert oracle.plonk_verify(cd, cap, pis, proof) == 0, "oracle verifier rejects the GPU proof"
    bad = list(pis)
    bad[-1] = (bad[-1] + 1) % P
    assert oracle.plonk_verify(cd, cap, bad, proof) != 0
    if cd.degree_bits <= 9:  # the CPU prover is slow beyond that
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        want = oracle.plonk_prove(cd, cs, wires, pis)
        diff = np.nonzero(proof != want)[0]
        assert diff.size == 0, "first differing proof word %d of %d" % (diff[0], len(proof))
    # host-pointer entry point gives the same proof
    if n_hashes == 20:
        assert (pc.prove(wires, pis) == proof).all()
    pc.free()


def test_broken_witness_gives_rejected_proof(pkg, circ, ctx, oracle):
    cd, wires, pis = make(pkg, circ, ctx, oracle, 10, seed_val=9)
    pc = pkg.ProverCircuit(ctx, cd)
    row = [i for i, (g, _) in enumerate(cd.builder.rows) if g == circ.GATE_POSEIDON][3]
    w = wires.copy()
    w[70, row] = (int(w[70, row]) + 1) % P
    proof = pc.prove(ctx.to_device(w), pis)
    assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, proof) != 0
    pc.free()


def test_all_gate_types_proof_matches_oracle(pkg, circ, ctx, oracle):
    """BaseSumGate<2>, ArithmeticExtensionGate, MulExtensionGate next to the other five gate types."""
    import tests.test_oracle_plonk as top

    cd, x, u, v = top.all_gates_circuit(circ)
    inputs = {x: 0x1234567890, u[0]: 3, u[1]: 4, v[0]: 5, v[1]: 6}
    wires, pis = cd.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
    pc = pkg.ProverCircuit(ctx, cd)
    proof = pc.prove(ctx.to_device(wires), pis)
    assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, proof) == 0
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    assert (proof == oracle.plonk_prove(cd, cs, wires, pis)).all()
    wbad = wires.copy()
    row = [i for i, (gid, _) in enumerate(cd.builder.rows) if gid == circ.GATE_ARITHMETIC_EXT][0]
    wbad[6, row] = (int(wbad[6, row]) + 1) % P
    assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, pc.prove(ctx.to_device(wbad), pis)) != 0
    pc.free()


def test_recursion_gate_types_proof_matches_oracle(pkg, circ, ctx, oracle):
    """RandomAccessGate, ReducingGate, ReducingExtensionGate, ExponentiationGate, PoseidonMdsGate, CosetInterpolationGate —
    the gates builder.verify_proof adds (reference src/recursion/gadgets/mod.rs:114-127): partial products, quotient
    chunks and the whole proof equal the oracle's; a broken wire in each new gate yields a rejected proof; out-of-range
    gate parameters are refused."""
    import tests.test_oracle_plonk as top

    cd, t = top.recursion_gates_circuit(circ)
    assert len(cd.gates) > 8  # more distinct gates than the first version of the kernel could hold
    inputs = top.recursion_gates_inputs(t, seed=9)
    wires, pis = cd.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
    pc = pkg.ProverCircuit(ctx, cd)
    proof = pc.prove(ctx.to_device(wires), pis)
    assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, proof) == 0
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    assert (proof == oracle.plonk_prove(cd, cs, wires, pis)).all()
    rows = {g["id"]: [i for i, (gid, _) in enumerate(cd.builder.rows) if gid == g["id"]][0] for g in cd.gates}
    for gid, col in ((circ.GATE_RANDOM_ACCESS, 75), (circ.GATE_REDUCING, 53), (circ.GATE_REDUCING_EXT, 73),
                     (circ.GATE_EXPONENTIATION, 108), (circ.GATE_POSEIDON_MDS, 29), (circ.GATE_COSET_INTERPOLATION, 38)):
        wbad = wires.copy()
        wbad[col, rows[gid]] = (int(wbad[col, rows[gid]]) + 1) % P
        bad = pc.prove(ctx.to_device(wbad), pis)
        assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, bad) != 0, (gid, col)
        assert (bad == oracle.plonk_prove(cd, cs, wbad, pis)).all()  # even the wrong quotient is the same wrong quotient
    pc.free()
    for gid, param, status in ((circ.GATE_RANDOM_ACCESS, 5 | 2 << 8, -5), (circ.GATE_COSET_INTERPOLATION, 5 | 6 << 8, -5),
                               (circ.GATE_REDUCING, 70, -1), (circ.GATE_EXPONENTIATION, 100, -1)):
        cd2, _ = top.recursion_gates_circuit(circ)
        [g for g in cd2.gates if g["id"] == gid][0]["param"] = param
        with pytest.raises(pkg.GlproverError) as e:
            pkg.ProverCircuit(ctx, cd2)
        assert e.value.status == status, (gid, param)


def test_zkdsa_circuit_on_the_gpu(pkg, circ, ctx, oracle, kat):
    """BASELINE configs[0] on the HIP path: the zkdsa "simple signature" circuit (reference src/zkdsa/circuits/mod.rs:24-43:
    pk = two_to_one(sk, sk), sig = two_to_one(sk, msg), public inputs [msg, pk, sig]) with the fixed sender key of
    src/bin/block_circuit.rs:157-164 — the GPU proof equals the oracle's word for word, its public key output is the
    sender address of test_cases/block1_info.json, and the default proof (sk = msg = 0, src/rollup/circuits/mod.rs:1301)
    carries the two_to_one(0, 0) golden."""
    import tests.test_oracle_plonk as top

    cd, sk, msg = top.zkdsa_circuit(circ)
    pc = pkg.ProverCircuit(ctx, cd)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
    for skv, msgv in ((kat["accounts"]["private_keys"][1], [11, 22, 33, 44]), ([0, 0, 0, 0], [0, 0, 0, 0])):
        inputs = dict(zip(sk, skv))
        inputs.update(zip(msg, msgv))
        wires, pis = cd.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
        assert pis[:4] == msgv
        assert pis[4:8] == [int(x) for x in oracle.two_to_one(skv, skv)]
        assert pis[8:12] == [int(x) for x in oracle.two_to_one(skv, msgv)]
        if any(skv):
            assert oracle.hashout_to_hex(pis[4:8]) == kat["accounts"]["sender_addresses"][1]
        else:
            assert pis[4:8] == pis[8:12] == [int(x) for x in kat["two_to_one_zero"]["digest"]]
        for w in (ctx.to_device(wires), wires):  # device-resident witness and the host-pointer entry point
            proof = pc.prove(w, pis)
            assert oracle.plonk_verify(cd, cs["cap"], pis, proof) == 0
            assert (proof == oracle.plonk_prove(cd, cs, wires, pis)).all()
        bad = list(pis)
        bad[5] = (bad[5] + 1) % P
        assert oracle.plonk_verify(cd, cs["cap"], bad, proof) != 0
    pc.free()


def test_mixed_gate_circuit_quotient_and_proof_match_oracle(pkg, circ, ctx, oracle):
    """the 14-gate bulk circuit (every gate `verify_proof` brings, hundreds of rows per gate type) at 2^12 rows: partial
    products, the 16 quotient chunks and the whole proof equal the oracle's; a broken row of each gate type is rejected"""
    mc = circ.MixedGateCircuit(12, 8)
    cd = mc.data
    assert len(cd.gates) == 14
    wd, pis = mc.witness_device(ctx, pkg, seed=4)
    wires = ctx.to_host(wd)
    want_w, want_pis = mc.witness_host(oracle.poseidon_gate_rows, seed=4)
    other = np.ones(cd.n, dtype=bool)
    other[mc.row_first:mc.row_pi_gate + 1] = False  # the Poseidon rows use different random inputs on the two paths
    assert (wires[:, other] == want_w[:, other]).all()
    pc = pkg.ProverCircuit(ctx, cd)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    rng = np.random.default_rng(5)
    betas, gammas, alphas = (rng.integers(0, P, 2, dtype=np.uint64) for _ in range(3))
    got_zs = ctx.to_host(pc.partial_products(wd, betas, gammas))
    want_zs = oracle.partial_products_and_zs(cd, wires, betas, gammas)
    assert (got_zs == want_zs).all()
    wb = pkg.PolynomialBatch.from_values(ctx, wd, 3, False, 4)
    zb = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(want_zs), 3, False, 4)
    pi_hash = oracle.hash_no_pad(pis)
    got_q = ctx.to_host(pc.quotient_chunks(wb, zb, betas, gammas, alphas, pi_hash))
    wo = oracle.batch_from_values(wires, 3, 4)
    zo = oracle.batch_from_values(want_zs, 3, 4)
    want_q = oracle.quotient_chunks(cd, cs, wo, zo, betas, gammas, alphas, pi_hash)
    assert (got_q == want_q).all()
    wb.free()
    zb.free()
    proof = pc.prove(wd, pis)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof) == 0
    assert (proof == oracle.plonk_prove(cd, cs, wires, pis)).all()
    cols = {circ.GATE_ARITHMETIC: 3, circ.GATE_BASE_SUM: 0, circ.GATE_ARITHMETIC_EXT: 6, circ.GATE_MUL_EXT: 4,
            circ.GATE_RANDOM_ACCESS: 1, circ.GATE_REDUCING: 0, circ.GATE_REDUCING_EXT: 0, circ.GATE_EXPONENTIATION: 67,
            circ.GATE_POSEIDON_MDS: 24, circ.GATE_COSET_INTERPOLATION: 35}
    for gid, (r0, cnt) in mc.blocks.items():
        w2 = wd.clone()
        w2[cols[gid], r0 + cnt // 2] += 1
        assert oracle.plonk_verify(cd, cs["cap"], pis, pc.prove(w2, pis)) != 0, gid
    pc.free()


@pytest.mark.parametrize("flags", [0, 1, 2, 3, 0x100, 0x200, 0x203, 0x101])
def test_transcript_variants_match_oracle(pkg, circ, ctx, oracle, flags):
    """the version-sensitive choices of the restated transcript (include/glprover.h GLP_FLAG_*: multiply-by-X of the FRI final
    polynomial, proof-of-work rule, circuit digest composition) are one-bit flips: under every setting the GPU proof equals
    the oracle's word for word and the restated verifier accepts it"""
    cd, wires, pis = make(pkg, circ, ctx, oracle, 30, seed_val=77, min_bits=9)
    fp = pkg.FriParams(cd.degree_bits, flags=flags)
    pc = pkg.ProverCircuit(ctx, cd, fp)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    assert (pc.circuit_digest == oracle.circuit_digest(cs["cap"], cd.degree_bits, flags)).all()
    proof = pc.prove(ctx.to_device(wires), pis)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp.as_tuple()) == 0
    assert (proof == oracle.plonk_prove(cd, cs, wires, pis, fp.as_tuple())).all()
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, pkg.FriParams(cd.degree_bits, flags=flags ^ 3).as_tuple()) != 0
    if flags == 0x203:  # a host that computes common_data.circuit_digest itself hands it over
        pc0 = pkg.ProverCircuit(ctx, cd, pkg.FriParams(cd.degree_bits, flags=3))
        pc0.set_digest(oracle.circuit_digest(cs["cap"], cd.degree_bits, flags))
        assert (pc0.prove(ctx.to_device(wires), pis) == proof).all()
        pc0.free()
    pc.free()


@pytest.mark.parametrize("fused", ["0", "1", "persistent"])
@pytest.mark.parametrize("kind", ["mixed", "chain", "tiny"])
def test_both_quotient_evaluators_match_oracle(pkg, circ, ctx, oracle, monkeypatch, fused, kind):
    """the constraint quotient has two evaluators — per-point kernels (quotient_kernel + poseidon_gate_kernel) and the
    LDS-tiled fused kernel (quotient_fused.h; default for circuits with the recursive gate set) — and both must give the
    oracle's quotient chunks for every circuit shape: 14 gates, 5 gates, and a 2^3-row circuit (8 LDE rows short of a tile).
    "persistent" = the fused kernel's opt-in persistent form (one workgroup per CU walking the tiles, GLP_FQ_PERSISTENT=1)."""
    monkeypatch.setenv("GLP_QUOT_FUSED", "1" if fused == "persistent" else fused)
    if fused == "persistent":
        monkeypatch.setenv("GLP_FQ_PERSISTENT", "1")
    if kind == "mixed":
        mc = circ.MixedGateCircuit(9, 3)
        cd = mc.data
        wires, pis = mc.witness_host(lambda ins: pkg.poseidon_gate_rows(ctx, ins), seed=6)
    elif kind == "chain":
        cd, wires, pis = make(pkg, circ, ctx, oracle, 40, seed_val=3)
    else:
        import tests.test_oracle_plonk as top

        cd, sk, msg = top.zkdsa_circuit(circ)
        inputs = dict(zip(sk, [9, 8, 7, 6]))
        inputs.update(zip(msg, [1, 2, 3, 4]))
        wires, pis = cd.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
    pc = pkg.ProverCircuit(ctx, cd)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    rng = np.random.default_rng(8)
    betas, gammas, alphas = (rng.integers(0, P, 2, dtype=np.uint64) for _ in range(3))
    want_zs = oracle.partial_products_and_zs(cd, wires, betas, gammas)
    wb = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(wires), 3, False, 4)
    zb = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(want_zs), 3, False, 4)
    pi_hash = oracle.hash_no_pad(pis)
    got_q = ctx.to_host(pc.quotient_chunks(wb, zb, betas, gammas, alphas, pi_hash))
    want_q = oracle.quotient_chunks(cd, cs, oracle.batch_from_values(wires, 3, 4), oracle.batch_from_values(want_zs, 3, 4), betas,
                                    gammas, alphas, pi_hash)
    assert (got_q == want_q).all()
    proof = pc.prove(ctx.to_device(wires), pis)
    assert (proof == oracle.plonk_prove(cd, cs, wires, pis)).all()
    wb.free()
    zb.free()
    pc.free()


@pytest.mark.parametrize("kind,K", [("zkdsa", 5), ("zkdsa", 67), ("chain", 3), ("mixed", 4), ("chain9", 2), ("layered12", 3),
                                    ("layered13", 2)])
def test_batched_proofs_equal_single_proofs(pkg, circ, ctx, oracle, kind, K):
    """glp_prove_batch_dev: K witnesses of one circuit proven together (K x C columns per commitment, a forest of K Merkle
    trees, K transcripts side by side) — every proof equals the single-proof output word for word, is accepted by the restated
    verifier, and (small sizes) equals the oracle's proof.  Sizes: 2^3 rows (no FRI reduction round), 2^6, 2^9 (LDS-tiled
    NTT passes), the 14-gate circuit (fused quotient kernel with a proof dimension), 2^12 (specialised passes).  K = 67: the
    proof-of-work grind runs in several short launches per look with the lanes re-dealt to the unsettled proofs (an odd number
    of lanes per proof) and must still return THE smallest witness of every proof."""
    import torch

    if kind == "zkdsa":
        import tests.test_oracle_plonk as top

        cd, sk, msg = top.zkdsa_circuit(circ)
        ws, ps = [], []
        for k in range(K):
            inputs = dict(zip(sk, [1 + k, 2, 3, 4 * k]))
            inputs.update(zip(msg, [5, 6 + k, 7, 8]))
            w, p_ = cd.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
            ws.append(w)
            ps.append(p_)
    elif kind in ("chain", "chain9"):
        ws, ps = [], []
        for k in range(K):
            cd, w, p_ = make(pkg, circ, ctx, oracle, 40, seed_val=50 + k, min_bits=9 if kind == "chain9" else 0)
            ws.append(w)
            ps.append(p_)
    elif kind == "mixed":
        mc = circ.MixedGateCircuit(8, 3)
        cd = mc.data
        ws, ps = zip(*[mc.witness_host(lambda ins: pkg.poseidon_gate_rows(ctx, ins), seed=k) for k in range(K)])
    else:  # 2^13 rows: the natural-order last pass of the inverse transform (7 + 6 bits) with a proof dimension
        lc = circ.LayeredPoseidonCircuit(13 if kind == "layered13" else 12, 8)
        cd = lc.data
        ws, ps = [], []
        for k in range(K):
            wd, p_ = lc.witness_device(ctx, pkg, seed=k)
            ws.append(ctx.to_host(wd))
            ps.append(p_)
    pc = pkg.ProverCircuit(ctx, cd)
    batch = torch.stack([ctx.to_device(w) for w in ws]).contiguous()
    proofs = pc.prove_batch(batch, ps)
    assert proofs.shape == (K, pc.proof_len())
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4) if cd.degree_bits <= 9 else None
    for k in range(K):
        single = pc.prove(ctx.to_device(ws[k]), ps[k])
        diff = np.nonzero(proofs[k] != single)[0]
        assert diff.size == 0, "proof %d differs from the single-proof output at word %d of %d" % (k, diff[0], single.size)
        assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, ps[k], proofs[k]) == 0
        if cs is not None:
            assert (proofs[k] == oracle.plonk_prove(cd, cs, ws[k], ps[k])).all()
    assert not (proofs[0] == proofs[1]).all()
    pc.free()


@pytest.mark.parametrize("fused", ["0", "1"])
def test_single_challenge_configuration(pkg, circ, ctx, oracle, fused):
    """num_challenges = 1 (the ABI accepts 1 or 2; the reference's standard_recursion_config uses 2): whole proof == oracle with
    both quotient evaluators (the fused one always combines two alpha rows: the second is zero here)"""
    old = circ.CircuitConfig.num_challenges
    circ.CircuitConfig.num_challenges = 1
    os.environ["GLP_QUOT_FUSED"] = fused
    try:
        cd, wires, pis = make(pkg, circ, ctx, oracle, 20, seed_val=3)
        assert cd.cfg.num_challenges == 1
        pc = pkg.ProverCircuit(ctx, cd)
        proof = pc.prove(ctx.to_device(wires), pis)
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        assert (proof == oracle.plonk_prove(cd, cs, wires, pis)).all()
        assert oracle.plonk_verify(cd, cs["cap"], pis, proof) == 0
        pc.free()
    finally:
        circ.CircuitConfig.num_challenges = old
        del os.environ["GLP_QUOT_FUSED"]


def test_prove_stream_of_host_witnesses(pkg, circ, ctx, oracle):
    """glp_prove_stream: host-resident witnesses (pinned and pageable) with the upload of witness k + 1 overlapping proof k give
    the proofs of the device-resident path"""
    import torch

    ws, ps = [], []
    for k in range(4):
        cd, w, p_ = make(pkg, circ, ctx, oracle, 60, seed_val=200 + k, min_bits=10)
        ws.append(w)
        ps.append(p_)
    pc = pkg.ProverCircuit(ctx, cd)
    want = np.stack([pc.prove(ctx.to_device(w), p_) for w, p_ in zip(ws, ps)])
    pinned = [torch.from_numpy(w.view(np.int64)).pin_memory() for w in ws]
    assert (pc.prove_stream(pinned, ps) == want).all()
    assert (pc.prove_stream(ws, ps) == want).all()          # pageable memory: synchronous copies, same proofs
    assert (pc.prove_stream(ws[:1], ps[:1]) == want[:1]).all()
    # groups: small circuits go through glp_prove_batch_dev a group at a time; force groups of 3 (3 + 1 witnesses) and of 1
    for grp in ("3", "1"):
        os.environ["GLP_STREAM_BATCH"] = grp
        try:
            assert (pc.prove_stream(pinned, ps) == want).all()
        finally:
            del os.environ["GLP_STREAM_BATCH"]
    pc.free()


def test_abi_error_paths(pkg, circ, ctx, oracle):
    """misuse is reported as a status, never a crash: unsupported gate id, mismatched oracle shapes, bad FRI parameters"""
    import ctypes

    cd, wires, pis = make(pkg, circ, ctx, oracle, 3, seed_val=1)
    pc = pkg.ProverCircuit(ctx, cd)
    # (1) a gate id the quotient kernel does not know (e.g. RandomAccessGate of the recursive circuits)
    bad = circ.CircuitBuilder()
    x = bad.virtual()
    bad.register_public_inputs(bad.hash_n_to_hash_no_pad([x]))
    cd2 = bad.build()
    cd2.gates[0]["id"] = 99
    with pytest.raises(pkg.GlproverError) as e:
        pkg.ProverCircuit(ctx, cd2)
    assert e.value.status == -5  # GLP_ERR_UNSUPPORTED
    # (2) wrong witness shape
    with pytest.raises((pkg.GlproverError, AssertionError)):
        pc.prove(ctx.to_device(wires[:, : cd.n // 2].copy()), pis)
    # (3) FRI over oracles whose degree does not match the parameters
    vals = np.zeros((3, 1 << 6), dtype=np.uint64)
    b = pkg.PolynomialBatch.from_values(ctx, vals, 3, False, 4)
    with pytest.raises(pkg.GlproverError) as e:
        pkg.prove_openings(ctx, [((1, 2), [(0, 0)], None)], [b], pkg.Challenger(), pkg.FriParams(7))
    assert e.value.status == -1  # GLP_ERR_INVALID
    with pytest.raises(pkg.GlproverError):
        pkg.prove_openings(ctx, [((1, 2), [(0, 5)], None)], [b], pkg.Challenger(), pkg.FriParams(6))  # polynomial index out of range
    b.free()
    # the context is still healthy afterwards
    proof = pc.prove(ctx.to_device(wires), pis)
    assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, proof) == 0
    pc.free()


def test_witness_at_an_odd_word_address_and_the_stride_rule(pkg, circ, ctx):
    """the boundary takes plain u64 pointers: a witness (and the input of a commitment) that is 8-byte but not 16-byte aligned gives the
    same proof / cap; a col_stride other than 2^degree_bits is refused with GLP_ERR_UNSUPPORTED before any work"""
    import ctypes

    import torch

    u64p = ctypes.POINTER(ctypes.c_uint64)
    for db in (10, 13):
        lc = circ.MixedGateCircuit(db, 16)
        pc = pkg.ProverCircuit(ctx, lc.data)
        wires, pis = lc.witness_device(ctx, pkg, seed=1)
        want = np.array(pc.prove(wires, pis), copy=True)
        n = lc.data.n
        buf = torch.zeros(135 * n + 3, dtype=torch.int64, device="cuda:0")
        for off in (1, 3):
            view = buf[off:off + 135 * n].view(135, n)
            view.copy_(wires)
            torch.cuda.synchronize()
            assert view.data_ptr() % 16 == 8
            assert (pc.prove(view, pis) == want).all(), (db, off)
        proof = np.zeros(pc.proof_len(), dtype=np.uint64)
        pi = np.asarray(pis, dtype=np.uint64)
        st = pkg.lib().glp_prove_dev(ctx._h, pc._h, ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(n + 1), pi.ctypes.data_as(u64p),
                                     ctypes.c_size_t(pi.size), proof.ctypes.data_as(u64p))
        assert st == -5 and not proof.any()  # GLP_ERR_UNSUPPORTED, nothing written
        pc.free()
    x = torch.randint(0, 2**62, (8 * 4096 + 1,), dtype=torch.int64, device="cuda:0")
    v = x[1:].view(8, 4096)
    assert v.data_ptr() % 16 == 8
    b1 = pkg.PolynomialBatch.from_values(ctx, v, 3, False, 4)
    b2 = pkg.PolynomialBatch.from_values(ctx, v.clone(), 3, False, 4)
    assert (b1.cap.elements == b2.cap.elements).all()
    b1.free()
    b2.free()
