"""The recursive verifier as a circuit (intmax_zkp_core_amd.recursion: plonky2's verify_proof restated on this repository's
CircuitBuilder behind the reference's RecursiveProofTarget, src/recursion/gadgets/mod.rs:26-127), checked by the oracle: the
reference's own two recursion tests (a zkdsa signature proof verified inside a circuit, enabled and default), a verifier of a
verifier of a verifier (all 14 gate evaluators, FRI with reduction rounds; the circuit reaches its fixed point at 2^12 rows, plonky2's
recursion threshold), the in-circuit challenges against the oracle's challenger, BlockBatchTarget, and proofs the circuit must refuse.
A witness of the verifier circuit exists only if every check of the verifier holds (copy constraints are enforced while the
witness is generated), and the outer proof is then proven and verified by the oracle."""
import importlib

import numpy as np
import pytest

from tests.conftest import timed_leg

P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def env(oracle):
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    rec = importlib.import_module("intmax_zkp_core_amd.recursion")
    rc = pkg.poseidon_round_constants()
    assert (rc == oracle.round_constants()[:360]).all()
    return circ, rec, rc


def attach(oracle, cd):
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"])
    return cs


def prove(oracle, cd, cs, wires, pis):
    with timed_leg(oracle, cd.degree_bits, threshold=10):
        proof = oracle.plonk_prove(cd, cs, wires, pis)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof) == 0
    return proof


def signature_proof(oracle, circ, sk, msg):
    sig = circ.SimpleSignatureCircuit()
    cs = attach(oracle, sig.data)
    wires, pis = sig.witness(sk, msg, oracle.poseidon_gate_rows)
    return sig, cs, prove(oracle, sig.data, cs, wires, pis), pis


def test_reference_recursion_tests_by_the_oracle(oracle, env):
    """src/recursion/gadgets/mod.rs::test_recursion_simple_signature and ::test_recursion_default_simple_signature"""
    circ, rec, rc = env
    rng = np.random.default_rng(3)
    sk, msg = rng.integers(0, 2**63, 4, dtype=np.uint64), rng.integers(0, 2**63, 4, dtype=np.uint64)
    sig, cs, proof, pis = signature_proof(oracle, circ, sk, msg)
    outer = rec.RecursionCircuit(sig.data, cs["cap"].reshape(-1, 4), sig.data.circuit_digest, rc)
    assert outer.data.degree_bits == 11  # (a 2^3-row inner circuit: no FRI reduction rounds, Merkle paths of two siblings)
    ocs = attach(oracle, outer.data)
    wires, opis = outer.witness(proof, pis, oracle.poseidon_gate_rows, enabled=True)
    assert opis == pis and outer.data.value_of(outer.target.enabled) == 1
    prove(oracle, outer.data, ocs, wires, opis)
    # the default signature (sk = 0, msg = 0), enabled = false
    _, _, dproof, dpis = signature_proof(oracle, circ, [0] * 4, [0] * 4)
    wires, opis = outer.witness(dproof, dpis, oracle.poseidon_gate_rows, enabled=False)
    assert opis == dpis and outer.data.value_of(outer.target.enabled) == 0
    prove(oracle, outer.data, ocs, wires, opis)
    # the in-circuit transcript against the oracle's challenger on the same words
    L = outer.target.layout
    cap = 4 << L.cap_height
    ch = oracle.Challenger()
    ch.observe(sig.data.circuit_digest)
    ch.observe(oracle.hash_no_pad(np.array(dpis, dtype=np.uint64)))
    ch.observe(dproof[:cap])
    want = dict(betas=ch.get_n(2), gammas=ch.get_n(2))
    ch.observe(dproof[cap:2 * cap])
    want["alphas"] = ch.get_n(2)
    ch.observe(dproof[2 * cap:3 * cap])
    want["zeta"] = ch.get_n(2)
    for name, ts in outer.target.challenges.items():
        assert [outer.data.value_of(t) for t in ts] == [int(x) for x in want[name]], name


def test_every_corruption_is_refused(oracle, env):
    """one word of the proof changed, region by region (caps, openings, FRI caps, a query's leaf / sibling / coset evaluations,
    final polynomial, proof-of-work witness), a wrong public input, and the proof of another circuit: no witness exists"""
    circ, rec, rc = env
    keys = np.array([[1, 0, 0, 0], [12, 0, 0, 0], [5, 0, 0, 0]], dtype=np.uint64)
    vals = np.array([[2, 0, 0, 0], [1, 0, 0, 0], [51, 0, 0, 0]], dtype=np.uint64)
    res = oracle.smt_proofs_of_set(keys, vals, keys[2:3])
    n = int(res["n_siblings"][0])
    pr = dict(root=res["root"], key=keys[2], value=res["value"][0], siblings=[tuple(int(x) for x in s) for s in res["siblings"][0][:n]],
              found=True, is_old0=bool(res["is_old0"][0]), not_found_key=res["not_found_key"][0], not_found_value=res["not_found_value"][0])
    c = circ.SmtInclusionCircuit(16, public_root=True)  # 2^7 rows: one FRI reduction round
    cs = attach(oracle, c.data)
    wires, pis = c.witness(pr, oracle.poseidon_gate_rows)
    proof = prove(oracle, c.data, cs, wires, pis)
    outer = rec.RecursionCircuit(c.data, cs["cap"].reshape(-1, 4), c.data.circuit_digest, rc)
    assert outer.target.arity_bits == [4] and outer.data.degree_bits == 12
    outer.witness(proof, pis, oracle.poseidon_gate_rows)  # the honest proof has a witness
    L = outer.target.layout
    view = rec._index_view(L)["proof"]
    q0 = view["opening_proof"]["query_round_proofs"][0]
    q27 = view["opening_proof"]["query_round_proofs"][27]
    spots = dict(wires_cap=view["wires_cap"][3][1], zs_cap=view["plonk_zs_partial_products_cap"][0][0], quotient_cap=view["quotient_polys_cap"][15][3],
                 constant_opening=view["openings"]["constants"][0][0], sigma_opening=view["openings"]["plonk_sigmas"][79][1],
                 wire_opening=view["openings"]["wires"][134][0], z_opening=view["openings"]["plonk_zs"][1][0],
                 z_next_opening=view["openings"]["plonk_zs_next"][0][1], pp_opening=view["openings"]["partial_products"][17][0],
                 quotient_opening=view["openings"]["quotient_polys"][15][1], fri_cap=view["opening_proof"]["commit_phase_merkle_caps"][0][7][2],
                 leaf_constants=q0["initial_trees_proof"][0]["evals"][5], leaf_wires=q27["initial_trees_proof"][1]["evals"][134],
                 leaf_zs=q0["initial_trees_proof"][2]["evals"][19], leaf_quotient=q27["initial_trees_proof"][3]["evals"][0],
                 sibling=q0["initial_trees_proof"][1]["siblings"][2][1], coset_eval=q0["steps"][0]["evals"][9][1],
                 step_sibling=q27["steps"][0]["siblings"][0][0], final_poly=view["opening_proof"]["final_poly"][3][0],
                 pow_witness=view["opening_proof"]["pow_witness"])
    for name, pos in spots.items():
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        with pytest.raises((ValueError, AssertionError)):
            outer.witness(bad, pis, oracle.poseidon_gate_rows)
    with pytest.raises((ValueError, AssertionError)):
        outer.witness(proof, [pis[0] ^ 1] + pis[1:], oracle.poseidon_gate_rows)
    # the same statement proven by a circuit with other constants (another tree depth): other constants_sigmas cap
    c2 = circ.SmtInclusionCircuit(17, public_root=True)
    assert c2.data.degree_bits == c.data.degree_bits
    cs2 = attach(oracle, c2.data)
    w2, p2 = c2.witness(pr, oracle.poseidon_gate_rows)
    proof2 = prove(oracle, c2.data, cs2, w2, p2)
    with pytest.raises((ValueError, AssertionError)):
        outer.witness(proof2, p2, oracle.poseidon_gate_rows)


def test_verifier_of_a_verifier_and_the_block_batch_target(oracle, env):
    """user-transaction proof (the reference's make_user_proof_circuit, 2^11 rows, two FRI rounds) -> verified in a circuit (2^12) ->
    that proof verified in a circuit (all 14 gate types inside the verified circuit: 2^12 again, the fixed point) -> two of those in
    a BlockBatchTarget (one enabled, one slot filled with the last proof, enabled = false)"""
    from tests.test_oracle_smt_circuit import _mp, user_transactions

    circ, rec, rc = env
    c = circ.UserTransactionCircuit()
    cs = attach(oracle, c.data)
    defaults = circ.MergeCircuit.defaults(c.constants["log_n_txs"], _mp(oracle), oracle.two_to_one)
    tx = user_transactions(oracle)[1]
    wires, pis = c.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], defaults, oracle.poseidon_gate_rows)
    proof = prove(oracle, c.data, cs, wires, pis)
    chain = [(c.data, cs, proof, pis)]
    for level in range(2):
        cd, ccs, pf, pp = chain[-1]
        outer = rec.RecursionCircuit(cd, ccs["cap"].reshape(-1, 4), cd.circuit_digest, rc)
        assert outer.data.degree_bits == 12
        ocs = attach(oracle, outer.data)
        w, op = outer.witness(pf, pp, oracle.poseidon_gate_rows)
        assert op == pis  # the statement travels up
        chain.append((outer.data, ocs, prove(oracle, outer.data, ocs, w, op), op))
    assert len({g["id"] for g in chain[1][0].gates}) == 14
    cd, ccs, pf, pp = chain[-1]
    batch = rec.BlockBatchCircuit(cd, ccs["cap"].reshape(-1, 4), cd.circuit_digest, rc, n_blocks=2)
    assert batch.data.degree_bits == 13
    bcs = attach(oracle, batch.data)
    w, bp = batch.witness([(pf, pp)], oracle.poseidon_gate_rows)
    assert bp == pp + pp + [1, 0]
    prove(oracle, batch.data, bcs, w, bp)
    with pytest.raises(AssertionError):
        batch.witness([], oracle.poseidon_gate_rows)


def test_every_gate_with_random_rows_and_blinded_proofs_inside_the_verifier(oracle, env):
    """the synthetic 14-gate circuit (every gate type with random valid rows: other constants, selector groups and wire patterns than
    the verifier circuit's own rows) verified in-circuit, as a plain proof and as a zero-knowledge proof (config.zero_knowledge:
    salted leaves in the three blinded oracles; the verifier circuit hashes them in the Merkle proofs and keeps them out of the
    batch sums); a zero-knowledge proof read with the plain layout, and the reverse, are refused"""
    circ, rec, rc = env
    m = circ.MixedGateCircuit(7, 2)
    cd = m.data
    cs = attach(oracle, cd)
    wires, pis = m.witness_host(oracle.poseidon_gate_rows, seed=5)
    assert len({g["id"] for g in cd.gates}) == 14
    plain = prove(oracle, cd, cs, wires, pis)
    outer = rec.RecursionCircuit(cd, cs["cap"].reshape(-1, 4), cd.circuit_digest, rc, n_public_inputs=len(pis))
    w, op = outer.witness(plain, pis, oracle.poseidon_gate_rows)
    assert op == pis
    prove(oracle, outer.data, attach(oracle, outer.data), w, op)
    fp = oracle.zk_params(cd)
    salts = np.random.default_rng(8).integers(0, P, (3, 4, cd.n << cd.cfg.rate_bits), dtype=np.uint64)
    zk = oracle.plonk_prove(cd, cs, wires, pis, fp, salts=salts)
    assert oracle.plonk_verify(cd, cs["cap"], pis, zk, fp) == 0 and zk.size == plain.size + 28 * 3 * 4
    outer_zk = rec.RecursionCircuit(cd, cs["cap"].reshape(-1, 4), cd.circuit_digest, rc, zero_knowledge=True, n_public_inputs=len(pis))
    w, op = outer_zk.witness(zk, pis, oracle.poseidon_gate_rows)
    prove(oracle, outer_zk.data, attach(oracle, outer_zk.data), w, op)
    bad = zk.copy()
    salt_word = rec._index_view(outer_zk.target.layout)["proof"]["opening_proof"]["query_round_proofs"][3]["initial_trees_proof"][1]["evals"][135]
    bad[salt_word] = (int(bad[salt_word]) + 1) % P  # a salt word: not in any sum, but in the leaf's hash
    with pytest.raises((ValueError, AssertionError)):
        outer_zk.witness(bad, pis, oracle.poseidon_gate_rows)
    with pytest.raises((ValueError, AssertionError)):
        outer.witness(zk, pis, oracle.poseidon_gate_rows)
    with pytest.raises((ValueError, AssertionError)):
        outer_zk.witness(plain, pis, oracle.poseidon_gate_rows)
