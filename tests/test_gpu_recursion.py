"""Recursive proofs ON THE GPU: the verifier circuit of intmax_zkp_core_amd.recursion (plonky2's verify_proof behind the reference's
RecursiveProofTarget, src/recursion/gadgets/mod.rs:26-127) proven by glp_prove — the first circuits of this suite whose PoseidonGate
rows use the swap wire and whose traces hold all six gates verify_proof adds at their real density.  Inner proofs come from the GPU
as well (their PoseidonGate rows from glp_poseidon_gate_witness_dev); every outer proof equals the oracle's proof of the same
witness word for word and the restated verifier accepts it."""
import importlib

import numpy as np
import pytest

from tests.conftest import timed_leg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    rec = importlib.import_module("intmax_zkp_core_amd.recursion")
    trees = importlib.import_module("intmax_zkp_core_amd.trees")
    ctx = pkg.Context(0)
    yield pkg, circ, rec, trees, ctx
    ctx.close()


def gpu_prove(pkg, ctx, oracle, cd, wires, pis, check_oracle=True):
    pc = pkg.ProverCircuit(ctx, cd)
    proof = pc.prove(wires, pis)
    cap = np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)
    cs = None
    if check_oracle:
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        assert (cap.reshape(-1) == cs["cap"].reshape(-1)).all()
        with timed_leg(oracle, cd.degree_bits, threshold=10):
            assert (proof == oracle.plonk_prove(cd, cs, wires, pis)).all()
        assert oracle.plonk_verify(cd, cs["cap"], pis, proof) == 0
    return pc, cap, proof


def test_reference_recursion_test_on_the_gpu(env, oracle):
    """src/recursion/gadgets/mod.rs::test_recursion_simple_signature: the zkdsa proof and the proof that verifies it, both by glp_prove"""
    pkg, circ, rec, trees, ctx = env
    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    rng = np.random.default_rng(11)
    sig = circ.SimpleSignatureCircuit()
    wires, pis = sig.witness(rng.integers(0, 2**63, 4, dtype=np.uint64), rng.integers(0, 2**63, 4, dtype=np.uint64), rows_fn)
    pc, cap, proof = gpu_prove(pkg, ctx, oracle, sig.data, wires, pis)
    outer = rec.RecursionCircuit(sig.data, cap, pc.circuit_digest, pkg.poseidon_round_constants())
    w, p = outer.witness(proof, pis, rows_fn)
    swaps = w[24, [r for r, (g, _) in enumerate(outer.data.builder.rows) if g == circ.GATE_POSEIDON]]
    assert 0 < int(swaps.sum()) < swaps.size  # PoseidonGate rows of both orders
    assert p == pis
    pc2, _, _ = gpu_prove(pkg, ctx, oracle, outer.data, w, p)
    # a corrupted inner proof has no witness
    bad = proof.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises((ValueError, AssertionError)):
        outer.witness(bad, pis, rows_fn)
    pc.free()
    pc2.free()


def test_user_transaction_proofs_aggregated_on_the_gpu(env, oracle):
    """the reference's user-transaction circuit (2^11 rows) proven for the two transactions of src/bin/block_circuit.rs, each proof
    verified inside a circuit (2^12 rows), the two verifier proofs verified together by a BlockBatchTarget circuit (2^13 rows, all
    14 gate types in the circuit AND in the circuits it verifies): seven GPU proofs, every one equal to the oracle's"""
    from tests.test_oracle_smt_circuit import _mp, user_transactions

    pkg, circ, rec, trees, ctx = env
    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    rc = pkg.poseidon_round_constants()
    c = circ.UserTransactionCircuit()
    defaults = circ.MergeCircuit.defaults(c.constants["log_n_txs"], _mp(oracle), oracle.two_to_one)
    pc = pkg.ProverCircuit(ctx, c.data)
    cap = np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)
    wrap = rec.RecursionCircuit(c.data, cap, pc.circuit_digest, rc)
    assert wrap.data.degree_bits == 12 and len({g["id"] for g in wrap.data.gates}) == 14
    pcw = pkg.ProverCircuit(ctx, wrap.data)
    capw = np.array(pcw.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)
    csw = oracle.batch_from_values(wrap.data.constants_sigmas_values(), 3, 4)
    assert (capw.reshape(-1) == csw["cap"].reshape(-1)).all()
    level1 = []
    for tx in user_transactions(oracle):
        wires, pis = c.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], defaults, rows_fn)
        proof = pc.prove(wires, pis)
        w, p = wrap.witness(proof, pis, rows_fn)
        assert p == pis
        outer = pcw.prove(w, p)
        with timed_leg(oracle, 12, threshold=10):
            assert (outer == oracle.plonk_prove(wrap.data, csw, w, p)).all()
        assert oracle.plonk_verify(wrap.data, csw["cap"], p, outer) == 0
        level1.append((outer, p))
    batch = rec.BlockBatchCircuit(wrap.data, capw, pcw.circuit_digest, rc, n_blocks=2)
    assert batch.data.degree_bits == 13
    w, p = batch.witness(level1, rows_fn)
    assert p == level1[0][1] + level1[1][1] + [1, 1]
    pcb, _, _ = gpu_prove(pkg, ctx, oracle, batch.data, w, p)
    w1, p1 = batch.witness(level1[:1], rows_fn)  # one block, the second slot disabled
    assert p1 == level1[0][1] + level1[0][1] + [1, 0]
    got = pcb.prove(w1, p1)
    csb = oracle.batch_from_values(batch.data.constants_sigmas_values(), 3, 4)
    assert oracle.plonk_verify(batch.data, csb["cap"], p1, got) == 0
    for x in (pc, pcw, pcb):
        x.free()


def test_a_blinded_gpu_proof_verified_inside_a_circuit(env, oracle):
    """a zero-knowledge proof (GLP_FLAG_ZERO_KNOWLEDGE, caller's salts) of the 14-gate circuit made by glp_prove_salted, verified by the
    recursive verifier circuit with the salted-leaf layout, and that circuit proven on the GPU"""
    pkg, circ, rec, trees, ctx = env
    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    m = circ.MixedGateCircuit(7, 2)
    wires, pis = m.witness_host(rows_fn, seed=5)
    pc = pkg.ProverCircuit(ctx, m.data, pkg.FriParams(m.data.degree_bits, flags=pkg.FLAG_ZERO_KNOWLEDGE))
    salts = np.random.default_rng(8).integers(0, 0xFFFFFFFF00000001, (3, 4, m.data.n << 3), dtype=np.uint64)
    zk = pc.prove(wires, pis, salts=salts)
    cs = oracle.batch_from_values(m.data.constants_sigmas_values(), 3, 4)
    fp = oracle.zk_params(m.data)
    assert (zk == oracle.plonk_prove(m.data, cs, wires, pis, fp, salts=salts)).all()
    cap = np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)
    outer = rec.RecursionCircuit(m.data, cap, pc.circuit_digest, pkg.poseidon_round_constants(), zero_knowledge=True, n_public_inputs=len(pis))
    w, p = outer.witness(zk, pis, rows_fn)
    pc2, _, _ = gpu_prove(pkg, ctx, oracle, outer.data, w, p)
    pc.free()
    pc2.free()
