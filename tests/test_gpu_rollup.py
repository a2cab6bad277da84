"""The reference's whole proving pipeline for one block ON THE GPU, as src/bin/block_circuit.rs runs it: two user-transaction proofs
and the default one (make_user_proof_circuit, 2^11 rows), a signature proof and the default one (zkdsa, 2^3 rows), the BLOCK proof
(make_block_proof_circuit: 2^15 rows, eight recursive verifications inside) and the batch proof over it (BlockBatchTarget, 2^14
rows) — seven glp_prove calls; PoseidonGate witness rows from glp_poseidon_gate_witness_dev.  The block proof and the batch proof
equal the oracle's proofs of the same witnesses word for word; the restated verifier accepts all of them; the block's public
input is the entry hash of the reference's BlockProductionPublicInputs."""
import importlib

import numpy as np
import pytest

from tests.conftest import timed_leg
from tests.test_oracle_rollup import Z, block2_detail
from tests.test_oracle_smt_circuit import _mp, user_transactions

pytestmark = pytest.mark.gpu


def test_one_block_of_the_reference_proven_end_to_end_on_the_gpu(oracle):
    import __graft_entry__ as g

    pkg = g.load_package()
    circ, rec, rollup = (importlib.import_module("intmax_zkp_core_amd." + m) for m in ("circuit", "recursion", "rollup"))
    ctx = pkg.Context(0)
    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    rc = pkg.poseidon_round_constants()
    cap_of = lambda pc: np.array(pc.constants_sigmas_cap.elements, dtype=np.uint64).reshape(-1, 4)  # noqa: E731
    user, sig = circ.UserTransactionCircuit(), circ.SimpleSignatureCircuit()
    upc, spc = pkg.ProverCircuit(ctx, user.data), pkg.ProverCircuit(ctx, sig.data)
    defaults = circ.MergeCircuit.defaults(user.constants["log_n_txs"], _mp(oracle), oracle.two_to_one)

    def user_proof(tx):
        w, pis = user.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], defaults, rows_fn)
        return upc.prove(w, pis), pis

    def sign(sk, message):
        w, pis = sig.witness(sk, message, rows_fn)
        return spc.prove(w, pis), pis

    user_proofs = [user_proof(tx) for tx in user_transactions(oracle)]
    default_user = user_proof(dict(sender_address=0, merges=[], ins=[], outs=[], nonce=Z, old_root=Z))
    detail, encoded = block2_detail(oracle, circ, user_proofs, default_user, sign)
    block = rollup.BlockProductionCircuit((user.data, cap_of(upc)), (sig.data, cap_of(spc)), rc)
    assert block.data.degree_bits == 15
    wires, pis = block.witness(detail, rows_fn)
    assert [block.data.value_of(t) for t in block.encoded_public_inputs] == encoded
    assert pis == [int(x) for x in pkg.hash_no_pad(np.array(encoded, dtype=np.uint64))]
    bpc = pkg.ProverCircuit(ctx, block.data)
    block_proof = bpc.prove(wires, pis)
    bcs = oracle.batch_from_values(block.data.constants_sigmas_values(), 3, 4)
    assert (cap_of(bpc).reshape(-1) == bcs["cap"].reshape(-1)).all()
    with timed_leg(oracle, 15):
        assert (block_proof == oracle.plonk_prove(block.data, bcs, wires, pis)).all()
    assert oracle.plonk_verify(block.data, bcs["cap"], pis, block_proof) == 0
    batch = rec.BlockBatchCircuit(block.data, cap_of(bpc), bpc.circuit_digest, rc, n_blocks=2)
    assert batch.data.degree_bits == 14
    w, bp = batch.witness([(block_proof, pis)], rows_fn)
    assert bp == pis + pis + [1, 0]
    tpc = pkg.ProverCircuit(ctx, batch.data)
    batch_proof = tpc.prove(w, bp)
    tcs = oracle.batch_from_values(batch.data.constants_sigmas_values(), 3, 4)
    with timed_leg(oracle, 14):
        assert (batch_proof == oracle.plonk_prove(batch.data, tcs, w, bp)).all()
    assert oracle.plonk_verify(batch.data, tcs["cap"], bp, batch_proof) == 0
    for pc in (upc, spc, bpc, tpc):
        pc.free()
    ctx.close()


def test_block_state_from_library_calls_equals_the_sequential_restatement(oracle):
    """state.reference_block (every tree operation answered by glp_smt_apply_updates / glp_smt_proofs_of_set / glp_two_to_one_batch_dev
    / glp_merkle_proof_with_zero_dev / glp_block_hash_batch_dev) against the oracle's restatement of the reference's sequential trees:
    the same transactions piece by piece, the same BlockDetail, and the pipeline's entry hash is the hash of the encoded public inputs"""
    import __graft_entry__ as g

    pkg = g.load_package()
    circ, rec, rollup, state, wl = (importlib.import_module("intmax_zkp_core_amd." + m) for m in ("circuit", "recursion", "rollup", "state", "workloads"))
    ctx = pkg.Context(0)
    blk = state.reference_block(pkg, ctx)
    norm = lambda p: {k: ([tuple(int(x) for x in s) for s in v] if k == "siblings" else (tuple(int(x) for x in v) if hasattr(v, "__len__") else v))  # noqa: E731
                      for k, v in p.items()}
    for got, want in zip(blk["transactions"], user_transactions(oracle)):
        assert got["sender_address"] == want["sender_address"] and list(got["nonce"]) == list(want["nonce"])
        for name in ("old_root", "middle_root", "new_root"):
            assert tuple(got[name]) == tuple(want[name]), name
        for g3, w3 in zip(got["ins"] + got["outs"], want["ins"] + want["outs"]):
            for gp, wp in zip(g3, w3):
                assert norm(gp) == norm(wp)
        for gm, wm in zip(got["merges"], want["merges"]):
            assert gm["merkle"] == wm["merkle"] and norm(gm["process"]) == norm(wm["process"]) and gm["header"] == wm["header"]
            assert norm(gm["diff_inclusion"]) == norm(wm["diff_inclusion"])
    want_defaults = circ.MergeCircuit.defaults(2, _mp(oracle), oracle.two_to_one)
    assert blk["defaults"] == want_defaults
    fake = lambda *_: ([0], [0] * 12)  # noqa: E731  (detail() only stores the proofs)
    proofs = [([0], [0] * 24)] * 2
    got = blk["detail"](proofs, ([0], [0] * 24), fake)
    want, _ = block2_detail(oracle, circ, proofs, ([0], [0] * 24), fake)
    for key in ("deposit_process_proofs",):
        for g3, w3 in zip(got[key], want[key]):
            for gp, wp in zip(g3, w3):
                assert norm(gp) == norm(wp)
    for key in ("world_state_process_proofs", "world_state_revert_proofs", "latest_account_process_proofs"):
        assert [norm(p) for p in got[key]] == [norm(p) for p in want[key]], key
    assert got["block_headers_proof_siblings"] == want["block_headers_proof_siblings"] and got["prev_block_header"] == want["prev_block_header"]
    out = wl.reference_block_pipeline(pkg, circ, rec, rollup, ctx, reps=1)
    assert [out[k]["degree_bits"] for k in ("user_transaction", "signature", "block", "batch")] == [11, 3, 15, 14]
    ctx.close()


def test_the_block_circuit_with_16_slots_equals_the_oracle_and_its_scalar_checker(oracle):
    """The headline circuit family at a size the default suite can afford: make_block_proof_circuit
    (src/rollup/circuits/mod.rs:389-624) with log_n_txs = 4 — 16 transaction slots, each verifying a user-transaction proof and a
    signature proof: a 2^17-row circuit with all 14 gate types — filled with the block of src/bin/block_circuit.rs.  Its GPU proof
    equals the oracle's word for word TWICE: the timed leg (packed Goldilocks) and the SCALAR CHECKER, which closes GPU == scalar above
    2^15 rows on the GPU box in every run of the suite; the witness also goes through glp_prove_cols (separate pageable columns)."""
    import __graft_entry__ as g

    pkg = g.load_package()
    circ, rec, rollup, wl = (importlib.import_module("intmax_zkp_core_amd." + m) for m in ("circuit", "recursion", "rollup", "workloads"))
    ctx = pkg.Context(0)
    rb = wl.ReferenceBlockCircuit(pkg, circ, rec, rollup, ctx, 4)
    assert rb.n_slots == 16 and rb.data.degree_bits == 17 and len({g_["id"] for g_ in rb.data.gates}) == 14
    wires, pis = rb.witness_host()
    pc = pkg.ProverCircuit(ctx, rb.data)
    proof = pc.prove(ctx.to_device(wires), pis)
    assert (pc.prove_cols([np.array(c_, copy=True) for c_ in wires], pis) == proof).all()
    cs = oracle.batch_from_values(rb.data.constants_sigmas_values(), 3, 4)
    assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
    with timed_leg(oracle, 17):
        fast = oracle.plonk_prove(rb.data, cs, wires, pis)
    bad = np.nonzero(proof != fast)[0]
    assert bad.size == 0, "timed leg: first differing proof word %d of %d" % (int(bad[0]), proof.size)
    with timed_leg(oracle, 17, threshold=99):  # the scalar checker itself
        scalar = oracle.plonk_prove(rb.data, cs, wires, pis)
    bad = np.nonzero(proof != scalar)[0]
    assert bad.size == 0, "scalar checker: first differing proof word %d of %d" % (int(bad[0]), proof.size)
    assert oracle.plonk_verify(rb.data, cs["cap"], pis, proof) == 0
    pc.free()
    ctx.close()


@pytest.mark.skipif(__import__("os").environ.get("GLP_BIG_BLOCK") != "1",
                    reason="minutes of host preparation (a 2^20-row circuit built and filled in Python): GLP_BIG_BLOCK=1 runs it; "
                           "profiles/r05_block128.md holds the recorded run")
def test_the_block_circuit_with_128_slots_is_a_2_20_row_circuit_and_its_gpu_proof_equals_the_oracle(oracle):
    """make_block_proof_circuit with log_n_txs = 7: BASELINE's headline size as the reference's own circuit.  The GPU proof of the block
    of src/bin/block_circuit.rs in it equals the oracle's (timed leg) word for word and verifies."""
    import __graft_entry__ as g

    pkg = g.load_package()
    circ, rec, rollup, wl = (importlib.import_module("intmax_zkp_core_amd." + m) for m in ("circuit", "recursion", "rollup", "workloads"))
    ctx = pkg.Context(0)
    rb = wl.ReferenceBlockCircuit(pkg, circ, rec, rollup, ctx, 7)
    assert rb.data.degree_bits == 20 and len({g_["id"] for g_ in rb.data.gates}) == 14
    wires, pis = rb.witness_host()
    pc = pkg.ProverCircuit(ctx, rb.data)
    proof = pc.prove(ctx.to_device(wires), pis)
    cs = oracle.batch_from_values(rb.data.constants_sigmas_values(), 3, 4)
    with timed_leg(oracle, 20):
        assert (proof == oracle.plonk_prove(rb.data, cs, wires, pis)).all()
    assert oracle.plonk_verify(rb.data, cs["cap"], pis, proof) == 0
    pc.free()
    ctx.close()
