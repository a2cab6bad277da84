"""The reference's BLOCK PRODUCTION circuit (intmax_zkp_core_amd.rollup.BlockProductionCircuit = make_block_proof_circuit,
src/rollup/circuits/mod.rs:389-624) on the block src/bin/block_circuit.rs produces: two user transactions (sender 1 purges two assets
and returns no signature; sender 2 merges a deposit of block 1, sends it on and signs the proposed world state), one deposit, the
world-state tree updated and reverted, the latest-account tree, the block-headers tree, the block hash — every user-transaction
and signature proof verified INSIDE the block circuit by the restated recursive verifier, then the block proof itself inside the
reference's BlockBatchTarget circuit.  All proofs by the oracle (timed leg), all verified by the restated verifier."""
import importlib

import numpy as np
import pytest

from tests.conftest import timed_leg
from tests.test_oracle_recursion import attach, prove
from tests.test_oracle_smt_circuit import _mp, user_transactions

Z = (0, 0, 0, 0)


def T(v):
    return tuple(int(x) for x in v)


@pytest.fixture(scope="module")
def env(oracle):
    import __graft_entry__ as g

    pkg = g.load_package()
    mods = [importlib.import_module("intmax_zkp_core_amd." + m) for m in ("circuit", "recursion", "rollup")]
    return (*mods, pkg.poseidon_round_constants())


def block2_detail(oracle, circ, user_proofs, default_user_proof, sign):
    """src/bin/block_circuit.rs:367-640 on the restated host trees.  user_proofs: [(flat, pis)] of the two transactions;
    sign(sk, message) -> (flat, pis) of a signature proof"""
    from oracle import smt_incremental as inc

    tx1, tx2 = user_transactions(oracle)
    a1, a2 = (tx1["sender_address"], 0, 0, 0), (tx2["sender_address"], 0, 0, 0)
    world = inc.SparseMerkleTree()
    world.set(a1, tx1["old_root"])
    prev_header = dict(tx2["merges"][0]["header"])
    assert T(prev_header["approved_world_state_digest"]) == world.root
    ws_process = [inc.set_with_proof(world, a1, tx1["new_root"]), inc.set_with_proof(world, a2, tx2["new_root"])]
    proposed = world.root
    sk2 = [15657143458229430356, 6012455030006979790, 4280058849535143691, 5153662694263190591]
    received = [None, sign(sk2, proposed)]
    block_number = prev_header["block_number"] + 1
    latest = inc.SparseMerkleTree()
    reverts, latest_proofs = [], []
    for r, tx, addr in zip(received, (tx1, tx2), (a1, a2)):
        if r is None:
            last, confirmed = latest.get(addr), tx["middle_root"]
        else:
            last, confirmed = (block_number, 0, 0, 0), tx["new_root"]
        latest_proofs.append(inc.set_with_proof(latest, addr, last))
        reverts.append(inc.set_with_proof(world, addr, confirmed))
    names = ("prev_block_hash", "block_headers_digest", "transactions_digest", "deposit_digest", "proposed_world_state_digest",
             "approved_world_state_digest", "latest_account_digest")
    prev_hash = T(oracle.block_hash(np.array([prev_header["block_number"]] + [x for nm in names for x in prev_header[nm]], dtype=np.uint64)))
    sib, headers_root = oracle.merkle_proof_with_zero(np.array([Z, prev_hash], dtype=np.uint64), prev_header["block_number"], 32)
    dep = inc.LayeredLayeredSparseMerkleTree()
    deposits = [inc.layered_layered_set_with_proofs(dep, a2, (1, 0, 0, 0), Z, (1, 0, 0, 0))]
    detail = dict(block_number=block_number, user_tx_proofs=user_proofs, default_user_tx_proof=default_user_proof,
                  deposit_process_proofs=deposits, scroll_process_proofs=[], polygon_process_proofs=[],
                  world_state_process_proofs=ws_process, world_state_revert_proofs=reverts, received_signature_proofs=received,
                  default_simple_signature_proof=sign([0] * 4, Z), latest_account_process_proofs=latest_proofs,
                  block_headers_proof_siblings=[T(s) for s in sib], prev_block_header=prev_header)
    tx_root = oracle.merkle_proof_with_zero(np.array([user_proofs[0][1][20:], user_proofs[1][1][20:]], dtype=np.uint64), 0, 2,
                                            default_user_proof[1][20:])[1]
    dep_root = oracle.merkle_proof_with_zero(np.array([dep.root], dtype=np.uint64), 0, 2)[1]
    header = dict(block_number=block_number, prev_block_hash=prev_hash, block_headers_digest=T(headers_root), transactions_digest=T(tx_root),
                  deposit_digest=T(dep_root), proposed_world_state_digest=proposed, approved_world_state_digest=world.root,
                  latest_account_digest=latest.root)
    block_hash = T(oracle.block_hash(np.array([block_number] + [x for nm in names for x in header[nm]], dtype=np.uint64)))
    encoded = ([a1[0], 0, 0, 0, 0, a2[0], 0, 0, 0, 1] + [0] * 10  # address list: (sender, signature came back), two empty slots
               + list(a2) + [1, 0, 0, 0] + list(Z) + [1] + [0] * 13 + [0] * 26 + [0] * 26  # deposit list, scroll flags, polygon flags
               + list(prev_header["latest_account_digest"]) + list(latest.root) + list(prev_header["approved_world_state_digest"])
               + list(world.root) + list(prev_header["block_headers_digest"]) + list(T(headers_root)) + list(block_hash))
    return detail, encoded


def test_block_production_circuit_on_the_reference_binarys_block(oracle, env):
    circ, rec, rollup, rc = env
    user = circ.UserTransactionCircuit()
    ucs = attach(oracle, user.data)
    sig = circ.SimpleSignatureCircuit()
    scs = attach(oracle, sig.data)
    defaults = circ.MergeCircuit.defaults(user.constants["log_n_txs"], _mp(oracle), oracle.two_to_one)

    def user_proof(tx):
        w, pis = user.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], defaults, oracle.poseidon_gate_rows)
        return prove(oracle, user.data, ucs, w, pis), pis

    def sign(sk, message):
        w, pis = sig.witness(sk, message, oracle.poseidon_gate_rows)
        return prove(oracle, sig.data, scs, w, pis), pis

    default_tx = dict(sender_address=0, merges=[], ins=[], outs=[], nonce=Z, old_root=Z)
    user_proofs = [user_proof(tx) for tx in user_transactions(oracle)]
    detail, encoded = block2_detail(oracle, circ, user_proofs, user_proof(default_tx), sign)
    block = rollup.BlockProductionCircuit((user.data, ucs["cap"].reshape(-1, 4)), (sig.data, scs["cap"].reshape(-1, 4)), rc)
    assert block.data.degree_bits == 15 and len({g["id"] for g in block.data.gates}) == 14
    wires, pis = block.witness(detail, oracle.poseidon_gate_rows)
    # BlockProductionPublicInputs::encode of what the reference's set_witness returns, and its entry hash
    assert [block.data.value_of(t) for t in block.encoded_public_inputs] == encoded
    assert len(encoded) == 5 * 4 + 13 * 6 + 28 and pis == [int(x) for x in oracle.hash_no_pad(np.array(encoded, dtype=np.uint64))]
    codec = importlib.import_module("intmax_zkp_core_amd.codec")  # ... and they decode as BlockProductionPublicInputs::decode reads them
    dec = codec.block_production_public_inputs_decode(encoded, 4, 2, 2, 2)
    assert [a["is_valid"] for a in dec["address_list"]] == [False, True, False, False] and dec["deposit_list"][0]["amount"] == 1
    assert dec["deposit_list"][0]["receiver_address"] == dec["address_list"][1]["sender_address"] and dec["deposit_list"][1]["amount"] == 0
    assert codec.block_production_public_inputs_encode(dec) == encoded
    bcs = attach(oracle, block.data)
    block_proof = prove(oracle, block.data, bcs, wires, pis)
    # BlockBatchTarget over the block proof (n_blocks = 2: the second slot repeats it, enabled = false)
    batch = rec.BlockBatchCircuit(block.data, bcs["cap"].reshape(-1, 4), block.data.circuit_digest, rc, n_blocks=2)
    assert batch.data.degree_bits == 14
    w, bp = batch.witness([(block_proof, pis)], oracle.poseidon_gate_rows)
    assert bp == pis + pis + [1, 0]
    prove(oracle, batch.data, attach(oracle, batch.data), w, bp)
    # what the block circuit must refuse: a signature over another message; a transaction proof in the wrong slot; a world-state
    # update that is not the transaction's new asset root
    bad = dict(detail, received_signature_proofs=[None, sign([15657143458229430356, 6012455030006979790, 4280058849535143691, 5153662694263190591], Z)])
    with pytest.raises((ValueError, AssertionError)):
        block.witness(bad, oracle.poseidon_gate_rows)
    bad = dict(detail, user_tx_proofs=user_proofs[::-1])
    with pytest.raises((ValueError, AssertionError)):
        block.witness(bad, oracle.poseidon_gate_rows)
    # the default block (src/rollup/circuits/mod.rs::test_prove_block_production: BlockDetail::new, no transactions at all)
    z32 = oracle.merkle_proof_with_zero(np.zeros((1, 4), dtype=np.uint64), 0, 32)
    h0 = dict(circ.MergeCircuit.defaults(2, _mp(oracle), oracle.two_to_one)["header"], prev_block_hash=Z)
    names = ("prev_block_hash", "block_headers_digest", "transactions_digest", "deposit_digest", "proposed_world_state_digest",
             "approved_world_state_digest", "latest_account_digest")
    assert T(z32[1]) == T(h0["block_headers_digest"])
    empty = dict(block_number=1, user_tx_proofs=[], default_user_tx_proof=detail["default_user_tx_proof"], deposit_process_proofs=[],
                 scroll_process_proofs=[], polygon_process_proofs=[], world_state_process_proofs=[], world_state_revert_proofs=[],
                 received_signature_proofs=[], default_simple_signature_proof=detail["default_simple_signature_proof"],
                 latest_account_process_proofs=[], block_headers_proof_siblings=[T(s) for s in z32[0]], prev_block_header=h0)
    wires, pis0 = block.witness(empty, oracle.poseidon_gate_rows)
    prove(oracle, block.data, bcs, wires, pis0)


def test_the_three_block_gadgets_as_circuits_of_their_own(oracle, env):
    """the reference's own tests of the gadgets the block circuit is assembled from, each a circuit of its own on the block of
    src/bin/block_circuit.rs: test_deposit_block (rollup/gadgets/deposit_block/mod.rs:353-471: public input = the interior deposit
    digest; one deposit, then none), test_proposal_block (proposal_block/mod.rs:241-615: public inputs = transactions digest and new
    world-state root) and test_approval_block (approval_block/mod.rs:355-843: no public inputs; the world state reverted for the
    sender without a signature, the latest-account tree updated for the other)"""
    circ, rec, rollup, rc = env
    user = circ.UserTransactionCircuit()
    defaults = circ.MergeCircuit.defaults(2, _mp(oracle), oracle.two_to_one)
    pis_of = lambda tx: user.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], defaults,  # noqa: E731
                                     oracle.poseidon_gate_rows)[1]
    default_tx = dict(sender_address=0, merges=[], ins=[], outs=[], nonce=Z, old_root=Z)
    user_pis = [pis_of(tx) for tx in user_transactions(oracle)]
    fake = lambda pis: ([0], pis)  # noqa: E731  (block2_detail only stores proofs; the gadgets read the public inputs)
    sk2 = [15657143458229430356, 6012455030006979790, 4280058849535143691, 5153662694263190591]
    sig_pis = lambda sk, msg: list(msg) + [int(x) for x in oracle.two_to_one(sk, sk)] + [int(x) for x in oracle.two_to_one(sk, msg)]  # noqa: E731
    detail, _ = block2_detail(oracle, circ, [fake(p) for p in user_pis], fake(pis_of(default_tx)), lambda sk, msg: fake(sig_pis(sk, msg)))
    default_user = detail["default_user_tx_proof"][1]

    # -- deposit block --
    b = circ.CircuitBuilder()
    dep = rollup.DepositBlockGadget(b, 3, 3, 3, 2)
    b.register_public_inputs(dep.interior_deposit_digest)
    cd = b.build()
    for proofs in (detail["deposit_process_proofs"], []):
        inputs = {}
        digest = dep.fill(inputs, proofs)
        wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
        assert pis == list(digest) and (proofs or pis == [0, 0, 0, 0])
        prove(oracle, cd, attach(oracle, cd), wires, pis)
    assert dep.fill({}, detail["deposit_process_proofs"]) == T(detail["deposit_process_proofs"][-1][0]["new_root"])

    # -- proposal block --
    b = circ.CircuitBuilder()
    prop = rollup.ProposalBlockGadget(b, 3, 4)
    b.register_public_inputs(prop.transactions_digest + prop.new_world_state_root)
    cd = b.build()
    old_ws = detail["prev_block_header"]["approved_world_state_digest"]
    inputs = {}
    proposed = prop.fill(inputs, detail["world_state_process_proofs"], user_pis, old_ws, default_user)
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    tx_root = oracle.merkle_proof_with_zero(np.array([p[20:] for p in user_pis], dtype=np.uint64), 0, 2, default_user[20:])[1]
    assert pis == [int(x) for x in tx_root] + list(proposed) and proposed == T(detail["world_state_process_proofs"][-1]["new_root"])
    prove(oracle, cd, attach(oracle, cd), wires, pis)
    inputs = {}
    assert prop.fill(inputs, [], [], old_ws, default_user) == T(old_ws)  # no transactions: the world state stays
    cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    with pytest.raises((ValueError, AssertionError)):  # a transaction whose new asset root is not what the world-state update stores
        bad = list(user_pis[0])
        bad[8] ^= 1
        inputs = {}
        prop.fill(inputs, detail["world_state_process_proofs"], [bad, user_pis[1]], old_ws, default_user)
        cd.generate_witness(inputs, oracle.poseidon_gate_rows)

    # -- approval block --
    b = circ.CircuitBuilder()
    appr = rollup.ApprovalBlockGadget(b, 3, 4)
    cd = b.build()
    received = [None, sig_pis(sk2, proposed)]
    default_sig = sig_pis([0] * 4, Z)
    inputs = {}
    ws, la = appr.fill(inputs, detail["block_number"], detail["world_state_revert_proofs"], user_pis, received, detail["latest_account_process_proofs"],
                       proposed, detail["prev_block_header"]["latest_account_digest"], default_user, default_sig)
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    assert pis == [] and ws == T(detail["world_state_revert_proofs"][-1]["new_root"]) and la == T(detail["latest_account_process_proofs"][-1]["new_root"])
    prove(oracle, cd, attach(oracle, cd), wires, pis)
    with pytest.raises((ValueError, AssertionError)):  # the signature of sender 2 over another message than the proposed world state
        inputs = {}
        appr.fill(inputs, detail["block_number"], detail["world_state_revert_proofs"], user_pis, [None, sig_pis(sk2, Z)],
                  detail["latest_account_process_proofs"], proposed, detail["prev_block_header"]["latest_account_digest"], default_user, default_sig)
        cd.generate_witness(inputs, oracle.poseidon_gate_rows)
