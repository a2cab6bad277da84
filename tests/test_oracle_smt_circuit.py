"""The reference's sparse-Merkle-tree (non-)inclusion gadget as a circuit (circuit.SmtInclusionCircuit: verify_smt_inclusion_proof,
src/sparse_merkle_tree/gadgets/verify/verify_smt.rs:209-307, restated statement for statement on this repository's CircuitBuilder),
fed with the `find` proofs of the restated tree (oracle.smt_proofs_of_set = SparseMerkleTree::find, tree.rs:588-690) and proven /
verified by the oracle: the reference's own test (gadgets/verify/mod.rs:3-52: keys 1, 12, 5, N_LEVELS = 16, find(key 5)), inclusion
and both kinds of non-inclusion proofs on larger sets, the `enabled = false` escape, and witnesses the circuit must refuse.  This
ties the tree row (N2: its leaf / node hashing is pinned by the reference's goldens) to the PLONK layer through a real gadget of the
reference: the circuit only closes if both restatements agree on hash_pad, key-bit order and sibling order."""
import importlib

import numpy as np
import pytest

P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def circ():
    import __graft_entry__ as g

    g.load_package()
    return importlib.import_module("intmax_zkp_core_amd.circuit")


def u128(x):  # GoldilocksHashOut::from_u128: the low 64 bits in limb 0, the high ones in limb 1
    return (x & (2**64 - 1), x >> 64, 0, 0)


def proof_of(res, i, key):
    n = int(res["n_siblings"][i])
    return dict(root=res["root"], key=key, value=res["value"][i], siblings=[tuple(int(x) for x in s) for s in res["siblings"][i][:n]],
                found=bool(res["found"][i]), is_old0=bool(res["is_old0"][i]), not_found_key=res["not_found_key"][i],
                not_found_value=res["not_found_value"][i])


def prove_and_verify(oracle, cd, wires, pis):
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"])
    fp = (cd.degree_bits, 3, 4, 8, 6, oracle.fri_reduction_arity_bits(cd.degree_bits, 3, 4))
    proof = oracle.plonk_prove(cd, cs, wires, pis, fp)
    return oracle.plonk_verify(cd, cs["cap"], pis, proof, fp)


def test_reference_inclusion_test_by_the_oracle(oracle, circ):
    """src/sparse_merkle_tree/gadgets/verify/mod.rs::test_verify_inclusion_proof_by_plonky2"""
    keys = np.array([u128(1), u128(12), u128(5)], dtype=np.uint64)
    vals = np.array([u128(2), u128(1), u128(51)], dtype=np.uint64)
    res = oracle.smt_proofs_of_set(keys, vals, keys[2:3])
    assert res["found"][0] and [int(x) for x in res["value"][0]] == list(u128(51))
    c = circ.SmtInclusionCircuit(16)
    assert c.data.degree_bits == 7 and {g["id"] for g in c.data.gates} == {circ.GATE_NOOP, circ.GATE_CONSTANT, circ.GATE_PUBLIC_INPUT,
                                                                          circ.GATE_BASE_SUM, circ.GATE_ARITHMETIC, circ.GATE_POSEIDON}
    wires, pis = c.witness(proof_of(res, 0, keys[2]), oracle.poseidon_gate_rows)
    assert pis == []
    assert prove_and_verify(oracle, c.data, wires, pis) == 0


def test_inclusion_and_non_inclusion_on_a_larger_set(oracle, circ):
    rng = np.random.default_rng(9)
    n = 300
    keys = rng.integers(0, 2**63, (n, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (n, 4), dtype=np.uint64)
    absent = rng.integers(0, 2**63, (40, 4), dtype=np.uint64)
    near = keys[:8].copy()
    near[:, 3] ^= np.uint64(1 << 60)  # share ~250 leading bits with a present key: ends on that key's leaf (is_old0 = 0)
    queries = np.concatenate([keys[:6], absent, near])
    res = oracle.smt_proofs_of_set(keys, vals, queries)
    kinds = {(bool(f), bool(z)) for f, z in zip(res["found"], res["is_old0"])}
    assert kinds == {(True, False), (False, True), (False, False)}
    c = circ.SmtInclusionCircuit(32, public_root=True)
    done = set()
    for i in range(len(queries)):
        kind = (bool(res["found"][i]), bool(res["is_old0"][i]))
        if kind in done and i % 9:
            continue
        done.add(kind)
        pr = proof_of(res, i, queries[i])
        wires, pis = c.witness(pr, oracle.poseidon_gate_rows)
        assert pis[:4] == [int(x) for x in res["root"]] and pis[4:8] == [int(x) for x in queries[i]]
        if len(done) <= 3 and kind not in list(done)[:-1]:
            assert prove_and_verify(oracle, c.data, wires, pis) == 0, kind
        # what the circuit must refuse: the witness generator meets a violated connect (plonky2: generate_partial_witness panics)
        bad = dict(pr, siblings=[tuple((int(x) + (j == 0)) % P for x in s) for j, s in enumerate(pr["siblings"])])
        if pr["siblings"]:
            with pytest.raises(ValueError):
                c.witness(bad, oracle.poseidon_gate_rows)
        with pytest.raises(ValueError):
            c.witness(dict(pr, root=[(int(x) + 1) % P for x in pr["root"]]), oracle.poseidon_gate_rows)
        if pr["found"]:
            with pytest.raises(ValueError):  # claiming non-inclusion of a present key
                c.witness(dict(pr, found=False, not_found_key=pr["key"], not_found_value=pr["value"]), oracle.poseidon_gate_rows)
        # enabled = false: nothing is enforced (the merge circuit's unused slots)
        c.witness(dict(pr, root=[1, 2, 3, 4]), oracle.poseidon_gate_rows, enabled=False)
    assert len(done) == 3


def test_wide_split_recombines(oracle, circ):
    """split_le_wide(x, 64): two BaseSumGate rows, sum_lo + 2^63 sum_hi == x"""
    b = circ.CircuitBuilder()
    x = b.virtual()
    bits = b.split_le_wide(x, 64)
    b.register_public_inputs([bits[0], bits[62], bits[63]])
    cd = b.build()
    for v in (0, 1, 2**63, 2**63 + 5, P - 1):
        wires, pis = cd.generate_witness({x: v}, oracle.poseidon_gate_rows)
        assert pis == [v & 1, (v >> 62) & 1, (v >> 63) & 1]


def test_process_proofs_of_all_four_kinds_close_the_reference_processor_circuit(oracle, circ):
    """circuit.SmtProcessCircuit = verify_smt_process_proof (src/sparse_merkle_tree/gadgets/process/process_smt.rs:117-341): the
    process proofs of sequential `set`s (oracle/smt_incremental.py: the reference's update / insert / remove / no-op, tree.rs:174-547)
    satisfy it — one proof of each kind is proven and verified by the oracle — and tampered ones do not generate a witness"""
    from oracle import smt_incremental as inc

    rng = np.random.default_rng(31)
    tree = inc.SparseMerkleTree()
    keys = [tuple(int(x) for x in rng.integers(0, 2**63, 4, dtype=np.uint64)) for _ in range(40)]
    for k in keys:
        tree.set(k, tuple(int(x) for x in rng.integers(1, 2**63, 4, dtype=np.uint64)))
    c = circ.SmtProcessCircuit(32, public_roots=True)
    assert c.data.degree_bits == 9
    near = list(keys[3])
    near[0] ^= 1 << 20  # an insert that meets keys[3]'s leaf and pushes it 20 levels down (is_old0 = 0; the keys must part within
    # n_levels bits: set_witness panics with "invalid `new_key`" otherwise, process_smt.rs:93-99)
    steps = [(keys[0], (5, 6, 7, 8)),            # update
             (tuple(near), (1, 1, 1, 1)),         # insert beside another leaf
             ((1, 2, 3, 4), (9, 9, 9, 9)),        # insert (most likely into an empty slot)
             (keys[7], (0, 0, 0, 0)),             # remove
             ((4, 3, 2, 1), (0, 0, 0, 0)),        # no-op
             (tuple(near), (0, 0, 0, 0))]         # remove: the lone sibling leaf moves back up
    seen = set()
    for key, value in steps:
        pr = inc.set_with_proof(tree, key, value)
        wires, pis = c.witness(pr, oracle.poseidon_gate_rows)
        assert pis == [int(x) for x in pr["old_root"]] + [int(x) for x in pr["new_root"]]
        if pr["fnc"] not in seen:
            seen.add(pr["fnc"])
            assert prove_and_verify(oracle, c.data, wires, pis) == 0, pr["fnc"]
        if pr["fnc"] != inc.NOOP:
            with pytest.raises(ValueError):
                c.witness(dict(pr, new_root=tuple((int(x) + 1) % P for x in pr["new_root"])), oracle.poseidon_gate_rows)
            with pytest.raises(ValueError):
                c.witness(dict(pr, old_value=tuple((int(x) + 1) % P for x in pr["old_value"])), oracle.poseidon_gate_rows)
    assert seen == {inc.NOOP, inc.UPDATE, inc.INSERT, inc.DELETE}
    # keys that part only after n_levels bits: the reference refuses to fill the witness; here the circuit cannot be satisfied
    far = list(keys[5])
    far[3] ^= 1 << 60
    pr = inc.set_with_proof(tree, tuple(far), (2, 2, 2, 2))
    assert pr["fnc"] == inc.INSERT and not pr["is_old0"]
    with pytest.raises((ValueError, AssertionError)):
        c.witness(pr, oracle.poseidon_gate_rows)


def test_reference_merkle_proof_test_by_the_oracle(oracle, circ):
    """src/merkle_tree/gadgets/mod.rs::test_verify_merkle_proof_by_plonky2: leaves [0, 10, 20, 30, 40, 0], the proof of index 5 at
    N_LEVELS = 10; the circuit's first four public inputs are the native root"""
    leaves = np.array([[v, 0, 0, 0] for v in (0, 10, 20, 30, 40, 0)], dtype=np.uint64)
    sib, root = oracle.merkle_proof_with_zero(leaves, 5, 10)
    c = circ.MerkleProofCircuit(10)
    wires, pis = c.witness(5, leaves[5], sib, oracle.poseidon_gate_rows)
    assert pis[:4] == [int(x) for x in root] and pis[4] == 5 and pis[5:] == [0, 0, 0, 0]
    assert [int(x) for x in oracle.merkle_root(5, leaves[5], sib)] == pis[:4]
    assert prove_and_verify(oracle, c.data, wires, pis) == 0
    rng = np.random.default_rng(2)
    lv = rng.integers(0, 2**63, (37, 4), dtype=np.uint64)
    for idx in (0, 18, 36):
        sib, root = oracle.merkle_proof_with_zero(lv, idx, 10)
        _, pis = c.witness(idx, lv[idx], sib, oracle.poseidon_gate_rows)
        assert pis[:4] == [int(x) for x in root]
    with pytest.raises(ValueError):  # an index outside the range check
        c.witness(1 << 10, lv[0], sib, oracle.poseidon_gate_rows)


def test_block_hash_circuit_equals_the_native_block_hash(oracle, circ, kat):
    """get_block_hash_target (src/transaction/gadgets/block_header/mod.rs:72-101) against BlockHeader::get_block_hash as restated
    (oracle.block_hash, pinned by the block1_info.json header): random headers and the default header"""
    c = circ.BlockHashCircuit()
    rng = np.random.default_rng(4)
    names = ("prev_block_hash",) + circ.BlockHashCircuit.FIELDS
    for trial in range(3):
        h = {"block_number": int(rng.integers(0, 2**32)) if trial else 0}
        for nm in names:
            h[nm] = [int(x) for x in rng.integers(0, 2**63, 4, dtype=np.uint64)] if trial else [0, 0, 0, 0]
        words = [h["block_number"]] + [x for nm in names for x in h[nm]]
        want = [int(x) for x in oracle.block_hash(np.array(words, dtype=np.uint64))]
        wires, pis = c.witness(h, oracle.poseidon_gate_rows)
        assert pis == want, trial
        if trial == 1:
            assert prove_and_verify(oracle, c.data, wires, pis) == 0
    with pytest.raises(ValueError):
        c.witness(dict(h, block_number=1 << 32), oracle.poseidon_gate_rows)  # range_check(block_number, 32)


def purge_scenario(oracle):
    """the scenario of src/transaction/gadgets/purge/mod.rs::test_purge_proof_by_plonky2 (:494-567): two assets in the sender's asset
    tree, both purged, two assets inserted into the transaction's diff tree"""
    from oracle import smt_incremental as inc

    key1, value1 = (u128(1), u128(3), u128(8)), u128(2)
    key2, value2 = (u128(12), u128(4), u128(8)), u128(1)
    key3, value3 = (u128(407), u128(3), u128(8)), u128(2)
    key4, value4 = (u128(832), u128(4), u128(8)), u128(1)
    sk = [15657143458229430356, 6012455030006979790, 4280058849535143691, 5153662694263190591]
    address = int(oracle.two_to_one(sk, sk)[0])  # public_key_to_address: the first limb of H(sk, sk) (src/zkdsa/account.rs:168-170)
    assets, diff = inc.UserAssetTree(), inc.LayeredLayeredSparseMerkleTree()
    inc.user_asset_set_with_proofs(assets, *key1, value1)
    inc.user_asset_set_with_proofs(assets, *key2, value2)
    old_root = assets.root
    zero = u128(0)
    proof1 = inc.user_asset_set_with_proofs(assets, *key2, zero)
    proof2 = inc.user_asset_set_with_proofs(assets, *key1, zero)
    proof3 = inc.layered_layered_set_with_proofs(diff, *key3, value3)
    proof4 = inc.layered_layered_set_with_proofs(diff, *key4, value4)
    nonce = [6657881311364026367, 11761473381903976612, 10768494808833234712, 3223267375194257474]
    return address, [proof1, proof2], [proof3, proof4], old_root, nonce, assets.root, diff.root


def test_reference_purge_test_by_the_oracle(oracle, circ):
    """PurgeCircuit = PurgeTransitionTarget (the purge half of the reference's user-transaction circuit) on the reference's own test:
    the witness closes, the public inputs are (new_user_asset_root, diff_root, two_to_one(diff_root, nonce)) as set_witness returns
    them, the oracle proves and verifies; the all-default witness of the test's second half closes too; unequal assets do not"""
    address, ins, outs, old_root, nonce, new_root, diff_root = purge_scenario(oracle)
    for w0, w1, w2 in ins:  # what set_witness asserts about the input proofs (:158-206)
        assert w0["fnc"] == 1 and w1["fnc"] in (1, 3) and w2["fnc"] == 3
    for w0, w1, w2 in outs:
        assert w0["fnc"] in (1, 2) and w1["fnc"] in (1, 2) and w2["fnc"] == 2
    c = circ.PurgeCircuit()
    assert c.data.degree_bits == 11 and {circ.GATE_ARITHMETIC_EXT, circ.GATE_MUL_EXT, circ.GATE_BASE_SUM} <= {g["id"] for g in c.data.gates}
    wires, pis = c.witness(address, ins, outs, old_root, nonce, oracle.poseidon_gate_rows)
    assert pis[:4] == list(new_root) and pis[4:8] == list(diff_root)
    assert pis[8:] == [int(x) for x in oracle.two_to_one(diff_root, nonce)]
    assert prove_and_verify(oracle, c.data, wires, pis) == 0
    # the default witness (no diffs at all)
    wires, pis = c.witness(0, [], [], (0, 0, 0, 0), (0, 0, 0, 0), oracle.poseidon_gate_rows)
    assert pis[:8] == [0] * 8 and pis[8:] == [int(x) for x in oracle.two_to_one([0] * 4, [0] * 4)]
    # one purged asset only, two inserted: the messes differ
    with pytest.raises((ValueError, AssertionError)):
        c.witness(address, ins[:1], outs, old_root, nonce, oracle.poseidon_gate_rows)


def _mp(oracle):
    return lambda leaves, index, depth, zero: oracle.merkle_proof_with_zero(np.array(leaves, dtype=np.uint64).reshape(-1, 4), index, depth, zero)


def deposit_merge_scenario(oracle, log_n_txs, other_tx_hashes=(), wide_header=False, world_state=(0, 0, 0, 0)):
    """the deposit the reference merges in src/transaction/gadgets/merge/mod.rs::test_merge_proof_by_plonky2 (:443-583) and, with
    wide_header, in src/bin/block_circuit.rs (:157-262): two assets (305, 8012, 2053), (471, 8012, 1111) deposited to the user's
    address in block 1 and merged into the user's (empty) asset tree -> (address, merge proof, user asset tree, merge key)"""
    from oracle import smt_incremental as inc

    sk = [15657143458229430356, 6012455030006979790, 4280058849535143691, 5153662694263190591]
    address = (int(oracle.two_to_one(sk, sk)[0]), 0, 0, 0)
    a1, a2 = (u128(305), u128(8012), u128(2053)), (u128(471), u128(8012), u128(1111))
    deposit_tree = inc.LayeredLayeredSparseMerkleTree()
    deposit_tree.set(address, *a1)
    deposit_tree.set(address, *a2)
    inclusion2 = inc.SparseMerkleTree(deposit_tree.nodes, deposit_tree.root).inclusion_proof(address)
    assert inclusion2["found"]
    nonce = (0, 0, 0, 0)
    tx_hash = tuple(int(x) for x in oracle.two_to_one(inclusion2["root"], nonce))
    leaves = [tx_hash] + list(other_tx_hashes)
    sib, mroot = oracle.merkle_proof_with_zero(np.array(leaves, dtype=np.uint64), 0, log_n_txs)
    default_root = oracle.merkle_proof_with_zero(np.zeros((0, 4), dtype=np.uint64), 0, log_n_txs)[1]
    z = (0, 0, 0, 0)
    header = dict(block_number=1, prev_block_hash=z, block_headers_digest=z, transactions_digest=tuple(int(x) for x in default_root),
                  deposit_digest=tuple(int(x) for x in mroot), proposed_world_state_digest=z, approved_world_state_digest=z,
                  latest_account_digest=z)
    if wide_header:  # block_circuit.rs:199-221
        header["block_headers_digest"] = tuple(int(x) for x in oracle.merkle_proof_with_zero(np.zeros((1, 4), dtype=np.uint64), 0, 32)[1])
        header["proposed_world_state_digest"] = header["approved_world_state_digest"] = tuple(world_state)
    names = ("prev_block_hash", "block_headers_digest", "transactions_digest", "deposit_digest", "proposed_world_state_digest",
             "approved_world_state_digest", "latest_account_digest")
    block_hash = oracle.block_hash(np.array([1] + [x for nm in names for x in header[nm]], dtype=np.uint64))
    merge_key = tuple(int(x) for x in oracle.two_to_one(tx_hash, block_hash))
    assets = inc.UserAssetTree()
    assets.set(merge_key, *a1)
    assets.set(merge_key, *a2)
    top = inc.SparseMerkleTree(assets.nodes, assets.root)
    asset_root = top.get(merge_key)
    assert asset_root == tuple(int(x) for x in oracle.two_to_one(inclusion2["value"], merge_key))
    inc.set_with_proof(top, merge_key, z)
    process = inc.set_with_proof(top, merge_key, asset_root)
    assert process["fnc"] == inc.INSERT and process["old_root"] == z and top.root == assets.root
    empty = dict(root=z, key=z, value=z, siblings=[], found=False, is_old0=True, not_found_key=z, not_found_value=z)
    proof = dict(is_deposit=True, header=header, merkle=(0, tx_hash, [tuple(int(x) for x in s) for s in sib]), diff_inclusion=inclusion2,
                 process=process, account_inclusion=empty, nonce=nonce)
    return address[0], proof, assets, merge_key, (a1, a2)


def test_reference_merge_test_by_the_oracle(oracle, circ):
    """MergeCircuit = MergeTransitionTarget (the merge half of the reference's user-transaction circuit) on the reference's own test
    (merge/mod.rs:405-612): one deposit merged, the other two slots default — public inputs (0^4, the user asset tree's root after
    the merge); then the all-default witness; then witnesses the circuit must refuse"""
    z = (0, 0, 0, 0)
    others = [tuple(int(x) for x in oracle.two_to_one(z, u128(1))), tuple(int(x) for x in oracle.two_to_one(z, u128(2)))]
    _, proof, assets, _, _ = deposit_merge_scenario(oracle, 3, others)
    c = circ.MergeCircuit(3, 3, 3, 3, 3)
    assert c.data.degree_bits == 9
    defaults = circ.MergeCircuit.defaults(3, _mp(oracle), oracle.two_to_one)
    assert circ.SmtInclusionCircuit.with_root(z) == proof["account_inclusion"]
    wires, pis = c.witness([proof], z, defaults, oracle.poseidon_gate_rows)
    assert pis == [0, 0, 0, 0] + list(assets.root)
    assert prove_and_verify(oracle, c.data, wires, pis) == 0
    wires, pis = c.witness([], z, defaults, oracle.poseidon_gate_rows)
    assert pis == [0] * 8
    assert prove_and_verify(oracle, c.data, wires, pis) == 0
    # a deposit whose nonce is not its index in the deposit tree; a merged value that is not two_to_one(asset_root, merge_key);
    # a transaction hash the block does not hold
    bad = dict(proof, nonce=(1, 0, 0, 0))
    with pytest.raises((ValueError, AssertionError)):
        c.witness([bad], z, defaults, oracle.poseidon_gate_rows)
    bad = dict(proof, diff_inclusion=dict(proof["diff_inclusion"], value=(1, 2, 3, 4)))
    with pytest.raises((ValueError, AssertionError)):
        c.witness([bad], z, defaults, oracle.poseidon_gate_rows)
    bad = dict(proof, merkle=(0, (5, 6, 7, 8), proof["merkle"][2]))
    with pytest.raises((ValueError, AssertionError)):
        c.witness([bad], z, defaults, oracle.poseidon_gate_rows)


def user_transactions(oracle):
    """the two user transactions of src/bin/block_circuit.rs (:79-367): sender1 purges two assets it holds (no merge); sender2 merges
    a deposit of block 1 and sends the merged assets on in the same transaction"""
    from oracle import smt_incremental as inc

    zero = u128(0)
    key1, value1 = (u128(12), u128(305), u128(8012)), u128(2053)
    key2, value2 = (u128(12), u128(471), u128(8012)), u128(1111)
    key3, value3 = (u128(407), u128(305), u128(8012)), u128(2053)
    key4, value4 = (u128(832), u128(471), u128(8012)), u128(1111)
    sk1 = [17426287337377512978, 8703645504073070742, 11984317793392655464, 9979414176933652180]
    address1 = int(oracle.two_to_one(sk1, sk1)[0])
    assets1, diff1 = inc.UserAssetTree(), inc.LayeredLayeredSparseMerkleTree()
    assets1.set(*key1, value1)
    assets1.set(*key2, value2)
    world = inc.SparseMerkleTree()
    world.set((address1, 0, 0, 0), assets1.root)
    old1 = assets1.root
    ins1 = [inc.user_asset_set_with_proofs(assets1, *key2, zero), inc.user_asset_set_with_proofs(assets1, *key1, zero)]
    outs1 = [inc.layered_layered_set_with_proofs(diff1, *key3, value3), inc.layered_layered_set_with_proofs(diff1, *key4, value4)]
    nonce1 = [7823975322825286183, 9539665429968124165, 6825628074508059665, 17852854585777218254]
    tx1 = dict(sender_address=address1, merges=[], ins=ins1, outs=outs1, nonce=nonce1, old_root=old1, middle_root=old1,
               new_root=assets1.root, diff_root=diff1.root)
    address2, merge_proof, assets2, merge_key, _ = deposit_merge_scenario(oracle, 2, (), wide_header=True, world_state=world.root)
    middle2 = assets2.root
    diff2 = inc.LayeredLayeredSparseMerkleTree()
    ins2 = [inc.user_asset_set_with_proofs(assets2, merge_key, key2[1], key2[2], zero),
            inc.user_asset_set_with_proofs(assets2, merge_key, key1[1], key1[2], zero)]
    outs2 = [inc.layered_layered_set_with_proofs(diff2, *key3, value3), inc.layered_layered_set_with_proofs(diff2, *key4, value4)]
    nonce2 = [6657881311364026367, 11761473381903976612, 10768494808833234712, 3223267375194257474]
    tx2 = dict(sender_address=address2, merges=[merge_proof], ins=ins2, outs=outs2, nonce=nonce2, old_root=(0, 0, 0, 0),
               middle_root=middle2, new_root=assets2.root, diff_root=diff2.root)
    return tx1, tx2


def test_user_transaction_circuit_on_the_reference_binarys_transactions(oracle, circ):
    """UserTransactionCircuit = make_user_proof_circuit with the RollupConstants of the reference's tests, on the two transactions
    src/bin/block_circuit.rs proves: the 24 public inputs are MergeAndPurgeTransitionPublicInputs::encode of what set_witness
    returns; the oracle proves and verifies both; the default transaction (Default::default(): no merges, no diffs) closes with the
    reference's golden default tx_hash (src/transaction/circuits/mod.rs:203-235)"""
    c = circ.UserTransactionCircuit()
    assert c.data.degree_bits == 11 and len(c.data.gates) == 8
    defaults = circ.MergeCircuit.defaults(c.constants["log_n_txs"], _mp(oracle), oracle.two_to_one)
    for tx in user_transactions(oracle):
        wires, pis = c.witness(tx["sender_address"], tx["merges"], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], defaults,
                               oracle.poseidon_gate_rows)
        want = (list(tx["old_root"]) + list(tx["middle_root"]) + list(tx["new_root"]) + list(tx["diff_root"]) + [tx["sender_address"], 0, 0, 0]
                + [int(x) for x in oracle.two_to_one(tx["diff_root"], tx["nonce"])])
        assert pis == want
        assert prove_and_verify(oracle, c.data, wires, pis) == 0
    z = (0, 0, 0, 0)
    wires, pis = c.witness(0, [], [], [], z, z, defaults, oracle.poseidon_gate_rows)
    assert pis[:20] == [0] * 20
    assert pis[20:] == [4330397376401421145, 14124799381142128323, 8742572140681234676, 14345658006221440202]
    assert prove_and_verify(oracle, c.data, wires, pis) == 0
    # the purge half must start from the root the merge half ends with
    tx = user_transactions(oracle)[1]
    with pytest.raises((ValueError, AssertionError)):
        c.witness(tx["sender_address"], [], tx["ins"], tx["outs"], tx["nonce"], tx["old_root"], defaults, oracle.poseidon_gate_rows)


def test_poseidon_rows_with_the_swap_wire_and_the_witness_scheduler(oracle, circ):
    """builder.poseidon(inputs, swap = bit) = plonky2's permute_swapped (PoseidonGate wires 24..28): with the bit set the row hashes
    (right, left), the oracle proves and verifies a circuit holding rows of both kinds; the dataflow witness generator runs operations
    in dependency order whatever order they were created in, refuses conflicting values and names an operation whose input never came"""
    b = circ.CircuitBuilder()
    left, right = [b.virtual() for _ in range(4)], [b.virtual() for _ in range(4)]
    bit0, bit1 = b.virtual_bool_safe(), b.virtual_bool_safe()
    zero = b.zero()
    outs = [b.poseidon(left + right + [zero] * 4, swap=bt)[:4] for bt in (bit0, bit1)]
    late = b.virtual()  # read by an operation created BEFORE the one that produces it
    user = b.mul(late, late)
    b.connect(late, b.add(outs[0][0], outs[1][0]))
    b.register_public_inputs(outs[0] + outs[1] + [user])
    cd = b.build()
    lv, rv = [1, 2, 3, 4], [5, 6, 7, 8]
    inputs = dict(zip(left + right + [bit0, bit1], lv + rv + [0, 1]))
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    h_lr, h_rl = [int(x) for x in oracle.two_to_one(lv, rv)], [int(x) for x in oracle.two_to_one(rv, lv)]
    assert pis[:4] == h_lr and pis[4:8] == h_rl and pis[8] == pow(h_lr[0] + h_rl[0], 2, P)
    rows = [r for r, (gid, _) in enumerate(b.rows) if gid == circ.GATE_POSEIDON][:2]
    assert [int(wires[24, r]) for r in rows] == [0, 1]
    assert [int(wires[25 + i, rows[1]]) for i in range(4)] == [(rv[i] - lv[i]) % P for i in range(4)]
    assert prove_and_verify(oracle, cd, wires, pis) == 0
    with pytest.raises(ValueError):  # a public-input class given a value the circuit contradicts
        cd.generate_witness({**inputs, user: 1}, oracle.poseidon_gate_rows)
    with pytest.raises(KeyError):  # an input target left unset
        cd.generate_witness({k: v for k, v in inputs.items() if k != bit1}, oracle.poseidon_gate_rows)


def test_reference_node_hash_and_logical_nor_tests(oracle, circ):
    """src/sparse_merkle_tree/gadgets/common.rs::test_calc_node_hash (calc_leaf_hash and calc_internal_hash in both orders against
    PoseidonNodeHash::calc_node_hash = oracle.smt_leaf_hash / two_to_one, key 1, value 2) and ::test_logical_nor (the four rows of
    the truth table as copy constraints on constants)"""
    G = circ.SmtInclusionCircuit
    b = circ.CircuitBuilder()
    key_t, value_t = [b.virtual() for _ in range(4)], [b.virtual() for _ in range(4)]
    out1 = G._leaf_hash(b, key_t, value_t)
    out2 = b.two_to_one(*G._reverse(b, key_t, value_t, b.zero()))  # calc_internal_hash(left, right, swap = false)
    out3 = b.two_to_one(*G._reverse(b, key_t, value_t, b.one()))
    b.register_public_inputs(out1 + out2 + out3)
    cd = b.build()
    key, value = u128(1), u128(2)
    wires, pis = cd.generate_witness(dict(zip(key_t + value_t, key + value)), oracle.poseidon_gate_rows)
    want = [oracle.smt_leaf_hash(key, value), oracle.two_to_one(key, value), oracle.two_to_one(value, key)]
    assert pis == [int(x) for h in want for x in h]
    assert prove_and_verify(oracle, cd, wires, pis) == 0
    b = circ.CircuitBuilder()
    t, f = b.one(), b.zero()
    nor = lambda x, y: G._and_not(b, b.not_(x), y)  # noqa: E731  (logical_nor = logical_and_not(not x, y))
    for x, y, want_t in ((t, t, f), (t, f, f), (f, t, f), (f, f, t)):
        b.connect(nor(x, y), want_t)
    cd = b.build()
    wires, pis = cd.generate_witness({}, oracle.poseidon_gate_rows)
    assert prove_and_verify(oracle, cd, wires, pis) == 0


def test_reference_process_proof_tests(oracle, circ):
    """src/sparse_merkle_tree/gadgets/process/mod.rs::test_verify_process_proof_by_plonky2 (a random walk of `set`s on a tree, the LAST
    process proof in the processor gadget at 16 levels, public inputs old key / value, new key / value, old / new root) and
    ::test_verify_process_proof2_by_plonky2 (the world-state tree of two senders at 32 levels: sender 1's asset root set, sender 2's
    set, sender 1's updated after its purge — the last update in the gadget)"""
    from oracle import smt_incremental as inc

    def processor(n_levels):
        b = circ.CircuitBuilder()
        g = circ.SmtProcessCircuit(n_levels, builder=b)
        b.register_public_inputs(g.old_key + g.old_value + g.new_key + g.new_value + g.old_root + g.new_root)
        return g, b.build()

    g, cd = processor(16)
    rng = np.random.default_rng(31)
    for walk in range(3):
        tree = inc.SparseMerkleTree()
        witness = inc.set_with_proof(tree, u128(1), u128(2))
        keys = [u128(1)]
        for _ in range(int(rng.integers(1, 40))):
            fresh = rng.random() < 0.5
            key = tuple(int(x) for x in rng.integers(0, 2**63, 4, dtype=np.uint64)) if fresh else keys[int(rng.integers(0, len(keys)))]
            value = (0, 0, 0, 0) if rng.random() < 0.4 else tuple(int(x) for x in rng.integers(1, 2**63, 4, dtype=np.uint64))
            witness = inc.set_with_proof(tree, key, value)
            keys.append(key)
        inc.verify_process_proof(witness)
        inputs = {}
        g.fill(inputs, witness)
        wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
        assert pis[16:20] == list(witness["old_root"]) and pis[20:] == list(witness["new_root"]) == list(tree.root)
        if walk == 0:
            assert prove_and_verify(oracle, cd, wires, pis) == 0
    # the second test: world state of two senders
    g, cd = processor(32)
    sk1 = [17426287337377512978, 8703645504073070742, 11984317793392655464, 9979414176933652180]
    sk2 = [17814943904840276189, 12088887497349422745, 1199609976110004574, 5153662694263190591]
    a1, a2 = ((int(oracle.two_to_one(sk, sk)[0]), 0, 0, 0) for sk in (sk1, sk2))
    key1, key2 = (u128(12), u128(305), u128(8012)), (u128(12), u128(471), u128(8012))
    assets = inc.LayeredLayeredSparseMerkleTree()
    assets.set(*key1, u128(2053))
    assets.set(*key2, u128(1111))
    world = inc.SparseMerkleTree()
    inc.verify_process_proof(inc.set_with_proof(world, a1, assets.root))
    assets.set(*key2, (0, 0, 0, 0))
    assets.set(*key1, (0, 0, 0, 0))
    inc.verify_process_proof(inc.set_with_proof(world, a2, tuple(int(x) for x in rng.integers(1, 2**63, 4, dtype=np.uint64))))
    witness = inc.set_with_proof(world, a1, assets.root)  # the asset tree is empty again: the world-state entry is REMOVED
    inc.verify_process_proof(witness)
    inputs = {}
    g.fill(inputs, witness)
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    assert pis[20:] == list(world.root) and witness["fnc"] == inc.DELETE
    assert prove_and_verify(oracle, cd, wires, pis) == 0
