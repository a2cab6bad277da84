"""The reference's SMT (non-)inclusion gadget (circuit.SmtInclusionCircuit = verify_smt_inclusion_proof, reference
src/sparse_merkle_tree/gadgets/verify/verify_smt.rs:209-307) END TO END ON THE GPU: the inclusion proofs come from the device-side
tree (glp_smt_proofs_of_set = SparseMerkleTree::find), the PoseidonGate rows of the witness from glp_poseidon_gate_witness_dev, the
proof from glp_prove / glp_prove_batch_dev — and equals the oracle's proof of the same witness word for word; the restated verifier
accepts.  Includes the reference's own test (gadgets/verify/mod.rs:3-52)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    trees = importlib.import_module("intmax_zkp_core_amd.trees")
    ctx = pkg.Context(0)
    yield pkg, circ, trees, ctx
    ctx.close()


def u128(x):
    return (x & (2**64 - 1), x >> 64, 0, 0)


def proof_of(res, i):
    return dict(root=res.root, key=res.keys[i], value=res.value[i], siblings=[tuple(int(x) for x in s) for s in res.siblings(i)],
                found=bool(res.found[i]), is_old0=bool(res.is_old0[i]), not_found_key=res.not_found_key[i],
                not_found_value=res.not_found_value[i])


def test_reference_inclusion_test_on_the_gpu(env, oracle):
    pkg, circ, trees, ctx = env
    keys = np.array([u128(1), u128(12), u128(5)], dtype=np.uint64)
    vals = np.array([u128(2), u128(1), u128(51)], dtype=np.uint64)
    res = trees.smt_find_batch(ctx, keys, vals, keys[2:3])
    assert res.found[0] and [int(x) for x in res.value[0]] == list(u128(51))
    want = oracle.smt_proofs_of_set(keys, vals, keys[2:3])
    assert (res.root == want["root"]).all() and (res.siblings(0) == want["siblings"][0][: int(want["n_siblings"][0])]).all()
    c = circ.SmtInclusionCircuit(16)
    wires, pis = c.witness(proof_of(res, 0), lambda ins: pkg.poseidon_gate_rows(ctx, ins))
    pc = pkg.ProverCircuit(ctx, c.data)
    proof = pc.prove(wires, pis)
    cs = oracle.batch_from_values(c.data.constants_sigmas_values(), 3, 4)
    assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
    assert (proof == oracle.plonk_prove(c.data, cs, wires, pis)).all()
    assert oracle.plonk_verify(c.data, cs["cap"], pis, proof) == 0
    pc.free()


def test_a_batch_of_inclusion_and_non_inclusion_proofs(env, oracle):
    """2^12 entries, 24 queries of all three kinds (found / ends in an empty subtree / ends on another leaf): one
    glp_prove_batch_dev call over the 24 witnesses; each proof equals the single proof, the oracle's proof of a sample, verifies"""
    pkg, circ, trees, ctx = env
    rng = np.random.default_rng(21)
    n = 1 << 12
    keys = rng.integers(0, 2**63, (n, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (n, 4), dtype=np.uint64)
    near = keys[:8].copy()
    near[:, 3] ^= np.uint64(1 << 60)
    queries = np.concatenate([keys[100:108], rng.integers(0, 2**63, (8, 4), dtype=np.uint64), near])
    res = trees.smt_find_batch(ctx, keys, vals, queries)
    kinds = {(bool(f), bool(z)) for f, z in zip(res.found, res.is_old0)}
    assert kinds == {(True, False), (False, True), (False, False)}
    c = circ.SmtInclusionCircuit(32, public_root=True)
    ws, ps = [], []
    for i in range(len(queries)):
        w, pis = c.witness(proof_of(res, i), lambda ins: pkg.poseidon_gate_rows(ctx, ins))
        assert pis[:4] == [int(x) for x in res.root]
        ws.append(w)
        ps.append(pis)
    pc = pkg.ProverCircuit(ctx, c.data)
    import torch

    batch = torch.stack([ctx.to_device(w) for w in ws]).contiguous()
    proofs = pc.prove_batch(batch, ps)
    cs = oracle.batch_from_values(c.data.constants_sigmas_values(), 3, 4)
    for i in range(len(queries)):
        assert oracle.plonk_verify(c.data, cs["cap"], ps[i], proofs[i]) == 0, i
        if i % 8 == 0:
            assert (proofs[i] == pc.prove(ws[i], ps[i])).all()
            assert (proofs[i] == oracle.plonk_prove(c.data, cs, ws[i], ps[i])).all()
    # a proof about one key does not verify as a proof about another
    assert oracle.plonk_verify(c.data, cs["cap"], ps[1], proofs[0]) != 0
    pc.free()


def test_a_block_of_state_updates_proven_with_the_reference_processor_gadget(env, oracle):
    """The chain a block producer needs for its state trees, entirely on the GPU: 48 sequential `set`s on a 2^12-entry tree as ONE
    glp_smt_apply_updates call (updates, inserts beside other leaves, removals, no-ops), each resulting SparseMerkleProcessProof fed to
    the reference's processor gadget (circuit.SmtProcessCircuit = verify_smt_process_proof, process_smt.rs:117-341), all 48 circuits
    proven by one glp_prove_batch_dev call.  Every proof verifies with [old_root, new_root] as public inputs, the roots chain
    (new_root of update i = old_root of update i + 1), samples equal the oracle's proof word for word."""
    pkg, circ, trees, ctx = env
    rng = np.random.default_rng(77)
    n, k = 1 << 12, 48
    keys = rng.integers(0, 2**63, (n, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (n, 4), dtype=np.uint64)
    uk = np.concatenate([keys[rng.integers(0, n, k // 2)], rng.integers(0, 2**63, (k - k // 2, 4), dtype=np.uint64)])
    rng.shuffle(uk)
    uv = rng.integers(1, 2**63, (k, 4), dtype=np.uint64)
    uv[rng.random(k) < 0.35] = 0
    proofs, roots = trees.smt_apply_updates(ctx, keys, vals, uk, uv)
    assert {p["fnc"] for p in proofs} == {trees.NOOP, trees.UPDATE, trees.INSERT, trees.DELETE}
    c = circ.SmtProcessCircuit(32, public_roots=True)
    ws, ps = [], []
    for i, p in enumerate(proofs):
        w, pis = c.witness(p, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
        assert pis[:4] == [int(x) for x in roots[i]] and pis[4:] == [int(x) for x in roots[i + 1]]
        ws.append(w)
        ps.append(pis)
    pc = pkg.ProverCircuit(ctx, c.data)
    import torch

    out = pc.prove_batch(torch.stack([ctx.to_device(w) for w in ws]).contiguous(), ps)
    cs = oracle.batch_from_values(c.data.constants_sigmas_values(), 3, 4)
    assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
    for i in range(k):
        assert oracle.plonk_verify(c.data, cs["cap"], ps[i], out[i]) == 0, i
        if i % 12 == 0:
            assert (out[i] == oracle.plonk_prove(c.data, cs, ws[i], ps[i])).all(), i
    assert oracle.plonk_verify(c.data, cs["cap"], ps[1], out[0]) != 0
    pc.free()


def test_merkle_proof_and_block_hash_gadgets_on_the_gpu(env, oracle):
    """the reference's fixed-depth Merkle proof circuit (its own test: leaves [0, 10, 20, 30, 40, 0], index 5, 10 levels,
    src/merkle_tree/gadgets/mod.rs:109-172) with the proof from glp_merkle_proof_with_zero_dev, and its block-hash gadget
    (src/transaction/gadgets/block_header/mod.rs:72-101) against glp_block_hash_batch_dev: the circuits' public inputs are the
    library's native results, the GPU proofs equal the oracle's and verify"""
    pkg, circ, trees, ctx = env
    rows_fn = lambda ins: pkg.poseidon_gate_rows(ctx, ins)  # noqa: E731
    leaves = np.array([[v, 0, 0, 0] for v in (0, 10, 20, 30, 40, 0)], dtype=np.uint64)
    sib, root = trees.get_merkle_proof(ctx, leaves, 5, 10)
    c = circ.MerkleProofCircuit(10)
    wires, pis = c.witness(5, leaves[5], sib, rows_fn)
    assert pis[:4] == [int(x) for x in root] == [int(x) for x in trees.get_merkle_root(ctx, [5], leaves[5:6], sib[None])[0]]
    pc = pkg.ProverCircuit(ctx, c.data)
    proof = pc.prove(wires, pis)
    cs = oracle.batch_from_values(c.data.constants_sigmas_values(), 3, 4)
    assert (proof == oracle.plonk_prove(c.data, cs, wires, pis)).all() and oracle.plonk_verify(c.data, cs["cap"], pis, proof) == 0
    pc.free()
    hc = circ.BlockHashCircuit()
    rng = np.random.default_rng(8)
    names = ("prev_block_hash",) + circ.BlockHashCircuit.FIELDS
    hdrs = []
    for _ in range(5):
        h = {"block_number": int(rng.integers(0, 2**32))}
        for nm in names:
            h[nm] = [int(x) for x in rng.integers(0, 2**63, 4, dtype=np.uint64)]
        hdrs.append(h)
    words = np.array([[h["block_number"]] + [x for nm in names for x in h[nm]] for h in hdrs], dtype=np.uint64)
    native = trees.get_block_hash(ctx, words)
    pc = pkg.ProverCircuit(ctx, hc.data)
    cs = oracle.batch_from_values(hc.data.constants_sigmas_values(), 3, 4)
    for h, want in zip(hdrs, native):
        wires, pis = hc.witness(h, rows_fn)
        assert pis == [int(x) for x in want]
        proof = pc.prove(wires, pis)
        assert oracle.plonk_verify(hc.data, cs["cap"], pis, proof) == 0
    assert (proof == oracle.plonk_prove(hc.data, cs, wires, pis)).all()
    pc.free()


def test_reference_purge_test_on_the_gpu(env, oracle):
    """PurgeCircuit (the purge half of the reference's user-transaction circuit) on the reference's own test scenario
    (src/transaction/gadgets/purge/mod.rs:438-610), every piece from the GPU: the eighteen process proofs of its witness come from
    glp_smt_apply_updates_multi — one call per layer of the sender's asset tree (variable, contract, merge-key layer, whose values are
    two_to_one(contract-layer root, merge_key) from glp_two_to_one_batch_dev) and of the diff tree — and equal the reference's
    sequential algorithm proof by proof; the circuit's public inputs are the trees' roots; the GPU proof (8 gate types incl.
    ArithmeticExtensionGate / MulExtensionGate) equals the oracle's word for word and verifies."""
    from tests.test_oracle_smt_circuit import purge_scenario

    pkg, circ, trees, ctx = env
    address, ins_want, outs_want, old_root, nonce, new_root, diff_root = purge_scenario(oracle)
    Z = (0, 0, 0, 0)
    T = lambda a: tuple(int(x) for x in a)  # noqa: E731
    arr = lambda rows: np.array(rows, dtype=np.uint64).reshape(-1, 4)  # noqa: E731
    key1, value1 = (u128(1), u128(3), u128(8)), u128(2)
    key2, value2 = (u128(12), u128(4), u128(8)), u128(1)
    # ---- inputs: purge key2's asset, then key1's, from the sender's asset tree (two different merge keys) ----
    order = [(key2, value2), (key1, value1)]
    r_var = trees.smt_apply_updates_multi(ctx, [(arr([k[2]]), arr([v])) for k, v in order], [(arr([k[2]]), arr([Z])) for k, _ in order])
    var_before = [T(r[1][0]) for r in r_var]
    var_after = [T(r[1][1]) for r in r_var]
    assert var_after == [Z, Z]
    r_con = trees.smt_apply_updates_multi(ctx, [(arr([k[1]]), arr([var_before[i]])) for i, (k, _) in enumerate(order)],
                                          [(arr([k[1]]), arr([var_after[i]])) for i, (k, _) in enumerate(order)])
    con_before = [T(r[1][0]) for r in r_con]
    con_after = [T(r[1][1]) for r in r_con]
    mk = arr([k[0] for k, _ in order])
    top_before = [T(x) for x in trees.two_to_one(ctx, arr(con_before), mk)]  # asset_root = two_to_one(contract-layer root, merge_key)
    top_after = [T(x) for x in trees.two_to_one(ctx, arr(con_after), mk)]
    r_top = trees.smt_apply_updates_multi(ctx, [(mk, arr(top_before))], [(mk, arr(top_after))])[0]
    assert T(r_top[1][0]) == tuple(old_root) and T(r_top[1][-1]) == tuple(new_root)
    ins = [(r_top[0][i], r_con[i][0][0], r_var[i][0][0]) for i in range(2)]
    # ---- outputs: two assets into the empty diff tree (LayeredLayeredSparseMerkleTree::set twice) ----
    key3, value3 = (u128(407), u128(3), u128(8)), u128(2)
    key4, value4 = (u128(832), u128(4), u128(8)), u128(1)
    empty = (np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64))
    outs_in = [(key3, value3), (key4, value4)]
    o3 = trees.smt_apply_updates_multi(ctx, [empty, empty], [(arr([k[2]]), arr([v])) for k, v in outs_in])
    o2 = trees.smt_apply_updates_multi(ctx, [empty, empty], [(arr([k[1]]), arr([T(o3[i][1][1])])) for i, (k, _) in enumerate(outs_in)])
    o1 = trees.smt_apply_updates_multi(ctx, [empty], [(arr([k[0] for k, _ in outs_in]), arr([T(o2[i][1][1]) for i in range(2)]))])[0]
    assert T(o1[1][-1]) == tuple(diff_root)
    outs = [(o1[0][i], o2[i][0][0], o3[i][0][0]) for i in range(2)]
    for got, want in ((ins, ins_want), (outs, outs_want)):
        for g3, w3 in zip(got, want):
            for gp, wp in zip(g3, w3):
                wp = dict(wp, siblings=[tuple(s) for s in wp["siblings"]])
                assert gp == wp
    c = circ.PurgeCircuit()
    wires, pis = c.witness(address, ins, outs, old_root, nonce, lambda x: pkg.poseidon_gate_rows(ctx, x))
    assert pis[:4] == list(new_root) and pis[4:8] == list(diff_root)
    assert pis[8:] == [int(x) for x in trees.two_to_one(ctx, arr([diff_root]), arr([nonce]))[0]]
    pc = pkg.ProverCircuit(ctx, c.data)
    proof = pc.prove(wires, pis)
    cs = oracle.batch_from_values(c.data.constants_sigmas_values(), 3, 4)
    assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
    assert (proof == oracle.plonk_prove(c.data, cs, wires, pis)).all()
    assert oracle.plonk_verify(c.data, cs["cap"], pis, proof) == 0
    pc.free()


def test_reference_merge_test_and_user_transactions_on_the_gpu(env, oracle):
    """MergeCircuit on the reference's own test (src/transaction/gadgets/merge/mod.rs:405-612) and UserTransactionCircuit
    (= make_user_proof_circuit, src/transaction/circuits/mod.rs:88-164) on the two transactions src/bin/block_circuit.rs proves plus
    the default transaction.  For the transaction that merges a deposit and sends it on, EVERY piece of the witness comes from the
    GPU library — the deposit tree's roots (glp_smt_roots_of_sets), the receiver's inclusion proof (glp_smt_proofs_of_set), the
    deposit's Merkle proof (glp_merkle_proof_with_zero_dev), the block hash (glp_block_hash_batch_dev), the merge / purge / diff
    process proofs (glp_smt_apply_updates[_multi]), hashes (glp_two_to_one_batch_dev), the unused slots' defaults — and equals the
    reference's sequential algorithm piece by piece.  The three user-transaction witnesses go through one glp_prove_batch_dev call;
    every proof equals the oracle's word for word and verifies."""
    from tests.test_oracle_smt_circuit import deposit_merge_scenario, user_transactions

    pkg, circ, trees, ctx = env
    rows_fn = lambda x: pkg.poseidon_gate_rows(ctx, x)  # noqa: E731
    Z = (0, 0, 0, 0)
    T = lambda a: tuple(int(x) for x in a)  # noqa: E731
    arr = lambda rows: np.array(rows, dtype=np.uint64).reshape(-1, 4)  # noqa: E731
    h2 = lambda l, r: T(trees.two_to_one(ctx, arr([l]), arr([r]))[0])  # noqa: E731
    mp = lambda leaves, index, depth, zero: trees.get_merkle_proof_with_zero(ctx, arr(leaves), index, depth, zero)  # noqa: E731
    empty = (np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64))

    # ---- the merge circuit of the reference's test ----
    others = [h2(Z, u128(1)), h2(Z, u128(2))]
    _, proof, assets, _, _ = deposit_merge_scenario(oracle, 3, others)
    mc = circ.MergeCircuit(3, 3, 3, 3, 3)
    defaults3 = circ.MergeCircuit.defaults(3, mp, h2)
    assert defaults3 == circ.MergeCircuit.defaults(3, lambda l, i, d, z: oracle.merkle_proof_with_zero(arr(l), i, d, z), oracle.two_to_one)
    wires, pis = mc.witness([proof], Z, defaults3, rows_fn)
    assert pis == [0, 0, 0, 0] + list(assets.root)
    pc = pkg.ProverCircuit(ctx, mc.data)
    got = pc.prove(wires, pis)
    cs = oracle.batch_from_values(mc.data.constants_sigmas_values(), 3, 4)
    assert (got == oracle.plonk_prove(mc.data, cs, wires, pis)).all() and oracle.plonk_verify(mc.data, cs["cap"], pis, got) == 0
    pc.free()

    # ---- sender2's transaction of block_circuit.rs, built from GPU calls ----
    tx1, tx2 = user_transactions(oracle)
    want = tx2["merges"][0]
    address = (tx2["sender_address"], 0, 0, 0)
    a1, a2 = (u128(305), u128(8012), u128(2053)), (u128(471), u128(8012), u128(1111))
    r3 = [T(r) for r in trees.smt_roots_of_sets(ctx, [(arr([a[1]]), arr([a[2]])) for a in (a1, a2)])]
    r2 = T(trees.smt_root_of_set(ctx, arr([a1[0], a2[0]]), arr(r3)))
    inc2 = proof_of(trees.smt_find_batch(ctx, arr([address]), arr([r2]), arr([address])), 0)
    inc2 = {k: (T(v) if k not in ("siblings", "found", "is_old0") else v) for k, v in inc2.items()}
    assert inc2 == want["diff_inclusion"]
    nonce = Z
    tx_hash = h2(inc2["root"], nonce)
    sib, mroot = trees.get_merkle_proof(ctx, arr([tx_hash]), 0, 2)
    header = dict(want["header"])
    assert T(mroot) == header["deposit_digest"] and T(mp([], 0, 2, Z)[1]) == header["transactions_digest"]
    assert T(mp([Z], 0, 32, Z)[1]) == header["block_headers_digest"]
    names = ("prev_block_hash",) + circ.BlockHashCircuit.FIELDS
    block_hash = T(trees.get_block_hash(ctx, np.array([[1] + [x for nm in names for x in header[nm]]], dtype=np.uint64))[0])
    merge_key = h2(tx_hash, block_hash)
    asset_root = h2(r2, merge_key)
    process = trees.smt_apply_updates(ctx, *empty, arr([merge_key]), arr([asset_root]))[0][0]
    assert process == dict(want["process"], siblings=[tuple(s) for s in want["process"]["siblings"]])
    merge = dict(is_deposit=True, header=header, merkle=(0, tx_hash, [T(s) for s in sib]), diff_inclusion=inc2, process=process,
                 account_inclusion=circ.SmtInclusionCircuit.with_root(Z), nonce=nonce)
    assert merge["merkle"] == want["merkle"]
    # purge: the two merged assets leave the asset tree again (variable layers, then the contract layer and the merge-key layer in order)
    order = [a2, a1]
    r_var = trees.smt_apply_updates_multi(ctx, [(arr([a[1]]), arr([a[2]])) for a in order], [(arr([a[1]]), arr([Z])) for a in order])
    r_con, con_roots = trees.smt_apply_updates(ctx, arr([a1[0], a2[0]]), arr(r3), arr([a[0] for a in order]), arr([T(r[1][1]) for r in r_var]))
    assert T(con_roots[0]) == r2 and T(con_roots[2]) == Z
    top_vals = [h2(T(con_roots[i]), merge_key) for i in (1, 2)]
    r_top, top_roots = trees.smt_apply_updates(ctx, arr([merge_key]), arr([asset_root]), arr([merge_key, merge_key]), arr(top_vals))
    assert T(top_roots[0]) == tuple(tx2["middle_root"]) and T(top_roots[2]) == tuple(tx2["new_root"])
    ins = [(r_top[i], r_con[i], r_var[i][0][0]) for i in range(2)]
    outs_in = [((u128(407), u128(305), u128(8012)), u128(2053)), ((u128(832), u128(471), u128(8012)), u128(1111))]
    o3 = trees.smt_apply_updates_multi(ctx, [empty, empty], [(arr([k[2]]), arr([v])) for k, v in outs_in])
    o2 = trees.smt_apply_updates_multi(ctx, [empty, empty], [(arr([k[1]]), arr([T(o3[i][1][1])])) for i, (k, _) in enumerate(outs_in)])
    o1 = trees.smt_apply_updates_multi(ctx, [empty], [(arr([k[0] for k, _ in outs_in]), arr([T(o2[i][1][1]) for i in range(2)]))])[0]
    assert T(o1[1][-1]) == tuple(tx2["diff_root"])
    outs = [(o1[0][i], o2[i][0][0], o3[i][0][0]) for i in range(2)]
    for got3, want3 in zip(ins + outs, tx2["ins"] + tx2["outs"]):
        for gp, wp in zip(got3, want3):
            assert gp == dict(wp, siblings=[tuple(s) for s in wp["siblings"]])

    c = circ.UserTransactionCircuit()
    defaults = circ.MergeCircuit.defaults(c.constants["log_n_txs"], mp, h2)
    ws, ps = [], []
    w, p = c.witness(tx2["sender_address"], [merge], ins, outs, tx2["nonce"], Z, defaults, rows_fn)
    assert p == (list(Z) + list(tx2["middle_root"]) + list(tx2["new_root"]) + list(tx2["diff_root"]) + list(address)
                 + list(h2(tx2["diff_root"], tx2["nonce"])))
    ws.append(w)
    ps.append(p)
    w, p = c.witness(tx1["sender_address"], [], tx1["ins"], tx1["outs"], tx1["nonce"], tx1["old_root"], defaults, rows_fn)
    ws.append(w)
    ps.append(p)
    w, p = c.witness(0, [], [], [], Z, Z, defaults, rows_fn)
    assert p[20:] == [4330397376401421145, 14124799381142128323, 8742572140681234676, 14345658006221440202]
    ws.append(w)
    ps.append(p)
    import torch

    pc = pkg.ProverCircuit(ctx, c.data)
    out = pc.prove_batch(torch.stack([ctx.to_device(w) for w in ws]).contiguous(), ps)
    cs = oracle.batch_from_values(c.data.constants_sigmas_values(), 3, 4)
    assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
    for i in range(3):
        assert (out[i] == oracle.plonk_prove(c.data, cs, ws[i], ps[i])).all(), i
        assert oracle.plonk_verify(c.data, cs["cap"], ps[i], out[i]) == 0, i
    assert (out[0] == pc.prove(ws[0], ps[0])).all()
    assert oracle.plonk_verify(c.data, cs["cap"], ps[1], out[0]) != 0
    pc.free()
