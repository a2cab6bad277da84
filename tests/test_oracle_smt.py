"""Oracle for the reference's native tree hashing against the reference's goldens and its own tests'
properties.  CPU only."""
import numpy as np
import pytest

P = 0xFFFFFFFF00000001


def u128_key(x):
    return [x & (2**64 - 1), x >> 64, 0, 0]


def h4(x):
    return [int(t) for t in x]


def test_tx_hash_goldens_through_incremental_layered_tree(oracle, kat):
    """Rebuild both tx hashes the way src/bin/block_circuit.rs does: LayeredLayered `set` on an empty diff tree, then
    tx_hash = two_to_one(diff_root, nonce) (src/transaction/gadgets/purge/mod.rs:296)."""
    from oracle.smt_incremental import LayeredLayeredSparseMerkleTree

    t = kat["transactions"]
    tree = LayeredLayeredSparseMerkleTree()
    for e in t["diff_tree_entries"]:
        k1, k2, k3 = (h4(np.array([k, 0, 0, 0], dtype=np.uint64)) for k in e["key"])
        tree.set(k1, k2, k3, [e["value"], 0, 0, 0])
    for nonce, want in zip(t["nonces"], t["tx_hashes"]):
        assert oracle.hashout_to_hex(oracle.two_to_one(tree.root, nonce)) == want


def test_zero_hash_chain_is_default_merkle_proof(oracle, kat):
    """BlockHeader::new: block_headers_digest = get_merkle_proof(&[], 0, 32).root (src/transaction/block_header.rs:139)"""
    z = kat["zero_hash_chain"]
    sib, root = oracle.merkle_proof_with_zero(np.zeros((0, 4), dtype=np.uint64), 0, 32)
    assert [oracle.hashout_to_hex(s) for s in sib] == z["siblings"]
    assert oracle.hashout_to_hex(root) == z["block_headers_digest"]
    # BlockDetail::new(N_TXS = 4) passes 4 as the depth: the default deposit tree is 0-filled (-> H_4), the default
    # transactions tree is filled with the default tx hash H_1 = two_to_one(0, 0) (-> H_5)   (block_header.rs:127-140)
    empty = np.zeros((0, 4), dtype=np.uint64)
    _, r4 = oracle.merkle_proof_with_zero(empty, 0, 4)
    assert oracle.hashout_to_hex(r4) == z["deposit_digest"]
    _, r5 = oracle.merkle_proof_with_zero(empty, 0, 4, zero=oracle.hashout_from_hex(z["siblings"][1]))
    assert oracle.hashout_to_hex(r5) == z["transactions_digest"]


def test_merkle_proof_properties_of_reference_test(oracle):
    """src/merkle_tree/tree.rs:130-172: siblings do not change when the index-th leaf changes, and get_merkle_root of the
    new leaf equals the root of the rebuilt tree; :174-196: too many leaves for the depth panics."""
    leaves = np.zeros((6, 4), dtype=np.uint64)
    leaves[:, 0] = [0, 10, 20, 30, 40, 0]
    index = 5
    sib, old_root = oracle.merkle_proof_with_zero(leaves, index, 10)
    assert (oracle.merkle_root(index, leaves[index], sib) == old_root).all()
    new_leaf = np.array([50, 0, 0, 0], dtype=np.uint64)
    new_root = oracle.merkle_root(index, new_leaf, sib)
    leaves[index] = new_leaf
    sib2, root2 = oracle.merkle_proof_with_zero(leaves, index, 10)
    assert (sib == sib2).all() and (new_root == root2).all()
    with pytest.raises(AssertionError):
        oracle.merkle_proof_with_zero(leaves, index, 2)


@pytest.mark.parametrize("n,key_bits", [(1, 256), (2, 256), (40, 256), (60, 6), (33, 9)])
def test_set_root_equals_incremental_root(oracle, n, key_bits):
    """the root is a function of the entry SET: inserts in any order, updates and removals lead to the same root as
    glo_smt_root_of_set.  Short keys (few distinct low bits, equal high limbs) force long shared prefixes."""
    from oracle.smt_incremental import SparseMerkleTree

    rng = np.random.default_rng(n * 1000 + key_bits)
    keys = set()
    while len(keys) < n:
        if key_bits >= 256:
            k = tuple(int(x) for x in rng.integers(0, P, 4, dtype=np.uint64))
        else:  # differ only in bits [100, 100 + key_bits): 100 shared levels, then a dense region
            k = (5, int(rng.integers(0, 1 << key_bits)) << 36, 7, 9)
        keys.add(k)
    keys = sorted(keys)
    vals = [tuple(int(x) for x in rng.integers(1, P, 4, dtype=np.uint64)) for _ in keys]
    tree = SparseMerkleTree()
    order = rng.permutation(n)
    for i in order:
        tree.set(keys[i], vals[i])
    want = oracle.smt_root_of_set(np.array(keys, dtype=np.uint64), np.array(vals, dtype=np.uint64))
    assert tree.root == tuple(int(x) for x in want)
    # update half, remove a quarter: still a function of the surviving set
    for j, i in enumerate(order):
        if j % 2 == 0:
            vals[i] = tuple(int(x) for x in rng.integers(1, P, 4, dtype=np.uint64))
            tree.set(keys[i], vals[i])
        elif j % 4 == 1:
            vals[i] = (0, 0, 0, 0)
            tree.set(keys[i], vals[i])
    want = oracle.smt_root_of_set(np.array(keys, dtype=np.uint64), np.array(vals, dtype=np.uint64))
    assert tree.root == tuple(int(x) for x in want)
    # every surviving key has an inclusion path that climbs to the root
    for k, v in zip(keys, vals):
        f = tree.find(k)
        assert f["found"] == (v != (0, 0, 0, 0))
        if f["found"]:
            sib = np.array(f["siblings"], dtype=np.uint64).reshape(-1, 4)
            got = oracle.smt_root_from_path(k, oracle.smt_leaf_hash(k, v), sib)
            assert tuple(int(x) for x in got) == tree.root


def test_block_hash_shape(oracle):
    """get_block_hash (src/transaction/block_header.rs:157-174) as a composition of two_to_one"""
    hdr = np.arange(29, dtype=np.uint64) + 1
    num, prev, bh, tx, dep, prop, appr, acc = hdr[0], hdr[1:5], hdr[5:9], hdr[9:13], hdr[13:17], hdr[17:21], hdr[21:25], hdr[25:29]
    a = oracle.two_to_one([num, 0, 0, 0], acc)
    b = oracle.two_to_one(dep, tx)
    e = oracle.two_to_one(oracle.two_to_one(a, b), oracle.two_to_one(prop, appr))
    assert (oracle.block_hash(hdr) == oracle.two_to_one(bh, e)).all()


def test_set_root_is_decided_on_canonical_words(oracle):
    """any u64 representative is accepted: k + p is the key k, a value limb p is 0; a key occurring twice is rejected
    whether or not one occurrence is absent"""
    rng = np.random.default_rng(3)
    n = 50
    keys = rng.integers(0, 2**32, (n, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (n, 4), dtype=np.uint64)
    want = oracle.smt_root_of_set(keys, vals)
    kk = keys.copy()
    kk[::2] += np.uint64(P)
    assert (oracle.smt_root_of_set(kk, vals) == want).all()
    v2, v3 = vals.copy(), vals.copy()
    v2[4] = [P, 0, P, P]
    v3[4] = 0
    assert (oracle.smt_root_of_set(keys, v2) == oracle.smt_root_of_set(keys, v3)).all()
    for extra_k, extra_v in ((keys[7], np.zeros(4, dtype=np.uint64)), (keys[7] + np.uint64(P), vals[8])):
        with pytest.raises(ValueError):
            oracle.smt_root_of_set(np.concatenate([keys, extra_k[None]]), np.concatenate([vals, extra_v[None]]))


def _set_cases(rng, n, clustered):
    """keys: spread over 256 bits, or clustered (long common prefixes -> unary chains and deep proofs)"""
    if clustered:
        keys = np.zeros((n, 4), dtype=np.uint64)
        keys[:, 0] = rng.integers(0, 2**6, n, dtype=np.uint64)          # few distinct low bits: deep sharing
        keys[:, 1] = rng.integers(0, 2**64, n, dtype=np.uint64) & np.uint64(0xFF00000000000000)
        keys[:, 3] = rng.integers(0, 2**63, n, dtype=np.uint64)
        keys[:, 2] = np.arange(n, dtype=np.uint64)                       # distinct
    else:
        keys = rng.integers(0, 2**63, (n, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (n, 4), dtype=np.uint64)
    vals[::7] = 0  # absent entries
    return keys, vals


@pytest.mark.parametrize("n,clustered", [(0, False), (1, False), (2, True), (40, False), (300, True), (1500, False)])
def test_proofs_of_set_equal_the_incremental_find(oracle, n, clustered):
    """glo_smt_proofs_of_set (the set-based restatement of `find`, reference src/sparse_merkle_tree/tree.rs:588-690) against the
    reference's own incremental algorithm restated in oracle/smt_incremental.py (a sequence of `set` calls, then `find`): same
    siblings in the same order, same found / is_old0 / other-leaf answers — for present keys, removed keys, keys that end on
    another leaf and keys that end in an empty subtree; every proof climbs back to the root."""
    from oracle.smt_incremental import SparseMerkleTree

    rng = np.random.default_rng(1000 + n)
    keys, vals = _set_cases(rng, n, clustered)
    tree = SparseMerkleTree()
    for k, v in zip(keys, vals):
        tree.set(h4(k), h4(v))
    # queries: every key of the set (present and absent), near misses (one high bit flipped: shares a long prefix), random keys
    near = keys.copy()
    if n:
        near[:, 3] ^= np.uint64(1 << 62)
    queries = np.concatenate([keys, near, rng.integers(0, 2**63, (50, 4), dtype=np.uint64)])
    got = oracle.smt_proofs_of_set(keys, vals, queries)
    assert h4(got["root"]) == list(tree.root) == h4(oracle.smt_root_of_set(keys, vals))
    seen = set()
    for i, q in enumerate(queries):
        want = tree.find(h4(q))
        ns = int(got["n_siblings"][i])
        assert ns == len(want["siblings"]), (i, ns, len(want["siblings"]))
        assert [h4(s) for s in got["siblings"][i, :ns]] == [list(s) for s in want["siblings"]], i
        assert bool(got["found"][i]) == want["found"] and bool(got["is_old0"][i]) == want["is_old0"], i
        assert h4(got["value"][i]) == list(want["value"]), i
        if want["other"] is None:
            assert not got["not_found_key"][i].any() and not got["not_found_value"][i].any()
            kind = "found" if want["found"] else "empty"
        else:
            assert h4(got["not_found_key"][i]) == list(want["other"][0]) and h4(got["not_found_value"][i]) == list(want["other"][1])
            kind = "other"
        seen.add(kind)
        # the proof climbs to the root: from the found leaf, from the other leaf, or from 0^4
        if want["found"]:
            start = oracle.smt_leaf_hash(q, got["value"][i])
        elif want["other"] is not None:
            start = oracle.smt_leaf_hash(got["not_found_key"][i], got["not_found_value"][i])
        else:
            start = np.zeros(4, dtype=np.uint64)
        assert h4(oracle.smt_root_from_path(q, start, got["siblings"][i, :ns])) == list(tree.root), (i, kind)
    if n >= 40:
        assert seen == {"found", "empty", "other"}, seen
    with pytest.raises(ValueError):
        oracle.smt_proofs_of_set(np.concatenate([keys, keys[:1]]) if n else np.zeros((2, 4), dtype=np.uint64),
                                 np.concatenate([vals, vals[:1]]) if n else np.ones((2, 4), dtype=np.uint64), queries[:1])


def test_process_proofs_pass_the_reference_check_and_tampering_does_not(oracle):
    """oracle/smt_incremental.py::set_with_proof (what SparseMerkleTree::set returns: reference tree.rs:174-559) against
    verify_process_proof (the reference's own verify_smt_process_proof, proof/process.rs:153-370, restated): 600 random `set`s on
    spread and on 6-bit keys — all four kinds occur and verify; a proof with a wrong sibling, root, value or role does not."""
    from oracle import smt_incremental as inc

    rng = np.random.default_rng(31)
    pool = [tuple(int(x) for x in rng.integers(0, 2**63, 4, dtype=np.uint64)) for _ in range(40)] + [(int(x), 0, 0, 0) for x in range(64)]
    tree = inc.SparseMerkleTree()
    kinds = {}
    kept = []
    for _ in range(600):
        key = pool[int(rng.integers(0, len(pool)))]
        value = (0, 0, 0, 0) if rng.random() < 0.35 else tuple(int(x) for x in rng.integers(1, 2**63, 4, dtype=np.uint64))
        p = inc.set_with_proof(tree, key, value)
        inc.verify_process_proof(p)
        kinds[p["fnc"]] = kinds.get(p["fnc"], 0) + 1
        if p["fnc"] != inc.NOOP and p["siblings"]:
            kept.append(p)
    assert set(kinds) == {inc.NOOP, inc.UPDATE, inc.INSERT, inc.DELETE} and min(kinds.values()) > 20, kinds
    bump = lambda h: ((h[0] + 1) % P,) + tuple(h[1:])  # noqa: E731
    for p in kept[::25]:
        # the value of the leaf that is written (update / insert) or removed (delete: reported as "old"); the OTHER value of an
        # is_old0 proof is ignored by the reference's state machine (status OldIsZero), so it is not tampered with here
        written = "old_value" if p["fnc"] == inc.DELETE else "new_value"
        for field in ("old_root", "new_root", written):
            bad = dict(p, **{field: bump(p[field])})
            with pytest.raises(AssertionError):
                inc.verify_process_proof(bad)
        bad = dict(p, siblings=[bump(p["siblings"][0])] + list(p["siblings"][1:]))
        with pytest.raises(AssertionError):
            inc.verify_process_proof(bad)
