"""GPU parity for the reference's native tree hashing through the C-ABI: every batched kernel against
the oracle on the same inputs, the reference's own goldens (tx hashes, zero-hash chain, default header digests), the
property test of src/merkle_tree/tree.rs:130-172, ragged / empty / single-item batches and both kernel shapes
(16 lanes per item up to 8192 items, one lane per item above)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g

    return g.load_package()


@pytest.fixture(scope="module")
def trees(pkg):
    import importlib

    return importlib.import_module(pkg.__name__ + ".trees")


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def rand(rng, *shape):
    return rng.integers(0, P, shape, dtype=np.uint64)


@pytest.mark.parametrize("n", [0, 1, 5, 300, 9000])
def test_two_to_one_batch(trees, ctx, oracle, n):
    rng = np.random.default_rng(n)
    l, r = rand(rng, n, 4), rand(rng, n, 4)
    got = trees.two_to_one(ctx, l, r)
    assert got.shape == (n, 4)
    step = max(1, n // 64)
    for i in list(range(0, n, step)) + ([n - 1] if n else []):
        assert (got[i] == oracle.two_to_one(l[i], r[i])).all()


@pytest.mark.parametrize("mode", ["no_pad", "pad", "or_noop"])
@pytest.mark.parametrize("length", [0, 1, 3, 4, 5, 8, 9, 10, 11, 12, 17, 22, 23, 135])
def test_hash_batch_modes(trees, ctx, oracle, mode, length):
    rng = np.random.default_rng(length)
    for n in (3, 8200):
        x = rand(rng, n, length)
        x[0, : min(length, 2)] = 2**64 - 1  # non-canonical representatives are accepted
        got = getattr(trees, "hash_" + mode)(ctx, x)
        ref = getattr(oracle, "hash_" + mode)
        for i in (0, 1, n - 1):
            assert (got[i] == ref(x[i])).all(), (mode, length, n, i)


def test_smt_leaf_hash_and_goldens(trees, ctx, oracle, kat):
    rng = np.random.default_rng(1)
    for n in (1, 100, 8300):
        k, v = rand(rng, n, 4), rand(rng, n, 4)
        got = trees.smt_leaf_hash(ctx, k, v)
        for i in (0, n // 2, n - 1):
            assert (got[i] == oracle.smt_leaf_hash(k[i], v[i])).all()
    # sender addresses = two_to_one(sk, sk) (src/zkdsa/account.rs:164-166)
    sk = np.array(kat["accounts"]["private_keys"], dtype=np.uint64)
    addr = trees.two_to_one(ctx, sk, sk)
    assert [oracle.hashout_to_hex(a) for a in addr] == kat["accounts"]["sender_addresses"]


def test_tx_hash_goldens_on_gpu(trees, ctx, oracle, kat):
    """both tx hashes of test_cases/block1_info.json from GPU calls only: three stacked one-entry trees (the root of a
    one-entry tree is its leaf hash), the two-entry top tree through smt_root_of_set, then two_to_one with the nonce"""
    t = kat["transactions"]
    k1s, roots2 = [], []
    for e in t["diff_tree_entries"]:
        k1, k2, k3 = e["key"]
        node = np.array([[e["value"], 0, 0, 0]], dtype=np.uint64)
        for k in (k3, k2):
            node = trees.smt_leaf_hash(ctx, np.array([[k, 0, 0, 0]], dtype=np.uint64), node)
        k1s.append([k1, 0, 0, 0])
        roots2.append(node[0])
    diff_root = trees.smt_root_of_set(ctx, np.array(k1s, dtype=np.uint64), np.array(roots2, dtype=np.uint64))
    nonces = np.array(t["nonces"], dtype=np.uint64)
    tx = trees.two_to_one(ctx, np.tile(diff_root, (2, 1)), nonces)
    assert [oracle.hashout_to_hex(h) for h in tx] == t["tx_hashes"]


def test_zero_chain_and_default_header_digests(trees, ctx, oracle, kat):
    z = kat["zero_hash_chain"]
    empty = np.zeros((0, 4), dtype=np.uint64)
    sib, root = trees.get_merkle_proof(ctx, empty, 0, 32)
    assert [oracle.hashout_to_hex(s) for s in sib] == z["siblings"]
    assert oracle.hashout_to_hex(root) == z["block_headers_digest"]
    _, r4 = trees.get_merkle_proof(ctx, empty, 0, 4)
    assert oracle.hashout_to_hex(r4) == z["deposit_digest"]
    _, r5 = trees.get_merkle_proof_with_zero(ctx, empty, 0, 4, zero=oracle.hashout_from_hex(z["siblings"][1]))
    assert oracle.hashout_to_hex(r5) == z["transactions_digest"]


@pytest.mark.parametrize("n_leaves,depth", [(0, 0), (1, 0), (1, 1), (2, 1), (6, 10), (8, 3), (33, 32), (5000, 13), (70000, 20)])
def test_merkle_proof_with_zero_matches_oracle(trees, ctx, oracle, n_leaves, depth):
    rng = np.random.default_rng(n_leaves + depth)
    leaves = rand(rng, n_leaves, 4)
    zero = rand(rng, 4)
    for index in sorted({0, max(n_leaves, 1) - 1, max(n_leaves, 1) // 2}):
        sib, root = trees.get_merkle_proof_with_zero(ctx, leaves, index, depth, zero)
        sib_o, root_o = oracle.merkle_proof_with_zero(leaves, index, depth, zero)
        assert (sib == sib_o).all() and (root == root_o).all()
        if n_leaves:
            assert (trees.get_merkle_root(ctx, [index], leaves[index][None], sib[None])[0] == root).all()


def test_merkle_proof_reference_property_and_asserts(pkg, trees, ctx):
    """src/merkle_tree/tree.rs:130-196"""
    leaves = np.zeros((6, 4), dtype=np.uint64)
    leaves[:, 0] = [0, 10, 20, 30, 40, 0]
    index = 5
    sib, _ = trees.get_merkle_proof(ctx, leaves, index, 10)
    new_leaf = np.array([50, 0, 0, 0], dtype=np.uint64)
    new_root = trees.get_merkle_root(ctx, [index], new_leaf[None], sib[None])[0]
    leaves[index] = new_leaf
    sib2, root2 = trees.get_merkle_proof(ctx, leaves, index, 10)
    assert (sib == sib2).all() and (new_root == root2).all()
    with pytest.raises(pkg.GlproverError) as e:  # should_panic in the reference: 6 leaves do not fit depth 2
        trees.get_merkle_proof(ctx, leaves, index, 2)
    assert e.value.status == -1
    with pytest.raises(pkg.GlproverError):  # index out of range
        trees.get_merkle_proof(ctx, leaves, 6, 10)


@pytest.mark.parametrize("n,depth", [(1, 0), (7, 5), (100, 32), (9000, 12)])
def test_merkle_root_batch(trees, ctx, oracle, n, depth):
    rng = np.random.default_rng(n)
    idx = rng.integers(0, 1 << min(depth, 62), n, dtype=np.uint64) if depth else np.zeros(n, dtype=np.uint64)
    val, sib = rand(rng, n, 4), rand(rng, n, depth, 4)
    got = trees.get_merkle_root(ctx, idx, val, sib)
    for i in sorted({0, n // 3, n - 1}):
        assert (got[i] == oracle.merkle_root(int(idx[i]), val[i], sib[i])).all()


@pytest.mark.parametrize("n", [1, 50, 8500])
def test_smt_roots_from_ragged_paths(trees, ctx, oracle, n):
    rng = np.random.default_rng(n)
    keys, starts = rand(rng, n, 4), rand(rng, n, 4)
    lens = rng.integers(0, 40, n)
    lens[0] = 0
    if n > 1:
        lens[1] = 200  # deep path: key bits from all four limbs
    sibs = [rand(rng, int(m), 4) for m in lens]
    got = trees.smt_roots_from_paths(ctx, keys, starts, sibs)
    for i in sorted({0, min(1, n - 1), n // 2, n - 1}):
        assert (got[i] == oracle.smt_root_from_path(keys[i], starts[i], sibs[i])).all()


@pytest.mark.parametrize("n", [1, 2, 64, 9000])
def test_block_hash_batch(trees, ctx, oracle, n):
    rng = np.random.default_rng(n)
    hdr = rand(rng, n, 29)
    hdr[:, 0] = rng.integers(0, 2**32, n, dtype=np.uint64)
    got = trees.get_block_hash(ctx, hdr)
    for i in sorted({0, n // 2, n - 1}):
        assert (got[i] == oracle.block_hash(hdr[i])).all()


@pytest.mark.parametrize("n,kind", [(0, "rand"), (1, "rand"), (2, "rand"), (3, "prefix"), (100, "rand"), (257, "prefix"),
                                    (5000, "dense"), (20000, "rand"), (20000, "prefix")])
def test_smt_root_of_set_matches_oracle(pkg, trees, ctx, oracle, n, kind):
    rng = np.random.default_rng(n)
    if kind == "rand":
        keys = rand(rng, n, 4)
    elif kind == "prefix":  # keys agree on the first 100 bits and differ in a narrow window: long unary chains
        keys = np.zeros((n, 4), dtype=np.uint64)
        keys[:, 0] = 5
        keys[:, 1] = rng.choice(1 << 20, n, replace=False).astype(np.uint64) << np.uint64(36)
        keys[:, 2], keys[:, 3] = 7, 9
    else:  # dense: consecutive integers in limb 0 (complete subtrees), plus far-away outliers in limb 3
        keys = np.zeros((n, 4), dtype=np.uint64)
        keys[:, 0] = np.arange(n, dtype=np.uint64)
        keys[n // 2:, 3] = rng.integers(1, 4, n - n // 2, dtype=np.uint64) << np.uint64(61)
    vals = rand(rng, n, 4)
    if n > 4:
        vals[3] = 0  # absent entry
    perm = rng.permutation(n)
    got = trees.smt_root_of_set(ctx, keys[perm], vals[perm])
    assert (got == oracle.smt_root_of_set(keys, vals)).all()
    if n >= 2:
        dup_k = np.concatenate([keys, keys[:1]])
        dup_v = np.concatenate([vals, vals[1:2]])
        with pytest.raises(pkg.GlproverError):
            trees.smt_root_of_set(ctx, dup_k, dup_v)


def test_smt_root_of_set_equals_incremental_sets(trees, ctx, oracle):
    """the GPU bulk root equals the root the reference's incremental walks reach (restated in oracle/smt_incremental.py)"""
    from oracle.smt_incremental import SparseMerkleTree

    rng = np.random.default_rng(7)
    n = 48
    keys = np.zeros((n, 4), dtype=np.uint64)
    keys[:, 0] = rng.choice(256, n, replace=False)  # 8-bit keys: deep shared prefixes among 48 of 256
    vals = rand(rng, n, 4)
    tree = SparseMerkleTree()
    for i in rng.permutation(n):
        tree.set(keys[i], vals[i])
    assert tuple(int(x) for x in trees.smt_root_of_set(ctx, keys, vals)) == tree.root
    for i in range(0, n, 3):
        vals[i] = 0
        tree.set(keys[i], vals[i])
    assert tuple(int(x) for x in trees.smt_root_of_set(ctx, keys, vals)) == tree.root


def test_smt_root_of_set_noncanonical_and_absent_duplicates(pkg, trees, ctx, oracle):
    """inputs may be any u64 representative (include/glprover.h): a value limb equal to p is a zero (absent entry), a key
    given as k + p is the same key; a key that occurs twice is rejected even when one occurrence is absent"""
    P = np.uint64(pkg.P)
    rng = np.random.default_rng(11)
    n = 300
    keys = rng.integers(0, 2**32, (n, 4), dtype=np.uint64)  # small limbs: k + p still fits in 64 bits
    vals = rand(rng, n, 4)
    want = oracle.smt_root_of_set(keys, vals)
    kk, vv = keys.copy(), vals.copy()
    kk[::3] += P  # non-canonical representatives of the same keys
    vv[vv < np.uint64(2**32)] += P
    assert (trees.smt_root_of_set(ctx, kk, vv) == want).all()
    # value (p, p, 0, p) is 0^4: the entry is absent
    v2 = vals.copy()
    v2[5] = [P, P, 0, P]
    v3 = vals.copy()
    v3[5] = 0
    assert (trees.smt_root_of_set(ctx, keys, v2) == oracle.smt_root_of_set(keys, v3)).all()
    # (present, absent) pair with the same key; and the same key under two representatives
    for extra_k, extra_v in ((keys[7], np.zeros(4, dtype=np.uint64)), (keys[7] + P, vals[8])):
        dk = np.concatenate([keys, extra_k[None]])
        dv = np.concatenate([vals, extra_v[None]])
        with pytest.raises(ValueError):
            oracle.smt_root_of_set(dk, dv)
        with pytest.raises(pkg.GlproverError):
            trees.smt_root_of_set(ctx, dk, dv)


# ------------------------------------------------------------------ batched `find` (glp_smt_proofs_of_set)
def _key_sets(rng, n, kind):
    if kind == "rand":
        return rand(rng, n, 4)
    if kind == "prefix":  # keys agree on the first 100 bits and differ in a narrow window: long unary chains
        keys = np.zeros((n, 4), dtype=np.uint64)
        keys[:, 0] = 5
        keys[:, 1] = rng.choice(1 << 20, n, replace=False).astype(np.uint64) << np.uint64(36)
        keys[:, 2], keys[:, 3] = 7, 9
        return keys
    keys = np.zeros((n, 4), dtype=np.uint64)  # dense: consecutive integers (complete subtrees) plus far-away outliers
    keys[:, 0] = np.arange(n, dtype=np.uint64)
    keys[n // 2:, 3] = rng.integers(1, 4, n - n // 2, dtype=np.uint64) << np.uint64(61)
    return keys


def _queries(rng, keys, m):
    """present keys, removed keys, near misses (a high bit flipped, a bit inside the shared prefix flipped) and random keys"""
    n = keys.shape[0]
    parts = [rand(rng, m // 4 + 1, 4)]
    if n:
        pick = keys[rng.integers(0, n, m // 4 + 1)]
        hi = pick.copy()
        hi[:, 3] ^= np.uint64(1 << 62)
        lo = keys[rng.integers(0, n, m // 4 + 1)].copy()
        lo[:, 0] ^= np.uint64(1) << rng.integers(0, 64, lo.shape[0]).astype(np.uint64)
        mid = keys[rng.integers(0, n, m // 4 + 1)].copy()
        mid[:, 1] ^= np.uint64(1) << rng.integers(0, 64, mid.shape[0]).astype(np.uint64)
        parts += [pick, hi, lo, mid]
    return np.concatenate(parts)[:max(m, 1)]


@pytest.mark.parametrize("n,kind,m", [(0, "rand", 5), (1, "rand", 9), (2, "prefix", 16), (3, "prefix", 40), (100, "rand", 300),
                                      (257, "prefix", 500), (5000, "dense", 2000), (20000, "prefix", 3000),
                                      (1 << 16, "rand", 1 << 12)])
def test_smt_find_batch_matches_oracle(pkg, trees, ctx, oracle, n, kind, m):
    """glp_smt_proofs_of_set: every proof (siblings in the reference's order incl. the 0^4 siblings of unary nodes, found /
    is_old0, value, the other leaf) equals the oracle's `find` (oracle/smt.c: glo_smt_proofs_of_set, itself checked against the
    reference's incremental algorithm in tests/test_oracle_smt.py) — up to 2^16 entries and 2^12 queries — and every proof climbs
    back to the root through the GPU's own path kernel (glp_smt_root_batch_dev)."""
    rng = np.random.default_rng(n + 1)
    keys = _key_sets(rng, n, kind)
    vals = rand(rng, n, 4)
    if n > 4:
        vals[3::11] = 0  # absent entries: their keys are queried too
    q = _queries(rng, keys, m)
    got = trees.smt_find_batch(ctx, keys, vals, q)
    want = oracle.smt_proofs_of_set(keys, vals, q)
    assert (got.root == want["root"]).all()
    assert (got.n_siblings == want["n_siblings"]).all()
    assert (got.found == want["found"]).all() and (got.is_old0 == want["is_old0"]).all()
    assert (got.value == want["value"]).all()
    assert (got.not_found_key == want["not_found_key"]).all() and (got.not_found_value == want["not_found_value"]).all()
    width = got.padded_siblings().shape[1]
    assert width == max(1, int(want["n_siblings"].max()))
    assert (got.padded_siblings() == want["siblings"][:, :width]).all(), "sibling lists differ from the oracle"
    if n >= 100:
        kinds = {("found" if f else "empty" if z else "other") for f, z in zip(got.found, got.is_old0)}
        assert {"found", "other"} <= kinds and (kind != "prefix" or "empty" in kinds), kinds  # prefix: queries leave the long unary chain
    # climb: from the found leaf, from the other leaf, or from 0^4 — on the GPU, ragged lengths
    start = np.zeros((len(got), 4), dtype=np.uint64)
    f = got.found
    other = ~got.found & ~got.is_old0
    if f.any():
        start[f] = trees.smt_leaf_hash(ctx, q[f], got.value[f])
    if other.any():
        start[other] = trees.smt_leaf_hash(ctx, got.not_found_key[other], got.not_found_value[other])
    roots = trees.smt_roots_from_paths(ctx, q, start, [got.siblings(i) for i in range(len(got))])
    assert (roots == got.root).all(), "a proof does not climb to the root"
    # a proof longer than the caller's buffer is an error, not a truncation; duplicates are rejected as by the root builder
    if int(got.n_siblings.max()) > 1:
        with pytest.raises(pkg.GlproverError) as e:
            trees.smt_find_batch(ctx, keys, vals, q, max_siblings=int(got.n_siblings.max()) - 1)
        assert e.value.status == -1
    if n >= 2:
        with pytest.raises(pkg.GlproverError):
            trees.smt_find_batch(ctx, np.concatenate([keys, keys[:1]]), np.concatenate([vals, vals[1:2]]), q[:1])


def test_smt_find_batch_equals_the_incremental_find(trees, ctx, oracle):
    """against the reference's own algorithm (a sequence of `set`s, then `find`: oracle/smt_incremental.py) on 8-bit keys (deep
    shared prefixes), with non-canonical representatives of keys and queries"""
    from oracle.smt_incremental import SparseMerkleTree

    rng = np.random.default_rng(21)
    n = 60
    keys = np.zeros((n, 4), dtype=np.uint64)
    keys[:, 0] = rng.choice(256, n, replace=False)
    vals = rand(rng, n, 4)
    tree = SparseMerkleTree()
    for i in rng.permutation(n):
        tree.set(keys[i], vals[i])
    for i in range(0, n, 4):
        vals[i] = 0
        tree.set(keys[i], vals[i])
    q = np.zeros((256, 4), dtype=np.uint64)
    q[:, 0] = np.arange(256)
    got = trees.smt_find_batch(ctx, keys + np.uint64(P) * (keys < 2**32), vals, q + np.uint64(P))
    assert tuple(int(x) for x in got.root) == tree.root
    for i in range(256):
        w = tree.find(q[i])
        assert [tuple(int(x) for x in s) for s in got.siblings(i)] == [tuple(s) for s in w["siblings"]], i
        assert bool(got.found[i]) == w["found"] and bool(got.is_old0[i]) == w["is_old0"]
        assert tuple(int(x) for x in got.value[i]) == tuple(w["value"])
        if w["other"] is not None:
            assert tuple(int(x) for x in got.not_found_key[i]) == tuple(w["other"][0])
            assert tuple(int(x) for x in got.not_found_value[i]) == tuple(w["other"][1])


# ------------------------------------------------------------------ sequential `set`s as one batch (glp_smt_apply_updates)
def _update_stream(rng, keys, k, fresh):
    """k updates: existing keys (new value or removal), keys removed earlier (re-insert), fresh keys (insert), absent keys set
    to zero (no-op); some keys several times"""
    pool = [tuple(int(x) for x in r) for r in keys] + fresh
    out_k, out_v = [], []
    for _ in range(k):
        key = pool[int(rng.integers(0, len(pool)))] if pool else tuple(int(x) for x in rng.integers(0, 2**63, 4, dtype=np.uint64))
        value = (0, 0, 0, 0) if rng.random() < 0.3 else tuple(int(x) for x in rng.integers(1, 2**63, 4, dtype=np.uint64))
        out_k.append(key)
        out_v.append(value)
    return np.array(out_k, dtype=np.uint64).reshape(-1, 4), np.array(out_v, dtype=np.uint64).reshape(-1, 4)


@pytest.mark.parametrize("n,kind,k", [(0, "rand", 12), (1, "rand", 20), (2, "prefix", 30), (60, "small", 400), (300, "prefix", 300),
                                      (1000, "rand", 600), (3000, "dense", 1000)])
def test_smt_apply_updates_equals_sequential_sets(pkg, trees, ctx, oracle, n, kind, k):
    """glp_smt_apply_updates — k `SparseMerkleTree::set` calls in order as ONE batch (base tree built on the device, the walks replayed
    structurally on the host, all hashing as levelled launches) — against the reference's own sequential algorithm restated in
    oracle/smt_incremental.py: every SparseMerkleProcessProof (old / new root, keys, values, the trimmed sibling list, is_old0,
    fnc) is equal field by field, passes the reference's own check (verify_smt_process_proof, proof/process.rs:153-244, restated),
    all four kinds occur, and the final root is the root of the final set (glp_smt_root_of_set and the oracle)."""
    from oracle import smt_incremental as inc

    rng = np.random.default_rng(7000 + n)
    if kind == "small":  # 8-bit keys: deep shared prefixes, leaves moving up and down all the time
        keys = np.zeros((n, 4), dtype=np.uint64)
        keys[:, 0] = rng.choice(256, n, replace=False)
        fresh = [(int(x), 0, 0, 0) for x in range(256)]
    else:
        keys = _key_sets(rng, n, kind)
        fresh = [tuple(int(x) for x in rng.integers(0, 2**63, 4, dtype=np.uint64)) for _ in range(max(4, k // 4))]
        if n:  # near misses of present keys: inserts that push an existing leaf down
            near = keys[rng.integers(0, n, max(4, k // 8))].copy()
            near[:, 3] ^= np.uint64(1 << 61)
            fresh += [tuple(int(x) for x in r) for r in near]
    vals = rand(rng, n, 4)
    if n > 4:
        vals[2::9] = 0  # absent base entries
    uk, uv = _update_stream(rng, keys, k, fresh)
    proofs, roots = trees.smt_apply_updates(ctx, keys, vals, uk, uv)
    tree = inc.SparseMerkleTree()
    for key, value in zip(keys, vals):
        tree.set(key, value)
    assert tuple(int(x) for x in roots[0]) == tree.root
    kinds = set()
    for i in range(k):
        want = inc.set_with_proof(tree, uk[i], uv[i])
        got = proofs[i]
        for f in ("fnc", "is_old0", "old_root", "new_root", "old_key", "old_value", "new_key", "new_value"):
            assert got[f] == want[f], (i, f, got[f], want[f])
        assert got["siblings"] == [tuple(s) for s in want["siblings"]], (i, "siblings")
        kinds.add(got["fnc"])
        if i % 7 == 0 or k <= 40:
            inc.verify_process_proof(got)  # the reference's own check of a process proof
        assert tuple(int(x) for x in roots[i + 1]) == tree.root
    if k >= 300:
        assert kinds == {inc.NOOP, inc.UPDATE, inc.INSERT, inc.DELETE}
    # the final root is the root of the final SET
    final = {tuple(int(x) for x in key): tuple(int(x) for x in value) for key, value in zip(keys, vals)}
    for key, value in zip(uk, uv):
        final[tuple(int(x) for x in key)] = tuple(int(x) for x in value)
    fk = np.array(list(final.keys()), dtype=np.uint64).reshape(-1, 4)
    fv = np.array(list(final.values()), dtype=np.uint64).reshape(-1, 4)
    assert tuple(int(x) for x in trees.smt_root_of_set(ctx, fk, fv)) == tree.root


def test_smt_apply_updates_edge_cases(pkg, trees, ctx, oracle):
    """non-canonical representatives of update keys / values (k + p is the same key, a value limb equal to p is zero), an empty
    update list, a sibling buffer that is too short (an error, not a truncation), a duplicate key in the BASE set"""
    import ctypes

    from oracle import smt_incremental as inc

    Pn = np.uint64(pkg.P)
    rng = np.random.default_rng(99)
    keys = rng.integers(0, 2**32, (50, 4), dtype=np.uint64)
    vals = rand(rng, 50, 4)
    uk = np.concatenate([keys[:6] + Pn, rng.integers(0, 2**32, (6, 4), dtype=np.uint64)])
    uv = rand(rng, 12, 4)
    uv[1] = [Pn, 0, Pn, 0]  # = 0^4: removes keys[1]
    uv[7] = 0               # no-op
    proofs, roots = trees.smt_apply_updates(ctx, keys, vals, uk, uv)
    tree = inc.SparseMerkleTree()
    for a, b in zip(keys, vals):
        tree.set(a, b)
    for i in range(12):
        want = inc.set_with_proof(tree, [int(x) % pkg.P for x in uk[i]], [int(x) % pkg.P for x in uv[i]])
        assert proofs[i] == {**want, "siblings": [tuple(s) for s in want["siblings"]]}, i
    assert proofs[1]["fnc"] == inc.DELETE and proofs[7]["fnc"] == inc.NOOP
    none, r0 = trees.smt_apply_updates(ctx, keys, vals, np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64))
    assert none == [] and (r0[0] == trees.smt_root_of_set(ctx, keys, vals)).all()
    L = pkg.lib()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    sib = np.zeros((12, 1, 4), dtype=np.uint64)
    st = L.glp_smt_apply_updates(ctx._h, keys.ctypes.data_as(u64p), vals.ctypes.data_as(u64p), ctypes.c_size_t(50),
                                 uk.ctypes.data_as(u64p), uv.ctypes.data_as(u64p), ctypes.c_size_t(12), ctypes.c_size_t(1),
                                 sib.ctypes.data_as(u64p), None, None, None, None, None, None, None, None)
    assert st == -1
    with pytest.raises(pkg.GlproverError):
        trees.smt_apply_updates(ctx, np.concatenate([keys, keys[:1]]), np.concatenate([vals, vals[1:2]]), uk, uv)


# ------------------------------------------------------------------ many small trees per call (glp_smt_roots_of_sets)
def test_smt_roots_of_sets_matches_oracle(pkg, trees, ctx, oracle):
    """700 independent sets of 0 .. 40 entries (random, clustered and 6-bit keys, absent entries, one large set of 5 000) in ONE
    call: every root equals the oracle's root of that set — and glp_smt_root_of_set's; a duplicate key inside a set is refused"""
    rng = np.random.default_rng(55)
    sets = []
    for s in range(700):
        n = int(rng.integers(0, 41))
        kind = s % 3
        if kind == 0:
            keys = rand(rng, n, 4)
        elif kind == 1:
            keys = np.zeros((n, 4), dtype=np.uint64)
            keys[:, 0] = rng.choice(64, n, replace=False)
        else:
            keys = np.zeros((n, 4), dtype=np.uint64)
            keys[:, 0] = 5
            keys[:, 2] = rng.choice(1 << 12, n, replace=False).astype(np.uint64) << np.uint64(20)
            keys[:, 3] = 9
        vals = rand(rng, n, 4)
        if n > 3:
            vals[1] = 0
        sets.append((keys, vals))
    big_k, big_v = rand(rng, 5000, 4), rand(rng, 5000, 4)
    sets.append((big_k, big_v))
    roots = trees.smt_roots_of_sets(ctx, sets)
    for s, (keys, vals) in enumerate(sets):
        assert (roots[s] == oracle.smt_root_of_set(keys, vals)).all(), s
    assert (roots[-1] == trees.smt_root_of_set(ctx, big_k, big_v)).all()
    assert not roots[[i for i, (k, _) in enumerate(sets) if len(k) == 0]].any()  # empty sets: 0^4
    assert trees.smt_roots_of_sets(ctx, []).shape == (0, 4)
    dup = (np.concatenate([sets[3][0], sets[3][0][:1]]), np.concatenate([sets[3][1], sets[3][1][2:3]]))
    if len(sets[3][0]):
        with pytest.raises(pkg.GlproverError):
            trees.smt_roots_of_sets(ctx, [sets[0], dup])


def test_tx_hash_goldens_through_batched_layered_trees(trees, ctx, oracle, kat):
    """both tx hashes of test_cases/block1_info.json (goldens) from three batched calls: the diff tree of a transaction is a
    LayeredLayeredSparseMerkleTree (src/sparse_merkle_tree/layered_layered_tree.rs:124-160); trees.layered_smt_roots builds every
    bottom tree, then every middle tree, then every top tree of ALL transactions with one glp_smt_roots_of_sets each;
    tx_hash = two_to_one(diff_root, nonce).  Also 300 random layered trees against the incremental oracle."""
    from oracle.smt_incremental import LayeredLayeredSparseMerkleTree

    t = kat["transactions"]
    K = lambda x: (int(x), 0, 0, 0)  # noqa: E731
    entries = [(K(e["key"][0]), K(e["key"][1]), K(e["key"][2]), K(e["value"])) for e in t["diff_tree_entries"]]
    # the two transactions of the fixture share the diff tree and differ in the nonce
    roots = trees.layered_smt_roots(ctx, [entries, entries])
    tx = trees.two_to_one(ctx, roots, np.array(t["nonces"], dtype=np.uint64))
    assert [oracle.hashout_to_hex(h) for h in tx] == t["tx_hashes"]
    rng = np.random.default_rng(8)
    many, want = [], []
    for _ in range(300):
        ref = LayeredLayeredSparseMerkleTree()
        es = {}
        for _ in range(int(rng.integers(1, 7))):
            k1, k2, k3 = (K(rng.integers(0, 3)), K(rng.integers(0, 3)), K(rng.integers(0, 4)))
            es[(k1, k2, k3)] = tuple(int(x) for x in rng.integers(1, 2**63, 4, dtype=np.uint64))
        for (k1, k2, k3), v in es.items():
            ref.set(k1, k2, k3, v)
        many.append([(k1, k2, k3, v) for (k1, k2, k3), v in es.items()])
        want.append(ref.root)
    got = trees.layered_smt_roots(ctx, many)
    assert [tuple(int(x) for x in r) for r in got] == want


def test_user_asset_trees_batched(trees, ctx, oracle):
    """trees.user_asset_roots — the asset trees of 200 users in four batched calls — against the reference's incremental
    UserAssetTree::set (src/transaction/tree/user_asset.rs:99-150, restated in oracle/smt_incremental.py), with purged (zero-amount)
    entries: a contract whose variables are all purged vanishes from its layer; a merge key that was ever touched stays (the
    reference stores two_to_one(0, merge_key) for it)"""
    from oracle.smt_incremental import UserAssetTree

    rng = np.random.default_rng(12)
    K = lambda x: (int(x), 0, 0, 0)  # noqa: E731
    users, want = [], []
    for u in range(200):
        ref = UserAssetTree()
        final = {}
        for _ in range(int(rng.integers(0, 9))):
            mk, ca, vi = K(rng.integers(1, 3)), K(rng.integers(0, 3)), K(rng.integers(0, 3))
            amount = (0, 0, 0, 0) if rng.random() < 0.3 else tuple(int(x) for x in rng.integers(1, 2**40, 4, dtype=np.uint64))
            ref.set(mk, ca, vi, amount)
            final[(mk, ca, vi)] = amount
        users.append([(mk, ca, vi, a) for (mk, ca, vi), a in final.items()])
        want.append(ref.root)
    got = trees.user_asset_roots(ctx, users)
    assert [tuple(int(x) for x in r) for r in got] == want


def _check_block_state_against_oracle(sc, out):
    """every root recomputed by the CPU oracle from the final SETS (the trees are canonical) and the incremental trees"""
    from oracle import oracle as orc
    from oracle.smt_incremental import LayeredLayeredSparseMerkleTree, UserAssetTree

    T = len(sc["senders"])
    for i in range(T):
        ua = UserAssetTree()
        for e in sc["users"][i]:
            ua.set(*e)
        assert tuple(int(x) for x in out["asset_roots"][i]) == ua.root, ("asset root", i)
        ll = LayeredLayeredSparseMerkleTree()
        for e in sc["diffs"][i]:
            ll.set(*e)
        assert tuple(int(x) for x in out["diff_roots"][i]) == ll.root, ("diff root", i)
        assert (out["tx_hashes"][i] == orc.two_to_one(ll.root, sc["nonces"][i])).all()
    wv = sc["world_vals"].copy()
    wv[sc["senders"]] = out["asset_roots"]
    assert (out["world_roots"][0] == orc.smt_root_of_set(sc["addr"], sc["world_vals"])).all()
    assert (out["world_roots"][-1] == orc.smt_root_of_set(sc["addr"], wv)).all()
    lv = sc["latest_vals"].copy()
    lv[sc["senders"]] = 0
    lv[sc["senders"], 0] = sc["block_number"]
    assert (out["latest_roots"][-1] == orc.smt_root_of_set(sc["addr"], lv)).all()
    depth = max(1, int(np.ceil(np.log2(max(2, T)))))
    _, want_digest = orc.merkle_proof_with_zero(out["tx_hashes"], 0, depth, orc.two_to_one([0] * 4, [0] * 4))
    assert (out["tx_digest"] == want_digest).all()
    assert (out["block_hash"] == orc.block_hash(out["header"][0])).all()
    assert all(p["fnc"] == 1 for p in out["world_proofs"]) and all(p["fnc"] == 1 for p in out["latest_proofs"])  # all updates


def test_block_state_pipeline_matches_oracle(pkg, trees, ctx, oracle):
    """tools/block_state_bench.py — a block's host-side state work as GPU batches (user-asset trees after the purge, diff trees and
    tx hashes, T sequential world-state and latest-account `set`s with their process proofs, transactions digest, block hash) — at
    T = 24 transactions over 300 accounts: every root and digest equals the CPU oracle's"""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("block_state_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                    "intmax-zkp-core_amd", "tools", "block_state_bench.py"))
    bsb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bsb)
    sc = bsb.scenario(np.random.default_rng(4), 300, 24)
    out = bsb.run_block(pkg, trees, ctx, sc)
    _check_block_state_against_oracle(sc, out)
    from oracle import smt_incremental as inc

    for p in out["world_proofs"][::5] + out["latest_proofs"][::5]:
        inc.verify_process_proof(p)
    # the same block with the world-state and latest-account trees RESIDENT on the device (the tool's --resident): same roots, same proofs
    sc2 = dict(sc, world_tree=trees.ResidentSparseMerkleTree(ctx, sc["addr"], sc["world_vals"]),
               latest_tree=trees.ResidentSparseMerkleTree(ctx, sc["addr"], sc["latest_vals"]))
    out2 = bsb.run_block(pkg, trees, ctx, sc2)
    _check_block_state_against_oracle(sc, out2)
    assert out2["world_proofs"] == out["world_proofs"] and out2["latest_proofs"] == out["latest_proofs"]
    assert (out2["block_hash"] == out["block_hash"]).all()
    assert (sc2["world_tree"].root() == out["world_roots"][-1]).all() and (sc2["latest_tree"].root() == out["latest_roots"][-1]).all()
    sc2["world_tree"].close()
    sc2["latest_tree"].close()


def test_smt_apply_updates_heavy_churn_across_a_limb_boundary(pkg, trees, ctx, oracle):
    """1 500 updates over a 2^8 key space spread over two limbs (4 bits in limb 0, the rest in limb 1): unary chains cross a limb
    boundary, leaves move up and down all the time, proofs reach 68 siblings — longer than the driver's one-call guess of 48, so the
    lengths-first path of trees.smt_apply_updates runs too.  Every process proof equals the sequential oracle's."""
    from oracle import smt_incremental as inc

    rng = np.random.default_rng(1)
    space, n, k = 256, 60, 1500
    pick = rng.choice(space, n, replace=False)
    keys = np.zeros((n, 4), dtype=np.uint64)
    keys[:, 0], keys[:, 1] = pick & 0xF, pick >> 4
    vals = rand(rng, n, 4)
    p2 = rng.integers(0, space, k)
    uk = np.zeros((k, 4), dtype=np.uint64)
    uk[:, 0], uk[:, 1] = p2 & 0xF, p2 >> 4
    uv = rand(rng, k, 4)
    uv[rng.random(k) < 0.45] = 0
    proofs, roots = trees.smt_apply_updates(ctx, keys, vals, uk, uv)
    tree = inc.SparseMerkleTree()
    for a, b in zip(keys, vals):
        tree.set(a, b)
    for i in range(k):
        want = inc.set_with_proof(tree, uk[i], uv[i])
        want["siblings"] = [tuple(s) for s in want["siblings"]]
        assert proofs[i] == want, i
    assert max(len(p["siblings"]) for p in proofs) > 48


def test_smt_apply_updates_multi_matches_sequential_sets(pkg, trees, ctx, oracle):
    """glp_smt_apply_updates_multi: 400 independent small trees — empty, one-entry, 6-bit-key and random bases of up to 30 entries,
    each with its own sequence of 0 .. 25 `set`s — in ONE call: every tree's process proofs equal the reference's sequential
    algorithm on that tree (oracle/smt_incremental.py), its roots chain, and a sample passes the reference's proof check"""
    from oracle import smt_incremental as inc

    rng = np.random.default_rng(77)
    bases, updates = [], []
    for s in range(400):
        n, k = int(rng.integers(0, 31)) if s % 5 else 0, int(rng.integers(0, 26))
        if s % 3 == 0:
            space = [(int(x), 0, 0, 0) for x in range(64)]
            keys = np.array([space[i] for i in rng.choice(64, min(n, 64), replace=False)], dtype=np.uint64).reshape(-1, 4)
        else:
            keys = rand(rng, n, 4)
            space = [tuple(int(x) for x in r) for r in keys] + [tuple(int(x) for x in rng.integers(0, 2**63, 4, dtype=np.uint64)) for _ in range(8)]
        vals = rand(rng, keys.shape[0], 4)
        uk = np.array([space[int(rng.integers(0, len(space)))] for _ in range(k)], dtype=np.uint64).reshape(-1, 4)
        uv = rand(rng, k, 4)
        uv[rng.random(k) < 0.4] = 0
        bases.append((keys, vals))
        updates.append((uk, uv))
    got = trees.smt_apply_updates_multi(ctx, bases, updates)
    kinds = set()
    for s, ((keys, vals), (uk, uv)) in enumerate(zip(bases, updates)):
        tree = inc.SparseMerkleTree()
        for a, b in zip(keys, vals):
            tree.set(a, b)
        proofs, roots = got[s]
        assert roots.shape == (len(uk) + 1, 4) and tuple(int(x) for x in roots[0]) == tree.root, s
        for i in range(len(uk)):
            want = inc.set_with_proof(tree, uk[i], uv[i])
            want["siblings"] = [tuple(x) for x in want["siblings"]]
            assert proofs[i] == want, (s, i)
            kinds.add(want["fnc"])
            assert tuple(int(x) for x in roots[i + 1]) == tree.root
        if s % 40 == 0:
            for p in proofs:
                inc.verify_process_proof(p)
    assert kinds == {inc.NOOP, inc.UPDATE, inc.INSERT, inc.DELETE}
    assert trees.smt_apply_updates_multi(ctx, [], []) == []
    with pytest.raises(pkg.GlproverError):
        trees.smt_apply_updates_multi(ctx, [(np.ones((2, 4), np.uint64), np.ones((2, 4), np.uint64))], [(np.zeros((0, 4), np.uint64),) * 2])


def test_smt_apply_updates_multi_refuses_offset_tables_not_starting_at_zero(pkg, trees, ctx):
    """the replay records and every output row are indexed by the absolute update / entry index, so an offset table whose
    first entry is not 0 is an argument error (GLP_ERR_INVALID) — not an out-of-bounds read.  Straight through the C ABI."""
    import ctypes

    L = pkg.lib()
    u64p, u32p, szp = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_size_t)
    rng = np.random.default_rng(4)
    keys = rng.integers(1, 2**62, (6, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**62, (6, 4), dtype=np.uint64)
    uk = rng.integers(1, 2**62, (5, 4), dtype=np.uint64)
    uv = rng.integers(1, 2**62, (5, 4), dtype=np.uint64)
    ns, fl = np.zeros(5, dtype=np.uint32), np.zeros(5, dtype=np.uint32)
    roots = np.zeros((5 + 2, 4), dtype=np.uint64)

    def call(soff, uoff):
        soff, uoff = np.asarray(soff, dtype=np.uint64), np.asarray(uoff, dtype=np.uint64)
        return L.glp_smt_apply_updates_multi(ctx._h, keys.ctypes.data_as(u64p), vals.ctypes.data_as(u64p), soff.ctypes.data_as(szp),
                                             uk.ctypes.data_as(u64p), uv.ctypes.data_as(u64p), uoff.ctypes.data_as(szp),
                                             ctypes.c_size_t(2), ctypes.c_size_t(0), None, ns.ctypes.data_as(u32p), fl.ctypes.data_as(u32p),
                                             None, None, None, None, None, roots.ctypes.data_as(u64p))

    assert call([0, 3, 6], [0, 2, 5]) == 0
    assert roots.any()
    assert call([0, 3, 6], [1, 2, 5]) == -1  # GLP_ERR_INVALID
    assert call([1, 3, 6], [0, 2, 5]) == -1
    assert call([0, 3, 6], [0, 3, 2]) == -1  # (not monotone: refused as before)


def test_layered_tree_process_proofs_in_three_calls(pkg, trees, ctx, oracle):
    """The three process proofs LayeredLayeredSparseMerkleTree::set returns (reference layered_layered_tree.rs:124-160: the set in the
    bottom tree, then set(key2, new bottom root) in the middle tree, then set(key1, new middle root) in the top tree) for the
    outputs of 120 transactions, each inserting 1 .. 5 entries into its own EMPTY diff tree — one glp_smt_apply_updates_multi call
    per layer for all transactions together (a lower layer's roots are the next layer's update values).  Every proof of every
    layer equals the one the reference's sequential algorithm gives."""
    from oracle import smt_incremental as inc

    rng = np.random.default_rng(5)
    K = lambda x: (int(x), 0, 0, 0)  # noqa: E731
    txs = [[(K(rng.integers(0, 3)), K(rng.integers(0, 3)), K(rng.integers(0, 50)),
             tuple(int(x) for x in rng.integers(1, 2**40, 4, dtype=np.uint64))) for _ in range(int(rng.integers(1, 6)))] for _ in range(120)]
    # group the sets by the small tree they touch, keeping the order inside each tree
    bottom, b_order = {}, []
    for t, entries in enumerate(txs):
        for j, (k1, k2, k3, v) in enumerate(entries):
            bottom.setdefault((t, k1, k2), []).append((j, k3, v))
            b_order.append((t, j))
    bkeys = list(bottom)
    empty = (np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64))
    r3 = trees.smt_apply_updates_multi(ctx, [empty] * len(bkeys), [([k3 for _, k3, _ in bottom[b]], [v for _, _, v in bottom[b]]) for b in bkeys])
    new_root3 = {}  # (t, j) -> bottom root after that set
    proof3 = {}
    for b, (proofs, roots) in zip(bkeys, r3):
        for (j, _, _), p, r in zip(bottom[b], proofs, roots[1:]):
            new_root3[(b[0], j)] = tuple(int(x) for x in r)
            proof3[(b[0], j)] = p
    middle = {}
    for t, entries in enumerate(txs):
        for j, (k1, k2, k3, v) in enumerate(entries):
            middle.setdefault((t, k1), []).append((j, k2, new_root3[(t, j)]))
    mkeys = list(middle)
    r2 = trees.smt_apply_updates_multi(ctx, [empty] * len(mkeys), [([k2 for _, k2, _ in middle[m]], [v for _, _, v in middle[m]]) for m in mkeys])
    new_root2, proof2 = {}, {}
    for m, (proofs, roots) in zip(mkeys, r2):
        for (j, _, _), p, r in zip(middle[m], proofs, roots[1:]):
            new_root2[(m[0], j)] = tuple(int(x) for x in r)
            proof2[(m[0], j)] = p
    r1 = trees.smt_apply_updates_multi(ctx, [empty] * len(txs), [([k1 for k1, _, _, _ in e], [new_root2[(t, j)] for j in range(len(e))]) for t, e in enumerate(txs)])
    # the reference's way, one set at a time
    for t, entries in enumerate(txs):
        nodes = {}
        top = inc.SparseMerkleTree(nodes)
        for j, (k1, k2, k3, v) in enumerate(entries):
            t2 = inc.SparseMerkleTree(nodes, top.get(k1))
            t3 = inc.SparseMerkleTree(nodes, t2.get(k2))
            w3 = inc.set_with_proof(t3, k3, v)
            w2 = inc.set_with_proof(t2, k2, t3.root)
            w1 = inc.set_with_proof(top, k1, t2.root)
            for want, got in ((w3, proof3[(t, j)]), (w2, proof2[(t, j)]), (w1, r1[t][0][j])):
                want["siblings"] = [tuple(x) for x in want["siblings"]]
                assert got == want, (t, j)
        assert tuple(int(x) for x in r1[t][1][-1]) == top.root
    # and the final roots are what the set-based batched builder gives
    assert [tuple(int(x) for x in r) for r in trees.layered_smt_roots(ctx, txs)] == [tuple(int(x) for x in r1[t][1][-1]) for t in range(len(txs))]


def test_random_instances_of_the_tree_primitives(pkg, ctx, oracle):
    """100 seeded random instances per primitive: the three sponge modes over 0 .. 44-word items, `get_merkle_proof_with_zero` over ragged
    leaf counts (0 .. 1 000), depths up to 39 and random zero hashes, the root and the inclusion / non-inclusion proofs of clustered key
    sets (3- to 64-bit key spaces in one limb) — each equal to the oracle's (300 more per primitive in a sweep of this generator)"""
    import importlib

    trees = importlib.import_module(pkg.__name__ + ".trees")
    bad = []

    def check(ok, what):
        if not ok:
            bad.append(what)

    import os

    for case in range(int(os.environ.get("GLP_SWEEP_CASES", "100"))):  # a soak raises it: GLP_SWEEP_CASES=1000 GLP_SWEEP_SEED0=7
        rng = np.random.default_rng(61000 + int(os.environ.get("GLP_SWEEP_SEED0", "0")) + case)
        # hash_batch: n items of `length` words, three modes
        n, length = int(rng.integers(1, 300)), int(rng.integers(0, 45))
        x = rng.integers(0, 2**64, (n, length), dtype=np.uint64)
        for name, gf, of in (("no_pad", trees.hash_no_pad, oracle.hash_no_pad), ("pad", trees.hash_pad, oracle.hash_pad),
                             ("or_noop", trees.hash_or_noop, oracle.hash_or_noop)):
            got = gf(ctx, x)
            want = np.array([of(r) for r in x[: min(n, 12)]])
            check((np.asarray(got)[: min(n, 12)] == want).all(), "case %d hash_%s n %d len %d" % (61000 + case, name, n, length))
        # merkle proof with zero: ragged leaf counts, depths, indices
        nl = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 64, 100, 257, 1000]))
        depth = int(rng.integers(max(1, int(np.ceil(np.log2(max(nl, 2))))), 40))
        idx = int(rng.integers(0, max(nl, 1)))
        leaves = rng.integers(0, P, (nl, 4), dtype=np.uint64)
        zero = tuple(int(v) for v in rng.integers(0, P, 4, dtype=np.uint64)) if rng.random() < 0.5 else (0, 0, 0, 0)
        sib, root = trees.get_merkle_proof_with_zero(ctx, leaves, idx, depth, zero)
        osib, oroot = oracle.merkle_proof_with_zero(leaves, idx, depth, zero)
        check((np.asarray(sib) == np.asarray(osib)).all() and (np.asarray(root) == np.asarray(oroot)).all(),
              "case %d merkle_proof_with_zero leaves %d depth %d index %d" % (61000 + case, nl, depth, idx))
        # set root / proofs on clustered keys
        m = int(rng.choice([0, 1, 2, 3, 17, 100, 1000]))
        bits = int(rng.choice([3, 8, 20, 64]))
        keys = np.zeros((m, 4), dtype=np.uint64)
        if m:
            space = min(1 << bits, 1 << 62)
            pick = rng.choice(space, m, replace=False) if space >= m and space <= (1 << 24) else rng.integers(0, 2**62, m)
            keys[:, int(rng.integers(0, 4))] = pick
            keys = np.unique(keys, axis=0)
        vals = rng.integers(1, P, (keys.shape[0], 4), dtype=np.uint64)
        check((np.asarray(trees.smt_root_of_set(ctx, keys, vals)) == np.asarray(oracle.smt_root_of_set(keys, vals))).all(),
              "case %d smt_root_of_set m %d bits %d" % (61000 + case, keys.shape[0], bits))
        q = np.concatenate([keys[: min(5, keys.shape[0])], rng.integers(0, 2**20, (4, 4), dtype=np.uint64)])
        pr = trees.smt_find_batch(ctx, keys, vals, q)
        o = oracle.smt_proofs_of_set(keys, vals, q)
        same = (np.asarray(pr.root) == np.asarray(o["root"])).all() and (pr.n_siblings == o["n_siblings"]).all() and (pr.found == o["found"]).all() and (pr.is_old0 == o["is_old0"]).all()
        if same:
            for i in range(q.shape[0]):
                same &= (np.asarray(pr.siblings(i)) == np.asarray(o["siblings"][i][: int(o["n_siblings"][i])])).all()
        check(same, "case %d smt_find_batch m %d" % (61000 + case, keys.shape[0]))
    assert not bad, bad


def test_resident_tree_follows_block_after_block_of_updates(pkg, trees, ctx, oracle):
    """glp_smt_open / glp_smt_update / glp_smt_find / glp_smt_entries: a tree that stays on the device takes 25 successive batches of
    `set`s (updates, inserts, removals, noops; a small clustered key space and a spread one; batches with the same key several times;
    an empty batch; a batch that empties the tree) — every process proof equals the sequential oracle's on the evolving tree, and after
    every batch the root, the entry set and the `find` proofs equal what a freshly built tree of the oracle's entry set gives."""
    from oracle import smt_incremental as inc

    for variant in range(3):
        rng = np.random.default_rng(400 + variant)
        if variant == 0:  # clustered: 2^9 keys over two limbs
            def draw(cnt):
                p = rng.integers(0, 512, cnt)
                a = np.zeros((cnt, 4), dtype=np.uint64)
                a[:, 0], a[:, 1] = p & 0xF, p >> 4
                return a
            n0 = 120
        elif variant == 1:  # spread keys, a larger tree
            def draw(cnt):
                return rand(rng, cnt, 4)
            n0 = 5000
        else:  # starts empty
            def draw(cnt):
                a = np.zeros((cnt, 4), dtype=np.uint64)
                a[:, 2] = rng.integers(0, 64, cnt)
                return a
            n0 = 0
        keys = np.unique(draw(n0), axis=0) if n0 else np.zeros((0, 4), dtype=np.uint64)
        vals = rand(rng, keys.shape[0], 4)
        vals[:, 0] |= 1
        rt = trees.ResidentSparseMerkleTree(ctx, keys, vals)
        tree = inc.SparseMerkleTree()
        cur = {}
        for a, b in zip(keys, vals):
            tree.set(a, b)
            cur[tuple(int(x) for x in a)] = tuple(int(x) for x in b)
        assert (rt.root() == np.asarray(oracle.smt_root_of_set(keys, vals))).all() and len(rt) == keys.shape[0]
        for batch in range(25):
            k = 0 if batch == 7 else int(rng.integers(1, 120))
            if batch == 19 and cur:  # remove everything that is there, each key once
                uk = np.array(sorted(cur), dtype=np.uint64).reshape(-1, 4)
                uv = np.zeros_like(uk)
            else:
                uk = draw(k)
                have = np.array(sorted(cur), dtype=np.uint64).reshape(-1, 4)
                for i in range(k):  # about a third of the updates hit existing keys
                    if have.shape[0] and rng.random() < 0.35:
                        uk[i] = have[int(rng.integers(0, have.shape[0]))]
                uv = rand(rng, uk.shape[0], 4)
                uv[:, 0] |= 1
                uv[rng.random(uk.shape[0]) < 0.3] = 0
            proofs, roots = rt.update(uk, uv)
            assert roots.shape[0] == uk.shape[0] + 1
            for i in range(uk.shape[0]):
                want = inc.set_with_proof(tree, uk[i], uv[i])
                want["siblings"] = [tuple(s) for s in want["siblings"]]
                assert proofs[i] == want, (variant, batch, i)
                kk, vv = tuple(int(x) for x in uk[i]), tuple(int(x) for x in uv[i])
                if any(vv):
                    cur[kk] = vv
                else:
                    cur.pop(kk, None)
            ck = np.array(sorted(cur), dtype=np.uint64).reshape(-1, 4)
            cv = np.array([cur[tuple(int(x) for x in a)] for a in ck], dtype=np.uint64).reshape(-1, 4)
            want_root = np.asarray(oracle.smt_root_of_set(ck, cv))
            assert (rt.root() == want_root).all(), (variant, batch)
            assert len(rt) == len(cur)
            ek, ev = rt.entries()
            assert sorted(map(tuple, np.concatenate([ek, ev], axis=1).tolist())) == sorted(map(tuple, np.concatenate([ck, cv], axis=1).tolist()))
            if batch % 6 == 0:  # find on the resident tree == find on a tree built from the oracle's entry set
                q = np.concatenate([ck[: min(6, ck.shape[0])], draw(5)])
                got, ref = rt.find(q), trees.smt_find_batch(ctx, ck, cv, q)
                assert (got.root == ref.root).all() and (got.n_siblings == ref.n_siblings).all() and (got.found == ref.found).all()
                assert (got.is_old0 == ref.is_old0).all() and (got.value == ref.value).all()
                for i in range(q.shape[0]):
                    assert (np.asarray(got.siblings(i)) == np.asarray(ref.siblings(i))).all()
        rt.close()


def test_resident_tree_error_paths_leave_it_unchanged(pkg, trees, ctx, oracle):
    """a proof longer than max_siblings (GLP_ERR_INVALID) applies nothing; the wrapper's retry then applies the batch once; a duplicate
    key at open is refused; closing twice / a NULL tree is harmless"""
    import ctypes

    L = pkg.lib()
    rng = np.random.default_rng(9)
    # keys that agree in their first 70 bits: every proof has > 64 siblings
    keys = np.zeros((8, 4), dtype=np.uint64)
    keys[:, 0] = 0x0123456789ABCDEF
    keys[:, 1] = (np.arange(8, dtype=np.uint64) << np.uint64(6)) | np.uint64(0x15)
    vals = rand(rng, 8, 4)
    vals[:, 0] |= 1
    rt = trees.ResidentSparseMerkleTree(ctx, keys, vals)
    root0 = rt.root().copy()
    uk, uv = keys[:3].copy(), rand(rng, 3, 4)
    u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
    sib = np.zeros((3, 16, 4), dtype=np.uint64)
    ns, fl = np.zeros(3, dtype=np.uint32), np.zeros(3, dtype=np.uint32)
    st = L.glp_smt_update(rt._h, uk.ctypes.data_as(u64p), uv.ctypes.data_as(u64p), ctypes.c_size_t(3), ctypes.c_size_t(16),
                          sib.ctypes.data_as(u64p), ns.ctypes.data_as(u32p), fl.ctypes.data_as(u32p), None, None, None, None, None, None)
    assert st == -1 and (rt.root() == root0).all() and len(rt) == 8  # nothing applied
    proofs, roots = rt.update(uk, uv)  # the wrapper starts at 64 siblings, is refused, comes back with 256: applied ONCE
    assert len(proofs[0]["siblings"]) > 64 and (roots[0] == root0).all()
    nv = vals.copy()
    nv[:3] = uv % np.uint64(0xFFFFFFFF00000001)
    assert (rt.root() == np.asarray(oracle.smt_root_of_set(keys, nv))).all()
    rt.close()
    rt.close()
    L.glp_smt_close(None)
    with pytest.raises(pkg.GlproverError):
        trees.ResidentSparseMerkleTree(ctx, np.concatenate([keys, keys[:1]]), np.concatenate([vals, vals[:1]]))


def test_partitioned_replay_gives_the_same_process_proofs(pkg, trees, ctx, oracle):
    """Round 5: the host replay of a batch of `set`s runs the key prefixes side by side on the worker pool
    (csrc/smt_replay.h::replay_partitioned) where the tree's shape allows it.  In child processes with the thresholds forced
    (GLP_SMT_REPLAY_BITS / GLP_SMT_REPLAY_MIN_K; 0 bits = always sequential): 3 000 sets on a 2^14-entry tree of random keys —
    updates, inserts, removals, no-ops, repeated keys — every output array of glp_smt_apply_updates is identical between the two
    forms, the first 200 process proofs equal the reference's sequential algorithm, and the trace says which form ran."""
    import subprocess
    import sys

    code = r'''
import hashlib, importlib, sys
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
trees = importlib.import_module("intmax_zkp_core_amd.trees")
ctx = pkg.Context(0)
rng = np.random.default_rng(55)
n, k = 1 << 14, 3000
keys = rng.integers(0, 2**63, (n, 4), dtype=np.uint64)
vals = rng.integers(1, 2**63, (n, 4), dtype=np.uint64)
uk = np.concatenate([keys[rng.integers(0, n, k // 2)], rng.integers(0, 2**63, (k - k // 2, 4), dtype=np.uint64)])
rng.shuffle(uk)
uk[100:140] = uk[60:100]  # repeated keys
uv = rng.integers(1, 2**63, (k, 4), dtype=np.uint64)
uv[rng.random(k) < 0.3] = 0
proofs, roots = trees.smt_apply_updates(ctx, keys, vals, uk, uv)
h = hashlib.sha256()
h.update(np.ascontiguousarray(roots).tobytes())
for p in proofs:
    h.update(repr(sorted(p.items())).encode())
print("DIGEST", h.hexdigest())
if len(sys.argv) > 1:
    from oracle import smt_incremental as inc
    tree = inc.SparseMerkleTree()
    for key, value in zip(keys, vals):
        tree.set(key, value)
    for i in range(200):
        want = inc.set_with_proof(tree, uk[i], uv[i])
        for f in ("fnc", "is_old0", "old_root", "new_root", "old_key", "old_value", "new_key", "new_value"):
            assert proofs[i][f] == want[f], (i, f)
        assert proofs[i]["siblings"] == [tuple(s) for s in want["siblings"]], i
    print("ORACLE OK")
ctx.close()
'''
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, env in (("partitioned", {"GLP_SMT_REPLAY_BITS": "4", "GLP_SMT_REPLAY_MIN_K": "1"}),
                      ("partitioned6", {"GLP_SMT_REPLAY_BITS": "6", "GLP_SMT_REPLAY_MIN_K": "1"}),
                      ("sequential", {"GLP_SMT_REPLAY_BITS": "0"})):
        r = subprocess.run([sys.executable, "-c", code] + (["oracle"] if name == "partitioned" else []), cwd=root,
                           env=dict(os.environ, GLP_TRACE_STEPS="1", **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0]
        want_form = "sequential" if name == "sequential" else "partitioned by key prefix"
        assert ("replay: " + want_form) in r.stderr, (name, r.stderr[-1500:])
        if name == "partitioned":
            assert "ORACLE OK" in r.stdout
    assert outs["partitioned"] == outs["sequential"] == outs["partitioned6"]
