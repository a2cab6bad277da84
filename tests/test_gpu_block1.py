"""The reference's block fixture `test_cases/block1_info.json` recomputed with GPU calls only: every tree
state of the scenario in src/bin/block_circuit.rs is rebuilt from its entry SET with glp_smt_root_of_set (the trees are
canonical), leaf / pair hashes with the batched kernels, the fixed-depth trees with glp_merkle_proof_with_zero_dev and the
block hash with glp_block_hash_batch_dev.  All eight header fields must equal the fixture."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZERO = (0, 0, 0, 0)


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


@pytest.fixture(scope="module")
def block1():
    with open(os.path.join(ROOT, "tests", "golden", "block1_kat.json")) as f:
        return json.load(f)


def K(x):
    return (int(x), 0, 0, 0)


def T(h):
    return tuple(int(x) for x in h)


def test_block1_info_header_from_gpu_calls(trees, ctx, oracle, kat, block1):
    e = block1["asset_entries"]
    depth, max_blocks, prev_number = block1["log_n_txs"], block1["log_max_n_blocks"], block1["prev_block_number"]

    def smt(entries):  # {key: value} -> root of the sparse tree holding exactly these entries
        ks = np.array(list(entries.keys()), dtype=np.uint64).reshape(-1, 4)
        vs = np.array(list(entries.values()), dtype=np.uint64).reshape(-1, 4)
        return T(trees.smt_root_of_set(ctx, ks, vs))

    def h2(a, b):
        return T(trees.two_to_one(ctx, [a], [b])[0])

    def layered(entries):  # {k1: {k2: {k3: value}}} -> root (LayeredLayeredSparseMerkleTree)
        return smt({k1: smt({k2: smt(l3) for k2, l3 in l2.items()}) for k1, l2 in entries.items()})

    def user_asset(merge_key, assets):  # UserAssetTree with one merge key: {contract: {variable: amount}}
        layer1 = smt({c: smt(vs) for c, vs in assets.items()})
        return smt({merge_key: h2(layer1, merge_key)})

    def merkle_root(leaves, index, d, zero=ZERO):
        return T(trees.get_merkle_proof_with_zero(ctx, np.array(leaves, dtype=np.uint64).reshape(-1, 4), index, d, zero)[1])

    sk = np.array(kat["accounts"]["private_keys"], dtype=np.uint64)
    addr1, addr2 = (T(a) for a in trees.two_to_one(ctx, sk, sk))  # 32-byte address convention of the fixture
    (m1, c1, v1), (m2, c2, v2) = e["key1"], e["key2"]
    assert m1 == m2
    funded = {K(c1): {K(v1): K(e["value1"])}, K(c2): {K(v2): K(e["value2"])}}
    purged = {K(c1): {K(v1): ZERO}, K(c2): {K(v2): ZERO}}
    user1_funded, user1_purged = user_asset(K(m1), funded), user_asset(K(m1), purged)
    world0 = smt({addr1: user1_funded})
    deposit_tx_hash = h2(layered({addr2: funded}), ZERO)
    prev = [prev_number, *ZERO, *merkle_root([ZERO], 0, max_blocks), *merkle_root([], 0, depth),
            *merkle_root([deposit_tx_hash], 0, depth), *world0, *world0, *ZERO]
    prev_block_hash = T(trees.get_block_hash(ctx, [prev])[0])
    merge_key = h2(deposit_tx_hash, prev_block_hash)
    user2_purged = user_asset(merge_key, purged)
    d = block1["block2_deposit"]
    tx_hashes = [oracle.hashout_from_hex(t) for t in block1["block_info"]["transactions"]]
    header = {
        "block_number": "0x%08x" % (prev_number + 1),
        "prev_block_hash": prev_block_hash,
        "block_headers_digest": merkle_root([ZERO, prev_block_hash], prev_number, max_blocks),
        "transactions_digest": merkle_root(tx_hashes, 0, depth, zero=h2(ZERO, ZERO)),
        "deposit_digest": merkle_root([layered({addr2: {K(d["contract_address"]): {K(d["variable_index"]): K(d["amount"])}}})], 0, depth),
        "proposed_world_state_digest": smt({addr1: user1_purged, addr2: user2_purged}),
        "approved_world_state_digest": smt({addr1: user1_funded, addr2: user2_purged}),
        "latest_account_digest": smt({addr1: K(0), addr2: K(prev_number + 1)}),
    }
    got = {k: (v if isinstance(v, str) else oracle.hashout_to_hex(v)) for k, v in header.items()}
    assert got == block1["block_info"]["header"]


def test_block1_sender_and_asset_proofs_from_the_gpu(trees, ctx, oracle, kat, block1):
    """The inclusion proofs the scenario of src/bin/block_circuit.rs takes out of its trees, from one batched call per tree
    (glp_smt_proofs_of_set): both senders in the proposed and the approved world state (their digests are fields of the
    fixture's header), and user 1's asset entries in her asset tree.  Each proof must (a) equal the reference's incremental
    `find` (oracle/smt_incremental.py replaying the scenario's `set` calls) and (b) climb to the golden digest."""
    from oracle.smt_incremental import SparseMerkleTree

    e = block1["asset_entries"]
    prev_number = block1["prev_block_number"]
    hdr = block1["block_info"]["header"]

    def smt(entries):
        ks = np.array(list(entries.keys()), dtype=np.uint64).reshape(-1, 4)
        vs = np.array(list(entries.values()), dtype=np.uint64).reshape(-1, 4)
        return T(trees.smt_root_of_set(ctx, ks, vs))

    def h2(a, b):
        return T(trees.two_to_one(ctx, [a], [b])[0])

    def user_asset(merge_key, assets):
        layer1 = smt({c: smt(vs) for c, vs in assets.items()})
        return smt({merge_key: h2(layer1, merge_key)})

    def layered(entries):
        return smt({k1: smt({k2: smt(l3) for k2, l3 in l2.items()}) for k1, l2 in entries.items()})

    sk = np.array(kat["accounts"]["private_keys"], dtype=np.uint64)
    addr1, addr2 = (T(a) for a in trees.two_to_one(ctx, sk, sk))
    (m1, c1, v1), (m2, c2, v2) = e["key1"], e["key2"]
    funded = {K(c1): {K(v1): K(e["value1"])}, K(c2): {K(v2): K(e["value2"])}}
    purged = {K(c1): {K(v1): ZERO}, K(c2): {K(v2): ZERO}}
    user1_funded, user1_purged = user_asset(K(m1), funded), user_asset(K(m1), purged)
    world0 = smt({addr1: user1_funded})
    deposit_tx_hash = h2(layered({addr2: funded}), ZERO)

    def merkle_root(leaves, index, d, zero=ZERO):
        return T(trees.get_merkle_proof_with_zero(ctx, np.array(leaves, dtype=np.uint64).reshape(-1, 4), index, d, zero)[1])

    prev = [prev_number, *ZERO, *merkle_root([ZERO], 0, block1["log_max_n_blocks"]), *merkle_root([], 0, block1["log_n_txs"]),
            *merkle_root([deposit_tx_hash], 0, block1["log_n_txs"]), *world0, *world0, *ZERO]
    merge_key = h2(deposit_tx_hash, T(trees.get_block_hash(ctx, [prev])[0]))
    user2_purged = user_asset(merge_key, purged)
    worlds = {"proposed_world_state_digest": {addr1: user1_purged, addr2: user2_purged},
              "approved_world_state_digest": {addr1: user1_funded, addr2: user2_purged}}
    stranger = (1, 2, 3, 4)
    for field, entries in worlds.items():
        ks = np.array(list(entries.keys()), dtype=np.uint64)
        vs = np.array(list(entries.values()), dtype=np.uint64)
        q = np.array([addr1, addr2, stranger], dtype=np.uint64)
        got = trees.smt_find_batch(ctx, ks, vs, q)
        assert oracle.hashout_to_hex(got.root) == hdr[field]
        ref = SparseMerkleTree()
        for k, v in entries.items():
            ref.set(k, v)
        for i in range(3):
            w = ref.find(q[i])
            assert [T(s) for s in got.siblings(i)] == [tuple(s) for s in w["siblings"]]
            assert bool(got.found[i]) == w["found"] == (i < 2) and bool(got.is_old0[i]) == w["is_old0"]
            assert T(got.value[i]) == tuple(w["value"])
            if got.found[i]:  # (b) the sender's proof climbs to the header's digest
                leaf = trees.smt_leaf_hash(ctx, q[i:i + 1], got.value[i:i + 1])
                root = trees.smt_roots_from_paths(ctx, q[i:i + 1], leaf, [got.siblings(i)])[0]
                assert oracle.hashout_to_hex(root) == hdr[field]
    # user 1's asset tree: the contract layer holds both contracts; each contract's variable layer holds its amount
    layer1_entries = {c: smt(vs) for c, vs in funded.items()}
    ks = np.array(list(layer1_entries.keys()), dtype=np.uint64)
    vs = np.array(list(layer1_entries.values()), dtype=np.uint64)
    got = trees.smt_find_batch(ctx, ks, vs, ks)
    ref = SparseMerkleTree()
    for k, v in layer1_entries.items():
        ref.set(k, v)
    assert T(got.root) == ref.root and got.found.all()
    for i in range(len(ks)):
        assert [T(s) for s in got.siblings(i)] == [tuple(s) for s in ref.find(ks[i])["siblings"]]
        assert T(got.value[i]) == layer1_entries[T(ks[i])]
