"""The reference's block fixture `test_cases/block1_info.json` as a known-answer test for its native tree hashing
(SURVEY.md 8f N2): all eight header fields and both tx hashes are recomputed from the fixed scenario inputs of
src/bin/block_circuit.rs with the oracle (C hashing + the restated incremental trees).  CPU only."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def block1():
    with open(os.path.join(ROOT, "tests", "golden", "block1_kat.json")) as f:
        return json.load(f)


def K(x):
    return (int(x), 0, 0, 0)


def scenario(block1, kat, oracle):
    """replays src/bin/block_circuit.rs:81-640 on the host trees; returns the BlockHeader fields as hex"""
    from oracle.smt_incremental import ZERO, LayeredLayeredSparseMerkleTree, SparseMerkleTree, UserAssetTree, _t

    O = oracle
    e = block1["asset_entries"]
    depth, max_blocks = block1["log_n_txs"], block1["log_max_n_blocks"]
    sk1, sk2 = kat["accounts"]["private_keys"]
    addr1, addr2 = _t(O.two_to_one(sk1, sk1)), _t(O.two_to_one(sk2, sk2))  # 32-byte address convention of the fixture
    (m1, c1, v1), (m2, c2, v2) = e["key1"], e["key2"]
    empty = np.zeros((0, 4), dtype=np.uint64)

    user1 = UserAssetTree()
    user1.set(K(m1), K(c1), K(v1), K(e["value1"]))
    user1.set(K(m2), K(c2), K(v2), K(e["value2"]))
    world = SparseMerkleTree()
    world.set(addr1, user1.root)
    user1_before_purge = user1.root
    user1.set(K(m2), K(c2), K(v2), ZERO)
    user1.set(K(m1), K(c1), K(v1), ZERO)

    dep1 = LayeredLayeredSparseMerkleTree()
    dep1.set(addr2, K(c1), K(v1), K(e["value1"]))
    dep1.set(addr2, K(c2), K(v2), K(e["value2"]))
    deposit_tx_hash = O.two_to_one(dep1.root, ZERO)
    prev = {
        "block_number": block1["prev_block_number"],
        "prev_block_hash": ZERO,
        "block_headers_digest": O.merkle_proof_with_zero(np.zeros((1, 4), dtype=np.uint64), 0, max_blocks)[1],
        "transactions_digest": O.merkle_proof_with_zero(empty, 0, depth)[1],
        "deposit_digest": O.merkle_proof_with_zero(np.array([deposit_tx_hash]), 0, depth)[1],
        "proposed_world_state_digest": world.root,
        "approved_world_state_digest": world.root,
        "latest_account_digest": ZERO,
    }
    order = ("prev_block_hash", "block_headers_digest", "transactions_digest", "deposit_digest",
             "proposed_world_state_digest", "approved_world_state_digest", "latest_account_digest")
    words = [prev["block_number"]] + [int(x) for k in order for x in prev[k]]
    prev_block_hash = O.block_hash(np.array(words, dtype=np.uint64))

    merge_key = _t(O.two_to_one(deposit_tx_hash, prev_block_hash))
    user2 = UserAssetTree()
    user2.set(merge_key, K(c1), K(v1), K(e["value1"]))
    user2.set(merge_key, K(c2), K(v2), K(e["value2"]))
    user2.set(merge_key, K(c2), K(v2), ZERO)
    user2.set(merge_key, K(c1), K(v1), ZERO)
    world.set(addr1, user1.root)
    world.set(addr2, user2.root)
    proposed = world.root
    world.set(addr1, user1_before_purge)  # no received signature for sender 1: revert to the root before its purge
    world.set(addr2, user2.root)
    approved = world.root
    latest = SparseMerkleTree()
    latest.set(addr1, K(0))
    latest.set(addr2, K(block1["prev_block_number"] + 1))

    d = block1["block2_deposit"]
    dep2 = LayeredLayeredSparseMerkleTree()
    dep2.set(addr2, K(d["contract_address"]), K(d["variable_index"]), K(d["amount"]))
    tx_hashes = np.array([O.hashout_from_hex(t) for t in block1["block_info"]["transactions"]])
    default_tx_hash = O.two_to_one(ZERO, ZERO)
    header = {
        "block_number": "0x%08x" % (block1["prev_block_number"] + 1),
        "prev_block_hash": prev_block_hash,
        "block_headers_digest": O.merkle_proof_with_zero(np.array([ZERO, _t(prev_block_hash)], dtype=np.uint64),
                                                         block1["prev_block_number"], max_blocks)[1],
        "transactions_digest": O.merkle_proof_with_zero(tx_hashes, 0, depth, default_tx_hash)[1],
        "deposit_digest": O.merkle_proof_with_zero(np.array([dep2.root], dtype=np.uint64), 0, depth)[1],
        "proposed_world_state_digest": proposed,
        "approved_world_state_digest": approved,
        "latest_account_digest": latest.root,
    }
    return {k: (v if isinstance(v, str) else O.hashout_to_hex(v)) for k, v in header.items()}


def test_block1_info_header_is_reproduced(oracle, kat, block1):
    assert scenario(block1, kat, oracle) == block1["block_info"]["header"]


def test_block1_info_lists(oracle, kat, block1):
    """the non-header fields: tx hashes (already pinned in test_oracle_smt), deposit list and address list"""
    info = block1["block_info"]
    sk1, sk2 = kat["accounts"]["private_keys"]
    addrs = [oracle.hashout_to_hex(oracle.two_to_one(s, s)) for s in (sk1, sk2)]
    assert [a["sender_address"] for a in info["address_list"]] == addrs
    assert [a["is_valid"] for a in info["address_list"]] == [False, True]  # only sender 2 returned a signature
    assert info["deposit_list"][0]["receiver_address"] == addrs[1]
    assert info["transactions"] == kat["transactions"]["tx_hashes"]
