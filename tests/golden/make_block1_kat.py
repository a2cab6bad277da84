#!/usr/bin/env python3
"""Extract the block-2 fixture of the reference into tests/golden/block1_kat.json.

`test_cases/block1_info.json` is a DATA file the reference's own binary writes and re-reads
(src/bin/block_circuit.rs:649-661): the `BlockInfo` (header with its seven digests, tx hashes, deposit list, address
list) of the block that binary builds.  Together with the fixed inputs of that scenario (asset-tree keys/values, deposit
entry, rollup constants) it is a known-answer test for the reference's NATIVE tree hashing: every header digest is a
function of those inputs through get_block_hash / get_merkle_proof[_with_zero] / the sparse Merkle trees.

Only data is stored: the JSON file's content and numbers parsed out of the scenario; no reference source text.
Run once in the build container (needs /root/reference); nothing under tests/ touches /root/reference at test time.

NOTE (schema drift): the JSON was written by a revision in which an address was the whole public-key HashOut (32-byte
hex); the current `Address` is its first limb (8 bytes, src/zkdsa/account.rs:168-170).  The digests below reproduce
with the 32-byte convention, i.e. tree keys = two_to_one(sk, sk).
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def read(p):
    with open(os.path.join(REF, p)) as f:
        return f.read()


bc = read("src/bin/block_circuit.rs")


def u128s_after(anchor, count):
    i = bc.index(anchor)
    return [int(v) for v in re.findall(r"from_u128\((\d+)\)", bc[i:])[:count]]


def const_field(name):
    return int(re.search(name + r":\s*(\d+)", bc).group(1))


out = {
    "_generated_by": "tests/golden/make_block1_kat.py",
    "_reference": "InternetMaximalism/intmax-zkp-core",
    "source": "test_cases/block1_info.json; scenario src/bin/block_circuit.rs:81-330,444-640",
    "block_info": json.loads(read("test_cases/block1_info.json")),
    # (merge key / recipient, contract address, variable index) -> amount, as GoldilocksHashOut::from_u128 values
    "asset_entries": {
        "key1": u128s_after("let key1 = (", 3), "value1": u128s_after("let value1 = ", 1)[0],
        "key2": u128s_after("let key2 = (", 3), "value2": u128s_after("let value2 = ", 1)[0],
        "key3": u128s_after("let key3 = (", 3), "value3": u128s_after("let value3 = ", 1)[0],
        "key4": u128s_after("let key4 = (", 3), "value4": u128s_after("let value4 = ", 1)[0],
    },
    "block2_deposit": {"contract_address": int(re.search(r"contract_address: Address\(GoldilocksField::from_canonical_u64\((\d+)\)\)", bc).group(1)),
                       "variable_index": int(re.search(r"variable_index: (\d+)u8\.into\(\)", bc).group(1)),
                       "amount": int(re.search(r"amount: F::from_noncanonical_u64\((\d+)\)", bc).group(1))},
    "log_n_txs": const_field("log_n_txs"),
    "log_max_n_blocks": int(re.search(r"const LOG_MAX_N_BLOCKS: usize = (\d+);", bc).group(1)),
    "prev_block_number": int(re.search(r"let prev_block_number = (\d+)u32;", bc).group(1)),
    "scenario": [
        "sender1 user-asset tree: set key1 -> value1, key2 -> value2; world_state[addr1] = that root (block_circuit.rs:125-138)",
        "block-1 deposit tree (LayeredLayered): (addr2, key1.1, key1.2) -> value1, (addr2, key2.1, key2.2) -> value2; "
        "deposit_tx_hash = two_to_one(root, 0); prev header deposit_digest = get_merkle_proof([deposit_tx_hash], 0, log_n_txs).root (:176-199)",
        "prev header = {number 1, prev_block_hash 0, block_headers_digest = get_merkle_proof([0], 0, 32).root, transactions_digest = "
        "get_merkle_proof([], 0, log_n_txs).root, deposit_digest, proposed = approved = world-state root, latest_account 0}; "
        "header.prev_block_hash = get_block_hash(prev header) (:203-227)",
        "sender2 merges the deposit under merge_key = two_to_one(deposit_tx_hash, prev_block_hash), then both senders purge "
        "everything (set to 0) (:229-265); world_state[addr1], world_state[addr2] = the purged roots -> proposed digest (:379-398)",
        "sender1 has no received signature: its world-state entry reverts to the root before its purge; latest_account[addr2] = 2 "
        "-> approved digest, latest_account_digest (:456-492)",
        "block_headers_digest = get_merkle_proof([0, prev_block_hash], 1, 32).root (:494-499); deposit_digest from the block-2 deposit "
        "tree entry (addr2, contract, variable) -> amount; transactions_digest = get_merkle_proof_with_zero(tx hashes, 0, log_n_txs, "
        "default tx hash).root (:592-612)",
    ],
}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "block1_kat.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print("wrote", dst)
