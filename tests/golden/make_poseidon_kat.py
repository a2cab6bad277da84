#!/usr/bin/env python3
"""Extract the golden values that the reference's OWN tests hold for the hashing path into a small JSON fixture.

Run once in the build container (needs /root/reference); the output `poseidon_kat.json` is committed and is
what the test-suite reads — nothing under tests/ touches /root/reference at test time.

Only DATA is extracted (expected digests, fixed input keys); no reference source text is stored.
Sources (reference file:line):
  * src/zkdsa/circuits/mod.rs:85-100        two_to_one(0^4, 0^4) as four decimal limbs
  * src/transaction/circuits/mod.rs:211-218 the same digest (tx_hash of the default transaction)
  * src/zkdsa/circuits/mod.rs:143,149       its hex form (HashOut codec)
  * src/rollup/circuits/mod.rs:104          32 zero-hash siblings H_0..H_31 + default header digests (H_32, H_5, H_4)
  * src/bin/block_circuit.rs:81-88,157-164  the two fixed private keys
  * test_cases/block1_info.json             sender addresses (= two_to_one(sk, sk)) and the two tx hashes
  * src/sparse_merkle_tree/goldilocks_poseidon/hash/mod.rs:63-79,121-138  codec vectors
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def read(p):
    with open(os.path.join(REF, p)) as f:
        return f.read()


def limbs_after(text, anchor, count=4):
    i = text.index(anchor)
    vals = re.findall(r"from_canonical_u64\((\d+)\)", text[i:])
    return [int(v) for v in vals[:count]]


out = {"_generated_by": "tests/golden/make_poseidon_kat.py", "_reference": "InternetMaximalism/intmax-zkp-core"}

zk = read("src/zkdsa/circuits/mod.rs")
out["two_to_one_zero"] = {
    "source": "src/zkdsa/circuits/mod.rs:85-100",
    "left": [0, 0, 0, 0],
    "right": [0, 0, 0, 0],
    "digest": limbs_after(zk, "fn test_default_simple_signature"),
}
tx = read("src/transaction/circuits/mod.rs")
out["default_tx_hash"] = {
    "source": "src/transaction/circuits/mod.rs:211-218",
    "digest": limbs_after(tx, "fn test_default_user_transaction"),
}
m = re.search(r"0xc71603f3[0-9a-f]{56}", zk)
out["two_to_one_zero_hex"] = {"source": "src/zkdsa/circuits/mod.rs:143,149", "hex": m.group(0)}

ru = read("src/rollup/circuits/mod.rs")
i = ru.index("fn test_serde_block_detail")
line = ru[i:]
sib = re.search(r'block_headers_proof_siblings\\":\[(.*?)\]', line).group(1)
siblings = re.findall(r"0x[0-9a-f]{64}", sib)
hdr = {k: re.search(k + r'\\":\\"(0x[0-9a-f]{64})', line).group(1)
       for k in ("block_headers_digest", "transactions_digest", "deposit_digest")}
out["zero_hash_chain"] = {
    "source": "src/rollup/circuits/mod.rs:104",
    "rule": "H_0 = 0^4, H_{i+1} = two_to_one(H_i, H_i)",
    "siblings": siblings,  # H_0 .. H_31
    "block_headers_digest": hdr["block_headers_digest"],  # H_32
    "transactions_digest": hdr["transactions_digest"],  # H_5
    "deposit_digest": hdr["deposit_digest"],  # H_4
}

bc = read("src/bin/block_circuit.rs")
k1 = limbs_after(bc, "let sender1_private_key")
k2 = limbs_after(bc, "let sender2_private_key")
info = json.loads(read("test_cases/block1_info.json"))
out["accounts"] = {
    "source": "src/bin/block_circuit.rs:81-88,157-164 + test_cases/block1_info.json (address = two_to_one(sk, sk), "
              "src/zkdsa/account.rs:164-166)",
    "private_keys": [k1, k2],
    "sender_addresses": [a["sender_address"] for a in info["address_list"]],
}
# the purge outputs of both senders: LayeredLayered SMT entries (key1, key2, key3) -> value, and the two nonces
def u128_tuple(name):
    i = bc.index("let %s = (" % name)
    return [int(v) for v in re.findall(r"from_u128\((\d+)\)", bc[i:])[:3]]


def u128_value(name):
    i = bc.index("let %s = " % name)
    return int(re.findall(r"from_u128\((\d+)\)", bc[i:])[0])


out["transactions"] = {
    "source": "test_cases/block1_info.json (tx hashes); src/bin/block_circuit.rs:111-122,147-152,270-275,284-291,316-323 "
              "(diff-tree entries and nonces); rules: src/sparse_merkle_tree/goldilocks_poseidon/mod.rs:161-184, "
              "src/sparse_merkle_tree/tree.rs:255-350, src/transaction/gadgets/purge/mod.rs:296",
    "rule": "leaf = hash_pad(key[4] ++ value[4] ++ [1]); internal = two_to_one(left, right), key bit (LSB first) = 1 -> right; "
            "layered tree: root of the lower tree is the value of the upper; tx_hash = two_to_one(diff_root, nonce)",
    "diff_tree_entries": [{"key": u128_tuple("key3"), "value": u128_value("value3")},
                          {"key": u128_tuple("key4"), "value": u128_value("value4")}],
    "nonces": [limbs_after(bc, "let sender1_nonce"), limbs_after(bc, "let sender2_nonce")],
    "tx_hashes": info["transactions"],
}

out["codec"] = {
    "source": "src/sparse_merkle_tree/goldilocks_poseidon/hash/mod.rs:63-79,121-138",
    "cases": [
        {"limbs": [1, 0, 0, 0], "hex": "0x" + "0" * 63 + "1", "short_hex": "0x01"},
    ],
}

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "poseidon_kat.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
n = 1 + 1 + 1 + len(siblings) + 3 + 2 + 2
print(f"wrote {dst}: {n + 2} golden values")
