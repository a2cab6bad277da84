"""Wire / on-disk formats of the reference, host side: what a harness needs to consume the reference's
fixtures (`test_cases/block1_info.json`) and hand its own results back in the same shape.  Pure host logic — strings and
integers, no field arithmetic, no GPU work.

    HashOut        "0x" + 64 hex digits: 4 limbs as little-endian u64, the whole 32 bytes reversed (big-endian number);
                   shorter inputs are accepted and left-padded     reference src/sparse_merkle_tree/goldilocks_poseidon/hash/mod.rs:84-119
    Address        "0x" + 16 hex digits, big-endian u64; at most 8 bytes on input        src/zkdsa/account.rs:63-99
    VariableIndex  "0x" + 2 hex digits; the FIRST byte on input                         src/rollup/gadgets/deposit_block/mod.rs:97-131
    block number   "0x" + 8 hex digits (u32)                                            src/transaction/block_header.rs:52-101
    DepositInfo / ContributedAsset, TransactionSenderWithValidity, BlockHeader, BlockInfo, TokenKind::to_bytes

JSON is emitted exactly as serde_json does (field order of the Rust structs, no whitespace) so that the reference's
golden strings compare equal.
"""
import json

P = 0xFFFFFFFF00000001
BLOCK_HEADER_DIGESTS = ("prev_block_hash", "block_headers_digest", "transactions_digest", "deposit_digest",
                        "proposed_world_state_digest", "approved_world_state_digest", "latest_account_digest")


class CodecError(ValueError):
    pass


def _strip(s):
    if not isinstance(s, str) or not s.startswith("0x"):
        raise CodecError("fail to strip 0x-prefix: given value %r does not start with 0x" % (s,))
    body = s[2:]
    if len(body) % 2:
        raise CodecError("fail to parse a hex string: odd number of digits")
    try:
        return bytes.fromhex(body)
    except ValueError as e:
        raise CodecError("fail to parse a hex string: %s" % e)


# ---- HashOut ----
def hashout_to_hex(limbs):
    limbs = [int(x) for x in limbs]
    assert len(limbs) == 4
    le = b"".join((x % P).to_bytes(8, "little") for x in limbs)
    return "0x" + le[::-1].hex()


def hashout_from_hex(s):
    b = _strip(s)
    if len(b) > 32:
        raise CodecError("too long hexadecimal sequence")
    le = b[::-1].ljust(32, b"\0")
    limbs = [int.from_bytes(le[8 * i:8 * i + 8], "little") for i in range(4)]
    if any(x >= P for x in limbs):  # HashOut::from_bytes takes canonical limbs
        raise CodecError("limb is not a canonical field element")
    return limbs


# ---- Address (one field element) ----
def address_to_hex(a):
    return "0x" + (int(a) % P).to_bytes(8, "big").hex()


def address_from_hex(s):
    b = _strip(s)
    if len(b) > 8:
        raise CodecError("too long hexadecimal sequence")
    v = int.from_bytes(b.rjust(8, b"\0"), "big")
    if v >= P:
        raise CodecError("not a canonical field element")
    return v


def address_to_hash_out(a):
    """Address::to_hash_out = HashOut::from_partial(&[a])   (src/zkdsa/account.rs:128-130): the SMT key of an account"""
    return [int(a) % P, 0, 0, 0]


# ---- VariableIndex (u8) ----
def variable_index_to_hex(v):
    v = int(v)
    if not 0 <= v < 256:
        raise CodecError("variable index does not fit a byte")
    return "0x%02x" % v


def variable_index_from_hex(s):
    b = _strip(s)
    if not b:
        raise CodecError("out of index: given value %r is too short" % (s,))
    return b[0]


def block_number_to_hex(n):
    return "0x%08x" % int(n)


def block_number_from_hex(s):
    b = _strip(s)
    if len(b) > 4:
        raise CodecError("too long hexadecimal sequence")
    return int.from_bytes(b, "big")


def _dumps(obj):
    return json.dumps(obj, separators=(",", ":"))


# ---- DepositInfo (== flattened ContributedAsset: same keys, src/transaction/asset.rs:105-160) ----
def deposit_info_to_obj(d):
    return {"receiver_address": address_to_hex(d["receiver_address"]), "contract_address": address_to_hex(d["contract_address"]),
            "variable_index": variable_index_to_hex(d["variable_index"]), "amount": int(d["amount"])}


def deposit_info_from_obj(o):
    return {"receiver_address": address_from_hex(o["receiver_address"]), "contract_address": address_from_hex(o["contract_address"]),
            "variable_index": variable_index_from_hex(o["variable_index"]), "amount": int(o["amount"])}


def deposit_info_to_json(d):
    return _dumps(deposit_info_to_obj(d))


def deposit_info_from_json(s):
    return deposit_info_from_obj(json.loads(s))


def token_kind_to_bytes(contract_address, variable_index):
    """TokenKind::to_bytes (src/transaction/asset.rs:47-58): 24 bytes of the address hash-out, 8 of the index, little-endian"""
    a = b"".join(int(x).to_bytes(8, "little") for x in address_to_hash_out(contract_address))[:24]
    v = int(variable_index).to_bytes(8, "little")
    return a + v


def token_kind_from_bytes(b):
    assert len(b) == 32
    return int.from_bytes(b[0:8], "little"), b[24]


# ---- BlockHeader ----
def block_header_to_obj(h):
    o = {"block_number": block_number_to_hex(h["block_number"])}
    for k in BLOCK_HEADER_DIGESTS:
        o[k] = hashout_to_hex(h[k])
    return o


def block_header_from_obj(o):
    h = {"block_number": block_number_from_hex(o["block_number"])}
    for k in BLOCK_HEADER_DIGESTS:
        h[k] = hashout_from_hex(o[k])
    return h


def block_header_to_json(h):
    return _dumps(block_header_to_obj(h))


def block_header_from_json(s):
    return block_header_from_obj(json.loads(s))


def block_header_words(h):
    """the 29 u64 words glp_block_hash_batch_dev reads (include/glprover.h)"""
    return [int(h["block_number"])] + [int(x) for k in BLOCK_HEADER_DIGESTS for x in h[k]]


# ---- SimpleSignaturePublicInputs: plonky2's derive form and the reference's hex form (src/zkdsa/circuits/mod.rs:100-153) ----
SIGNATURE_FIELDS = ("message", "public_key", "signature")


def signature_public_inputs_to_json(pi, hex_form=True):
    if hex_form:
        return _dumps({k: hashout_to_hex(pi[k]) for k in SIGNATURE_FIELDS})
    return _dumps({k: {"elements": [int(x) for x in pi[k]]} for k in SIGNATURE_FIELDS})


def signature_public_inputs_from_json(s):
    o = json.loads(s)
    return {k: (hashout_from_hex(o[k]) if isinstance(o[k], str) else [int(x) for x in o[k]["elements"]]) for k in SIGNATURE_FIELDS}


# ---- BlockInfo (src/rollup/block.rs:10-44) ----
def block_info_to_json(info, address_codec=(address_to_hex, address_from_hex)):
    enc = address_codec[0]

    def dep(d):
        o = deposit_info_to_obj({**d, "receiver_address": 0, "contract_address": 0})
        o["receiver_address"], o["contract_address"] = enc(d["receiver_address"]), enc(d["contract_address"])
        return o

    o = {"header": block_header_to_obj(info["header"]), "transactions": [hashout_to_hex(t) for t in info["transactions"]],
         "deposit_list": [dep(d) for d in info["deposit_list"]]}
    for k in ("scroll_flag_list", "polygon_flag_list"):
        if k in info:
            o[k] = [dep(d) for d in info[k]]
    o["address_list"] = [{"sender_address": enc(a["sender_address"]), "is_valid": bool(a["is_valid"])} for a in info["address_list"]]
    return _dumps(o)


def block_info_from_json(s, address_codec=(address_to_hex, address_from_hex)):
    """`address_codec`: the fixture test_cases/block1_info.json predates the 8-byte Address and stores addresses as whole
    public-key digests — pass (hashout_to_hex, hashout_from_hex) to read it"""
    dec = address_codec[1]
    o = json.loads(s) if isinstance(s, str) else s

    def dep(d):
        return {"receiver_address": dec(d["receiver_address"]), "contract_address": dec(d["contract_address"]),
                "variable_index": variable_index_from_hex(d["variable_index"]), "amount": int(d["amount"])}

    info = {"header": block_header_from_obj(o["header"]), "transactions": [hashout_from_hex(t) for t in o["transactions"]],
            "deposit_list": [dep(d) for d in o["deposit_list"]]}
    for k in ("scroll_flag_list", "polygon_flag_list"):
        if k in o:
            info[k] = [dep(d) for d in o[k]]
    info["address_list"] = [{"sender_address": dec(a["sender_address"]), "is_valid": bool(a["is_valid"])} for a in o["address_list"]]
    return info


# ---- BlockProductionPublicInputs::encode / decode (src/rollup/circuits/mod.rs:711-856): the field elements whose hash_no_pad is the
# block proof's only public input (its "entry hash") ----
BLOCK_PI_ROOTS = ("old_account_tree_root", "new_account_tree_root", "old_world_state_root", "new_world_state_root",
                  "old_prev_block_header_digest", "new_prev_block_header_digest", "block_hash")


def block_production_public_inputs_encode(pi):
    """pi: dict(address_list [{sender_address (int), is_valid}], deposit_list / scroll_flag_list / polygon_flag_list [{receiver_address,
    contract_address (ints), variable_index (u8), amount}], the seven roots [4 words each]) -> the flat word list; an Address and a
    VariableIndex are written as a HashOut with the value in limb 0"""
    out = []
    for a in pi["address_list"]:
        out += [int(a["sender_address"]), 0, 0, 0, 1 if a["is_valid"] else 0]
    for name in ("deposit_list", "scroll_flag_list", "polygon_flag_list"):
        for d in pi[name]:
            out += [int(d["receiver_address"]), 0, 0, 0, int(d["contract_address"]), 0, 0, 0, int(d["variable_index"]), 0, 0, 0, int(d["amount"])]
    for name in BLOCK_PI_ROOTS:
        out += [int(x) for x in pi[name]]
    return out


def block_production_public_inputs_decode(words, n_txs, n_deposits, n_scroll_flags, n_polygon_flags):
    words = [int(x) for x in words]
    if len(words) != 5 * n_txs + 13 * (n_deposits + n_scroll_flags + n_polygon_flags) + 28:
        raise CodecError("block production public inputs: %d words" % len(words))
    it = iter(words)
    take = lambda n: [next(it) for _ in range(n)]  # noqa: E731

    def address():  # Address::read (src/zkdsa/account.rs:140) takes limb 0 and skips the other three unchecked: so does this
        return take(4)[0]

    pi = {"address_list": [{"sender_address": address(), "is_valid": bool(next(it))} for _ in range(n_txs)]}
    for name, cnt in (("deposit_list", n_deposits), ("scroll_flag_list", n_scroll_flags), ("polygon_flag_list", n_polygon_flags)):
        pi[name] = [{"receiver_address": address(), "contract_address": address(), "variable_index": take(4)[0] & 0xFF, "amount": next(it)}
                    for _ in range(cnt)]
    for name in BLOCK_PI_ROOTS:
        pi[name] = take(4)
    return pi


# ---------------------------------------------------------------------------------------------------------------
# ProofWithPublicInputs: the flat u64 proof of glp_prove (layout at glp_proof_len / glp_fri_proof_len in
# include/glprover.h) <-> a structured view <-> plonky2's byte format (`Buffer::write_proof_with_public_inputs`,
# plonky2 @ f99ed9c src/util/serialization.rs).  The byte format is every field element as 8 little-endian bytes in
# plonky2's struct order — the same order as the flat proof — plus ONE length byte in front of each Merkle path
# (`write_merkle_proof`: `siblings.len() as u8`), and the public inputs at the end.  "Parity unpinned": the reference holds
# no serialized proof; restated from the upstream source.
class ProofLayout:
    """segment sizes of a proof, from the circuit's common data (a `circuit.CircuitData` or any object with the same
    attributes: degree_bits, num_constants, quotient_degree_factor, num_partial_products, cfg.{num_wires, num_routed_wires,
    num_challenges, rate_bits, cap_height, num_query_rounds}) and the FRI reduction arity bits"""

    def __init__(self, cd, reduction_arity_bits, num_query_rounds=None, cap_height=None, zero_knowledge=False):
        cfg = cd.cfg
        nch = cfg.num_challenges
        self.cap_height = cfg.cap_height if cap_height is None else cap_height
        self.rate_bits = cfg.rate_bits
        self.degree_bits = cd.degree_bits
        self.num_query_rounds = cfg.num_query_rounds if num_query_rounds is None else num_query_rounds
        self.arity_bits = list(reduction_arity_bits)
        self.oracle_cols = [cd.num_constants + cfg.num_routed_wires, cfg.num_wires, nch * (1 + cd.num_partial_products),
                            nch * cd.quotient_degree_factor]
        if zero_knowledge:  # config.zero_knowledge: the `evals` of the three blinded oracles carry SALT_SIZE = 4 salt words
            self.oracle_cols = [self.oracle_cols[0]] + [c + 4 for c in self.oracle_cols[1:]]
        self.openings = [("constants", cd.num_constants), ("plonk_sigmas", cfg.num_routed_wires), ("wires", cfg.num_wires),
                         ("plonk_zs", nch), ("plonk_zs_next", nch), ("partial_products", nch * cd.num_partial_products),
                         ("quotient_polys", nch * cd.quotient_degree_factor)]
        self.lde_bits = self.degree_bits + self.rate_bits
        self.final_poly_len = 1 << (self.degree_bits - sum(self.arity_bits))

    def path_len(self, tree_bits):
        return max(tree_bits - self.cap_height, 0)


def proof_from_flat(flat, layout):
    """flat u64 proof -> dict mirroring plonky2's Proof { wires_cap, plonk_zs_partial_products_cap, quotient_polys_cap,
    openings, opening_proof: FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness } }"""
    flat = [int(x) for x in flat]
    pos = [0]

    def take(n):
        out = flat[pos[0]:pos[0] + n]
        if len(out) != n:
            raise CodecError("flat proof too short")
        pos[0] += n
        return out

    def hashes(count):
        w = take(4 * count)
        return [w[4 * i:4 * i + 4] for i in range(count)]

    def exts(count):
        w = take(2 * count)
        return [w[2 * i:2 * i + 2] for i in range(count)]

    L = layout
    cap = 1 << L.cap_height
    proof = {"wires_cap": hashes(cap), "plonk_zs_partial_products_cap": hashes(cap), "quotient_polys_cap": hashes(cap),
             "openings": {name: exts(n) for name, n in L.openings}}
    fri = {"commit_phase_merkle_caps": [hashes(cap) for _ in L.arity_bits], "query_round_proofs": []}
    for _ in range(L.num_query_rounds):
        initial = [{"evals": take(cols), "siblings": hashes(L.path_len(L.lde_bits))} for cols in L.oracle_cols]
        steps, bits = [], L.lde_bits
        for a in L.arity_bits:
            bits -= a
            steps.append({"evals": exts(1 << a), "siblings": hashes(L.path_len(bits))})
        fri["query_round_proofs"].append({"initial_trees_proof": initial, "steps": steps})
    fri["final_poly"] = exts(L.final_poly_len)
    fri["pow_witness"] = take(1)[0]
    if pos[0] != len(flat):
        raise CodecError("flat proof has %d trailing words" % (len(flat) - pos[0]))
    proof["opening_proof"] = fri
    return proof


def proof_to_flat(proof):
    out = []
    for k in ("wires_cap", "plonk_zs_partial_products_cap", "quotient_polys_cap"):
        for h in proof[k]:
            out += h
    for vals in proof["openings"].values():
        for e in vals:
            out += e
    fri = proof["opening_proof"]
    for c in fri["commit_phase_merkle_caps"]:
        for h in c:
            out += h
    for q in fri["query_round_proofs"]:
        for t in q["initial_trees_proof"]:
            out += t["evals"]
            for h in t["siblings"]:
                out += h
        for s in q["steps"]:
            for e in s["evals"]:
                out += e
            for h in s["siblings"]:
                out += h
    for e in fri["final_poly"]:
        out += e
    out.append(fri["pow_witness"])
    return out


def _u64s(words):
    return b"".join(int(w).to_bytes(8, "little") for w in words)


def proof_with_public_inputs_to_bytes(proof, public_inputs):
    b = bytearray()
    for k in ("wires_cap", "plonk_zs_partial_products_cap", "quotient_polys_cap"):
        for h in proof[k]:
            b += _u64s(h)
    for vals in proof["openings"].values():
        for e in vals:
            b += _u64s(e)
    fri = proof["opening_proof"]
    for c in fri["commit_phase_merkle_caps"]:
        for h in c:
            b += _u64s(h)

    def merkle_proof(sib):
        if len(sib) > 255:
            raise CodecError("Merkle path longer than one length byte")
        b.append(len(sib))
        for h in sib:
            b.extend(_u64s(h))

    for q in fri["query_round_proofs"]:
        for t in q["initial_trees_proof"]:
            b += _u64s(t["evals"])
            merkle_proof(t["siblings"])
        for s in q["steps"]:
            for e in s["evals"]:
                b += _u64s(e)
            merkle_proof(s["siblings"])
    for e in fri["final_poly"]:
        b += _u64s(e)
    b += _u64s([fri["pow_witness"]])
    b += _u64s(public_inputs)
    return bytes(b)


def proof_with_public_inputs_from_bytes(data, layout, n_public_inputs):
    """inverse of proof_with_public_inputs_to_bytes -> (proof dict, public_inputs); checks the path length bytes"""
    pos = [0]

    def words(n):
        chunk = data[pos[0]:pos[0] + 8 * n]
        if len(chunk) != 8 * n:
            raise CodecError("proof bytes too short")
        pos[0] += 8 * n
        ws = [int.from_bytes(chunk[8 * i:8 * i + 8], "little") for i in range(n)]
        if any(w >= P for w in ws):
            raise CodecError("non-canonical field element in proof bytes")
        return ws

    def hashes(count):
        w = words(4 * count)
        return [w[4 * i:4 * i + 4] for i in range(count)]

    def exts(count):
        w = words(2 * count)
        return [w[2 * i:2 * i + 2] for i in range(count)]

    def merkle_proof(expect):
        if pos[0] >= len(data):
            raise CodecError("proof bytes too short")
        ln = data[pos[0]]
        pos[0] += 1
        if ln != expect:
            raise CodecError("Merkle path of %d siblings where the circuit expects %d" % (ln, expect))
        return hashes(ln)

    L = layout
    cap = 1 << L.cap_height
    proof = {"wires_cap": hashes(cap), "plonk_zs_partial_products_cap": hashes(cap), "quotient_polys_cap": hashes(cap),
             "openings": {name: exts(n) for name, n in L.openings}}
    fri = {"commit_phase_merkle_caps": [hashes(cap) for _ in L.arity_bits], "query_round_proofs": []}
    for _ in range(L.num_query_rounds):
        initial = []
        for cols in L.oracle_cols:
            ev = words(cols)
            initial.append({"evals": ev, "siblings": merkle_proof(L.path_len(L.lde_bits))})
        steps, bits = [], L.lde_bits
        for a in L.arity_bits:
            bits -= a
            ev = exts(1 << a)
            steps.append({"evals": ev, "siblings": merkle_proof(L.path_len(bits))})
        fri["query_round_proofs"].append({"initial_trees_proof": initial, "steps": steps})
    fri["final_poly"] = exts(L.final_poly_len)
    fri["pow_witness"] = words(1)[0]
    proof["opening_proof"] = fri
    public_inputs = words(n_public_inputs)
    if pos[0] != len(data):
        raise CodecError("%d trailing bytes after the proof" % (len(data) - pos[0]))
    return proof, public_inputs


def _path_marks(layout):
    """[(word offset in the flat proof, path length)] of every Merkle path, in order — the byte form differs from the flat
    words only by one length byte in front of each path"""
    L = layout
    cap = 4 << L.cap_height
    pos = 3 * cap + 2 * sum(n for _, n in L.openings) + len(L.arity_bits) * cap
    marks = []
    for _ in range(L.num_query_rounds):
        for cols in L.oracle_cols:
            pos += cols
            pl = L.path_len(L.lde_bits)
            marks.append((pos, pl))
            pos += 4 * pl
        bits = L.lde_bits
        for a in L.arity_bits:
            bits -= a
            pos += 2 << a
            pl = L.path_len(bits)
            marks.append((pos, pl))
            pos += 4 * pl
    return marks, pos + 2 * L.final_poly_len + 1


def flat_proof_to_bytes(flat, layout, public_inputs):
    """the flat proof (numpy u64, as the prover returns it) + public inputs -> plonky2's byte form; the same bytes as
    proof_with_public_inputs_to_bytes(proof_from_flat(flat, layout), public_inputs) without the detour through Python lists
    (the aggregation workloads serialise hundreds of proofs)"""
    import numpy as np

    words = np.ascontiguousarray(np.asarray(flat, dtype="<u8"))
    marks, total = _path_marks(layout)
    if words.size != total:
        raise CodecError("flat proof has %d words, the layout says %d" % (words.size, total))
    raw = words.tobytes()
    out, prev = bytearray(), 0
    for pos, pl in marks:
        if pl > 255:
            raise CodecError("Merkle path longer than one length byte")
        out += raw[8 * prev:8 * pos]
        out.append(pl)
        prev = pos
    out += raw[8 * prev:]
    out += np.asarray([int(x) for x in public_inputs], dtype="<u8").tobytes()
    return bytes(out)


def flat_proof_from_bytes(data, layout, n_public_inputs):
    """inverse of flat_proof_to_bytes -> (numpy u64 flat proof, public inputs); checks lengths, path length bytes and canonicity"""
    import numpy as np

    marks, total = _path_marks(layout)
    if len(data) != 8 * (total + n_public_inputs) + len(marks):
        raise CodecError("proof bytes: %d, expected %d" % (len(data), 8 * (total + n_public_inputs) + len(marks)))
    parts, prev, off = [], 0, 0
    for pos, pl in marks:
        parts.append(data[off:off + 8 * (pos - prev)])
        off += 8 * (pos - prev)
        if data[off] != pl:
            raise CodecError("Merkle path of %d siblings where the circuit expects %d" % (data[off], pl))
        off += 1
        prev = pos
    parts.append(data[off:])
    words = np.frombuffer(b"".join(parts), dtype="<u8")
    if (words >= np.uint64(P)).any():
        raise CodecError("non-canonical field element in proof bytes")
    return words[:total].copy(), [int(x) for x in words[total:]]


# ---------------------------------------------------------------------------------------------------------------
# serde-JSON form of a proof — what `serde_json::to_string(&proof)` yields for plonky2's `Proof` / `ProofWithPublicInputs`
# (both `#[derive(Serialize, Deserialize)]`; the reference relies on it for `SimpleSignatureProofWithPublicInputs`,
# src/zkdsa/circuits/mod.rs:256-265, and clones the same fields one by one at src/recursion/gadgets/mod.rs:33-58):
#   GoldilocksField            -> a JSON number (canonical u64)
#   QuadraticExtension([F; 2]) -> [a, b]
#   HashOut { elements }       -> {"elements": [4]}
#   MerkleCap(Vec<HashOut>)    -> [HashOut, ...]
#   MerkleProof { siblings }   -> {"siblings": [HashOut, ...]}
#   OpeningSet                 -> {"constants": [...], "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products",
#                                  "quotient_polys"}
#   FriInitialTreeProof        -> {"evals_proofs": [[evals, MerkleProof], ...]}   (a Rust tuple is a 2-element array)
#   FriQueryStep               -> {"evals": [ext...], "merkle_proof": MerkleProof}
#   PolynomialCoeffs           -> {"coeffs": [...]}
# The struct form of this module (proof_from_flat) differs only in naming: siblings lists are bare lists of 4-word lists.
OPENING_FIELDS = ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products", "quotient_polys")


def _ho(h):
    return {"elements": [int(x) for x in h]}


def _ho_in(o):
    e = o["elements"] if isinstance(o, dict) else o
    if len(e) != 4 or any(not (0 <= int(x) < P) for x in e):
        raise CodecError("HashOut needs four canonical field elements")
    return [int(x) for x in e]


def _fe(x):
    x = int(x)
    if not 0 <= x < P:
        raise CodecError("non-canonical field element %d" % x)
    return x


def proof_to_serde_json(proof):
    """struct form (proof_from_flat) -> the object plonky2's derived Serialize produces for `Proof`"""
    fri = proof["opening_proof"]
    return {
        "wires_cap": [_ho(h) for h in proof["wires_cap"]],
        "plonk_zs_partial_products_cap": [_ho(h) for h in proof["plonk_zs_partial_products_cap"]],
        "quotient_polys_cap": [_ho(h) for h in proof["quotient_polys_cap"]],
        "openings": {k: [[int(e[0]), int(e[1])] for e in proof["openings"][k]] for k in OPENING_FIELDS},
        "opening_proof": {
            "commit_phase_merkle_caps": [[_ho(h) for h in cap] for cap in fri["commit_phase_merkle_caps"]],
            "query_round_proofs": [
                {"initial_trees_proof": {"evals_proofs": [[[int(x) for x in t["evals"]],
                                                           {"siblings": [_ho(h) for h in t["siblings"]]}]
                                                          for t in q["initial_trees_proof"]]},
                 "steps": [{"evals": [[int(e[0]), int(e[1])] for e in s["evals"]],
                            "merkle_proof": {"siblings": [_ho(h) for h in s["siblings"]]}} for s in q["steps"]]}
                for q in fri["query_round_proofs"]],
            "final_poly": {"coeffs": [[int(e[0]), int(e[1])] for e in fri["final_poly"]]},
            "pow_witness": int(fri["pow_witness"]),
        },
    }


def proof_from_serde_json(obj):
    """inverse of proof_to_serde_json; rejects non-canonical elements and malformed extension pairs"""
    def ext(e):
        if len(e) != 2:
            raise CodecError("extension element needs two limbs")
        return [_fe(e[0]), _fe(e[1])]

    try:
        fri = obj["opening_proof"]
        out = {
            "wires_cap": [_ho_in(h) for h in obj["wires_cap"]],
            "plonk_zs_partial_products_cap": [_ho_in(h) for h in obj["plonk_zs_partial_products_cap"]],
            "quotient_polys_cap": [_ho_in(h) for h in obj["quotient_polys_cap"]],
            "openings": {k: [ext(e) for e in obj["openings"][k]] for k in OPENING_FIELDS},
            "opening_proof": {
                "commit_phase_merkle_caps": [[_ho_in(h) for h in cap] for cap in fri["commit_phase_merkle_caps"]],
                "query_round_proofs": [
                    {"initial_trees_proof": [{"evals": [_fe(x) for x in ep[0]],
                                              "siblings": [_ho_in(h) for h in ep[1]["siblings"]]}
                                             for ep in q["initial_trees_proof"]["evals_proofs"]],
                     "steps": [{"evals": [ext(e) for e in s["evals"]],
                                "siblings": [_ho_in(h) for h in s["merkle_proof"]["siblings"]]} for s in q["steps"]]}
                    for q in fri["query_round_proofs"]],
                "final_poly": [ext(e) for e in fri["final_poly"]["coeffs"]],
                "pow_witness": _fe(fri["pow_witness"]),
            },
        }
    except (KeyError, TypeError, IndexError) as e:
        raise CodecError("malformed proof JSON: %r" % (e,))
    return out


def proof_with_public_inputs_to_json(proof, public_inputs):
    """ProofWithPublicInputs { proof, public_inputs: Vec<F> } as a JSON string"""
    import json

    return json.dumps({"proof": proof_to_serde_json(proof), "public_inputs": [_fe(x) for x in public_inputs]})


def proof_with_public_inputs_from_json(text):
    import json

    try:
        obj = json.loads(text)
    except ValueError as e:
        raise CodecError("not JSON: %s" % e)
    if not isinstance(obj, dict) or "proof" not in obj or "public_inputs" not in obj:
        raise CodecError("ProofWithPublicInputs needs `proof` and `public_inputs`")
    return proof_from_serde_json(obj["proof"]), [_fe(x) for x in obj["public_inputs"]]


def simple_signature_proof_to_json(proof, public_inputs):
    """SimpleSignatureProofWithPublicInputs (reference src/zkdsa/circuits/mod.rs:54-60,256-265): the twelve public inputs
    [message, public_key, signature] are three `HashOut`s"""
    import json

    if len(public_inputs) != 12:
        raise CodecError("a simple signature has 12 public inputs")
    pi = [_fe(x) for x in public_inputs]
    return json.dumps({"proof": proof_to_serde_json(proof),
                       "public_inputs": {"message": _ho(pi[0:4]), "public_key": _ho(pi[4:8]), "signature": _ho(pi[8:12])}})


def simple_signature_proof_from_json(text):
    """-> (proof struct, public inputs in `encode()` order: message, public_key, signature)"""
    import json

    try:
        obj = json.loads(text)
        pi = obj["public_inputs"]
        flat = _ho_in(pi["message"]) + _ho_in(pi["public_key"]) + _ho_in(pi["signature"])
        return proof_from_serde_json(obj["proof"]), flat
    except (ValueError, KeyError, TypeError) as e:
        raise CodecError("malformed SimpleSignatureProofWithPublicInputs: %r" % (e,))
