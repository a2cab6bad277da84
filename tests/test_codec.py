"""Wire formats: the host codec against the strings the reference's own serde tests compare with
(tests/golden/serde_kat.json) and against its block fixture.  CPU only."""
import importlib
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def codec():
    import __graft_entry__ as g

    g.load_package()
    return importlib.import_module("intmax_zkp_core_amd.codec")


@pytest.fixture(scope="module")
def serde():
    with open(os.path.join(ROOT, "tests", "golden", "serde_kat.json")) as f:
        return json.load(f)


def test_hashout_address_variable_index(codec, serde, oracle):
    h = serde["hashout"]
    assert codec.hashout_to_hex(h["limbs"]) == h["display"] and json.dumps(h["display"]) == h["json"]
    assert codec.hashout_from_hex(h["short"]) == h["limbs"] == codec.hashout_from_hex(json.loads(h["json_short"]))
    a = serde["address"]
    assert codec.address_to_hex(a["value"]) == a["display"] and json.dumps(a["display"]) == a["json"]
    assert codec.address_from_hex(a["short"]) == a["value"] == codec.address_from_hex(json.loads(a["json_short"]))
    v = serde["variable_index"]
    assert codec.variable_index_to_hex(v["value"]) == v["display"] and codec.variable_index_from_hex(v["display"]) == v["value"]
    # agreement with the oracle's C codec on non-trivial digests, and the error cases of the Rust deserialisers
    limbs = [int(x) for x in oracle.two_to_one([0] * 4, [0] * 4)]
    assert codec.hashout_to_hex(limbs) == oracle.hashout_to_hex(limbs)
    assert codec.hashout_from_hex(codec.hashout_to_hex(limbs)) == limbs
    for bad in ("01", "0x0", "0xzz", "0x" + "00" * 33):
        with pytest.raises(codec.CodecError):
            codec.hashout_from_hex(bad)
    with pytest.raises(codec.CodecError):
        codec.address_from_hex("0x" + "00" * 9)
    with pytest.raises(codec.CodecError):
        codec.variable_index_from_hex("0x")
    assert codec.variable_index_from_hex("0x14ff") == 20  # the first byte wins (deposit_block/mod.rs:117-119)


def test_deposit_info_block_header_signature(codec, serde):
    d = {"receiver_address": 0, "contract_address": 0, "variable_index": 0, "amount": 0}
    assert codec.deposit_info_to_json(d) == serde["deposit_info_default"]
    assert codec.deposit_info_from_json(serde["deposit_info_default"]) == d
    bh = serde["block_header"]
    hdr = {"block_number": bh["block_number"]}
    for k, v in zip(codec.BLOCK_HEADER_DIGESTS, bh["digests_from_u32"]):
        hdr[k] = [v, 0, 0, 0]
    assert codec.block_header_to_json(hdr) == bh["json"]
    assert codec.block_header_from_json(bh["json"]) == hdr
    assert codec.block_header_words(hdr) == [0, 1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5, 0, 0, 0, 6, 0, 0, 0, 7, 0, 0, 0]
    sg = serde["simple_signature_public_inputs"]
    pi = codec.signature_public_inputs_from_json(sg["elements_form"])
    assert pi == codec.signature_public_inputs_from_json(sg["hex_form"])
    assert codec.signature_public_inputs_to_json(pi, hex_form=True) == sg["hex_form"]
    assert codec.signature_public_inputs_to_json(pi, hex_form=False) == sg["elements_form"]
    assert codec.token_kind_from_bytes(codec.token_kind_to_bytes(0x1234, 7)) == (0x1234, 7)


def test_block_detail_default_json(codec, serde, oracle, kat):
    """BlockDetail::new(4) (src/rollup/circuits/mod.rs:93-131): the siblings and the previous header inside the golden
    string are exactly what the tree functions produce"""
    import numpy as np

    o = json.loads(serde["block_detail_new_4"])
    hdr = codec.block_header_from_obj(o["prev_block_header"])
    empty = np.zeros((0, 4), dtype=np.uint64)
    h1 = oracle.two_to_one([0] * 4, [0] * 4)
    want = {"block_number": 0, "prev_block_hash": [0] * 4,
            "block_headers_digest": [int(x) for x in oracle.merkle_proof_with_zero(empty, 0, 32)[1]],
            "transactions_digest": [int(x) for x in oracle.merkle_proof_with_zero(empty, 0, 4, h1)[1]],
            "deposit_digest": [int(x) for x in oracle.merkle_proof_with_zero(empty, 0, 4)[1]],
            "proposed_world_state_digest": [0] * 4, "approved_world_state_digest": [0] * 4, "latest_account_digest": [0] * 4}
    assert hdr == want
    # block_headers = [get_block_hash(prev header)], index 0, depth 32: the siblings are the zero-hash chain
    block_hash = oracle.block_hash(np.array(codec.block_header_words(hdr), dtype=np.uint64))
    sib, _ = oracle.merkle_proof_with_zero(np.array([block_hash]), 0, 32)
    assert [codec.hashout_to_hex(s) for s in sib] == o["block_headers_proof_siblings"]
    # re-emitting the object in the struct's field order gives the golden string back
    o2 = {k: o[k] for k in o if k not in ("block_headers_proof_siblings", "prev_block_header")}
    o2["block_headers_proof_siblings"] = [codec.hashout_to_hex(s) for s in sib]
    o2["prev_block_header"] = codec.block_header_to_obj(hdr)
    assert json.dumps(o2, separators=(",", ":")) == serde["block_detail_new_4"]


def test_block1_info_roundtrip(codec):
    """test_cases/block1_info.json (addresses in the 32-byte form it was written with) parses and re-serialises unchanged"""
    with open(os.path.join(ROOT, "tests", "golden", "block1_kat.json")) as f:
        fixture = json.load(f)["block_info"]
    wide = (codec.hashout_to_hex, codec.hashout_from_hex)
    info = codec.block_info_from_json(fixture, address_codec=wide)
    assert info["header"]["block_number"] == 2 and len(info["transactions"]) == 2 and info["address_list"][1]["is_valid"]
    assert json.loads(codec.block_info_to_json(info, address_codec=wide)) == fixture
    with pytest.raises(codec.CodecError):  # the current 8-byte Address codec rejects that file, as the reference's would
        codec.block_info_from_json(fixture)


def test_proof_flat_struct_bytes_roundtrip(codec, oracle):
    """the flat proof of the prover <-> plonky2's Proof structure <-> its byte format: lengths agree with the library's own
    layout (glo_plonk_proof_len), every Merkle path carries its length byte, and all three forms round-trip"""
    import importlib

    import numpy as np

    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    b = circ.CircuitBuilder()
    x = [b.virtual() for _ in range(4)]
    h = b.two_to_one(x, x)
    for _ in range(6):
        h = b.two_to_one(h, x)
    b.register_public_inputs(h)
    cd = b.build()
    wires, pis = cd.generate_witness({t: 5 + i for i, t in enumerate(x)}, oracle.poseidon_gate_rows)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"])
    arity = oracle.fri_reduction_arity_bits(cd.degree_bits, 3, 4)
    fp = (cd.degree_bits, 3, 4, 6, 5, arity)  # 6 PoW bits, 5 queries: a quick proof
    flat = oracle.plonk_prove(cd, cs, wires, pis, fp)
    layout = codec.ProofLayout(cd, arity, num_query_rounds=5)
    proof = codec.proof_from_flat(flat, layout)
    assert codec.proof_to_flat(proof) == [int(v) for v in flat]
    assert len(proof["openings"]["wires"]) == 135 and len(proof["opening_proof"]["query_round_proofs"]) == 5
    q0 = proof["opening_proof"]["query_round_proofs"][0]
    lde_bits = cd.degree_bits + 3
    assert [len(t["evals"]) for t in q0["initial_trees_proof"]] == layout.oracle_cols
    assert all(len(t["siblings"]) == max(lde_bits - 4, 0) for t in q0["initial_trees_proof"])
    data = codec.proof_with_public_inputs_to_bytes(proof, pis)
    n_paths = 5 * (4 + len(arity))
    assert len(data) == 8 * (len(flat) + len(pis)) + n_paths  # one length byte per Merkle path, nothing else added
    proof2, pis2 = codec.proof_with_public_inputs_from_bytes(data, layout, len(pis))
    assert proof2 == proof and pis2 == [int(p) for p in pis]
    assert oracle.plonk_verify(cd, cs["cap"], pis2, np.array(codec.proof_to_flat(proof2), dtype=np.uint64), fp) == 0
    # the array form used by the aggregation workloads: same bytes, same words back
    assert codec.flat_proof_to_bytes(flat, layout, pis) == data
    flat2, pis3 = codec.flat_proof_from_bytes(data, layout, len(pis))
    assert (flat2 == flat).all() and pis3 == [int(p) for p in pis]
    with pytest.raises(codec.CodecError):
        codec.flat_proof_from_bytes(data[:-1], layout, len(pis))
    with pytest.raises(codec.CodecError):
        codec.proof_with_public_inputs_from_bytes(data[:-1], layout, len(pis))
    bad = bytearray(data)
    bad[8 * (3 * 16 * 4 + 2 * sum(n for _, n in layout.openings) + len(arity) * 16 * 4 + layout.oracle_cols[0])] ^= 1  # a length byte
    with pytest.raises(codec.CodecError):
        codec.proof_with_public_inputs_from_bytes(bytes(bad), layout, len(pis))
    with pytest.raises(codec.CodecError):
        codec.flat_proof_from_bytes(bytes(bad), layout, len(pis))


def test_proof_serde_json_roundtrip(codec, oracle):
    """flat proof -> struct -> plonky2's serde-JSON shape (Proof / ProofWithPublicInputs and the reference's
    SimpleSignatureProofWithPublicInputs, src/zkdsa/circuits/mod.rs:256-265) -> back: same words, still verifies"""
    import importlib

    import numpy as np

    import tests.test_oracle_plonk as top

    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    cd, sk, msg = top.zkdsa_circuit(circ)
    inputs = dict(zip(sk, [1, 2, 3, 4]))
    inputs.update(zip(msg, [5, 6, 7, 8]))
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"])
    arity = oracle.fri_reduction_arity_bits(cd.degree_bits, 3, 4)
    fp = (cd.degree_bits, 3, 4, 6, 3, arity)
    flat = oracle.plonk_prove(cd, cs, wires, pis, fp)
    layout = codec.ProofLayout(cd, arity, num_query_rounds=3)
    proof = codec.proof_from_flat(flat, layout)
    obj = codec.proof_to_serde_json(proof)
    # plonky2's field names and shapes
    assert list(obj) == ["wires_cap", "plonk_zs_partial_products_cap", "quotient_polys_cap", "openings", "opening_proof"]
    assert list(obj["openings"]) == list(codec.OPENING_FIELDS)
    assert set(obj["wires_cap"][0]) == {"elements"} and len(obj["wires_cap"]) == 16
    fri = obj["opening_proof"]
    assert list(fri) == ["commit_phase_merkle_caps", "query_round_proofs", "final_poly", "pow_witness"]
    ep = fri["query_round_proofs"][0]["initial_trees_proof"]["evals_proofs"]
    assert len(ep) == 4 and len(ep[1][0]) == 135 and set(ep[1][1]) == {"siblings"}
    assert set(fri["final_poly"]) == {"coeffs"} and isinstance(fri["pow_witness"], int)
    assert codec.proof_from_serde_json(obj) == proof
    text = codec.proof_with_public_inputs_to_json(proof, pis)
    p2, pis2 = codec.proof_with_public_inputs_from_json(text)
    assert p2 == proof and pis2 == [int(x) for x in pis]
    assert oracle.plonk_verify(cd, cs["cap"], pis2, np.array(codec.proof_to_flat(p2), dtype=np.uint64), fp) == 0
    # the reference's wrapper: public inputs as three HashOuts
    text = codec.simple_signature_proof_to_json(proof, pis)
    assert set(json.loads(text)["public_inputs"]) == {"message", "public_key", "signature"}
    p3, pis3 = codec.simple_signature_proof_from_json(text)
    assert p3 == proof and pis3 == [int(x) for x in pis]
    # malformed input is reported, not swallowed
    bad = json.loads(codec.proof_with_public_inputs_to_json(proof, pis))
    bad["proof"]["openings"]["wires"][0] = [1]
    with pytest.raises(codec.CodecError):
        codec.proof_with_public_inputs_from_json(json.dumps(bad))
    bad = json.loads(codec.proof_with_public_inputs_to_json(proof, pis))
    bad["proof"]["wires_cap"][0]["elements"][0] = codec.P if hasattr(codec, "P") else 0xFFFFFFFF00000001
    with pytest.raises(codec.CodecError):
        codec.proof_with_public_inputs_from_json(json.dumps(bad))
    del bad["proof"]["opening_proof"]["pow_witness"]
    with pytest.raises(codec.CodecError):
        codec.proof_with_public_inputs_from_json(json.dumps(bad))


def test_proof_forms_roundtrip_under_random_layouts(codec):
    """CPU only, no prover: 200 random layouts (degree, cap height, blow-up 8, query rounds, challenges, wire / constant counts, FRI
    reduction schedules incl. none) filled with random canonical words — flat <-> struct <-> bytes <-> serde JSON all round-trip, the two
    byte encoders agree, and a truncated or length-corrupted blob is refused"""
    import types

    import numpy as np

    P = 0xFFFFFFFF00000001
    for case in range(200):
        rng = np.random.default_rng(88000 + case)
        db = int(rng.integers(1, 14))
        cap_h = int(rng.integers(0, min(5, db + 3) + 1))
        arity, d, lde = [], db, db + 3
        stop = float(rng.choice([0.0, 0.2, 0.6]))
        while d > 0 and rng.random() >= stop:
            a = int(rng.integers(1, 5))
            if a > d or lde - a < cap_h:
                break
            arity.append(a)
            d, lde = d - a, lde - a
        cfg = types.SimpleNamespace(num_wires=int(rng.integers(4, 140)), num_routed_wires=int(rng.integers(1, 81)),
                                    num_challenges=int(rng.integers(1, 3)), rate_bits=3, cap_height=cap_h,
                                    num_query_rounds=int(rng.integers(1, 9)))
        cd = types.SimpleNamespace(cfg=cfg, degree_bits=db, num_constants=int(rng.integers(1, 9)), quotient_degree_factor=8,
                                   num_partial_products=int(rng.integers(0, 10)))
        layout = codec.ProofLayout(cd, arity)
        _, total = codec._path_marks(layout)
        # the library's own proof length for the same shape (host-only entry point: no GPU)
        import ctypes

        import intmax_zkp_core_amd as pkg

        cc = pkg._CircuitC(db, cfg.num_wires, cfg.num_routed_wires, cd.num_constants, 1, cfg.num_challenges, 8, cd.num_partial_products, 0, 0,
                           None, None)
        cc.fri = pkg.FriParams(db, 3, cap_h, 16, cfg.num_query_rounds, reduction_arity_bits=arity).to_c()
        f = pkg.lib().glp_proof_len
        f.restype = ctypes.c_size_t
        assert int(f(ctypes.byref(cc))) == total, (case, db, cap_h, arity)
        flat = rng.integers(0, P, total, dtype=np.uint64)
        pis = [int(x) for x in rng.integers(0, P, int(rng.integers(0, 6)), dtype=np.uint64)]
        what = (case, db, cap_h, arity, cfg.num_query_rounds)
        proof = codec.proof_from_flat(flat, layout)
        assert codec.proof_to_flat(proof) == [int(v) for v in flat], what
        data = codec.proof_with_public_inputs_to_bytes(proof, pis)
        assert data == codec.flat_proof_to_bytes(flat, layout, pis), what
        back, pis2 = codec.flat_proof_from_bytes(data, layout, len(pis))
        assert (back == flat).all() and pis2 == pis, what
        proof2, pis3 = codec.proof_with_public_inputs_from_bytes(data, layout, len(pis))
        assert proof2 == proof and pis3 == pis, what
        proof3, pis4 = codec.proof_with_public_inputs_from_json(codec.proof_with_public_inputs_to_json(proof, pis))
        assert codec.proof_to_flat(proof3) == [int(v) for v in flat] and [int(x) for x in pis4] == pis, what
        with pytest.raises(codec.CodecError):
            codec.flat_proof_from_bytes(data + b"\\x00", layout, len(pis))
        marks, _ = codec._path_marks(layout)
        if marks and marks[0][1] < 255:
            bad = bytearray(data)
            bad[8 * marks[0][0]] ^= 0x80  # the first path's length byte
            with pytest.raises(codec.CodecError):
                codec.flat_proof_from_bytes(bytes(bad), layout, len(pis))


def test_block_production_public_inputs_roundtrip():
    """src/rollup/circuits/mod.rs::test_encode_block_production_public_inputs: random inputs, 4 transactions and 2 + 2 + 2 deposits /
    flags survive encode -> decode; the length is 5 n_txs + 13 (deposits + flags) + 28"""
    import importlib

    import numpy as np

    import __graft_entry__ as g

    g.load_package()
    codec = importlib.import_module("intmax_zkp_core_amd.codec")
    rng = np.random.default_rng(12)
    P = 0xFFFFFFFF00000001
    dep = lambda: {"receiver_address": int(rng.integers(0, P, dtype=np.uint64)), "contract_address": int(rng.integers(0, P, dtype=np.uint64)),  # noqa: E731
                   "variable_index": int(rng.integers(0, 256)), "amount": int(rng.integers(0, P, dtype=np.uint64))}
    pi = {"address_list": [{"sender_address": int(rng.integers(0, P, dtype=np.uint64)), "is_valid": bool(rng.integers(0, 2))} for _ in range(4)],
          "deposit_list": [dep(), dep()], "scroll_flag_list": [dep(), dep()], "polygon_flag_list": [dep(), dep()]}
    for name in codec.BLOCK_PI_ROOTS:
        pi[name] = [int(x) for x in rng.integers(0, P, 4, dtype=np.uint64)]
    words = codec.block_production_public_inputs_encode(pi)
    assert len(words) == 5 * 4 + 13 * 6 + 28
    assert codec.block_production_public_inputs_decode(words, 4, 2, 2, 2) == pi
    import pytest

    with pytest.raises(codec.CodecError):
        codec.block_production_public_inputs_decode(words[:-1], 4, 2, 2, 2)
    # Address::read (src/zkdsa/account.rs:140) reads limb 0 and skips limbs 1..3 without looking at them: words the reference decodes
    # are decoded here too, to the same value
    loose = list(words)
    loose[1], loose[2], loose[3] = 7, 8, 9
    assert codec.block_production_public_inputs_decode(loose, 4, 2, 2, 2) == pi
