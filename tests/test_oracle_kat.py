"""Pins the CPU oracle against every golden value the reference's own tests hold for the hashing path
(tests/golden/poseidon_kat.json; sources listed in tests/golden/make_poseidon_kat.py) and against
size-independent mathematical properties where the reference holds no vectors."""
import numpy as np
import pytest

P = 0xFFFFFFFF00000001


def test_round_constants_head(oracle):
    rc = oracle.round_constants()
    assert [int(x) for x in rc[:4]] == [0xB585F766F2144405, 0x7746A55F43921AD7, 0xB2FB0D31CEE799B4, 0x0F6760A4803427D7]
    assert all(int(x) < P for x in rc)


def test_two_to_one_zero(oracle, kat):
    k = kat["two_to_one_zero"]
    got = oracle.two_to_one(k["left"], k["right"])
    assert [int(x) for x in got] == k["digest"]
    assert [int(x) for x in got] == kat["default_tx_hash"]["digest"]
    assert oracle.hashout_to_hex(got) == kat["two_to_one_zero_hex"]["hex"]


def test_zero_hash_chain(oracle, kat):
    z = kat["zero_hash_chain"]
    h = np.zeros(4, dtype=np.uint64)
    chain = [h]
    for _ in range(32):
        h = oracle.two_to_one(h, h)
        chain.append(h)
    assert len(z["siblings"]) == 32
    for i, s in enumerate(z["siblings"]):
        assert oracle.hashout_to_hex(chain[i]) == s, i
    assert oracle.hashout_to_hex(chain[32]) == z["block_headers_digest"]
    assert oracle.hashout_to_hex(chain[5]) == z["transactions_digest"]
    assert oracle.hashout_to_hex(chain[4]) == z["deposit_digest"]


def test_sender_addresses(oracle, kat):
    a = kat["accounts"]
    for sk, addr in zip(a["private_keys"], a["sender_addresses"]):
        assert oracle.hashout_to_hex(oracle.two_to_one(sk, sk)) == addr


def test_codec(oracle, kat):
    for c in kat["codec"]["cases"]:
        assert oracle.hashout_to_hex(c["limbs"]) == c["hex"]
        assert [int(x) for x in oracle.hashout_from_hex(c["short_hex"])] == c["limbs"]
        assert [int(x) for x in oracle.hashout_from_hex(c["hex"])] == c["limbs"]
    with pytest.raises(ValueError):
        oracle.hashout_from_hex("c71603")


def test_field_mul_matches_slow(oracle):
    rng = np.random.default_rng(1)
    L = oracle.lib()
    vals = [0, 1, P - 1, P, 2**64 - 1, 2**32, 2**32 - 1] + [int(x) for x in rng.integers(0, 2**64, 200, dtype=np.uint64)]
    for a in vals:
        for b in vals[:20]:
            assert L.glo_mul(a, b) == (a * b) % P == L.glo_mul_slow(a, b)
            assert L.glo_add(a, b) == (a + b) % P
            assert L.glo_sub(a, b) == (a - b) % P
    assert L.glo_mul(L.glo_inv(12345), 12345) == 1
    # two-adic root: exact order 2^32 and equal to 7^((p-1)/2^32)
    g = L.glo_root_of_unity(32)
    assert g == 1753635133440165772 == pow(7, (P - 1) >> 32, P)
    assert pow(g, 2**31, P) == P - 1


def test_fast_and_textbook_mds_agree(oracle):
    rng = np.random.default_rng(11)
    for _ in range(100):
        s = rng.integers(0, 2**64, 12, dtype=np.uint64)
        assert (oracle.permute(s) == oracle.permute_slow(s)).all()


def test_sponge_rules(oracle):
    rng = np.random.default_rng(2)
    x = rng.integers(0, P, 135, dtype=np.uint64)
    # overwrite-mode sponge by hand
    st = np.zeros(12, dtype=np.uint64)
    for off in range(0, 135, 8):
        ch = x[off:off + 8]
        st[:len(ch)] = ch
        st = oracle.permute(st)
    assert (oracle.hash_no_pad(x) == st[:4]).all()
    # hash_or_noop copies <= 4 elements
    assert [int(v) for v in oracle.hash_or_noop([5, 6])] == [5, 6, 0, 0]
    assert (oracle.hash_or_noop(x[:5]) == oracle.hash_no_pad(x[:5])).all()
    # hash_pad: [k, v, 1] -> 9 inputs pad to 12: push 1, zeros, 1
    kv = rng.integers(0, P, 9, dtype=np.uint64)
    padded = np.concatenate([kv, np.array([1, 0, 1], dtype=np.uint64)])
    assert (oracle.hash_pad(kv) == oracle.hash_no_pad(padded)).all()
    # non-canonical input is the same field element
    y = x.copy()
    y[3] = np.uint64((int(x[3]) + P) % 2**64) if int(x[3]) + P < 2**64 else x[3]
    assert (oracle.hash_no_pad(y) == oracle.hash_no_pad(x)).all()


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 10])
def test_fft_is_dft(oracle, log_n):
    rng = np.random.default_rng(log_n)
    c = rng.integers(0, P, 1 << log_n, dtype=np.uint64)
    v = oracle.fft(c)
    assert (v == oracle.dft_naive(c)).all()
    assert (oracle.ifft(v) == c).all()
    assert (oracle.coset_ifft(oracle.coset_fft(c)) == c).all()
    # coset_fft(c)[i] = poly(7 * w^i)
    if log_n <= 5:
        w = oracle.lib().glo_root_of_unity(log_n)
        cf = oracle.coset_fft(c)
        for i in range(1 << log_n):
            x = 7 * pow(w, i, P) % P
            assert int(cf[i]) == sum(int(cj) * pow(x, j, P) for j, cj in enumerate(c)) % P


def test_ifft_roundtrip_2_20(oracle):
    rng = np.random.default_rng(20)
    v = rng.integers(0, P, 1 << 20, dtype=np.uint64)
    assert (oracle.fft(oracle.ifft(v)) == v).all()


def test_merkle_and_batch(oracle):
    rng = np.random.default_rng(3)
    C, log_n, rate_bits, cap_h = 7, 5, 3, 2
    vals = rng.integers(0, P, (C, 1 << log_n), dtype=np.uint64)
    b = oracle.batch_from_values(vals, rate_bits, cap_h)
    N = 1 << (log_n + rate_bits)
    # coefficients interpolate the values
    for c in range(C):
        assert (oracle.fft(b["coeffs"][c]) == vals[c]).all()
    # leaves: row rev(i) holds every column evaluated at 7*w_N^i
    w = oracle.lib().glo_root_of_unity(log_n + rate_bits)
    for i in (0, 1, 5, N - 1):
        x = 7 * pow(w, i, P) % P
        row = b["leaves"][oracle.reverse_bits(i, log_n + rate_bits)]
        for c in range(C):
            assert int(row[c]) == sum(int(cj) * pow(x, j, P) for j, cj in enumerate(b["coeffs"][c])) % P
    # every Merkle proof verifies against the cap; a corrupted leaf does not
    for i in (0, 3, N // 2, N - 1):
        sib = oracle.merkle_prove(b["digests"], N, cap_h, i)
        assert sib.shape == (log_n + rate_bits - cap_h, 4)
        assert oracle.merkle_verify(b["leaves"][i], i, sib, b["cap"], cap_h)
        bad = b["leaves"][i].copy()
        bad[0] ^= np.uint64(1)
        assert not oracle.merkle_verify(bad, i, sib, b["cap"], cap_h)
    # cap[i] is the root of the i-th contiguous block of leaves
    sub = b["leaves"][: N >> cap_h]
    _, subcap = oracle.merkle_build(sub, 0)
    assert (subcap[0] == b["cap"][0]).all()
    # from_coeffs agrees with from_values
    b2 = oracle.batch_from_coeffs(b["coeffs"], rate_bits, cap_h)
    assert (b2["cap"] == b["cap"]).all() and (b2["leaves"] == b["leaves"]).all()
    # narrow leaves (<= 4 columns) are copied, not hashed
    v3 = rng.integers(0, P, (3, 4), dtype=np.uint64)
    b3 = oracle.batch_from_values(v3, 1, 0)
    assert (b3["digests"][:3] == b3["leaves"][0]).all() and int(b3["digests"][3]) == 0


def smt_layered_leaf(hash_pad, entry):
    """LayeredLayered SMT with a single entry (k1, k2, k3) -> v: each lower root is the value of the tree above
    (reference src/sparse_merkle_tree/layered_layered_tree.rs; a one-leaf tree's root is the leaf hash, tree.rs:255-350)"""
    k1, k2, k3 = entry["key"]
    node = [entry["value"], 0, 0, 0]
    for k in (k3, k2, k1):
        node = [int(t) for t in hash_pad([k, 0, 0, 0] + node + [1])]
    return node


def test_tx_hash_goldens_pin_hash_pad_and_smt_rules(oracle, kat):
    t = kat["transactions"]
    la, lb = (smt_layered_leaf(oracle.hash_pad, e) for e in t["diff_tree_entries"])
    # top-level keys 407 (bit 0 = 1 -> right) and 832 (bit 0 = 0 -> left) split at depth 0
    k_a, k_b = t["diff_tree_entries"][0]["key"][0], t["diff_tree_entries"][1]["key"][0]
    assert (k_a & 1, k_b & 1) == (1, 0)
    diff_root = oracle.two_to_one(lb, la)
    for nonce, want in zip(t["nonces"], t["tx_hashes"]):
        assert oracle.hashout_to_hex(oracle.two_to_one(diff_root, nonce)) == want
