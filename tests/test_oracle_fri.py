"""Self-consistency of the oracle's transcript / openings / FRI restatement (the reference holds no vectors for them,
SURVEY.md §8c: 'parity unpinned'): a proof made by the restated prover is accepted by the restated verifier, and any
corruption is rejected at the expected check."""
import numpy as np
import pytest

P = 0xFFFFFFFF00000001


def test_challenger_rules(oracle):
    ch = oracle.Challenger()
    # no input yet: get_challenge duplexes the zero state; challenges pop from the END of state[0..8]
    z = oracle.permute(np.zeros(12, dtype=np.uint64))
    assert ch.get() == int(z[7]) and ch.get() == int(z[6])
    # observing invalidates buffered outputs; 3 pending inputs overwrite state[0..3] at the next squeeze
    ch.observe([1, 2, 3])
    st = z.copy()
    st[:3] = [1, 2, 3]
    st = oracle.permute(st)
    assert [int(x) for x in ch.get_n(3)] == [int(st[7]), int(st[6]), int(st[5])]
    # 8 observations trigger a duplex immediately
    ch2 = oracle.Challenger()
    ch2.observe(list(range(10, 18)))
    s = np.zeros(12, dtype=np.uint64)
    s[:8] = list(range(10, 18))
    s = oracle.permute(s)
    assert ch2.state_words()[0] == [int(x) for x in s]
    assert ch2.get() == int(s[7])
    # extension challenge = 2 pops: [first, second]
    ch3, ch4 = oracle.Challenger(), oracle.Challenger()
    e = ch3.get_ext()
    assert [int(e[0]), int(e[1])] == [ch4.get(), ch4.get()]


def test_poly_eval_ext(oracle):
    rng = np.random.default_rng(1)
    c = rng.integers(0, P, 33, dtype=np.uint64)
    z = rng.integers(0, P, 2, dtype=np.uint64)
    # base-field point embeds: p((x,0)) = (p(x), 0)
    x = int(z[0])
    got = oracle.poly_eval_ext(c, [x, 0])
    assert int(got[0]) == oracle.poly_eval(c, x) and int(got[1]) == 0
    # linear poly: c0 + c1 z
    got = oracle.poly_eval_ext(c[:2], z)
    assert int(got[0]) == (int(c[0]) + int(c[1]) * int(z[0])) % P and int(got[1]) == int(c[1]) * int(z[1]) % P


def make_instance(oracle, degree_bits, cols, seed, rate_bits=3, cap_h=2, pow_bits=4, num_queries=5):
    rng = np.random.default_rng(seed)
    n = 1 << degree_bits
    oracles = [oracle.batch_from_values(rng.integers(0, P, (c, n), dtype=np.uint64), rate_bits, cap_h) for c in cols]
    ch = oracle.Challenger()
    for o in oracles:
        ch.observe(o["cap"])
    zeta = ch.get_ext()
    g = oracle.lib().glo_root_of_unity(degree_bits)
    gz = np.array([int(zeta[0]) * g % P, int(zeta[1]) * g % P], dtype=np.uint64)
    all_polys = [(o, i) for o, c in enumerate(cols) for i in range(c)]
    next_polys = [(len(cols) - 1, 0), (len(cols) - 1, 1)] if cols[-1] >= 2 else [(len(cols) - 1, 0)]
    batches = [(zeta, all_polys), (gz, next_polys)]
    openings = [np.array([oracle.poly_eval_ext(oracles[o]["coeffs"][i], pt) for o, i in polys]) for pt, polys in batches]
    for op in openings:
        ch.observe(op)
    arity = oracle.fri_reduction_arity_bits(degree_bits, rate_bits, cap_h)
    params = (degree_bits, rate_bits, cap_h, pow_bits, num_queries, arity)
    return oracles, batches, openings, params, ch


@pytest.mark.parametrize("degree_bits,cols", [(3, [2, 3]), (6, [3, 5, 2]), (10, [4, 9, 2, 2])])
def test_fri_prove_verify_roundtrip(oracle, degree_bits, cols):
    oracles, batches, openings, params, ch = make_instance(oracle, degree_bits, cols, seed=degree_bits)
    ch_v = ch.clone()
    proof = oracle.fri_prove(batches, oracles, params, ch)
    assert len(proof) == oracle.fri_proof_len(params, cols)
    caps = [o["cap"] for o in oracles]
    assert oracle.fri_verify(batches, openings, caps, cols, params, ch_v.clone(), proof) == 0
    # prover and verifier transcripts end in the same state
    chk = ch_v.clone()
    oracle.fri_verify(batches, openings, caps, cols, params, chk, proof)
    assert chk.state_words() == ch.state_words()
    # deterministic (smallest pow witness)
    oracles2, batches2, openings2, params2, ch2 = make_instance(oracle, degree_bits, cols, seed=degree_bits)
    assert (oracle.fri_prove(batches2, oracles2, params2, ch2) == proof).all()
    # corruption is rejected
    n_rounds = len(params[5])
    bad = proof.copy()
    bad[-1] = (int(bad[-1]) + 1) % P  # pow witness
    assert oracle.fri_verify(batches, openings, caps, cols, params, ch_v.clone(), bad) != 0
    bad = proof.copy()
    bad[-2] ^= np.uint64(1)  # final poly
    assert oracle.fri_verify(batches, openings, caps, cols, params, ch_v.clone(), bad) != 0
    bad = proof.copy()
    off = n_rounds * (4 << params[2])
    bad[off] ^= np.uint64(1)  # first opened row element
    assert oracle.fri_verify(batches, openings, caps, cols, params, ch_v.clone(), bad) in (-2, -1)
    bad_open = [o.copy() for o in openings]
    bad_open[0][0][0] ^= np.uint64(1)  # claimed opening value
    assert oracle.fri_verify(batches, bad_open, caps, cols, params, ch_v.clone(), proof) != 0
    if n_rounds:
        bad = proof.copy()
        bad[0] ^= np.uint64(1)  # commit-phase cap -> different beta
        assert oracle.fri_verify(batches, openings, caps, cols, params, ch_v.clone(), bad) != 0


def test_reduction_strategy(oracle):
    assert oracle.fri_reduction_arity_bits(20, 3, 4) == [4, 4, 4, 4]
    assert oracle.fri_reduction_arity_bits(12, 3, 4) == [4, 4]
    assert oracle.fri_reduction_arity_bits(5, 3, 4) == []
    assert oracle.fri_reduction_arity_bits(3, 3, 4) == []
