"""GPU parity for the opening / FRI phase through the C-ABI: openings at an extension
point and the complete FriProof must equal the oracle's word for word, and the oracle's restated verifier must accept
the GPU proof (the reference itself only has prove -> verify round trips for this phase)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g

    return g.load_package()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def build(pkg, ctx, oracle, degree_bits, cols, seed, rate_bits=3, cap_h=4, pow_bits=8, num_queries=6):
    rng = np.random.default_rng(seed)
    n = 1 << degree_bits
    cap_h = min(cap_h, degree_bits + rate_bits)
    vals = [rng.integers(0, P, (c, n), dtype=np.uint64) for c in cols]
    ref = [oracle.batch_from_values(v, rate_bits, cap_h) for v in vals]
    gpu = [pkg.PolynomialBatch.from_values(ctx, ctx.to_device(v), rate_bits, False, cap_h) for v in vals]
    cho, chg = oracle.Challenger(), pkg.Challenger()
    for r, g in zip(ref, gpu):
        assert (r["cap"] == g.cap.elements).all()
        cho.observe(r["cap"])
        chg.observe_cap(g.cap)
    zeta = cho.get_ext()
    assert (zeta == chg.get_extension_challenge()).all()
    gen = oracle.lib().glo_root_of_unity(degree_bits)
    gz = np.array([int(zeta[0]) * gen % P, int(zeta[1]) * gen % P], dtype=np.uint64)
    all_polys = [(o, i) for o, c in enumerate(cols) for i in range(c)]
    nxt = [(len(cols) - 1, i) for i in range(min(2, cols[-1]))]
    return ref, gpu, cho, chg, zeta, gz, all_polys, nxt, (degree_bits, rate_bits, cap_h, pow_bits, num_queries)


@pytest.mark.parametrize("degree_bits,cols", [(2, [3]), (5, [2, 3]), (9, [4, 6, 2]), (12, [9, 20, 3, 2])])
def test_openings_match_oracle(pkg, ctx, oracle, degree_bits, cols):
    ref, gpu, cho, chg, zeta, gz, all_polys, nxt, _ = build(pkg, ctx, oracle, degree_bits, cols, seed=degree_bits)
    for r, g in zip(ref, gpu):
        got = g.eval_ext(zeta)
        want = np.array([oracle.poly_eval_ext(r["coeffs"][i], zeta) for i in range(g.n_cols)])
        assert (got == want).all()
    # a sub-range at the second point (plonk_zs_next)
    g, r = gpu[-1], ref[-1]
    k = min(2, g.n_cols)
    got = g.eval_ext(gz, 0, k)
    want = np.array([oracle.poly_eval_ext(r["coeffs"][i], gz) for i in range(k)])
    assert (got == want).all()
    for g in gpu:
        g.free()


@pytest.mark.parametrize("degree_bits,cols", [(3, [2, 3]), (6, [3, 5, 2]), (8, [4, 2]), (9, [3, 4, 2]), (10, [4, 9, 2, 2]),
                                              (13, [6, 11, 4, 3])])
@pytest.mark.parametrize("give_openings", [True, False])
def test_fri_proof_matches_oracle(pkg, ctx, oracle, degree_bits, cols, give_openings):
    if not give_openings and degree_bits not in (6, 10):
        pytest.skip("internal-evaluation variant covered on two sizes")
    ref, gpu, cho, chg, zeta, gz, all_polys, nxt, base = build(pkg, ctx, oracle, degree_bits, cols, seed=40 + degree_bits)
    batches = [(zeta, all_polys), (gz, nxt)]
    openings = [np.array([oracle.poly_eval_ext(ref[o]["coeffs"][i], pt) for o, i in polys]) for pt, polys in batches]
    for op in openings:
        cho.observe(op)
        chg.observe_extension_elements(op)
    fp = pkg.FriParams(base[0], base[1], base[2], base[3], base[4])
    params = fp.as_tuple()
    ch_verify = cho.clone()
    want = oracle.fri_prove(batches, ref, params, cho)
    inst = [(pt, polys, openings[i] if give_openings else None) for i, (pt, polys) in enumerate(batches)]
    got = pkg.prove_openings(ctx, inst, gpu, chg, fp)
    assert len(got) == len(want)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing proof word %d of %d" % (bad[0], len(got))
    assert chg.state_words() == cho.state_words()  # transcripts end in the same state
    assert oracle.fri_verify(batches, openings, [r["cap"] for r in ref], cols, params, ch_verify, got) == 0
    for g in gpu:
        g.free()


@pytest.mark.parametrize("degree_bits,rate_bits,cap_h,pow_bits,num_queries", [(7, 1, 0, 0, 3), (9, 2, 2, 20, 5), (10, 4, 3, 5, 9),
                                                                             (11, 3, 4, 12, 28), (6, 3, 6, 1, 4)])
def test_fri_other_parameters(pkg, ctx, oracle, degree_bits, rate_bits, cap_h, pow_bits, num_queries):
    """FRI parameters away from the standard configuration: other blow-ups and cap heights, no proof of work at all (witness 0),
    a 20-bit grind (the search runs over several looks), one bit; GPU proof == oracle proof, verifier accepts."""
    cols = [3, 5, 2]
    ref, gpu, cho, chg, zeta, gz, all_polys, nxt, base = build(pkg, ctx, oracle, degree_bits, cols, seed=70 + degree_bits,
                                                               rate_bits=rate_bits, cap_h=cap_h, pow_bits=pow_bits, num_queries=num_queries)
    batches = [(zeta, all_polys), (gz, nxt)]
    openings = [np.array([oracle.poly_eval_ext(ref[o]["coeffs"][i], pt) for o, i in polys]) for pt, polys in batches]
    for op in openings:
        cho.observe(op)
        chg.observe_extension_elements(op)
    fp = pkg.FriParams(base[0], base[1], base[2], base[3], base[4])
    params = fp.as_tuple()
    ch_verify = cho.clone()
    want = oracle.fri_prove(batches, ref, params, cho)
    got = pkg.prove_openings(ctx, [(pt, polys, openings[i]) for i, (pt, polys) in enumerate(batches)], gpu, chg, fp)
    assert len(got) == len(want) and (got == want).all()
    assert oracle.fri_verify(batches, openings, [r["cap"] for r in ref], cols, params, ch_verify, got) == 0
    for g in gpu:
        g.free()


def test_fri_standard_config_2_12(pkg, ctx, oracle):
    """standard_recursion_config shape (135 wires-like, 20, 16 columns; 16 PoW bits; 28 queries) at n = 2^12:
    the GPU proof is accepted by the oracle verifier and equals the oracle proof."""
    cols = [10, 135, 20, 16]
    ref, gpu, cho, chg, zeta, gz, all_polys, nxt, _ = build(pkg, ctx, oracle, 12, cols, seed=7, pow_bits=16, num_queries=28)
    batches = [(zeta, all_polys), (gz, [(2, 0), (2, 1)])]
    openings = [np.concatenate([gpu[o].eval_ext(pt) for o in range(4)]) if b == 0 else gpu[2].eval_ext(pt, 0, 2)
                for b, (pt, _) in enumerate(batches)]
    for op in openings:
        cho.observe(op)
        chg.observe_extension_elements(op)
    fp = pkg.FriParams(12)
    ch_verify = cho.clone()
    got = pkg.prove_openings(ctx, [(pt, polys, openings[i]) for i, (pt, polys) in enumerate(batches)], gpu, chg, fp)
    assert oracle.fri_verify(batches, openings, [r["cap"] for r in ref], cols, fp.as_tuple(), ch_verify, got) == 0
    want = oracle.fri_prove(batches, ref, fp.as_tuple(), cho)
    assert (got == want).all()
    for g in gpu:
        g.free()


def test_fri_random_instances(pkg, ctx, oracle):
    """60 seeded random FRI instances — 2^1 .. 2^12 coefficients, blow-ups 2 .. 32, cap heights 0 .. 5, grinding 0 .. 17 bits, 1 .. 40
    queries, one to four oracles of 1 .. 11 columns, reduction schedules of arities 2 .. 32 with early stops: GPU proof == oracle proof
    and the restated verifier accepts (300 more in a sweep of the same generator, round 4)"""
    import os

    for case in range(int(os.environ.get("GLP_SWEEP_CASES", "60"))):  # a soak raises it: GLP_SWEEP_CASES=500 GLP_SWEEP_SEED0=7
        rng = np.random.default_rng(52000 + int(os.environ.get("GLP_SWEEP_SEED0", "0")) + case)
        db, rate = int(rng.integers(1, 13)), int(rng.integers(1, 6))
        cap_h = int(rng.integers(0, min(5, db + rate) + 1))
        pow_bits, nq = int(rng.choice([0, 1, 3, 8, 13, 17])), int(rng.choice([1, 2, 7, 28, 40]))
        cols = [int(rng.integers(1, 12)) for _ in range(int(rng.integers(1, 5)))]
        arities, d, lde = [], db, db + rate
        stop = float(rng.choice([0.0, 0.2, 0.5]))
        while d > 0 and rng.random() >= stop:
            a = int(rng.integers(1, 6))
            if a > d or lde - a < cap_h:
                break
            arities.append(a)
            d, lde = d - a, lde - a
        what = (case, db, rate, cap_h, pow_bits, nq, cols, arities)
        ref, gpu, cho, chg, zeta, gz, all_polys, nxt, base = build(pkg, ctx, oracle, db, cols, seed=52000 + case, rate_bits=rate, cap_h=cap_h,
                                                                   pow_bits=pow_bits, num_queries=nq)
        batches = [(zeta, all_polys), (gz, nxt)]
        openings = [np.array([oracle.poly_eval_ext(ref[o]["coeffs"][i], pt) for o, i in polys]) for pt, polys in batches]
        for op in openings:
            cho.observe(op)
            chg.observe_extension_elements(op)
        fp = pkg.FriParams(base[0], base[1], base[2], base[3], base[4], reduction_arity_bits=arities)
        ch_verify = cho.clone()
        want = oracle.fri_prove(batches, ref, fp.as_tuple(), cho)
        got = pkg.prove_openings(ctx, [(pt, polys, openings[i]) for i, (pt, polys) in enumerate(batches)], gpu, chg, fp)
        assert len(got) == len(want) and (got == want).all(), what
        assert oracle.fri_verify(batches, openings, [r["cap"] for r in ref], cols, fp.as_tuple(), ch_verify, got) == 0, what
        for b in gpu:
            b.free()
