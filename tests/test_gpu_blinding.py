"""blinding = true through the C ABI (the TODOs at reference src/zkdsa/circuits/mod.rs:412,
src/rollup/circuits/mod.rs:1283,1298 point at the zero-knowledge configuration): PolynomialBatch::from_values / from_coeffs with
the CALLER's salts equal the oracle's salted commitments bit for bit (cap, leaves with their salt words, paths), the library's own
salts give valid trees, and CircuitData::prove under config.zero_knowledge equals the oracle's proof word for word at 2^12 and 2^16
rows with fixed salts; the restated verifier accepts."""
import importlib

import numpy as np
import pytest

from tests.conftest import timed_leg

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g

    return g.load_package()


@pytest.fixture(scope="module")
def circ(pkg):
    return importlib.import_module("intmax_zkp_core_amd.circuit")


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


SHAPES = [(1, 0, 1, 0), (3, 2, 3, 1), (5, 4, 3, 2), (20, 7, 3, 4), (135, 9, 3, 4), (135, 12, 3, 4), (16, 13, 3, 4), (2, 15, 1, 0),
          (84, 10, 2, 3)]


@pytest.mark.parametrize("cols,log_n,rb,cap", SHAPES)
def test_salted_commitments_equal_oracle(pkg, ctx, oracle, cols, log_n, rb, cap):
    """every leaf-hash path (16 lanes per leaf, 4 lanes, one lane) with leaves of n_cols + 4 words; host and device inputs"""
    rng = np.random.default_rng(cols * 100 + log_n)
    n, N = 1 << log_n, 1 << (log_n + rb)
    v = rng.integers(0, 2**64, (cols, n), dtype=np.uint64)
    salts = rng.integers(0, 2**64, (4, N), dtype=np.uint64)
    with timed_leg(oracle, log_n, threshold=12):
        want = oracle.batch_from_values(v, rb, cap, salts=salts)
        want_c = oracle.batch_from_coeffs(want["coeffs"], rb, cap, salts=salts)
    idx = sorted(set([0, N - 1, N // 2] + [int(x) for x in rng.integers(0, N, 9)]))
    for form in ("host", "device", "coeffs_host", "coeffs_device"):
        data = want["coeffs"] if form.startswith("coeffs") else v
        if form.endswith("device"):
            b = (pkg.PolynomialBatch.from_coeffs if form.startswith("coeffs") else pkg.PolynomialBatch.from_values)(
                ctx, ctx.to_device(data), rb, True, cap, salts=ctx.to_device(salts))
        else:
            b = (pkg.PolynomialBatch.from_coeffs if form.startswith("coeffs") else pkg.PolynomialBatch.from_values)(
                ctx, data, rb, True, cap, salts=salts)
        assert b.blinding and b.salt_cols == 4 and b.n_cols == cols
        assert (b.cap.elements == (want_c if form.startswith("coeffs") else want)["cap"]).all(), form
        rows, paths = b.leaves(idx)
        assert rows.shape == (len(idx), cols + 4)
        assert (rows == want["leaves"][idx]).all(), form
        for k, p in enumerate(idx):
            assert (paths[k] == oracle.merkle_prove(want["digests"], N, cap, p)).all()
        assert (b.get_lde_values(3 % N) == want["leaves"][int(format(3 % N, "0%db" % (log_n + rb))[::-1], 2) if log_n + rb else 0][:cols]).all()
        b.free()


def test_library_salts_are_fresh_valid_and_reproducible(pkg, ctx, oracle):
    """blinding = 1 without caller salts (what plonky2 does): no GLP_ERR_UNSUPPORTED any more; the polynomial part of the leaves is
    the plain commitment's, the salt words are field elements that differ from leaf to leaf and from commitment to commitment, every
    path verifies against the cap; glp_ctx_set_blinding_seed makes the stream reproducible"""
    rng = np.random.default_rng(3)
    cols, log_n, rb, cap = 7, 10, 3, 4
    N = 1 << (log_n + rb)
    v = rng.integers(0, P, (cols, 1 << log_n), dtype=np.uint64)
    plain = pkg.PolynomialBatch.from_values(ctx, v, rb, False, cap)
    a = pkg.PolynomialBatch.from_values(ctx, v, rb, True, cap)
    b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(v), rb, True, cap)
    idx = list(range(0, N, 97))
    ra, pa = a.leaves(idx)
    rb_, _ = b.leaves(idx)
    rp, _ = plain.leaves(idx)
    assert (ra[:, :cols] == rp).all() and (rb_[:, :cols] == rp).all()
    assert (ra[:, cols:] < np.uint64(P)).all()
    assert len({tuple(r) for r in ra[:, cols:]}) == len(idx) and not (ra[:, cols:] == rb_[:, cols:]).any()
    assert not (a.cap.elements == b.cap.elements).all() and not (a.cap.elements == plain.cap.elements).all()
    for k, p in enumerate(idx[:20]):
        assert oracle.merkle_verify(ra[k], p, pa[k], a.cap.elements, cap)
    # the whole salt columns reproduce the cap through the oracle
    all_rows, _ = a.leaves(list(range(N)))
    nat = np.zeros((4, N), dtype=np.uint64)
    for p in range(N):
        nat[:, int(format(p, "0%db" % (log_n + rb))[::-1], 2)] = all_rows[p, cols:]
    assert (oracle.batch_from_values(v, rb, cap, salts=nat)["cap"] == a.cap.elements).all()
    for x in (a, b, plain):
        x.free()
    ctx.set_blinding_seed([1, 2, 3, 4])
    c1 = pkg.PolynomialBatch.from_values(ctx, v, rb, True, cap)
    c2 = pkg.PolynomialBatch.from_values(ctx, v, rb, True, cap)
    ctx.set_blinding_seed([1, 2, 3, 4])
    d1 = pkg.PolynomialBatch.from_values(ctx, v, rb, True, cap)
    assert (c1.cap.elements == d1.cap.elements).all() and not (c1.cap.elements == c2.cap.elements).all()
    for x in (c1, c2, d1):
        x.free()


@pytest.mark.parametrize("bits,layers", [(6, 2), (12, 8), (16, 16)])
def test_zero_knowledge_proof_equals_oracle(pkg, circ, ctx, oracle, bits, layers):
    """config.zero_knowledge with fixed salts: GPU proof == oracle proof word for word (device and host witnesses), the verifier
    accepts; with the library's own salts the verifier accepts and two proofs differ; K proofs in one call are valid too"""
    mc = circ.MixedGateCircuit(bits, layers)
    cd = mc.data
    fp = pkg.FriParams(cd.degree_bits, flags=pkg.FLAG_ZERO_KNOWLEDGE)
    pc = pkg.ProverCircuit(ctx, cd, fp)
    assert pc.proof_len() == pkg.ProverCircuit(ctx, cd).proof_len() + 28 * 3 * 4
    wires, pis = mc.witness_device(ctx, pkg, seed=bits)
    N = cd.n << 3
    rng = np.random.default_rng(bits)
    salts = rng.integers(0, P, (3, 4, N), dtype=np.uint64)
    proof = pc.prove(wires, pis, salts=ctx.to_device(salts))
    w = ctx.to_host(wires)
    with timed_leg(oracle, bits):
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
        want = oracle.plonk_prove(cd, cs, w, pis, fp.as_tuple(), salts=salts)
    bad = np.nonzero(proof != want)[0]
    assert bad.size == 0, "first differing proof word %d of %d" % (int(bad[0]), proof.size)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fp.as_tuple()) == 0
    if bits <= 12:
        assert (pc.prove(w, pis, salts=salts) == proof).all()  # host pointers (glp_prove_salted)
    # the library's own salts
    p1, p2 = pc.prove(wires, pis), pc.prove(wires, pis)
    assert not (p1 == p2).all()
    for p_ in (p1, p2):
        assert oracle.plonk_verify(cd, cs["cap"], pis, p_, fp.as_tuple()) == 0
    if bits <= 12:
        import torch

        ws = torch.stack([wires, wires, wires])
        many = pc.prove_batch(ws, [pis] * 3)
        for k in range(3):
            assert oracle.plonk_verify(cd, cs["cap"], pis, many[k], fp.as_tuple()) == 0
        # a plain circuit refuses salts, a zero-knowledge proof is not a plain proof
        plain = pkg.ProverCircuit(ctx, cd)
        with pytest.raises(pkg.GlproverError):
            plain.prove(wires, pis, salts=ctx.to_device(salts))
        assert oracle.plonk_verify(cd, cs["cap"], pis, proof[:plain.proof_len()]) != 0
        plain.free()
    pc.free()


def test_zero_knowledge_proof_codec_roundtrip(pkg, circ, ctx, oracle):
    """plonky2's proof structure / byte form of a zero-knowledge proof: the initial-tree `evals` of the blinded oracles are 4 words
    longer (codec.ProofLayout(zero_knowledge=True))"""
    codec = importlib.import_module("intmax_zkp_core_amd.codec")
    mc = circ.MixedGateCircuit(7, 2)
    cd = mc.data
    fp = pkg.FriParams(cd.degree_bits, flags=pkg.FLAG_ZERO_KNOWLEDGE)
    pc = pkg.ProverCircuit(ctx, cd, fp)
    wires, pis = mc.witness_device(ctx, pkg, seed=1)
    flat = pc.prove(wires, pis)
    lay = codec.ProofLayout(cd, fp.reduction_arity_bits, zero_knowledge=True)
    st = codec.proof_from_flat(flat, lay)
    q0 = st["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"]
    assert [len(e["evals"]) for e in q0] == lay.oracle_cols == [cd.num_constants + 80, 139, 24, 20]
    assert (np.asarray(codec.proof_to_flat(st), dtype=np.uint64) == flat).all()
    data = codec.flat_proof_to_bytes(flat, lay, pis)
    back, pis2 = codec.flat_proof_from_bytes(data, lay, len(pis))
    assert (back == flat).all() and list(pis2) == list(pis)
    pc.free()
