"""The vectorised form of the oracle (oracle/fast*.{c,h}: packed Goldilocks on AVX-512 / AVX2 / plain C — what bench.py's
`cpu_baseline` times) against the scalar restatement, which stays the checker: the reference's golden digests, 10^5 random
permutation states, transforms, commitments, FRI instances and whole proofs of every gate type come out identical, under
each backend this CPU can run.  """
import importlib

import numpy as np
import pytest

from tests import test_oracle_fri as tf
from tests import test_oracle_plonk as tp

P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def circ():
    import __graft_entry__ as g

    g.load_package()
    return importlib.import_module("intmax_zkp_core_amd.circuit")


def backends(oracle):
    """the instantiations this CPU runs (a forced backend above the CPU's best falls back to the best)"""
    names = []
    for isa in (512, 256, 1):
        name = oracle.set_fast(False, isa)
        if name not in [n for _, n in names]:
            names.append((isa, name))
    oracle.set_fast(False, 0)
    return names


@pytest.fixture(autouse=True)
def scalar_after(oracle):
    yield
    oracle.set_fast(False, 0)  # never leave the timed leg switched on for the other test modules


def test_backends_present(oracle):
    bs = backends(oracle)
    assert ("scalar" in [n for _, n in bs]) and len(bs) >= 1
    # the GPU box (EPYC 9575F) and this container both have AVX-512; a machine without it still runs the two others
    print("backends:", bs)


def test_goldens_through_the_vector_permutation(oracle, kat):
    """two_to_one(0, 0) and the 32-level zero-hash chain of the reference (src/zkdsa/circuits/mod.rs:85-100,
    src/rollup/circuits/mod.rs:104) recomputed by the packed permutation, the lanes filled with copies and with decoys"""
    z = kat["zero_hash_chain"]
    for isa, name in backends(oracle):
        oracle.set_fast(True, isa)
        st = np.zeros((19, 12), dtype=np.uint64)  # 19: two full AVX-512 groups + a scalar tail
        st[1::2, 0] = np.arange(9, dtype=np.uint64) + np.uint64(5)  # decoys in the odd rows
        out = oracle.permute_many(st)
        for r in range(0, 19, 2):
            assert [int(x) for x in out[r, :4]] == kat["two_to_one_zero"]["digest"], name
        cur = np.zeros(4, dtype=np.uint64)
        for i in range(33):
            hx = oracle.hashout_to_hex(cur)
            if i < 32:
                assert hx == z["siblings"][i], (name, i)
            else:
                assert hx == z["block_headers_digest"], name
            st = np.zeros((8, 12), dtype=np.uint64)
            st[:, :4] = cur
            st[:, 4:8] = cur
            st[3, 5] = np.uint64(77)  # one decoy lane
            got = oracle.permute_many(st)[:, :4]
            assert (got[[0, 1, 2, 4, 5, 6, 7]] == got[0]).all() and not (got[3] == got[0]).all()
            cur = got[0]


def test_golden_trees_through_the_vector_merkle_build(oracle, kat):
    """the zero-hash chain again, as the levels of a tree of 2^10 all-zero leaves built by the packed two_to_one; and the
    sponge over ragged leaf lengths against the scalar hash_or_noop"""
    z = kat["zero_hash_chain"]
    rng = np.random.default_rng(5)
    for isa, name in backends(oracle):
        oracle.set_fast(True, isa)
        dig, cap = oracle.merkle_build(np.zeros((1 << 10, 4), dtype=np.uint64), 0)
        assert oracle.hashout_to_hex(cap[0]) == z["siblings"][10], name
        for leaf_len in (1, 4, 5, 8, 9, 16, 17, 135):
            leaves = rng.integers(0, 2**64, (32, leaf_len), dtype=np.uint64)
            got, _ = oracle.merkle_build(leaves, 2)
            oracle.set_fast(False)
            want, _ = oracle.merkle_build(leaves, 2)
            oracle.set_fast(True, isa)
            assert (got == want).all(), (name, leaf_len)
            assert (got[:4] == oracle.hash_or_noop(leaves[0])).all()


def test_1e5_random_states_equal_scalar(oracle):
    rng = np.random.default_rng(20)
    st = rng.integers(0, 2**64, (100_000, 12), dtype=np.uint64)
    # representation edge cases: non-canonical words, p - 1, p, 2^64 - 1, 2^32 - 1, 2^32, 0
    edge = np.array([P - 1, P, 2**64 - 1, 2**32 - 1, 2**32, 0, 1, P + 5, 0xFFFFFFFF00000000], dtype=np.uint64)
    st[:4096] = rng.choice(edge, (4096, 12))
    ref = oracle.permute_many(st, fast=False)
    assert (ref[:64] == np.array([oracle.permute_slow(s) for s in st[:64]])).all()  # the checker against the textbook form
    for isa, name in backends(oracle):
        oracle.set_fast(True, isa)
        assert (oracle.permute_many(st) == ref).all(), name


@pytest.mark.parametrize("log_n", list(range(0, 17)) + [18])
def test_transforms_equal_scalar(oracle, log_n):
    rng = np.random.default_rng(log_n)
    a = rng.integers(0, 2**64, 1 << log_n, dtype=np.uint64)
    oracle.set_fast(False)
    want = {f: getattr(oracle, f)(a) for f in ("fft", "ifft")}
    want["coset_fft"] = oracle.coset_fft(a, 7)
    want["coset_ifft"] = oracle.coset_ifft(a, 7)
    for isa, name in backends(oracle):
        oracle.set_fast(True, isa)
        assert (oracle.fft(a) == want["fft"]).all(), name
        assert (oracle.ifft(a) == want["ifft"]).all(), name
        assert (oracle.coset_fft(a, 7) == want["coset_fft"]).all(), name
        assert (oracle.coset_ifft(a, 7) == want["coset_ifft"]).all(), name


def test_commitments_equal_scalar_random_shapes(oracle):
    rng = np.random.default_rng(7)
    shapes = [(1, 0, 1, 0), (3, 1, 3, 0), (5, 3, 3, 2), (9, 6, 2, 3), (135, 7, 3, 4), (20, 11, 3, 4), (2, 13, 1, 1), (4, 11, 4, 5)]
    for _ in range(12):
        log_n = int(rng.integers(0, 12))
        rb = int(rng.integers(1, 5))
        shapes.append((int(rng.integers(1, 40)), log_n, rb, int(rng.integers(0, min(6, log_n + rb) + 1))))
    oracle.set_threads(4)
    for cols, log_n, rb, cap in shapes:
        v = rng.integers(0, 2**64, (cols, 1 << log_n), dtype=np.uint64)
        oracle.set_fast(False)
        want = oracle.batch_from_values(v, rb, cap)
        for isa, name in backends(oracle):
            oracle.set_fast(True, isa)
            got = oracle.batch_from_values(v, rb, cap)
            for k in want:
                assert (np.asarray(got[k]) == np.asarray(want[k])).all(), (name, k, cols, log_n, rb, cap)
            got2 = oracle.batch_from_coeffs(np.asarray(want["coeffs"]), rb, cap)
            assert (np.asarray(got2["cap"]) == np.asarray(want["cap"])).all(), (name, cols, log_n, rb, cap)


@pytest.mark.parametrize("degree_bits,cols", [(3, [2, 3]), (6, [4, 1, 2]), (9, [3, 5]), (11, [7])])
def test_fri_proofs_equal_scalar(oracle, degree_bits, cols):
    oracle.set_fast(False)
    oracles, batches, openings, params, ch = tf.make_instance(oracle, degree_bits, cols, seed=degree_bits, pow_bits=9)
    want = oracle.fri_prove(batches, oracles, params, ch.clone())
    for isa, name in backends(oracle):
        oracle.set_fast(True, isa)
        got = oracle.fri_prove(batches, oracles, params, ch.clone())
        assert (np.asarray(got) == np.asarray(want)).all(), name


def whole_proof_cases(oracle, circ):
    cd, sk, msg = tp.zkdsa_circuit(circ)
    inputs = dict(zip(sk, [1, 2, 3, 4]))
    inputs.update(zip(msg, [5, 6, 7, 8]))
    yield ("zkdsa",) + (cd,) + cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    cd, x, u, v = tp.all_gates_circuit(circ)
    yield ("all_gates", cd) + cd.generate_witness({x: 0xABCDE12345, u[0]: 5, u[1]: 7, v[0]: 11, v[1]: 13}, oracle.poseidon_gate_rows)
    cd, t = tp.recursion_gates_circuit(circ)
    yield ("recursion_gates", cd) + cd.generate_witness(tp.recursion_gates_inputs(t), oracle.poseidon_gate_rows)
    for bits, layers in ((6, 2), (7, 4), (10, 8)):
        mc = circ.MixedGateCircuit(bits, layers)
        yield ("mixed_%d" % bits, mc.data) + mc.witness_host(oracle.poseidon_gate_rows, seed=bits)


def test_whole_proofs_equal_scalar(oracle, circ):
    """every gate type, standard FRI parameters (28 queries, 16 grinding bits) and reduced ones: the proof of the timed leg
    is the checker's proof byte for byte, and the restated verifier accepts it"""
    oracle.set_threads(4)
    for name, cd, wires, pis in whole_proof_cases(oracle, circ):
        oracle.set_fast(False)
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
        cd.circuit_digest = oracle.circuit_digest(cs["cap"])
        fps = [None, (cd.degree_bits, 3, 4, 6, 4, oracle.fri_reduction_arity_bits(cd.degree_bits, 3, 4))]
        if cd.degree_bits >= 9:
            fps = fps[:1]
        for fp in fps:
            want = oracle.plonk_prove(cd, cs, wires, pis, fp)
            assert oracle.plonk_verify(cd, cs["cap"], pis, want, fp) == 0
            for isa, bname in backends(oracle):
                oracle.set_fast(True, isa)
                cs_f = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
                got = oracle.plonk_prove(cd, cs_f, wires, pis, fp)
                assert (np.asarray(got) == np.asarray(want)).all(), (name, bname, fp is None)
            oracle.set_fast(False)


@pytest.mark.parametrize("flags", tp.VARIANT_FLAGS)
def test_transcript_variants_equal_scalar(oracle, circ, flags):
    cd, sk, msg = tp.zkdsa_circuit(circ)
    inputs = dict(zip(sk, [9, 2, 3, 4]))
    inputs.update(zip(msg, [5, 6, 7, 1]))
    wires, pis = cd.generate_witness(inputs, oracle.poseidon_gate_rows)
    oracle.set_fast(False)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    fp = (cd.degree_bits, 3, 4, 9, 4, oracle.fri_reduction_arity_bits(cd.degree_bits, 3, 4), flags)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"], cd.degree_bits, flags)
    want = oracle.plonk_prove(cd, cs, wires, pis, fp)
    for isa, name in backends(oracle):
        oracle.set_fast(True, isa)
        assert (np.asarray(oracle.plonk_prove(cd, cs, wires, pis, fp)) == np.asarray(want)).all(), (name, flags)


def test_broken_witness_gives_the_same_rejected_proof(oracle, circ):
    """a zero denominator / unsatisfied constraint takes the same path in both forms (inv(0) = 0 in the chunk quotients)"""
    mc = circ.MixedGateCircuit(6, 2)
    cd = mc.data
    wires, pis = mc.witness_host(oracle.poseidon_gate_rows, seed=1)
    wires = wires.copy()
    wires[3, 17] = (int(wires[3, 17]) + 1) % P
    oracle.set_fast(False)
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
    cd.circuit_digest = oracle.circuit_digest(cs["cap"])
    fp = (cd.degree_bits, 3, 4, 5, 4, oracle.fri_reduction_arity_bits(cd.degree_bits, 3, 4))
    want = oracle.plonk_prove(cd, cs, wires, pis, fp)
    assert oracle.plonk_verify(cd, cs["cap"], pis, want, fp) != 0
    for isa, name in backends(oracle):
        oracle.set_fast(True, isa)
        assert (np.asarray(oracle.plonk_prove(cd, cs, wires, pis, fp)) == np.asarray(want)).all(), name


def test_chunk_quotient_with_a_zero_denominator(oracle, circ):
    """force den = 0 in one chunk of one row (gamma := -(w + beta sigma)): quotient 0 there in both forms"""
    mc = circ.MixedGateCircuit(6, 2)
    cd = mc.data
    wires, _ = mc.witness_host(oracle.poseidon_gate_rows, seed=2)
    sig = cd.sigmas
    beta = 12345
    row, col = 9, 3
    gamma = (-(int(wires[col, row]) + beta * int(sig[col, row]))) % P
    oracle.set_fast(False)
    want = oracle.partial_products_and_zs(cd, wires, [beta, beta + 1], [gamma, gamma + 1])
    for isa, name in backends(oracle):
        oracle.set_fast(True, isa)
        got = oracle.partial_products_and_zs(cd, wires, [beta, beta + 1], [gamma, gamma + 1])
        assert (np.asarray(got) == np.asarray(want)).all(), name
    # the quotient of that chunk really is zero there: pp_0 (challenge 0) at that row is Z * 0
    assert int(want[2, row]) == 0
