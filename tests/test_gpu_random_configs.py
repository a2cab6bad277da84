"""Whole proofs under RANDOM prover configurations: circuit size, cap height, grinding bits, number of query
rounds, number of challenges and — the part the standard configuration never varies — the FRI reduction schedule (arities of 2, 4, 8
and 16 mixed in one proof, early stop with a long final polynomial, no reduction at all).  For every drawn configuration the GPU
proof equals the oracle's word for word and the restated verifier accepts it.  The draws are seeded: a failure names its case."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
N_CASES = int(os.environ.get("GLP_RANDOM_CASES", "28"))  # a soak run raises it: GLP_RANDOM_CASES=600 GLP_RANDOM_SEED0=5000
SEED0 = int(os.environ.get("GLP_RANDOM_SEED0", "1000"))


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    ctx = pkg.Context(0)
    yield pkg, circ, ctx
    ctx.close()


def draw(case):
    rng = np.random.default_rng(SEED0 + case)
    n_hashes = int(rng.choice([2, 5, 11, 23, 40, 90, 150, 260]))
    cfgd = dict(n_hashes=n_hashes, with_arith=bool(rng.integers(0, 2)), num_challenges=int(rng.choice([1, 2, 2])),
                cap_height=int(rng.integers(0, 6)), pow_bits=int(rng.choice([0, 1, 7, 12, 16, 18])),
                num_queries=int(rng.choice([1, 2, 5, 13, 28, 33])), min_bits=int(rng.choice([0, 0, 6, 9])),
                quot_fused=str(int(rng.integers(0, 2))), seed=int(rng.integers(0, 1 << 30)))
    return cfgd, rng


def schedule(rng, degree_bits, rate_bits, cap_height):
    """a random reduction schedule plonky2 would accept: every round's tree keeps at least 2^cap_height leaves, the final polynomial
    at least one coefficient; sometimes no rounds at all, sometimes reduced to the end"""
    out, d, lde = [], degree_bits, degree_bits + rate_bits
    stop_p = float(rng.choice([0.0, 0.15, 0.4]))
    while d > 0 and rng.random() >= stop_p:
        a = int(rng.integers(1, 5))
        if a > d or lde - a < cap_height:
            break
        out.append(a)
        d -= a
        lde -= a
    return out


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_configuration(env, oracle, case):
    from tests.test_gpu_plonk import hash_chain_circuit

    pkg, circ, ctx = env
    cfgd, rng = draw(case)
    saved = (circ.CircuitConfig.num_challenges, circ.CircuitConfig.cap_height)
    os.environ["GLP_QUOT_FUSED"] = cfgd["quot_fused"]
    try:
        circ.CircuitConfig.num_challenges = cfgd["num_challenges"]
        b, seed, sibs = hash_chain_circuit(circ, cfgd["n_hashes"], cfgd["with_arith"])
        cd0 = b.build(cfgd["min_bits"])  # to learn degree_bits: the cap cannot be taller than the LDE tree
        cap_h = min(cfgd["cap_height"], cd0.degree_bits + 3)
        circ.CircuitConfig.cap_height = cap_h
        b, seed, sibs = hash_chain_circuit(circ, cfgd["n_hashes"], cfgd["with_arith"])
        cd = b.build(cfgd["min_bits"])
        assert cd.cfg.cap_height == cap_h and cd.cfg.num_challenges == cfgd["num_challenges"]
        wr = np.random.default_rng(cfgd["seed"])
        inputs = {t: int(v) for t, v in zip(seed, wr.integers(0, P, 4, dtype=np.uint64))}
        for s in sibs:
            inputs.update({t: int(v) for t, v in zip(s, wr.integers(0, P, 4, dtype=np.uint64))})
        wires, pis = cd.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
        arities = schedule(rng, cd.degree_bits, 3, cap_h)
        fp = pkg.FriParams(cd.degree_bits, 3, cap_h, cfgd["pow_bits"], cfgd["num_queries"], reduction_arity_bits=arities)
        what = "case %d: 2^%d rows, %s, arities %s" % (case, cd.degree_bits, cfgd, arities)
        pc = pkg.ProverCircuit(ctx, cd, fri_params=fp)
        proof = pc.prove(ctx.to_device(wires), pis)
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, cap_h)
        assert (pc.constants_sigmas_cap.elements == cs["cap"]).all(), what
        want = oracle.plonk_prove(cd, cs, wires, pis, fri_params=fp.as_tuple())
        assert proof.shape == want.shape and (proof == want).all(), what
        assert oracle.plonk_verify(cd, cs["cap"], pis, proof, fri_params=fp.as_tuple()) == 0, what
        # plonky2's proof forms under this configuration (N1): flat -> struct -> bytes -> struct -> flat, flat -> bytes -> flat, serde JSON
        codec = importlib.import_module(pkg.__name__ + ".codec")
        layout = codec.ProofLayout(cd, arities, num_query_rounds=cfgd["num_queries"], cap_height=cap_h)
        st = codec.proof_from_flat(proof, layout)
        assert (codec.proof_to_flat(st) == proof).all(), what
        blob = codec.proof_with_public_inputs_to_bytes(st, pis)
        assert blob == codec.flat_proof_to_bytes(proof, layout, pis), what
        back, pis_back = codec.flat_proof_from_bytes(blob, layout, len(pis))
        assert (back == proof).all() and [int(x) for x in pis_back] == [int(x) for x in pis], what
        st2, pis2 = codec.proof_with_public_inputs_from_json(codec.proof_with_public_inputs_to_json(st, pis))
        assert (codec.proof_to_flat(st2) == proof).all() and [int(x) for x in pis2] == [int(x) for x in pis], what
        # the same circuit and witness through the batched prover (two copies): same bytes
        import torch

        wd = ctx.to_device(wires)
        both = pc.prove_batch(torch.stack([wd, wd]).contiguous(), [pis, pis])
        assert (both[0] == proof).all() and (both[1] == proof).all(), what
        pc.free()
    finally:
        circ.CircuitConfig.num_challenges, circ.CircuitConfig.cap_height = saved
        del os.environ["GLP_QUOT_FUSED"]
