"""The bulk synthetic circuit used by bench.py (layered PoseidonGate trace with copy constraints and a public-input
hash) proves and verifies; at small size the GPU proof also equals the oracle's."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    ctx = pkg.Context(0)
    yield pkg, circ, ctx
    ctx.close()


@pytest.mark.parametrize("degree_bits,layers", [(5, 3), (9, 4), (12, 8), (14, 16), (17, 16)])
def test_layered_circuit_proves_and_verifies(env, oracle, degree_bits, layers):
    pkg, circ, ctx = env
    lc = circ.LayeredPoseidonCircuit(degree_bits, layers)
    cd = lc.data
    pc = pkg.ProverCircuit(ctx, cd)
    wires, pis = lc.witness_device(ctx, pkg, seed=degree_bits)
    # public inputs really are [H-chain digest, first inputs] and the in-circuit hash feeds the PublicInputGate
    w = ctx.to_host(wires)
    assert [int(v) for v in w[0:4, lc.row_pi_gate]] == [int(v) for v in oracle.hash_no_pad(pis)]
    proof = pc.prove(wires, pis)
    assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, proof) == 0
    bad = list(pis)
    bad[0] = (bad[0] + 1) % pkg.P
    assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, bad, proof) != 0
    if degree_bits <= 17:  # word for word against the CPU prover (2^14 rows and up: the specialised NTT passes, the tiled
        # reorders and the lane-per-leaf sponges inside a whole proof; 2^17 rows = 2^20 LDE points per column)
        try:
            oracle.set_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
        except AttributeError:
            pass
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        want = oracle.plonk_prove(cd, cs, w, pis)
        assert (proof == want).all()
    # breaking one copy constraint between layers must make the proof fail
    if layers > 1 and degree_bits <= 14:
        w2 = wires.clone()
        w2[0, lc.row_first + lc.width] += 1
        proof2 = pc.prove(w2, pis)
        assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, proof2) != 0
    pc.free()
