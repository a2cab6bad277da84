"""The bulk synthetic circuits used by bench.py — the layered PoseidonGate trace (4 gate types) and the 14-gate trace of a
recursive circuit, both with copy constraints and a public-input hash — prove and verify; the GPU proof equals the oracle's
word for word up to 2^17 rows (2^20 LDE points per column: the launch shapes of the headline configuration, among them the
fused quotient kernel on grids of >= 2^19 points)."""
import importlib
import os

import numpy as np
import pytest

from tests.conftest import timed_leg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    ctx = pkg.Context(0)
    yield pkg, circ, ctx
    ctx.close()


@pytest.mark.parametrize("degree_bits,layers", [(5, 3), (9, 4), (12, 8), (14, 16), (16, 16), (17, 16)])
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
        with timed_leg(oracle, degree_bits):
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


@pytest.mark.parametrize("degree_bits,layers", [(13, 8), (16, 16), (17, 16)])
def test_mixed_gate_circuit_proof_equals_oracle(env, oracle, degree_bits, layers):
    """the 14-gate circuit bench.py times (circuit.MixedGateCircuit: every gate verify_proof brings, reference
    src/recursion/gadgets/mod.rs:103) at 2^16 / 2^17 rows — per launch 2^19 / 2^20 LDE points: the fused quotient kernel, the
    specialised NTT passes and the lane-per-leaf sponges in the grid shapes of the 2^20-row proof — equals the oracle's proof
    word for word and is accepted by the restated verifier"""
    pkg, circ, ctx = env
    mc = circ.MixedGateCircuit(degree_bits, layers)
    cd = mc.data
    assert len(cd.gates) == 14
    pc = pkg.ProverCircuit(ctx, cd)
    wires, pis = mc.witness_device(ctx, pkg, seed=degree_bits)
    proof = pc.prove(wires, pis)
    assert oracle.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, proof) == 0
    with timed_leg(oracle, degree_bits):
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        assert (pc.constants_sigmas_cap.elements == cs["cap"]).all()
        want = oracle.plonk_prove(cd, cs, ctx.to_host(wires), pis)
    bad = np.nonzero(proof != want)[0]
    assert bad.size == 0, "first differing proof word %d of %d" % (int(bad[0]), proof.size)
    pc.free()


def test_stream_and_batch_at_2_16_rows(env, oracle):
    """glp_prove_stream (host-resident pinned witnesses, upload overlapped) and glp_prove_batch_dev at 2^16 rows of the 14-gate
    circuit — the block-proof size of BASELINE configs[3] — give the single-proof output word for word; one of them is checked
    against the oracle's proof"""
    import torch

    pkg, circ, ctx = env
    mc = circ.MixedGateCircuit(16, 16)
    cd = mc.data
    pc = pkg.ProverCircuit(ctx, cd)
    ws = [mc.witness_device(ctx, pkg, seed=k) for k in range(3)]
    singles = np.stack([pc.prove(w, p) for w, p in ws])
    assert not (singles[0] == singles[1]).all()
    batch = torch.stack([w for w, _ in ws]).contiguous()
    got = pc.prove_batch(batch, [p for _, p in ws])
    assert (got == singles).all(), "batched proofs differ from single proofs at 2^16 rows"
    del batch
    pinned = [w.cpu().pin_memory() for w, _ in ws]
    got = pc.prove_stream(pinned, [p for _, p in ws])
    assert (got == singles).all(), "streamed proofs differ from single proofs at 2^16 rows"
    for grp in ("2", "1"):  # groups of 2 + 1: the second group's upload overlaps the first group's proofs; groups of 1: the
        # large-circuit shape — the FIRST witness goes up in column chunks and its wires commitment starts on the first chunk
        os.environ["GLP_STREAM_BATCH"] = grp
        try:
            assert (pc.prove_stream(pinned, [p for _, p in ws]) == singles).all(), grp
        finally:
            del os.environ["GLP_STREAM_BATCH"]
    # glp_prove with a host pointer (70 MB witness: chunked upload, inverse NTT + LDE per chunk under the copy), pinned and pageable
    assert (pc.prove(pinned[1].numpy().view(np.uint64), ws[1][1]) == singles[1]).all()
    assert (pc.prove(np.array(pinned[2].numpy().view(np.uint64)), ws[2][1]) == singles[2]).all()
    with timed_leg(oracle, 16):
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        assert (singles[2] == oracle.plonk_prove(cd, cs, ctx.to_host(ws[2][0]), ws[2][1])).all()
    pc.free()


def test_timed_leg_equals_scalar_checker_on_this_host(oracle):
    """The oracle's timed leg against its scalar checker at 2^15 rows of the 14-gate circuit, ON THIS HOST's cores and vector
    unit (the CPU suite holds the same for every backend up to 2^10 rows on the build container): whole proofs identical.
    This is what licenses `timed_leg` for the large parity cases of this suite."""
    import __graft_entry__ as g

    g.load_package()
    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    mc = circ.MixedGateCircuit(15, 16)
    cd = mc.data
    wires, pis = mc.witness_host(oracle.poseidon_gate_rows, seed=15)
    with timed_leg(oracle, 0, threshold=99):  # scalar checker, <= 64 threads
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        cd.circuit_digest = oracle.circuit_digest(cs["cap"])
        want = oracle.plonk_prove(cd, cs, wires, pis)
    with timed_leg(oracle, 15):
        cs_f = oracle.batch_from_values(cd.constants_sigmas_values(), 3, 4)
        for k in cs:
            assert (np.asarray(cs_f[k]) == np.asarray(cs[k])).all(), k
        got = oracle.plonk_prove(cd, cs_f, wires, pis)
    assert (np.asarray(got) == np.asarray(want)).all()
    assert oracle.plonk_verify(cd, cs["cap"], pis, got) == 0


def test_a_batch_beyond_2_to_32_lde_elements_equals_single_proofs(env):
    """64 proofs of the 2^16-row 14-gate circuit in ONE glp_prove_batch_dev call: the wires forest is 64 x 135 columns x 2^19 points =
    4.5e9 elements (36 GB) — every flat index of the batched kernels has to be 64-bit.  Each of the 64 proofs equals the single proof
    of its witness."""
    import torch

    pkg, circ, ctx = env
    if torch.cuda.get_device_properties(0).total_memory < 150 * (1 << 30):
        pytest.skip("needs ~80 GB of device memory")
    mc = circ.MixedGateCircuit(16, 16)
    pc = pkg.ProverCircuit(ctx, mc.data)
    ws = [mc.witness_device(ctx, pkg, seed=k) for k in range(4)]
    singles = [np.array(pc.prove(w, p), copy=True) for w, p in ws]
    K = 64
    batch = torch.stack([ws[k % 4][0] for k in range(K)]).contiguous()
    torch.cuda.synchronize()
    got = pc.prove_batch(batch, [ws[k % 4][1] for k in range(K)])
    bad = [k for k in range(K) if not (got[k] == singles[k % 4]).all()]
    assert not bad, bad
    del batch
    pc.free()
