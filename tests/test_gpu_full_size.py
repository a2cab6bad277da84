"""BASELINE's full size, once: the GPU proof of the 2^20-row circuit equals the CPU oracle's proof word for word.
The oracle needs a few minutes of all host cores and ~40 GB of host memory for that, so the test only runs when
GLP_FULL_SIZE_PARITY=1 (GLP_FULL_SIZE_CIRCUIT=mixed for the 14-gate trace; the committed records of runs are
profiles/r01k_full_size_parity.md and profiles/r02f_full_size_parity.md); the default suite covers
2^14 rows word for word and 2^20 rows through the restated verifier (bench.py)."""
import importlib
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("GLP_FULL_SIZE_PARITY") != "1", reason="set GLP_FULL_SIZE_PARITY=1 (minutes of CPU, ~40 GB of host memory)")
def test_full_size_proof_equals_oracle(oracle):
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    log_n = int(os.environ.get("GLP_FULL_SIZE_LOG_N", "20"))
    ctx = pkg.Context(0)
    kind = os.environ.get("GLP_FULL_SIZE_CIRCUIT", "poseidon")  # "mixed": the 14-gate trace bench.py proves by default
    lc = circ.MixedGateCircuit(log_n, 16) if kind == "mixed" else circ.LayeredPoseidonCircuit(log_n, 16)
    cd = lc.data
    pc = pkg.ProverCircuit(ctx, cd)
    wires, pis = lc.witness_device(ctx, pkg, seed=7)
    t0 = time.perf_counter()
    proof = pc.prove(wires, pis)
    t_gpu = time.perf_counter() - t0
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    oracle.set_threads(max(1, min(threads, 64)))
    w = ctx.to_host(wires)
    t0 = time.perf_counter()
    cs = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
    assert (pc.constants_sigmas_cap.elements == cs["cap"]).all(), "constants_sigmas cap differs"
    cd.circuit_digest = oracle.circuit_digest(cs["cap"])
    want = oracle.plonk_prove(cd, cs, w, pis)
    t_cpu = time.perf_counter() - t0
    print("2^%d rows, %s circuit (%d gate types): GPU %.3f s, oracle (%d threads) %.1f s, proof %d words" % (
        log_n, kind, len(cd.gates), t_gpu, threads, t_cpu, proof.size))
    bad = np.nonzero(proof != want)[0]
    assert bad.size == 0, "first differing word %d of %d" % (int(bad[0]), proof.size)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof) == 0
    pc.free()
    ctx.close()
