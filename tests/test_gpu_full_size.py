"""BASELINE's full size under the driver's eyes: the GPU proof of the 2^20-row synthetic 14-gate circuit (bench.py --circuit mixed;
the default bench circuit since the end of round 5 is the reference's own 128-slot block circuit of the same size, whose GPU proof
bench.py's cpu_baseline leg and the opt-in tests/test_gpu_rollup.py::test_the_block_circuit_with_128_slots_... hold to the oracle)
equals the CPU oracle's proof word for word.  Since round 5 the oracle side is its TIMED LEG (tests/conftest.py::timed_leg:
seconds instead of the scalar checker's ~3 minutes; GLP_SCALAR_ORACLE=1 runs the scalar checker as rounds 1-4 did — recorded runs
below).  ~40 GB of host memory, so the test runs BY DEFAULT only on a host with >= 32 hardware threads and >= 64 GB of available
memory and skips elsewhere; GLP_FULL_SIZE_PARITY=1 forces it, =0 disables it.
GLP_FULL_SIZE_CIRCUIT=mixed (default: the 14-gate trace) | poseidon (the 4-gate trace); GLP_FULL_SIZE_LOG_N (default 20).
Earlier recorded runs: profiles/r01k_full_size_parity.md, profiles/r02f_full_size_parity.md."""
import importlib
import os
import time

import numpy as np
import pytest

from tests.conftest import timed_leg

pytestmark = pytest.mark.gpu


def _host_can_run_the_oracle_at_full_size():
    """(ok, reason): >= 32 hardware threads and >= 64 GB of available host memory, or the environment override"""
    force = os.environ.get("GLP_FULL_SIZE_PARITY")
    if force == "1":
        return True, "forced"
    if force == "0":
        return False, "GLP_FULL_SIZE_PARITY=0"
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    avail_kb = 0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail_kb = int(line.split()[1])
    except OSError:
        pass
    if threads < 32 or avail_kb < 64 * 1024 * 1024:
        return False, "host has %d threads and %.0f GB available; the full-size oracle run needs >= 32 threads and >= 64 GB " \
                      "(GLP_FULL_SIZE_PARITY=1 forces it)" % (threads, avail_kb / 1048576.0)
    return True, "%d threads, %.0f GB available" % (threads, avail_kb / 1048576.0)


def test_full_size_proof_equals_oracle(oracle):
    ok, why = _host_can_run_the_oracle_at_full_size()
    if not ok:
        pytest.skip(why)
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    log_n = int(os.environ.get("GLP_FULL_SIZE_LOG_N", "20"))
    ctx = pkg.Context(0)
    kind = os.environ.get("GLP_FULL_SIZE_CIRCUIT", "mixed")  # "mixed": the synthetic 14-gate trace (bench.py --circuit mixed)
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
    w = ctx.to_host(wires)
    with timed_leg(oracle, log_n) as leg:
        scalar = not leg.on
        t0 = time.perf_counter()
        cs = oracle.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
        assert (pc.constants_sigmas_cap.elements == cs["cap"]).all(), "constants_sigmas cap differs"
        cd.circuit_digest = oracle.circuit_digest(cs["cap"])
        t_build = time.perf_counter() - t0  # CircuitBuilder::build's commitment: once per circuit, not part of a proof
        want = oracle.plonk_prove(cd, cs, w, pis)
        t_cpu = time.perf_counter() - t0
    print("2^%d rows, %s circuit (%d gate types): GPU %.3f s, oracle (%d threads) %.1f s, proof %d words" % (
        log_n, kind, len(cd.gates), t_gpu, threads, t_cpu, proof.size))
    bad = np.nonzero(proof != want)[0]
    assert bad.size == 0, "first differing word %d of %d" % (int(bad[0]), proof.size)
    assert oracle.plonk_verify(cd, cs["cap"], pis, proof) == 0
    # the same proof from the HOST with the witness as plonky2 holds it — 135 separate pageable columns, 1.13 GB at 2^20 rows: the
    # production chunk plan through the pinned ring, the leaf sponge behind the arriving chunks (glp_prove_cols), and from one array
    cols = [np.array(c_, copy=True) for c_ in w]
    t0 = time.perf_counter()
    from_cols = pc.prove_cols(cols, pis)
    t_cols = time.perf_counter() - t0
    assert (from_cols == proof).all(), "glp_prove_cols differs from the device-resident proof"
    assert (pc.prove(w, pis) == proof).all(), "glp_prove from one pageable array differs from the device-resident proof"
    print("   from 135 pageable host columns: %.3f s" % t_cols)
    del cols
    # a record for bench.py's cpu_baseline ("full_size_measured_s") and for profiles/: written under gpurun_out/ (scratch)
    try:
        import json

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "full_size_parity.json"), "w") as f:
            json.dump({"log_n": log_n, "circuit": kind, "gate_types": len(cd.gates), "gpu_s": t_gpu, "oracle_s": t_cpu, "oracle_build_s": t_build, "oracle_prove_s": t_cpu - t_build,
                       "oracle_threads": min(threads, 64) if scalar else threads, "oracle_leg": "scalar checker" if scalar else "timed leg", "proof_words": int(proof.size), "equal": True,
                       "cpu_model": next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")),
                                         "unknown")}, f)
    except OSError:
        pass
    pc.free()
    ctx.close()


def test_largest_supported_circuit_proves_and_verifies(oracle):
    """2^21 rows is the largest circuit the library takes at rate 1/8 (2^24-point transforms = the master root table): the only size
    whose LDE runs a LAZY strided pass between the canonical coset pass and the last sweep (plan 7 + 7 + 10), 2^24-leaf trees, 36 GB
    of pool memory.  The oracle would need ~80 GB of host memory to PROVE at this size, so the check here is the restated verifier
    (28 queries: cheap) on the GPU's proof — FRI ties the 2^24-point LDEs to the openings, the vanishing identity at zeta ties the
    openings to each other — plus determinism (a second proof is the same bytes) and rejection of a flipped word.  The LDE of this
    size against the oracle's, value by value: tests/test_gpu_lde_lazy_pass.py."""
    if os.environ.get("GLP_FULL_SIZE_PARITY") == "0":
        pytest.skip("GLP_FULL_SIZE_PARITY=0")
    import torch

    import __graft_entry__ as g

    if torch.cuda.get_device_properties(0).total_memory < 100 * (1 << 30):
        pytest.skip("needs ~40 GB of device memory")
    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    ctx = pkg.Context(0)
    lc = circ.MixedGateCircuit(21, 16)
    cd = lc.data
    pc = pkg.ProverCircuit(ctx, cd)
    wires, pis = lc.witness_device(ctx, pkg, seed=11)
    proof = np.array(pc.prove(wires, pis), copy=True)
    t0 = time.perf_counter()
    again = pc.prove(wires, pis)
    ctx.synchronize()
    t_gpu = time.perf_counter() - t0
    assert (again == proof).all()
    cap = pc.constants_sigmas_cap.elements
    assert oracle.plonk_verify(cd, cap, pis, proof) == 0
    bad = proof.copy()
    bad[bad.size // 2] ^= 1
    assert oracle.plonk_verify(cd, cap, pis, bad) != 0
    print("2^21 rows, %d gate types: GPU %.3f s per proof, %d words, verified by the restated verifier" % (len(cd.gates), t_gpu, proof.size))
    pc.free()
    ctx.close()
