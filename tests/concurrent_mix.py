"""The concurrent-contexts workload behind tests/test_gpu_concurrent_contexts.py (a helper, not a test module; also runnable
as `python tests/concurrent_mix.py ROUNDS` so that a child process can repeat it under GLP_POOL_CANARY=2).

SURVEY.md §8b "Callers & threading": a context is single-threaded, several contexts may be driven from different host
threads concurrently, the library is re-entrant across contexts — what the reference's `parallel` feature (Cargo.toml:18-21:
rayon under plonky2) would do to a backend that sits behind `CircuitData::prove`.  Eight host threads, each with its OWN
glp_ctx (own stream, own pool) on GPU 0, run at the same time:

    T0  the zkdsa "simple signature" circuit (reference src/zkdsa/circuits/mod.rs:24-43), one proof per round
    T1  a 2^12-row 14-gate proof per round (device-resident witness)
    T2  a 2^16-row 14-gate proof per round
    T3  a batch of 8 proofs of the 2^14-row 14-gate circuit per round (glp_prove_batch_dev)
    T4  glp_smt_root_of_set over 2^12 entries per round (rocPRIM sort + the tree kernels)

    T5  64 sequential `set`s on that set per round (glp_smt_apply_updates)
    T6  three host-resident 2^12-row witnesses per round through glp_prove_stream
    T7  a 2^16-row witness as 135 separate pageable columns per round through glp_prove_cols (the pinned ring)

Every result of every round must equal, word for word, what the same context gave when it ran alone; the sequential
results are themselves checked with the restated verifier (oracle/) and the oracle's set root.  Shared process state this
exercises under contention: the host worker pool (csrc/host_pool.h, try-lock fallback), the per-process env caches, the
per-context hipFuncSetAttribute flags, the pinned arenas, hipMalloc / hipFree from several threads."""
import importlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(rounds=20, verify=True, log=print):
    import torch

    import __graft_entry__ as g
    from oracle import oracle as orc
    from tests.test_oracle_plonk import zkdsa_circuit

    orc.set_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    trees = importlib.import_module(pkg.__name__ + ".trees")

    jobs = []  # (name, ctx, fn() -> numpy array, cleanup)

    def add_prover(name, cd, witnesses, batch):
        ctx = pkg.Context(0, use_torch_stream=False)  # its own non-blocking stream: nothing serialises on the null stream
        pc = pkg.ProverCircuit(ctx, cd)
        if batch:
            wires = torch.stack([w for w, _ in witnesses]).contiguous()
            pis = [p for _, p in witnesses]
            fn = lambda: pc.prove_batch(wires, pis)
        else:
            wires, pis = witnesses[0]
            fn = lambda: pc.prove(wires, pis)
        jobs.append((name, ctx, fn, pc, cd, witnesses))

    setup = pkg.Context(0)
    # T0: zkdsa-shaped
    cd0, sk, msg = zkdsa_circuit(circ)
    inputs = dict(zip(sk, [5, 6, 7, 8]))
    inputs.update(zip(msg, [11, 22, 33, 44]))
    w0, pis0 = cd0.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(setup, ins))
    add_prover("zkdsa", cd0, [(setup.to_device(w0), pis0)], False)
    # T1, T2: 14-gate circuits of two sizes; T3: a batch of 8 at 2^14
    for name, bits, layers, k in (("mixed12", 12, 8, 1), ("mixed16", 16, 16, 1), ("batch8x14", 14, 8, 8)):
        mc = circ.MixedGateCircuit(bits, layers)
        ws = [mc.witness_device(setup, pkg, seed=100 * bits + s) for s in range(k)]
        add_prover(name, mc.data, ws, k > 1)
    torch.cuda.synchronize()
    # T4: set root
    rng = np.random.default_rng(7)
    keys = rng.integers(0, 2**63, (1 << 12, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (1 << 12, 4), dtype=np.uint64)
    tctx = pkg.Context(0, use_torch_stream=False)
    jobs.append(("smt_root_of_set", tctx, lambda: np.asarray(trees.smt_root_of_set(tctx, keys, vals)), None, None, None))

    # T5: 64 sequential `set`s on the same set (device find passes + host replay + levelled hash jobs); T6: three host witnesses
    # through glp_prove_stream (pinned staging, copy stream)
    uk = np.concatenate([keys[:40], rng.integers(0, 2**63, (24, 4), dtype=np.uint64)])
    uv = rng.integers(0, 2**63, (64, 4), dtype=np.uint64)
    uv[::7] = 0
    uctx = pkg.Context(0, use_torch_stream=False)

    def updates():
        proofs, roots = trees.smt_apply_updates(uctx, keys, vals, uk, uv)
        flat = [np.asarray(roots, dtype=np.uint64).ravel()]
        for p in proofs:
            flat.append(np.array([x for s in p["siblings"] for x in s] + list(p["old_root"]) + list(p["new_root"]) + [p["fnc"]], dtype=np.uint64))
        return np.concatenate(flat)

    jobs.append(("smt_apply_updates", uctx, updates, None, None, "updates"))
    mc12 = circ.MixedGateCircuit(12, 8)
    sctx = pkg.Context(0, use_torch_stream=False)
    spc = pkg.ProverCircuit(sctx, mc12.data)
    host_w = [mc12.witness_device(setup, pkg, seed=900 + s) for s in range(3)]
    torch.cuda.synchronize()
    host_ws = [(w.cpu().pin_memory(), p) for w, p in host_w]
    jobs.append(("stream3x12", sctx, lambda: spc.prove_stream([w for w, _ in host_ws], [p for _, p in host_ws]), spc, mc12.data, host_ws))

    # T7: a 2^16-row witness as SEPARATE PAGEABLE COLUMNS through glp_prove_cols (70 MB: column chunks through the pinned ring, staged by
    # the process-wide worker pool the other threads are using too, the leaf sponge behind the chunks)
    mc16 = circ.MixedGateCircuit(16, 16)
    cctx = pkg.Context(0, use_torch_stream=False)
    cpc = pkg.ProverCircuit(cctx, mc16.data)
    cw, cpis = mc16.witness_device(setup, pkg, seed=1700)
    torch.cuda.synchronize()
    ccols = [np.array(c_, copy=True) for c_ in setup.to_host(cw)]
    jobs.append(("cols16", cctx, lambda: cpc.prove_cols(ccols, cpis), cpc, mc16.data, [(cw, cpis)]))

    # sequential pass: the expected results, each context alone
    want = []
    for name, ctx, fn, pc, cd, wit in jobs:
        r = fn().copy()
        want.append(r)
        if not verify:
            continue
        if wit == "updates":
            continue  # (checked against the incremental oracle in tests/test_gpu_trees.py)
        if pc is None:
            assert (r == np.asarray(orc.smt_root_of_set(keys, vals))).all(), "sequential set root differs from the oracle"
        else:
            proofs = r.reshape(len(wit), -1)
            for k, (_, pis) in enumerate(wit):
                assert orc.plonk_verify(cd, pc.constants_sigmas_cap.elements, pis, proofs[k]) == 0, \
                    "%s: the restated verifier rejects sequential proof %d" % (name, k)

    # concurrent rounds
    start = threading.Barrier(len(jobs))
    errors, mismatches = [], []

    def worker(i):
        name, ctx, fn = jobs[i][0], jobs[i][1], jobs[i][2]
        try:
            start.wait(timeout=120)
            for r in range(rounds):
                got = fn()
                if got.shape != want[i].shape or not (got == want[i]).all():
                    if got.shape == want[i].shape:
                        g1, w1 = np.asarray(got).ravel(), np.asarray(want[i]).ravel()
                        idx = np.nonzero(g1 != w1)[0]
                        again = np.asarray(fn()).ravel()  # the same call once more, still under contention
                        mismatches.append((name, r, int(idx.size), "of %d words; first at %s: got %s want %s; the call repeated %s" % (
                            g1.size, idx[:6].tolist(), [hex(int(x)) for x in g1[idx[:6]]], [hex(int(x)) for x in w1[idx[:6]]],
                            "is right" if (again == w1).all() else "differs again in %d words" % int((again != w1).sum()))))
                    else:
                        mismatches.append((name, r, -1))
        except Exception as e:  # noqa: BLE001 — reported below, with the thread's name
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,), name=jobs[i][0]) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    L = pkg.lib()
    import ctypes

    canary = 0
    for name, ctx, *_ in jobs:
        n = ctypes.c_uint64()
        assert L.glp_ctx_canary_violations(ctx._h, ctypes.byref(n)) == 0
        canary += n.value
    for name, ctx, fn, pc, *_ in jobs:
        if pc is not None:
            pc.free()
        ctx.close()
    setup.close()
    log("CONCURRENT threads=%d rounds=%d errors=%d mismatches=%d canary=%d" % (len(jobs), rounds, len(errors), len(mismatches), canary))
    for e in errors + mismatches:
        log("  ", e)
    return errors, mismatches, canary


if __name__ == "__main__":
    errs, mism, can = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, verify=False)
    sys.exit(1 if errs or mism or can else 0)
