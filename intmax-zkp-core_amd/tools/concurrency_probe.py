#!/usr/bin/env python3
"""How much whole-GPU throughput do W concurrent provers (one context + stream each) add over one?  The latency-bound
parts of a proof (tree tops, host round trips of the transcript, PoW) leave the GPU partly idle; a second proof in
flight fills them.  Usage: python concurrency_probe.py [log_n] [workers...]"""
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
circ = importlib.import_module("intmax_zkp_core_amd.circuit")
import torch  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
workers_list = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
lc = (circ.MixedGateCircuit if os.environ.get("GLP_PROBE_CIRCUIT", "mixed") == "mixed" else circ.LayeredPoseidonCircuit)(log_n, 16)
K = 4
for W in workers_list:
    ctxs, pcs, wit = [], [], []
    for w in range(W):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            c = pkg.Context(0)
            pc = pkg.ProverCircuit(c, lc.data)
            wires, pis = lc.witness_device(c, pkg, seed=w)
            pc.prove(wires, pis)  # warm-up
        ctxs.append(c)
        pcs.append(pc)
        wit.append((wires, pis))
    torch.cuda.synchronize()

    def run(i):
        for _ in range(K):
            pcs[i].prove(*wit[i])

    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i,)) for i in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("workers %d: %.2f proofs/s (%.1f ms per proof per worker)" % (W, W * K / dt, 1e3 * dt / K), flush=True)
    for pc in pcs:
        pc.free()
    for c in ctxs:
        c.close()
