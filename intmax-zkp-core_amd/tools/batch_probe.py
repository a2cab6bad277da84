#!/usr/bin/env python3
"""Throughput of batched proving at the reference's real circuit sizes (its circuits are 2^4 .. 2^16 rows: SURVEY.md §8d):
proofs/s of glp_prove_dev one proof at a time vs glp_prove_batch_dev with K witnesses per call.
    python intmax-zkp-core_amd/tools/batch_probe.py [--sizes 4,8,12,16] [--json out.json]"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="4,8,12,16")
ap.add_argument("--circuit", default="poseidon", choices=("poseidon", "mixed"))
ap.add_argument("--json")
args = ap.parse_args()
pkg = g.load_package()
import torch  # noqa: E402

circ = importlib.import_module("intmax_zkp_core_amd.circuit")
ctx = pkg.Context(0)
out = []
for bits in [int(x) for x in args.sizes.split(",")]:
    mk = circ.MixedGateCircuit if args.circuit == "mixed" and bits >= 5 else circ.LayeredPoseidonCircuit
    lc = mk(bits, min(16, max(1, (1 << bits) // 8)))
    pc = pkg.ProverCircuit(ctx, lc.data)
    Kmax = {4: 1024, 8: 512, 12: 128, 14: 64, 16: 32}.get(bits, 16)
    ws, ps = [], []
    for k in range(Kmax):  # distinct witnesses: every proof of a batch has its own proof-of-work grind
        w, p = lc.witness_device(ctx, pkg, seed=k)
        ws.append(w)
        ps.append(p)
    for _ in range(3):
        pc.prove(ws[0], ps[0])
    ctx.synchronize()
    reps = max(5, 200 >> max(0, bits - 8))
    t0 = time.perf_counter()
    for i in range(reps):
        pc.prove(ws[i % len(ws)], ps[i % len(ps)])
    ctx.synchronize()
    single = reps / (time.perf_counter() - t0)
    row = {"degree_bits": bits, "circuit": args.circuit, "single_proofs_per_s": single, "single_ms": 1e3 / single, "batch": {}}
    K = 4
    while K <= Kmax:
        batch = torch.stack([ws[k % len(ws)] for k in range(K)]).contiguous()
        pis = [ps[k % len(ps)] for k in range(K)]
        proofs = pc.prove_batch(batch, pis)
        assert (proofs[0] == pc.prove(ws[0], ps[0])).all()
        ctx.synchronize()
        nrep = max(2, int(0.5 * single / K) or 2)
        t0 = time.perf_counter()
        for _ in range(nrep):
            pc.prove_batch(batch, pis, out=proofs)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / nrep
        row["batch"][K] = {"proofs_per_s": K / dt, "ms_per_batch": 1e3 * dt, "speedup": K / dt / single}
        del batch
        K *= 4
    out.append(row)
    best = max(row["batch"].items(), key=lambda kv: kv[1]["proofs_per_s"])
    print("2^%-2d rows: single %8.1f proofs/s (%.2f ms) | best batch K=%d: %9.1f proofs/s (x%.1f)  %s" % (
        bits, single, 1e3 / single, best[0], best[1]["proofs_per_s"], best[1]["speedup"],
        " ".join("K%d:%.0f" % (k, v["proofs_per_s"]) for k, v in row["batch"].items())), flush=True)
    pc.free()
if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
