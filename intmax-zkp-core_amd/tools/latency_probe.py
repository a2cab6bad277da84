#!/usr/bin/env python3
"""Single-proof latency at small circuit sizes (the reference's own circuits are 2^3 .. 2^16 rows): wall
time per proof next to the sum of kernel time (hipEvent pairs) and the launch count, to show how much of the latency is
launch / synchronisation overhead rather than GPU work."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import importlib  # noqa: E402

circ = importlib.import_module(pkg.__name__ + ".circuit")
ctx = pkg.Context(0)
print("| degree_bits | wall ms/proof | kernel ms/proof | launches/proof | by kind (ms) |")
print("|---|---|---|---|---|")
for db in [int(a) for a in sys.argv[1:]] or [4, 8, 12, 14, 16]:
    lc = circ.LayeredPoseidonCircuit(db, layers=min(16, (1 << db) // 4))
    pc = pkg.ProverCircuit(ctx, lc.data)
    wires, pis = lc.witness_device(ctx, pkg, seed=db)
    for _ in range(3):
        pc.prove(wires, pis)
    ctx.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        pc.prove(wires, pis)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(reps):
        pc.prove(wires, pis)
    t = ctx.timing_read()
    ctx.timing_enable(False)
    kms = sum(v["ms"] for v in t.values()) / reps
    nl = sum(v["launches"] for v in t.values()) / reps
    kinds = ", ".join("%s %.2f" % (k, v["ms"] / reps) for k, v in t.items() if v["launches"])
    print("| %d | %.2f | %.2f | %.0f | %s |" % (db, wall, kms, nl, kinds))
    pc.free()
