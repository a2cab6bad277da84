"""A/B of the mapped read-backs (GLP_MAPPED_CAPS = 0 / 1 / 2): single-proof wall time at 2^3, 2^11, 2^14 rows, best of 5 x 40 proofs; one process per setting."""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
circ = importlib.import_module(pkg.__name__ + ".circuit")
ctx = pkg.Context(0)
out = []
for db in (3, 11, 14):
    lc = circ.MixedGateCircuit(db, min(16, max(1, (1 << db) // 64))) if db >= 6 else circ.LayeredPoseidonCircuit(db, 1)
    pc = pkg.ProverCircuit(ctx, lc.data)
    w, pis = lc.witness_device(ctx, pkg, seed=db)
    for _ in range(5):
        pc.prove(w, pis)
    ctx.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(40):
            pc.prove(w, pis)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40)
    out.append("2^%d %.4f ms" % (db, 1e3 * best))
    pc.free()
print(os.environ.get("GLP_MAPPED_CAPS"), " | ".join(out))
