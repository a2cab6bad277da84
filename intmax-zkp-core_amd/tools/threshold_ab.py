"""Single-proof and small-batch wall time around the lane-per-leaf / four-lanes-per-leaf threshold of the leaf sponge (GLP_COOP4_LEAF_LOG):
14-gate circuits of 2^12 and 2^13 rows (2^15 / 2^16 leaves), one proof and three proofs per call; one process per setting."""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
circ = importlib.import_module(pkg.__name__ + ".circuit")
import torch
ctx = pkg.Context(0)
out = []
for db, K in ((12, 1), (13, 1), (11, 3), (12, 2), (12, 3), (13, 2)):
    lc = circ.MixedGateCircuit(db, 8)
    pc = pkg.ProverCircuit(ctx, lc.data)
    ws = [lc.witness_device(ctx, pkg, seed=s) for s in range(K)]
    if K == 1:
        fn = lambda: pc.prove(*ws[0])
    else:
        batch = torch.stack([w for w, _ in ws]).contiguous()
        pis = [p for _, p in ws]
        fn = lambda: pc.prove_batch(batch, pis)
    for _ in range(5):
        fn()
    ctx.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    out.append("2^%d x%d %.3f" % (db, K, 1e3 * best))
    pc.free()
print(os.environ.get("GLP_COOP4_LEAF_LOG"), " | ".join(out))
