"""glp_prove_batch_dev of K proofs of a 2^bits-row circuit: min / median / p90 of 60 calls and the per-kind kernel time (the A/B tool for the host side of the batched prover; GLPROVER_LIB = the other build).  usage: batch_trace.py <bits> <K>"""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
import torch
circ = importlib.import_module(pkg.__name__ + ".circuit")
ctx = pkg.Context(0)
bits, K = int(sys.argv[1]), int(sys.argv[2])
lc = (circ.MixedGateCircuit if bits >= 6 else circ.LayeredPoseidonCircuit)(bits, min(8, max(1, (1 << bits) // 64)))
pc = pkg.ProverCircuit(ctx, lc.data)
ws = [lc.witness_device(ctx, pkg, seed=k) for k in range(K)]
batch = torch.stack([w for w, _ in ws]).contiguous()
pis = [p for _, p in ws]
proofs = pc.prove_batch(batch, pis)
for _ in range(3):
    pc.prove_batch(batch, pis, out=proofs)
ctx.synchronize()
ts = []
for _ in range(60):
    t0 = time.perf_counter()
    pc.prove_batch(batch, pis, out=proofs)
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
ts.sort()
dt = ts[len(ts) // 2]
print("   min %.3f median %.3f p90 %.3f ms" % (1e3 * ts[0], 1e3 * dt, 1e3 * ts[int(0.9 * len(ts))]))
ctx.timing_enable(True); ctx.timing_reset()
pc.prove_batch(batch, pis, out=proofs)
t = ctx.timing_read(); ctx.timing_enable(False)
print("2^%d x %d: %.3f ms per call = %.0f proofs/s; kernel ms %.3f; %s" % (bits, K, 1e3 * dt, K / dt, sum(v["ms"] for v in t.values()), {k: round(v["ms"], 2) for k, v in t.items() if v["launches"]}))
os.environ["X"] = "1"
