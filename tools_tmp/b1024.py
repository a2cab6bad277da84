import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
import torch
circ = importlib.import_module("intmax_zkp_core_amd.circuit")
ctx = pkg.Context(0)
bits = int(sys.argv[1]); K = int(sys.argv[2])
lc = circ.LayeredPoseidonCircuit(bits, min(16, max(1, (1 << bits) // 8)))
pc = pkg.ProverCircuit(ctx, lc.data)
ws, ps = [], []
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for k in range(NW):
    w, p = lc.witness_device(ctx, pkg, seed=k); ws.append(w); ps.append(p)
batch = torch.stack([ws[k % NW] for k in range(K)]).contiguous()
pis = [ps[k % NW] for k in range(K)]
pc.prove_batch(batch, pis); ctx.synchronize()
ctx.timing_enable(True) if hasattr(ctx, "timing_enable") else None
t0 = time.perf_counter()
for _ in range(5): pc.prove_batch(batch, pis)
ctx.synchronize()
dt = (time.perf_counter() - t0) / 5
print("proof_len", pc.proof_len()); print("bits %d K %d: %.2f ms per batch, %.0f proofs/s" % (bits, K, dt * 1e3, K / dt))
