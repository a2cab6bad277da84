import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
import torch, numpy as np
circ = importlib.import_module("intmax_zkp_core_amd.circuit")
ctx = pkg.Context(0)
bits = int(sys.argv[1]); K = int(sys.argv[2])
lc = circ.LayeredPoseidonCircuit(bits, min(16, max(1, (1 << bits) // 8)))
pc = pkg.ProverCircuit(ctx, lc.data)
ws, ps = [], []
for k in range(K):
    w, p = lc.witness_device(ctx, pkg, seed=k); ws.append(ctx.to_host(w)); ps.append(p)
pinned = [torch.from_numpy(w.view(np.int64)).pin_memory() for w in ws]
pc.prove_stream(pinned, ps); ctx.synchronize()
for grp in (None, "1"):
    if grp: os.environ["GLP_STREAM_BATCH"] = grp
    t0 = time.perf_counter()
    for _ in range(3): pc.prove_stream(pinned, ps)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("bits %d, %d host witnesses, group %s: %.2f ms, %.0f proofs/s" % (bits, K, grp or "auto", dt * 1e3, K / dt))
