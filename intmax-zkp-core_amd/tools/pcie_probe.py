import sys, os, time, importlib
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package(); circ = importlib.import_module("intmax_zkp_core_amd.circuit")
import torch, numpy as np
ctx = pkg.Context(0)
lc = circ.LayeredPoseidonCircuit(20, 16)
t=time.perf_counter(); pc = pkg.ProverCircuit(ctx, lc.data); ctx.synchronize(); print("CircuitBuilder::build commitment (84 cols, incl. H2D): %.1f ms" % (1e3*(time.perf_counter()-t)))
wires, pis = lc.witness_device(ctx, pkg, seed=0)
wh = ctx.to_host(wires).copy()
for name, w in (("device-resident witness", wires), ("host-resident witness (pageable, H2D inside)", wh)):
    pc.prove(w, pis)
    t=time.perf_counter()
    for _ in range(3): pc.prove(w, pis)
    print("%s: %.1f ms per proof" % (name, 1e3*(time.perf_counter()-t)/3))
