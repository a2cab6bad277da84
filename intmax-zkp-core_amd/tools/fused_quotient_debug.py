import importlib, sys
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
circ = importlib.import_module("intmax_zkp_core_amd.circuit")
ctx = pkg.Context(0)
for mk, name in ((circ.MixedGateCircuit, "mixed"), (circ.LayeredPoseidonCircuit, "poseidon")):
    lc = mk(16, 16)
    pc = pkg.ProverCircuit(ctx, lc.data)
    wires, pis = lc.witness_device(ctx, pkg, seed=1)
    pc.prove(wires, pis)
    print("=====", name, file=sys.stderr)
    pc.prove(wires, pis)
    pc.free()
