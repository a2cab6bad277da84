"""Time the partial-products step alone (wires_permutation_partial_products_and_zs of the 14-gate circuit, glp_partial_products_dev:
the chunk-quotient kernel + the three scan kernels) — the A/B tool for plonk::pp_* (use GLPROVER_LIB for the other build).
Run on the GPU box:  python intmax-zkp-core_amd/tools/pp_probe.py [--log-n 20] [--reps 10]"""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import numpy as np

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    ctx = pkg.Context(0)
    lc = circ.MixedGateCircuit(args.log_n, 16)
    pc = pkg.ProverCircuit(ctx, lc.data)
    wires, pis = lc.witness_device(ctx, pkg, seed=0)
    rng = np.random.default_rng(1)
    betas, gammas = (rng.integers(0, pkg.P, 2, dtype=np.uint64) for _ in range(2))
    z0 = ctx.to_host(pc.partial_products(wires, betas, gammas))
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(args.reps):
        pc.partial_products(wires, betas, gammas)
    t = ctx.timing_read()
    ctx.timing_enable(False)
    print(json.dumps({"log_n": args.log_n, "lib": os.environ.get("GLPROVER_LIB", "in-tree"), "partial_products_ms": t["plonk"]["ms"] / args.reps,
                      "checksum": int(np.bitwise_xor.reduce(z0.ravel()))}))


if __name__ == "__main__":
    main()
