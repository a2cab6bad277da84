"""Time the constraint-quotient evaluation alone (compute_quotient_polys of the 14-gate circuit, glp_quotient_dev) and a whole
proof, under the environment switches of the fused kernel (GLP_QUOT_FUSED, GLP_FQ_DYNAMIC, GLP_FQ_PIECE) — the A/B tool for
csrc/quotient_fused.h.  Run on the GPU box:  python intmax-zkp-core_amd/tools/quotient_probe.py [--log-n 20] [--reps 5]"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--circuit", default="mixed")
    ap.add_argument("--proofs", type=int, default=3)
    ap.add_argument("--no-proof", action="store_true", help="time the quotient evaluation only (timing builds whose arithmetic is wrong)")
    args = ap.parse_args()
    import numpy as np

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    ctx = pkg.Context(0)
    mk = circ.MixedGateCircuit if args.circuit == "mixed" else circ.LayeredPoseidonCircuit
    lc = mk(args.log_n, 16)
    cd = lc.data
    pc = pkg.ProverCircuit(ctx, cd)
    wires, pis = lc.witness_device(ctx, pkg, seed=0)
    rng = np.random.default_rng(1)
    betas, gammas, alphas = (rng.integers(0, pkg.P, 2, dtype=np.uint64) for _ in range(3))
    wb = pkg.PolynomialBatch.from_values(ctx, wires, 3, False, 4)
    zs = pc.partial_products(wires, betas, gammas)
    zb = pkg.PolynomialBatch.from_values(ctx, zs, 3, False, 4)
    pih = np.zeros(4, dtype=np.uint64)
    q0 = ctx.to_host(pc.quotient_chunks(wb, zb, betas, gammas, alphas, pih))
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(args.reps):
        pc.quotient_chunks(wb, zb, betas, gammas, alphas, pih)
    t = ctx.timing_read()
    ctx.timing_enable(False)
    out = {"log_n": args.log_n, "circuit": args.circuit, "env": {k: os.environ.get(k) for k in ("GLP_QUOT_FUSED", "GLP_FQ_DYNAMIC", "GLP_FQ_PIECE", "GLP_FQ_SUBSET")},
           "quotient_eval_ms": t["plonk"]["ms"] / args.reps, "plonk_launches_per_call": t["plonk"]["launches"] / args.reps,
           "quotient_checksum": int(np.bitwise_xor.reduce(q0.ravel()))}
    wb.free()
    zb.free()
    if os.environ.get("GLP_FQ_SUBSET") or args.no_proof:  # a partial quotient: no proof can be made with it
        print(json.dumps(out))
        return
    pc.prove(wires, pis)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.proofs):
        pc.prove(wires, pis)
    ctx.synchronize()
    out["proof_ms"] = 1e3 * (time.perf_counter() - t0) / args.proofs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
