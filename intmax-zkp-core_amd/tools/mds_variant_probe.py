"""The Poseidon MDS variants of libglprover side by side (glp_ctx_set_mds_variant): 0 = integer multiplies (the default and
measured configuration), 2 = OPT-IN int8 matrix-core layers (csrc/poseidon.h MDS_MFMA) — Merkle tree over the 2^23 x C LDE matrix
(C = 135: the wires commitment's hashing) and a whole 2^20-row 14-gate proof under each.  Prints one JSON object."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    ctx = pkg.Context(0)
    log_n = int(os.environ.get("GLP_PROBE_LOG_N", "20"))
    lc = circ.MixedGateCircuit(log_n, 16)
    pc = pkg.ProverCircuit(ctx, lc.data)
    wires, pis = lc.witness_device(ctx, pkg, seed=0)
    out = {"log_n": log_n}
    proofs = {}
    for v in (0, 2, 0, 2):
        ctx.set_mds_variant(v)
        ctx.timing_enable(True)
        ctx.timing_reset()
        b = pkg.PolynomialBatch.from_values(ctx, wires, 3, False, 4)
        t = ctx.timing_read()
        ctx.timing_enable(False)
        cap = b.cap.elements.copy()
        b.free()
        pc.prove(wires, pis)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            p = pc.prove(wires, pis)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 3
        proofs.setdefault(v, p)
        out.setdefault("variant_%d" % v, []).append({"leaf_hash_ms_C135": t["leaf_hash"]["ms"], "merkle_levels_ms": t["merkle_levels"]["ms"],
                                                     "gperm_per_s_leaf": (1 << (log_n + 3)) * 17 / (t["leaf_hash"]["ms"] * 1e-3) / 1e9,
                                                     "proof_ms": 1e3 * dt, "cap_word0": int(cap[0][0])})
    ctx.set_mds_variant(0)
    out["proofs_identical"] = bool((proofs[0] == proofs[2]).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
