"""A measurement OF the oracle's timed leg (oracle/fast*.{c,h}) on this host's cores: complete proofs of the 14-gate circuit
at one size, per thread count, with the per-phase wall times — the data behind bench.py's cpu_baseline and
profiles/r05_cpu_baseline.md.  CPU only; lives under tests/ (nothing in the product tree touches oracle/):
    python tests/oracle_fast_probe.py --log-n 20 --threads 128,256 [--scalar-too]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--threads", default="")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--isa", type=int, default=0)
    ap.add_argument("--scalar-too", action="store_true")
    args = ap.parse_args()
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    from oracle import oracle as orc

    logical, quota, usable = orc.host_cpus()
    threads = [int(t) for t in args.threads.split(",") if t] or [usable]
    lc = circ.MixedGateCircuit(args.log_n, 16)
    cd = lc.data
    isa = orc.set_fast(True, args.isa)
    st = np.random.default_rng(0).integers(0, 2**63, (1 << 16, 12), dtype=np.uint64)
    orc.permute_many(st[:4096])
    t0 = time.perf_counter()
    orc.permute_many(st)
    ns_fast = (time.perf_counter() - t0) / len(st) * 1e9
    t0 = time.perf_counter()
    orc.permute_many(st[:8192], fast=False)
    ns_scalar = (time.perf_counter() - t0) / 8192 * 1e9
    t0 = time.perf_counter()
    wires, pis = lc.witness_host(orc.poseidon_gate_rows, seed=0)
    t_wit = time.perf_counter() - t0
    orc.set_threads(max(threads))
    t0 = time.perf_counter()
    cs = orc.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
    t_build = time.perf_counter() - t0
    cd.circuit_digest = orc.circuit_digest(cs["cap"])
    out = {"log_n": args.log_n, "isa": isa, "logical_cpus": logical, "cgroup_cpu_quota": quota, "inner_log": os.environ.get("GLO_FAST_INNER_LOG", "default"),
           "ns_per_permutation_fast": ns_fast, "ns_per_permutation_scalar": ns_scalar, "witness_s": t_wit, "build_s": t_build, "runs": []}
    proof = None
    for th in threads:
        orc.set_threads(th)
        for _ in range(args.reps):
            t0 = time.perf_counter()
            proof = orc.plonk_prove(cd, cs, wires, pis)
            dt = time.perf_counter() - t0
            out["runs"].append({"threads": th, "prove_s": round(dt, 3), "phases_s": {k: round(v, 3) for k, v in orc.last_phase_seconds().items()}})
            sys.stderr.write("T=%d: %.2f s %s\n" % (th, dt, out["runs"][-1]["phases_s"]))
    orc.set_fast(False)
    out["verified"] = orc.plonk_verify(cd, cs["cap"], pis, proof) == 0
    if args.scalar_too:
        orc.set_threads(max(threads))
        t0 = time.perf_counter()
        ref = orc.plonk_prove(cd, cs, wires, pis)
        out["scalar_checker_prove_s"] = time.perf_counter() - t0
        out["scalar_checker_phases_s"] = orc.last_phase_seconds()
        out["equal_to_scalar_checker"] = bool((np.asarray(ref) == np.asarray(proof)).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
