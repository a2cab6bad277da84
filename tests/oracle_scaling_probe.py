"""How the CPU oracle's proving time grows with the circuit size on this host (T = all threads): the data behind the
extrapolation in bench.py's cpu_baseline (linear-in-rows scaling of a 2^15-row sample understated the
measured 2^20-row time by ~1.4x).  Prints one JSON object; run on the GPU box's host cores:
    python tests/oracle_scaling_probe.py [--max-log 18] [--kind mixed]
(a measurement OF the oracle: it lives under tests/, not in the product tree — nothing under intmax-zkp-core_amd/ touches oracle/)"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-log", type=int, default=13)
    ap.add_argument("--max-log", type=int, default=18)
    ap.add_argument("--kind", default="mixed")
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    from oracle import oracle as orc

    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 64))
    orc.set_threads(threads)
    rows = []
    for log_n in range(args.min_log, args.max_log + 1):
        mk = circ.MixedGateCircuit if args.kind == "mixed" else circ.LayeredPoseidonCircuit
        lc = mk(log_n, 16)
        cd = lc.data
        wires, pis = lc.witness_host(orc.poseidon_gate_rows, seed=0)
        t0 = time.perf_counter()
        cs = orc.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)
        t_build = time.perf_counter() - t0
        cd.circuit_digest = orc.circuit_digest(cs["cap"])
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            orc.plonk_prove(cd, cs, wires, pis)
            ts.append(time.perf_counter() - t0)
        rows.append({"log_n": log_n, "build_s": round(t_build, 3), "prove_s": [round(t, 3) for t in ts]})
        sys.stderr.write("2^%d rows: build %.2f s, prove %s\n" % (log_n, t_build, ts))
    print(json.dumps({"threads": threads, "kind": args.kind, "rows": rows}))


if __name__ == "__main__":
    main()
