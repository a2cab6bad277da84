"""Host-resident witnesses at BASELINE's size: glp_prove_stream over K pinned 2^20-row witnesses (1.13 GB each), K = 1, 2, 4, 8,
next to the device-resident proof.  Answers VERDICT r02 item 9: is the "+5.7 ms per proof" of the pinned + pipelined path a
cost of the overlap (copy engine vs the VALU-saturated CUs) or the first, un-overlapped upload amortised over a short stream?
    python intmax-zkp-core_amd/tools/stream_probe.py            # prints one JSON object
    tools/profile_stream.sh                                      # the same under rocprofv3 --kernel-trace --memory-copy-trace"""
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
    ap.add_argument("--ks", default="1,2,4,8")
    args = ap.parse_args()
    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    ctx = pkg.Context(0)
    lc = circ.MixedGateCircuit(args.log_n, 16)
    pc = pkg.ProverCircuit(ctx, lc.data)
    wires, pis = lc.witness_device(ctx, pkg, seed=0)
    pinned = wires.cpu().pin_memory()
    pc.prove(wires, pis)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        pc.prove(wires, pis)
    ctx.synchronize()
    t_dev = (time.perf_counter() - t0) / 3
    out = {"log_n": args.log_n, "witness_bytes": int(pinned.numel() * 8), "device_resident_ms_per_proof": 1e3 * t_dev, "stream": []}
    pc.prove_stream([pinned, pinned], [pis, pis])  # staging buffers and pinned arena sized
    for K in [int(x) for x in args.ks.split(",")]:
        ctx.synchronize()
        t0 = time.perf_counter()
        pc.prove_stream([pinned] * K, [pis] * K)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        out["stream"].append({"K": K, "total_ms": 1e3 * dt, "ms_per_proof": 1e3 * dt / K,
                              "total_minus_K_device_resident_ms": 1e3 * (dt - K * t_dev)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
