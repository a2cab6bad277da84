#!/usr/bin/env python3
"""BASELINE configs[4]: the recursive-aggregation tree over N independent block proofs, one process per GPU — a thin command
line over intmax_zkp_core_amd.workloads.run_aggtree (the same code bench.py --workload aggtree runs).

    python aggregation_tree.py [--proofs 64] [--arity 8] [--block-log-n 16] [--agg-log-n 16]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 aggregation_tree.py ...
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=64)
    ap.add_argument("--arity", type=int, default=8)
    ap.add_argument("--block-log-n", type=int, default=16)
    ap.add_argument("--agg-log-n", type=int, default=16)
    ap.add_argument("--circuit", default="mixed", choices=("mixed", "poseidon"))
    ap.add_argument("--steps", type=int, default=1)
    args = ap.parse_args()

    import torch

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    codec = importlib.import_module(pkg.__name__ + ".codec")
    dist_mod = importlib.import_module(pkg.__name__ + ".dist")
    workloads = importlib.import_module(pkg.__name__ + ".workloads")
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: libglprover has no CPU fallback")
    rank, local_rank, world = dist_mod.env_world()
    dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    rank, local_rank, world = dist_mod.init(device_index=dev)
    ctx = pkg.Context(dev)
    res = workloads.run_aggtree(pkg, circ, codec, dist_mod, ctx, rank, world, args.proofs, args.arity, args.block_log_n,
                                args.agg_log_n, args.steps, 1, args.circuit, device="cuda:%d" % dev if world > 1 else None)
    if rank == 0:
        print("| level | circuit | proofs | prove ms (max over ranks) | exchange ms | bytes exchanged | wait for the slowest rank ms |")
        print("|---|---|---|---|---|---|---|")
        for r in res["levels"]:
            print("| %d | %s | %d | %.1f | %.2f | %d | %.2f |" % r)
        total = res["elapsed_s"] / args.steps
        print("\n%d GPU(s): tree over %d block proofs, arity %d: %.1f ms per tree (%.2f block proofs/s aggregated); root proof %d bytes"
              % (world, args.proofs, args.arity, total * 1e3, args.proofs / total, res["root_bytes"]))
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
