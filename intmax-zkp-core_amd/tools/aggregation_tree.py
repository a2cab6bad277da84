#!/usr/bin/env python3
"""BASELINE configs[4] (and [3] as its first level): a recursive-aggregation tree over N independent block proofs, one
process per GPU.

    level 0   N block proofs, sharded round-robin over the ranks (no collective: SURVEY.md 8e)
    exchange  every rank serialises its proofs in plonky2's byte format (codec.py) and the blobs are all-gathered
              (RCCL all_gather of padded uint8 buffers on GPUs, gloo on CPU; ~190 kB per proof: latency-bound)
    level k   ceil(n / arity) aggregation proofs, again sharded round-robin; the witness of parent j is seeded by a digest
              of its children's bytes, so a parent cannot be proven before its children have arrived
    ...       until one root proof remains (on rank 0)

The block and aggregation circuits are synthetic PoseidonGate traces of the estimated sizes (the Rust circuits cannot be
built here): what is exercised and timed is the prover, the serialisation and the one real exchange step of
the product.  Works with one process as well (every level on one GPU).

    python aggregation_tree.py [--proofs 64] [--arity 8] [--block-log-n 16] [--agg-log-n 16] [--contexts 8]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 aggregation_tree.py ...
"""
import argparse
import hashlib
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=64)
    ap.add_argument("--arity", type=int, default=8)
    ap.add_argument("--block-log-n", type=int, default=16)
    ap.add_argument("--agg-log-n", type=int, default=16)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--contexts", type=int, default=8, help="provers (context + stream + thread) per GPU")
    args = ap.parse_args()

    import torch

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    codec = importlib.import_module(pkg.__name__ + ".codec")
    dist_mod = importlib.import_module(pkg.__name__ + ".dist")
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: libglprover has no CPU fallback")
    rank, local_rank, world = dist_mod.env_world()
    dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    rank, local_rank, world = dist_mod.init(device_index=dev)
    device = "cuda:%d" % dev

    # K provers per GPU, each with its own context and stream: at these circuit sizes a single proof is latency-bound
    # (dependent permutations of the tree levels, host round trips of the transcript) and several in flight fill the GPU
    circuits = {name: circ.LayeredPoseidonCircuit(log_n, args.layers)
                for name, log_n in (("block", args.block_log_n), ("agg", args.agg_log_n))}
    layouts = {name: codec.ProofLayout(lc.data, pkg.FriParams(lc.data.degree_bits).reduction_arity_bits)
               for name, lc in circuits.items()}
    workers = []
    for _ in range(max(1, args.contexts)):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            c = pkg.Context(dev)
            workers.append((st, c, {name: pkg.ProverCircuit(c, lc.data) for name, lc in circuits.items()}))

    def prove(w, kind, seed):
        st, c, pcs = workers[w]
        with torch.cuda.stream(st):
            wires, pis = circuits[kind].witness_device(c, pkg, seed=seed)
            flat = pcs[kind].prove(wires, pis)
        return codec.proof_with_public_inputs_to_bytes(codec.proof_from_flat(flat, layouts[kind]), pis)

    def prove_units(kind, seeds):
        """{unit: seed} -> {unit: proof bytes}, the units dealt to the worker threads of this rank"""
        out, todo, lock = {}, sorted(seeds.items()), threading.Lock()

        def run(w):
            while True:
                with lock:
                    if not todo:
                        return
                    j, seed = todo.pop(0)
                blob = prove(w, kind, seed)
                with lock:
                    out[j] = blob

        th = [threading.Thread(target=run, args=(w,)) for w in range(len(workers))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return out

    if True:
        for w in range(len(workers)):  # warm-up (tables, pools)
            prove(w, "block", 0)
            prove(w, "agg", 0)
        torch.cuda.synchronize()
        dist_mod.barrier()

        rows = []
        t_start = time.perf_counter()
        n, level, kind = args.proofs, 0, "block"
        children = None  # list of the previous level's blobs, identical on every rank
        while True:
            mine = dist_mod.shard_units(n, rank, world)
            t0 = time.perf_counter()
            seeds = {}
            for j in mine:
                if children is None:
                    seeds[j] = j
                else:
                    h = hashlib.sha256()
                    for c in children[j * args.arity:(j + 1) * args.arity]:
                        h.update(c)
                    seeds[j] = int.from_bytes(h.digest()[:7], "little")
            blobs = prove_units(kind, seeds)
            torch.cuda.synchronize()
            t_prove = time.perf_counter() - t0
            dist_mod.barrier()
            t1 = time.perf_counter()
            level_blobs = dist_mod.gather_indexed(blobs, n, device=device if world > 1 else None)  # the one exchange per level
            t_xchg = time.perf_counter() - t1
            t_prove_max, _ = dist_mod.reduce_max_sum(t_prove, len(mine), device=device if world > 1 else None)
            rows.append((level, kind, n, t_prove_max * 1e3, t_xchg * 1e3, sum(len(b) for b in level_blobs)))
            if n == 1:
                break
            children, n, level, kind = level_blobs, -(-n // args.arity), level + 1, "agg"
        total = time.perf_counter() - t_start

    if rank == 0:
        print("| level | circuit | proofs | prove ms (max over ranks) | exchange ms | bytes exchanged |")
        print("|---|---|---|---|---|---|")
        for r in rows:
            print("| %d | %s (2^%d rows) | %d | %.1f | %.2f | %d |" % (
                r[0], r[1], args.block_log_n if r[1] == "block" else args.agg_log_n, r[2], r[3], r[4], r[5]))
        print("\n%d GPU(s) x %d provers: tree over %d block proofs, arity %d: %.1f ms in all (%.2f block proofs/s aggregated); root proof %d bytes"
              % (world, len(workers), args.proofs, args.arity, total * 1e3, args.proofs / total, rows[-1][5]))
    for _, c, pcs in workers:
        for pc in pcs.values():
            pc.free()
        c.close()
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
