#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on MI355X.

Workload (BASELINE.json configs[1], the microbench slice of the 2^20-row block proof):
one STEP = the polynomial-commitment phase of one 2^20-row proof under standard_recursion_config, i.e.
    PolynomialBatch::from_values(wires,               135 cols x 2^20, rate_bits 3, cap_height 4)
    PolynomialBatch::from_values(Z + partial products, 20 cols x 2^20, ...)
    PolynomialBatch::from_coeffs(quotient chunks,       16 cols x 2^20, ...)
(iNTT -> x8 coset LDE -> Poseidon leaf sponge -> Merkle levels -> 2^4 cap, for each).  Inputs are synthetic
(SplitMix64 columns, uniform in [0,p)) and already resident in HBM when the timed region starts.  This is NOT yet
a whole proof (quotient evaluation, openings and the FRI loop are not in the step); the metric name says so.

    python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 launch with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: one process per
GPU, independent proofs per rank (no data-path collective), barrier + max-over-ranks timing.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
RATE_BITS = 3
CAP_H = 4
COMMITS = (("wires", 135, False), ("zs_partial_products", 20, False), ("quotient_chunks", 16, True))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable by a streaming copy


def leaf_hash_algorithmic_bytes(n_cols, log_n=LOG_N, rate_bits=RATE_BITS):
    """compulsory traffic of ONE leaf-hash launch: read the N x C LDE matrix once, write N 32-byte digests"""
    N = 1 << (log_n + rate_bits)
    return 8 * N * n_cols + 32 * N


def perms_per_commit(n_cols, log_n=LOG_N, rate_bits=RATE_BITS, cap_h=CAP_H):
    N = 1 << (log_n + rate_bits)
    return N * (-(-n_cols // 8)) + N - (1 << cap_h)


def cpu_baseline(pkg, log_n_sample, threads):
    """Oracle (C restatement, `port`) timed on the host cores on a bounded sample of the same step."""
    import numpy as np
    from oracle import oracle as orc

    orc.set_threads(threads)
    t_total = 0.0
    for name, C, is_coeffs in COMMITS:
        data = pkg.splitmix_columns(C, 1 << log_n_sample)
        t0 = time.perf_counter()
        if is_coeffs:
            orc.batch_from_coeffs(data, RATE_BITS, CAP_H)
        else:
            orc.batch_from_values(data, RATE_BITS, CAP_H)
        t_total += time.perf_counter() - t0
    scale = 1 << (LOG_N - log_n_sample)
    est_step_s = t_total * scale  # linear scaling in n (ignores the log factor of the NTT: favours the CPU)
    return {
        "value": 1.0 / est_step_s,
        "unit": "commit-phases/s",
        "cores": threads,
        "kind": "port",
        "sample": "oracle from_values(135)+from_values(20)+from_coeffs(16) at n=2^%d, %d OpenMP threads, %.2f s measured, "
                  "scaled x%d to n=2^20" % (log_n_sample, threads, t_total, scale),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mds", type=int, default=int(os.environ.get("GLP_MDS", "-1")), help="Poseidon MDS variant (0 int, 1 f64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log-n", type=int, default=15)
    args = ap.parse_args()

    import numpy as np
    import torch

    import __graft_entry__ as g

    pkg = g.load_package()
    import importlib

    dist_mod = importlib.import_module("intmax_zkp_core_amd.dist")
    rank, local_rank, world = dist_mod.init()
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node = --gpus"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libglprover has no CPU fallback")
    dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = pkg.Context(dev)  # binds the context to this torch stream
        if args.mds >= 0:
            ctx.set_mds_variant(args.mds)
        inputs = []
        for name, C, is_coeffs in COMMITS:
            host = pkg.splitmix_columns(C, 1 << LOG_N)
            if rank:  # different synthetic proof per rank
                host = np.roll(host, rank, axis=1)
            inputs.append(ctx.to_device(host))
            del host

        def step():
            caps = []
            for (name, C, is_coeffs), d in zip(COMMITS, inputs):
                if is_coeffs:
                    b = pkg.PolynomialBatch.from_coeffs(ctx, d, RATE_BITS, False, CAP_H)
                else:
                    b = pkg.PolynomialBatch.from_values(ctx, d, RATE_BITS, False, CAP_H)
                caps.append(b.cap.elements)
                b.free()
            return caps

        for _ in range(args.warmup):
            step()
        ctx.timing_enable(True)
        ctx.timing_reset()
        torch.cuda.synchronize()
        dist_mod.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            caps = step()
        torch.cuda.synchronize()
        dist_mod.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timing = ctx.timing_read()
        ctx.timing_enable(False)

        # dominant kernel, isolated: the C = 135 leaf-hash launch (one per step), timed live with HIP events on the
        # context's stream
        ctx.timing_enable(True)
        ctx.timing_reset()
        b = pkg.PolynomialBatch.from_values(ctx, inputs[0], RATE_BITS, False, CAP_H)
        t135 = ctx.timing_read()
        ctx.timing_enable(False)
        b.free()

    total_elapsed, total_steps = dist_mod.reduce_max_sum(elapsed, args.steps, device="cuda:%d" % dev if world > 1 else None)

    if rank == 0:
        leaf_ms = t135["leaf_hash"]["ms"] / max(1, t135["leaf_hash"]["launches"])
        alg = leaf_hash_algorithmic_bytes(135)
        achieved = alg / (leaf_ms * 1e-3) / 1e9
        perms_leaf = (1 << (LOG_N + RATE_BITS)) * 17
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "leaf_hash_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "polynomial-commitment phase (3x PolynomialBatch commit: wires 135 + Z/pp 20 + quotient 16 cols) of a "
                      "2^20-row block proof, per second — BASELINE configs[1] slice; not yet a whole proof",
            "value": total_steps / total_elapsed,
            "unit": "commit-phases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * total_elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks field, integer modular)",
            "data": "synthetic (SplitMix64 columns uniform in [0,p), HBM-resident before timing)",
            "config": {"workload": "configs[1]: 2^20-point Goldilocks NTT/LDE + Poseidon Merkle-cap (lde_commit) for "
                                   "C = 135, 20, 16 columns; rate_bits 3, cap_height 4",
                       "log_n": LOG_N, "rate_bits": RATE_BITS, "cap_height": CAP_H,
                       "sharding": "independent proofs per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": "merkle::leaf_hash_kernel (C=135, 2^23 leaves)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": leaf_ms,
                         "note": "Poseidon hashing is integer-VALU-bound, not HBM-bound (17 permutations per 1080-byte "
                                 "leaf); perms/s is the meaningful rate",
                         "poseidon_gperms_per_s": perms_leaf / (leaf_ms * 1e-3) / 1e9},
            "phase_ms_per_step": {k: v["ms"] / args.steps for k, v in timing.items() if v["launches"]},
            "perms_per_step": sum(perms_per_commit(C) for _, C, _ in COMMITS),
        }
        if not args.no_cpu_baseline:
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                threads = os.cpu_count() or 1
            threads = max(1, min(threads, 64))  # the oracle's per-column / per-leaf split stops scaling beyond this
            out["cpu_baseline"] = cpu_baseline(pkg, args.cpu_log_n, threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
