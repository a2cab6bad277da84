#!/usr/bin/env python3
"""The HOST-side state work of one block — what the reference does with its trees before any proof is made
(src/bin/block_circuit.rs:81-292,444-648) — as GPU batches through the native tree entry points:

    1. every sender's user-asset tree after the purge of the spent assets   trees.user_asset_roots      (4 calls for all senders)
    2. every transaction's diff tree (LayeredLayered) and its hash           trees.layered_smt_roots + two_to_one batch
    3. the world-state tree: one `set(sender, new asset root)` per sender,   trees.smt_apply_updates     (k sequential sets, 1 call)
       in order, with the process proofs the block circuit consumes
    4. the latest-account tree: `set(sender, block_number)` per sender       trees.smt_apply_updates
    5. the transactions digest (fixed-depth Merkle tree over the tx hashes)  trees.get_merkle_proof_with_zero
    6. the block hash                                                         trees.get_block_hash

The scenario is synthetic and of the reference's shape (A accounts in the world state, T transactions by distinct senders, a few
assets per sender, a few recipients per transaction).  GPU only: the CPU oracle is test infrastructure and is not used here —
tests/test_gpu_trees.py::test_block_state_pipeline_matches_oracle checks every root of this pipeline against it at a small size.
    python intmax-zkp-core_amd/tools/block_state_bench.py [--accounts 65536] [--txs 1024] [--reps 3]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
P = 0xFFFFFFFF00000001


def scenario(rng, n_accounts, n_txs, assets_per_user=4, recipients_per_tx=3):
    K = lambda x: (int(x), 0, 0, 0)  # noqa: E731
    addr = rng.integers(0, P, (n_accounts, 4), dtype=np.uint64)
    world_vals = rng.integers(1, P, (n_accounts, 4), dtype=np.uint64)  # current asset roots (opaque here)
    latest_vals = np.zeros((n_accounts, 4), dtype=np.uint64)
    latest_vals[:, 0] = rng.integers(1, 100, n_accounts)
    senders = rng.choice(n_accounts, n_txs, replace=False)
    users, diffs, nonces = [], [], rng.integers(0, P, (n_txs, 4), dtype=np.uint64)
    for s in senders:
        mk = K(1 + int(rng.integers(0, 2)))
        ent = []
        for a in range(assets_per_user):
            spent = a < 2  # the first two assets are purged by this transaction
            amount = (0, 0, 0, 0) if spent else K(int(rng.integers(1, 1 << 40)))
            ent.append((mk, K(int(rng.integers(0, 8))), K(a), amount))
        users.append(ent)
        diffs.append([(tuple(int(x) for x in addr[int(rng.integers(0, n_accounts))]), K(int(rng.integers(0, 8))), K(int(rng.integers(0, 4))),
                       K(int(rng.integers(1, 1 << 40)))) for _ in range(recipients_per_tx)])
    return dict(addr=addr, world_vals=world_vals, latest_vals=latest_vals, senders=senders, users=users, diffs=diffs, nonces=nonces,
                block_number=101)


def run_block(pkg, trees, ctx, sc, timings=None):
    """-> dict of the block's roots / digests; timings (optional dict): ms per stage accumulated"""
    def stage(name, t0):
        ctx.synchronize()
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)

    T = len(sc["senders"])
    t0 = time.perf_counter()
    asset_roots = trees.user_asset_roots(ctx, sc["users"])
    stage("1 user-asset trees", t0)
    t0 = time.perf_counter()
    diff_roots = trees.layered_smt_roots(ctx, sc["diffs"])
    tx_hashes = trees.two_to_one(ctx, diff_roots, sc["nonces"])
    stage("2 diff trees + tx hashes", t0)
    t0 = time.perf_counter()
    s_addr = sc["addr"][sc["senders"]]
    if sc.get("world_tree") is not None:  # --resident: the tree stays on the device from block to block (glp_smt_update)
        world_proofs, world_roots = sc["world_tree"].update(s_addr, asset_roots)
    else:
        world_proofs, world_roots = trees.smt_apply_updates(ctx, sc["addr"], sc["world_vals"], s_addr, asset_roots)
    stage("3 world state: T sequential sets", t0)
    t0 = time.perf_counter()
    bn = np.zeros((T, 4), dtype=np.uint64)
    bn[:, 0] = sc["block_number"]
    if sc.get("latest_tree") is not None:
        latest_proofs, latest_roots = sc["latest_tree"].update(s_addr, bn)
    else:
        latest_proofs, latest_roots = trees.smt_apply_updates(ctx, sc["addr"], sc["latest_vals"], s_addr, bn)
    stage("4 latest accounts: T sequential sets", t0)
    t0 = time.perf_counter()
    depth = max(1, int(np.ceil(np.log2(max(2, T)))))
    zero_tx = trees.two_to_one(ctx, np.zeros((1, 4), np.uint64), np.zeros((1, 4), np.uint64))[0]
    _, tx_digest = trees.get_merkle_proof_with_zero(ctx, tx_hashes, 0, depth, zero_tx)
    stage("5 transactions digest", t0)
    t0 = time.perf_counter()
    header = np.zeros((1, 29), dtype=np.uint64)
    header[0, 0] = sc["block_number"]
    header[0, 9:13] = tx_digest
    header[0, 17:21] = world_roots[-1]
    header[0, 21:25] = world_roots[-1]
    header[0, 25:29] = latest_roots[-1]
    block_hash = trees.get_block_hash(ctx, header)[0]
    stage("6 block hash", t0)
    return dict(asset_roots=asset_roots, diff_roots=diff_roots, tx_hashes=tx_hashes, world_roots=world_roots, world_proofs=world_proofs,
                latest_roots=latest_roots, latest_proofs=latest_proofs, tx_digest=tx_digest, block_hash=block_hash, header=header)


class TimedLib:
    """wraps libglprover's entry points: seconds spent INSIDE the library (and waiting for its GPU work), as opposed to the Python
    glue of trees.py around them (grouping entries into dicts, tuples for the process proofs) — a Rust host does that natively"""

    def __init__(self, lib):
        self._lib, self.seconds, self.calls, self.by_name = lib, 0.0, 0, {}

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if not name.startswith("glp_"):
            return f

        def timed(*a):
            t0 = time.perf_counter()
            r = f(*a)
            dt = time.perf_counter() - t0
            self.seconds += dt
            self.calls += 1
            self.by_name[name] = self.by_name.get(name, 0.0) + dt
            return r

        return timed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--accounts", type=int, default=1 << 16)
    ap.add_argument("--txs", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--resident", action="store_true", help="the world-state and latest-account trees stay on the device between blocks "
                    "(glp_smt_open once, glp_smt_update per block) instead of being rebuilt from the host arrays in every call")
    ap.add_argument("--cpu-perm-us", type=float, default=1.2, help="one Poseidon permutation on one host core (plonky2's AVX2 code on Zen 5)")
    args = ap.parse_args()
    import __graft_entry__ as g

    pkg = g.load_package()
    trees = importlib.import_module(pkg.__name__ + ".trees")
    ctx = pkg.Context(0)
    sc = scenario(np.random.default_rng(3), args.accounts, args.txs)
    if args.resident:
        sc["world_tree"] = trees.ResidentSparseMerkleTree(ctx, sc["addr"], sc["world_vals"])
        sc["latest_tree"] = trees.ResidentSparseMerkleTree(ctx, sc["addr"], sc["latest_vals"])
    out = run_block(pkg, trees, ctx, sc)  # warm-up
    timings = {}
    tl = TimedLib(pkg.lib())
    trees.lib = lambda: tl  # the tree helpers call through the timing proxy from here on
    t0 = time.perf_counter()
    for _ in range(args.reps):
        out = run_block(pkg, trees, ctx, sc, timings)
    total = (time.perf_counter() - t0) / args.reps
    in_lib = tl.seconds / args.reps
    depth = float(np.mean([len(p["siblings"]) for p in out["world_proofs"]]))
    # the reference's sequential cost in permutations: a `set` = find + re-hash of the path (depth + 2 for the leaf); the small trees
    # ~3 permutations per entry and layer; the digest T; per transaction
    perms = args.txs * (2 * (depth + 2) + 4 * 3 * 3 + 3 * 3 * 3 + 2)
    print(json.dumps({"accounts": args.accounts, "txs": args.txs, "resident_trees": bool(args.resident),
                      "ms_per_block_inside_the_library": 1e3 * in_lib, "library_calls_per_block": tl.calls // args.reps,
                      "blocks_per_s_library_time": 1 / in_lib, "txs_per_s_library_time": args.txs / in_lib,
                      "library_ms_per_block_by_entry_point": {k: round(1e3 * v / args.reps, 3) for k, v in sorted(tl.by_name.items(), key=lambda kv: -kv[1])},
                      "ms_per_block_with_the_python_glue": 1e3 * total,
                      "stage_ms_with_the_python_glue": {k: v / args.reps for k, v in timings.items()},
                      "mean_world_path": depth, "reference_permutations_per_block_estimate": int(perms),
                      "one_core_ms_estimate_at_%.1f_us_per_permutation" % args.cpu_perm_us: perms * args.cpu_perm_us / 1e3,
                      }))


if __name__ == "__main__":
    main()
