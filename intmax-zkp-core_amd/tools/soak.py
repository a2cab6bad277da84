#!/usr/bin/env python3
"""Soak test: many proofs and tree-hashing calls of varying sizes through one context; reports device memory in use
after each phase (the pool must stay bounded: size classes + cache budget) and that results stay identical."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import torch  # noqa: E402

circ = importlib.import_module(pkg.__name__ + ".circuit")
trees = importlib.import_module(pkg.__name__ + ".trees")
P = 0xFFFFFFFF00000001
ctx = pkg.Context(0)
rng = np.random.default_rng(0)


def used_gb():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**30


def pool_live():
    """(blocks, bytes) the library's pool has handed out and not got back: 0 between iterations means nothing leaks; what the device
    reports as "in use" beyond that is the pool's cache of free blocks (bounded by GLP_POOL_CACHE_GB) and torch's own cache"""
    import ctypes

    a, h, b, by = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    assert pkg.lib().glp_ctx_debug_counters(ctx._h, ctypes.byref(a), ctypes.byref(h), ctypes.byref(b), ctypes.byref(by)) == 0
    return b.value, by.value


print("start: %.2f GiB in use" % used_gb())
ref = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    db = int(rng.integers(4, 17))
    lc = circ.LayeredPoseidonCircuit(db, layers=min(16, (1 << db) // 4))
    pc = pkg.ProverCircuit(ctx, lc.data)
    wires, pis = lc.witness_device(ctx, pkg, seed=db)
    proof = pc.prove(wires, pis)
    key = ("proof", db)
    if key in ref:
        assert (ref[key] == proof).all(), "proof changed between runs"
    ref[key] = proof.copy()
    if db <= 12:  # a batch of a random size through the same context: proof 0 is the single proof
        K = int(rng.integers(2, 40))
        ws = [wires] + [lc.witness_device(ctx, pkg, seed=1000 * it + k) for k in range(1, K)]
        batch = torch.stack([w if k == 0 else w[0] for k, w in enumerate(ws)]).contiguous()
        proofs = pc.prove_batch(batch, [pis] + [w[1] for w in ws[1:]])
        assert (proofs[0] == proof).all(), "batched proof 0 differs from the single proof"
        del batch, ws
    pc.free()
    n = int(rng.integers(1, 200000))
    keys, vals = rng.integers(0, P, (n, 4), dtype=np.uint64), rng.integers(1, P, (n, 4), dtype=np.uint64)
    r1 = trees.smt_root_of_set(ctx, keys, vals)
    perm = rng.permutation(n)
    assert (trees.smt_root_of_set(ctx, keys[perm], vals[perm]) == r1).all()
    trees.two_to_one(ctx, keys, vals)
    trees.get_merkle_proof(ctx, keys[: min(n, 50000)], 0, 20)
    live = pool_live()
    assert live == (0, 0), "pool blocks still handed out after iteration %d: %r" % (it, live)
    if it % 5 == 4:
        print("iter %d: degree_bits %d, %d tree entries, %.2f GiB in use on the device (%.2f GiB of it torch's cache), pool blocks handed out: %d"
              % (it, db, n, used_gb(), torch.cuda.memory_reserved(0) / 2**30, live[0]))
print("done: %.2f GiB in use, %.2f GiB reserved by torch, pool blocks handed out: %d" % (used_gb(), torch.cuda.memory_reserved(0) / 2**30, pool_live()[0]))
