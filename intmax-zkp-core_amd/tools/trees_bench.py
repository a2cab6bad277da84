#!/usr/bin/env python3
"""Throughput of the native tree hashing kernels on 1 x MI355X, device-resident inputs.  Prints a
markdown table (median of 5 after 2 warm-ups).  GPU only: the CPU oracle is test infrastructure and is not used here;
`--cpu-perm-us X` adds a ratio column from a per-permutation CPU cost measured elsewhere (bench.py's cpu_baseline leg
runs the oracle at ~4.5 us per permutation per thread)."""
import ctypes
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import importlib  # noqa: E402

trees = importlib.import_module(pkg.__name__ + ".trees")

P = 0xFFFFFFFF00000001
ctx = pkg.Context(0)
L = pkg.lib()
rng = np.random.default_rng(0)
sz = ctypes.c_size_t


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def dev(a):
    return ctx.to_device(a)


PERM_US = float(sys.argv[sys.argv.index("--cpu-perm-us") + 1]) if "--cpu-perm-us" in sys.argv else 0.0
print("| op | items | GPU ms | items/s | Poseidon perms/s | CPU us/item at %.2f us/perm (1 thread) | GPU/CPU-thread |" % PERM_US)
print("|---|---|---|---|---|---|---|")


def row(name, n, t, perms_per_item):
    cpu_us = perms_per_item * PERM_US
    print("| %s | %d | %.3f | %.3g | %.3g | %.2f | %.0fx |" % (name, n, t * 1e3, n / t, n * perms_per_item / t, cpu_us,
                                                           cpu_us * 1e-6 / (t / n)))


n = 1 << 22
l, r = rng.integers(0, P, (n, 4), dtype=np.uint64), rng.integers(0, P, (n, 4), dtype=np.uint64)
ld, rd, od = dev(l), dev(r), ctx.new_tensor((n, 4))
t = timed(lambda: L.glp_two_to_one_batch_dev(ctx._h, pkg._dev_ptr(ld), pkg._dev_ptr(rd), sz(n), pkg._dev_ptr(od)))
row("two_to_one batch", n, t, 1)
t = timed(lambda: L.glp_smt_leaf_hash_batch_dev(ctx._h, pkg._dev_ptr(ld), pkg._dev_ptr(rd), sz(n), pkg._dev_ptr(od)))
row("SMT leaf hash batch", n, t, 2)

n, depth = 1 << 18, 32
idx = rng.integers(0, 1 << 32, n, dtype=np.uint64)
val, sib = rng.integers(0, P, (n, 4), dtype=np.uint64), rng.integers(0, P, (n, depth, 4), dtype=np.uint64)
idd, vd, sd, od = dev(idx), dev(val), dev(sib), ctx.new_tensor((n, 4))
t = timed(lambda: L.glp_merkle_root_batch_dev(ctx._h, pkg._dev_ptr(idd), pkg._dev_ptr(vd), pkg._dev_ptr(sd), sz(n), sz(depth),
                                              pkg._dev_ptr(od)))
row("get_merkle_root, depth 32", n, t, depth)

n = 1 << 20
leaves = rng.integers(0, P, (n, 4), dtype=np.uint64)
lvd = dev(leaves)
z = (ctypes.c_uint64 * 4)(0, 0, 0, 0)
sib_out, root_out = np.zeros((32, 4), np.uint64), np.zeros(4, np.uint64)
u64p = ctypes.POINTER(ctypes.c_uint64)
t = timed(lambda: L.glp_merkle_proof_with_zero_dev(ctx._h, pkg._dev_ptr(lvd), sz(n), sz(n - 1), sz(32), z,
                                                   sib_out.ctypes.data_as(u64p), root_out.ctypes.data_as(u64p)))
row("get_merkle_proof (2^20 leaves, depth 32), per leaf", n, t, 1)

for n in (1 << 16, 1 << 20):
    keys, vals = rng.integers(0, P, (n, 4), dtype=np.uint64), rng.integers(1, P, (n, 4), dtype=np.uint64)
    t = timed(lambda: trees.smt_root_of_set(ctx, keys, vals), reps=3, warm=1)
    row("smt_root_of_set (upload + device sort included), per entry", n, t, 4)

# batched `find` (glp_smt_proofs_of_set): tree build + one walk kernel; per QUERY; the CPU column prices the reference's find as the
# ~depth sibling fetches it is (no hashing: the node store holds the hashes) — the work the batch replaces is the tree
# maintenance itself, so the honest comparison is the root_of_set row above plus this row's marginal cost over it
for n, m in ((1 << 16, 1 << 12), (1 << 20, 1 << 16)):
    keys, vals = rng.integers(0, P, (n, 4), dtype=np.uint64), rng.integers(1, P, (n, 4), dtype=np.uint64)
    q = np.concatenate([keys[rng.integers(0, n, m // 2)], rng.integers(0, P, (m - m // 2, 4), dtype=np.uint64)])
    t_root = timed(lambda: trees.smt_root_of_set(ctx, keys, vals), reps=3, warm=1)
    t = timed(lambda: trees.smt_find_batch(ctx, keys, vals, q, max_siblings=64), reps=3, warm=1)
    print("| smt_find_batch: %d proofs out of a %d-entry set (tree build included; + %.3f ms over smt_root_of_set) | %d | %.3f | %.3g | - | - | - |"
          % (m, n, (t - t_root) * 1e3, m, t * 1e3, m / t))

# k sequential `set`s as one batch (glp_smt_apply_updates): the reference walks and re-hashes one path per `set`, ~(depth + 2)
# permutations that depend on the previous `set`; the CPU column prices exactly that at the given per-permutation cost
for n, k in ((1 << 16, 1 << 10), (1 << 20, 1 << 12), (1 << 20, 1 << 14)):
    keys, vals = rng.integers(0, P, (n, 4), dtype=np.uint64), rng.integers(1, P, (n, 4), dtype=np.uint64)
    uk = np.concatenate([keys[rng.integers(0, n, k // 2)], rng.integers(0, P, (k - k // 2, 4), dtype=np.uint64)])
    uv = rng.integers(1, P, (k, 4), dtype=np.uint64)
    uv[::5] = 0
    perm = rng.permutation(k)
    uk, uv = uk[perm], uv[perm]
    t_root = timed(lambda: trees.smt_root_of_set(ctx, keys, vals), reps=3, warm=1)
    ns = np.zeros(k, dtype=np.uint32)
    roots = np.zeros((k + 1, 4), dtype=np.uint64)
    u32p = ctypes.POINTER(ctypes.c_uint32)

    def run():
        st = L.glp_smt_apply_updates(ctx._h, keys.ctypes.data_as(u64p), vals.ctypes.data_as(u64p), sz(n), uk.ctypes.data_as(u64p),
                                     uv.ctypes.data_as(u64p), sz(k), sz(0), None, ns.ctypes.data_as(u32p), None, None, None, None, None,
                                     None, roots.ctypes.data_as(u64p))
        assert st == 0

    t = timed(run, reps=3, warm=1)
    depth = float(ns.mean())
    cpu_us = (depth + 2) * PERM_US
    print("| smt_apply_updates: %d sequential sets on a %d-entry tree (tree build included; + %.3f ms over smt_root_of_set; mean path %.1f) | %d | %.3f | %.3g | - | %.2f | %.0fx (over the build: %.0fx) |"
          % (k, n, (t - t_root) * 1e3, depth, k, t * 1e3, k / t, cpu_us, cpu_us * 1e-6 * k / t, cpu_us * 1e-6 * k / max(1e-9, t - t_root)))

    # the same batch on a tree that stays on the device (glp_smt_open once, then glp_smt_update per block): no upload / sort / hashing
    # of the whole set per call; the tree is rebuilt from device data after the batch.  Successive calls keep changing the tree (the
    # batch is applied again and again: updates of existing keys, re-inserts after removals), which is what a node does.
    h = ctypes.c_void_p()
    assert L.glp_smt_open(ctx._h, keys.ctypes.data_as(u64p), vals.ctypes.data_as(u64p), sz(n), ctypes.byref(h)) == 0

    def run_resident():
        st = L.glp_smt_update(h, uk.ctypes.data_as(u64p), uv.ctypes.data_as(u64p), sz(k), sz(0), None, ns.ctypes.data_as(u32p), None, None,
                              None, None, None, None, roots.ctypes.data_as(u64p))
        assert st == 0

    tr = timed(run_resident, reps=3, warm=1)
    L.glp_smt_close(h)
    print("| glp_smt_update: the same %d sets on a RESIDENT %d-entry tree (no build per call; the tree rebuilt on the device afterwards) | %d | %.3f | %.3g | - | %.2f | %.0fx |"
          % (k, n, k, tr * 1e3, k / tr, cpu_us, cpu_us * 1e-6 * k / tr))

# many small trees per call (glp_smt_roots_of_sets): S sets of e entries each, against S calls of smt_root_of_set
for S, e in ((4096, 6), (65536, 4)):
    sets = [(rng.integers(0, P, (e, 4), dtype=np.uint64), rng.integers(1, P, (e, 4), dtype=np.uint64)) for _ in range(S)]
    t = timed(lambda: trees.smt_roots_of_sets(ctx, sets), reps=3, warm=1)
    sample = min(S, 200)
    t_one = timed(lambda: [trees.smt_root_of_set(ctx, k_, v_) for k_, v_ in sets[:sample]], reps=2, warm=1) / sample
    print("| smt_roots_of_sets: %d trees of %d entries in one call (one call per tree: %.3f ms each = %.1f ms) | %d | %.3f | %.3g | - | %.2f | %.0fx |"
          % (S, e, t_one * 1e3, t_one * S * 1e3, S, t * 1e3, S / t, (3 * e) * PERM_US, (3 * e) * PERM_US * 1e-6 * S / t))
