#!/usr/bin/env python3
"""Soak of the pipelined aggregation tree (workloads.run_aggtree(pipeline=True): level 0 of the next tree on a side thread / context):
N runs of four trees each; every level's serialized proofs equal the level-synchronous run's.  python tools/aggtree_soak.py [N]"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
m = {k: importlib.import_module(pkg.__name__ + "." + k) for k in ("workloads", "codec", "circuit", "dist")}
ctx = pkg.Context(0)
kw = dict(n_proofs=16, arity=4, block_log_n=10, agg_log_n=10, steps=4, warmup=0, keep=True)
plain = m["workloads"].run_aggtree(pkg, m["circuit"], m["codec"], m["dist"], ctx, 0, 1, **kw)
want = [[bytes(x) for x in b] for _, b in plain["level_blobs"]]
bad = 0
t0 = time.perf_counter()
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for it in range(n_iter):
    piped = m["workloads"].run_aggtree(pkg, m["circuit"], m["codec"], m["dist"], ctx, 0, 1, pipeline=True, **kw)
    got = [[bytes(x) for x in b] for _, b in piped["level_blobs"]]
    if got != want or not (piped["root_flat"] == plain["root_flat"]).all():
        bad += 1
        print("iteration %d: the pipelined tree differs" % it)
print("AGGTREE SOAK iterations=%d (4 trees each) bad=%d  %.1f s" % (n_iter, bad, time.perf_counter() - t0))
