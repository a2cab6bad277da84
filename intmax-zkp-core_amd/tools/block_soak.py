#!/usr/bin/env python3
"""Soak of the block workload (workloads.run_block at reduced sizes): N runs; every run's proofs equal the first run's byte for byte and
the interleaved / wave forms equal the sequential one inside every run (side thread, second context).  python tools/block_soak.py [N]"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
m = {k: importlib.import_module(pkg.__name__ + "." + k) for k in ("workloads", "codec", "circuit", "dist")}
W = m["workloads"]
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = None
t0 = time.perf_counter()
bad = 0
for it in range(n_iter):
    res = W.run_block(pkg, m["circuit"], m["codec"], m["dist"], 0, 0, 1, n_blocks=4, user_log_n=8, block_log_n=10, batch_log_n=9, steps=1,
                      warmup=0, interleave=True, keep=True)
    kept = res["kept"]
    snap = {}
    for k, v in kept.items():
        if isinstance(v, dict):
            snap[k] = (tuple(v["user"]), tuple(v["sig"]), v["block"])
        else:
            snap[k] = v
    res["bp"].free() if hasattr(res["bp"], "free") else None
    if first is None:
        first = snap
        forms = sorted({k[0] for k in snap})
        print("kept forms:", forms, len(snap))
        # the three forms agree inside one run
    for k in snap:
        if snap[k] != first[k]:
            bad += 1
            print("iteration %d: %r differs from the first run" % (it, k))
    for b in range(4):
        for form in ("block_i", "block_w"):
            if (form, b) in snap and snap[(form, b)] != snap[("block", b)]:
                bad += 1
                print("iteration %d block %d: form %s differs from the sequential form" % (it, b, form))
print("BLOCK SOAK iterations=%d bad=%d  %.1f s" % (n_iter, bad, time.perf_counter() - t0))
