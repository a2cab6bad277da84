#!/usr/bin/env python3
"""Soak of the host-witness paths of round 6 (glp_prove_cols, glp_prove_batch_cols, glp_prove from a flat array, glp_prove_stream,
PolynomialBatch::from_values from host columns): random circuit sizes (2^10 .. 2^16 rows, 4-gate and 14-gate), witnesses as separate
pageable columns / one pageable array / one pinned array, one to four witnesses per call, interleaved on ONE context; every result must
equal the device-resident one, and between iterations the pool must have nothing handed out.  Run with GLP_CHUNKED_UPLOAD_MIN_MB=0 so
that every witness of 16 columns and more arrives in chunks (the ring, the chunked sponge), optionally GLP_POOL_CANARY=2.
usage: arrival_soak.py [iterations] [seed]"""
import ctypes
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import torch  # noqa: E402

circ = importlib.import_module(pkg.__name__ + ".circuit")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pkg.Context(0)


def pool_live():
    a, h, b, by = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    assert pkg.lib().glp_ctx_debug_counters(ctx._h, ctypes.byref(a), ctypes.byref(h), ctypes.byref(b), ctypes.byref(by)) == 0
    return b.value


circuits = {}


def circuit(bits, mixed):
    key = (bits, mixed)
    if key not in circuits:
        lc = (circ.MixedGateCircuit if mixed else circ.LayeredPoseidonCircuit)(bits, min(8, max(1, (1 << bits) // 128)))
        circuits[key] = (lc, pkg.ProverCircuit(ctx, lc.data))
        if len(circuits) > 6:  # keep the resident set small: drop the oldest
            old = next(iter(circuits))
            circuits.pop(old)[1].free()
    return circuits[key]


counts = {}
t0 = time.time()
for it in range(iters):
    bits = int(rng.choice([10, 11, 12, 13, 14, 15, 16], p=[0.1, 0.15, 0.15, 0.2, 0.2, 0.1, 0.1]))
    mixed = bool(rng.integers(0, 2))
    lc, pc = circuit(bits, mixed)
    before = pool_live()  # (the resident circuits' own blocks)
    K = int(rng.integers(1, 5)) if bits <= 14 else 1
    ws = [lc.witness_device(ctx, pkg, seed=int(rng.integers(0, 1 << 30))) for _ in range(K)]
    want = [pc.prove(w, p) for w, p in ws]
    hosts = [np.array(ctx.to_host(w), copy=True) for w, _ in ws]
    mode = str(rng.choice(["cols", "batch_cols", "flat", "pinned", "stream", "from_values"]))
    counts[mode] = counts.get(mode, 0) + 1
    if mode == "cols":
        for k in range(K):
            cols = [np.array(c_, copy=True) for c_ in hosts[k]]
            if rng.integers(0, 2):  # some columns as views into one big allocation at odd offsets (8-byte aligned, not 16)
                big = np.zeros(hosts[k].size + 135, dtype=np.uint64)
                off = 1
                for j in range(len(cols)):
                    big[off:off + cols[j].size] = cols[j]
                    cols[j] = big[off:off + cols[j].size]
                    off += cols[j].size + 1
            assert (pc.prove_cols(cols, ws[k][1]) == want[k]).all(), (it, mode, bits, k)
    elif mode == "batch_cols":
        got = pc.prove_batch_cols([[np.array(c_, copy=True) for c_ in h] for h in hosts], [p for _, p in ws])
        for k in range(K):
            assert (got[k] == want[k]).all(), (it, mode, bits, k)
    elif mode == "flat":
        for k in range(K):
            assert (pc.prove(hosts[k], ws[k][1]) == want[k]).all(), (it, mode, bits, k)
    elif mode == "pinned":
        for k in range(K):
            pt = torch.from_numpy(hosts[k].view(np.int64)).pin_memory()
            assert (pc.prove(pt.numpy().view(np.uint64), ws[k][1]) == want[k]).all(), (it, mode, bits, k)
    elif mode == "stream":
        pts = [torch.from_numpy(h.view(np.int64)).pin_memory() for h in hosts]
        got = pc.prove_stream(pts, [p for _, p in ws])
        for k in range(K):
            assert (got[k] == want[k]).all(), (it, mode, bits, k)
    else:
        dev = pkg.PolynomialBatch.from_values(ctx, ws[0][0], 3, False, 4)
        for src in (hosts[0], [np.array(c_, copy=True) for c_ in hosts[0]]):
            hb = pkg.PolynomialBatch.from_values(ctx, src, 3, False, 4)
            assert (hb.cap.elements == dev.cap.elements).all() and (hb.digests_host() == dev.digests_host()).all(), (it, mode, bits)
            hb.free()
        dev.free()
    del ws, hosts
    ctx.synchronize()
    live = pool_live()
    assert live == before, "pool blocks handed out and not returned: %d (before the iteration: %d) at iteration %d, mode %s" % (live, before, it, mode)
n = ctypes.c_uint64()
assert pkg.lib().glp_ctx_canary_violations(ctx._h, ctypes.byref(n)) == 0
print("ARRIVAL SOAK OK: %d iterations in %.0f s, %s, canary violations %d" % (iters, time.time() - t0, counts, n.value))
sys.exit(1 if n.value else 0)
