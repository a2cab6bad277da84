"""Fault-injection sweep behind tests/test_gpu_fault_injection.py (a helper run in a child process: a crash here is one failed
test, not a dead test session).  SURVEY.md §5 "Failure detection": GPU OOM / hipError -> Err, caller may fall back — the
reference surfaces prover errors as `Err` (src/rollup/circuits/mod.rs:1296,1303,1319,1323).

For every entry point E of {glp_circuit_build, glp_prove_dev, glp_prove (host witness), glp_prove_batch_dev (K = 4),
glp_prove_stream} on a 2^10-row 14-gate circuit, and for both fault kinds (the k-th pool allocation fails with OOM / the k-th
checked HIP call fails), k sweeps over EVERY such event of one call of E:

    * E returns GLP_ERR_OOM / GLP_ERR_HIP (never success, never a crash);
    * handle out-parameters are untouched (glp_circuit_build's data_out);
    * the context's pool is back at its baseline (blocks in use), i.e. nothing leaked;
    * the next call of E on the same context succeeds and gives the oracle's proof word for word.

Warm sweep: caches (twiddle tables, coset plans, arenas) exist.  Cold sweep: a FRESH context per k, so the failures land in the
first-use paths too (table builds, hipFuncSetAttribute, arena creation)."""
import ctypes
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

POOL, HIP = 0, 1
WANT = {POOL: -4, HIP: -3}


def counters(L, ctx):
    a, h, b, by = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    assert L.glp_ctx_debug_counters(ctx._h, ctypes.byref(a), ctypes.byref(h), ctypes.byref(b), ctypes.byref(by)) == 0
    return a.value, h.value, b.value, by.value


def arm(L, ctx, kind, k):
    assert L.glp_ctx_debug_fail_at(ctx._h, ctypes.c_uint32(kind), ctypes.c_uint64(k)) == 0


def main(bits=10, stride=1, cold_stride=3, host_only=0):
    """host_only = 1: only the two entry points that take HOST witnesses, warm sweep only — run at 2^16 rows, where the witness
    (70 MB) goes up in column chunks and the wires commitment starts on the first of them (glp_ctx::arrival_*): the error paths
    of the chunked upload."""
    import torch

    import __graft_entry__ as g
    from oracle import oracle as orc

    t_start = time.perf_counter()
    orc.set_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))  # the oracle's proofs of the 2^16-row variant: 8 s, not 100 s
    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    L = pkg.lib()
    mc = circ.MixedGateCircuit(bits, 3)
    cd = mc.data
    setup = pkg.Context(0)
    K = 2 if host_only else 4
    wit = [mc.witness_device(setup, pkg, seed=s) for s in range(K)]
    host_wit = [setup.to_host(w).copy() for w, _ in wit]
    pinned = [torch.from_numpy(h.view(np.int64)).pin_memory() for h in host_wit]
    cols1 = [np.array(c_, copy=True) for c_ in host_wit[1]]
    batch = torch.stack([w for w, _ in wit]).contiguous()
    pis = [p for _, p in wit]
    torch.cuda.synchronize()
    cs = orc.batch_from_values(cd.constants_sigmas_values(), 3, 4)
    cd.circuit_digest = orc.circuit_digest(cs["cap"])  # (ProverCircuit sets the same value when it builds)
    want = np.stack([orc.plonk_prove(cd, cs, host_wit[k], pis[k]) for k in range(K)])
    print("  [%.1f s] circuit, witnesses and the oracle's proofs ready" % (time.perf_counter() - t_start), file=sys.stderr, flush=True)

    def entry_points(pc):
        eps = [
            ("glp_prove_dev", lambda: pc.prove(wit[0][0], pis[0]), want[0]),
            ("glp_prove", lambda: pc.prove(host_wit[1], pis[1]), want[1]),
            ("glp_prove_batch_dev", lambda: pc.prove_batch(batch, pis), want),
            ("glp_prove_stream", lambda: pc.prove_stream(pinned, pis), want),
            # the witness as separate pageable columns: through the pinned ring, the leaf sponge following the chunks
            ("glp_prove_cols", lambda: pc.prove_cols(cols1, pis[1]), want[1]),
        ]
        return [e for e in eps if e[0] in ("glp_prove", "glp_prove_stream", "glp_prove_cols")] if host_only else eps

    report = {}

    def expect_failure(fn, kind, where):
        try:
            fn()
        except pkg.GlproverError as e:
            assert e.status == WANT[kind], "%s: status %d, wanted %d" % (where, e.status, WANT[kind])
            return True
        return False  # the armed event was not reached: the call has fewer such events than k

    # ---------------- warm sweep ----------------
    ctx = pkg.Context(0, use_torch_stream=False)
    pc = pkg.ProverCircuit(ctx, cd)
    for name, fn, w in entry_points(pc):
        assert (fn() == w).all(), name + ": warm-up result differs from the oracle"
    for name, fn, w in entry_points(pc):
        a0, h0, base_blocks, _ = counters(L, ctx)
        assert (fn() == w).all()
        a1, h1, blocks1, _ = counters(L, ctx)
        assert blocks1 == base_blocks, "%s leaves %d pool blocks behind on SUCCESS" % (name, blocks1 - base_blocks)
        n_events = {POOL: a1 - a0, HIP: h1 - h0}
        for kind in (POOL, HIP):
            fired = 0
            for k in range(1, n_events[kind] + 1, stride):
                arm(L, ctx, kind, k)
                failed = expect_failure(fn, kind, "%s kind %d k %d" % (name, kind, k))
                arm(L, ctx, kind, 0)
                fired += failed
                assert failed, "%s: event %d of %d (kind %d) did not fail the call" % (name, k, n_events[kind], kind)
                blocks = counters(L, ctx)[2]
                assert blocks == base_blocks, "%s kind %d k %d: %d pool blocks leaked" % (name, kind, k, blocks - base_blocks)
                got = fn()
                assert (got == w).all(), "%s kind %d k %d: the NEXT call's result differs from the oracle" % (name, kind, k)
            report["warm %s kind %d" % (name, kind)] = (n_events[kind], fired)
            print("  [%.1f s] warm %s kind %d: %d events, %d injected" % (time.perf_counter() - t_start, name, kind, n_events[kind], fired),
                  file=sys.stderr, flush=True)
    pc.free()
    ctx.close()
    if host_only:
        setup.close()
        for k, v in report.items():
            print("FAULT %-40s events %4d  injected %4d" % (k, v[0], v[1]))
        print("FAULT SWEEP OK")
        return

    # ---------------- cold sweep: circuit build + first proof on a fresh context per k ----------------
    probe = pkg.Context(0, use_torch_stream=False)
    a0, h0, b0, _ = counters(L, probe)
    pcp = pkg.ProverCircuit(probe, cd)
    a1, h1, b_circ, _ = counters(L, probe)
    assert (pcp.prove(wit[0][0], pis[0]) == want[0]).all()
    a2, h2, _, _ = counters(L, probe)
    pcp.free()
    probe.close()
    n_build = {POOL: a1 - a0, HIP: h1 - h0}
    n_first = {POOL: a2 - a1, HIP: h2 - h1}
    for kind in (POOL, HIP):
        fired = 0
        for k in range(1, n_build[kind] + 1, cold_stride):  # failures inside glp_circuit_build
            c2 = pkg.Context(0, use_torch_stream=False)
            base = counters(L, c2)[2]
            arm(L, c2, kind, k)
            try:
                pc2 = pkg.ProverCircuit(c2, cd)
                built = True
            except pkg.GlproverError as e:
                built = False
                assert e.status == WANT[kind], ("glp_circuit_build", kind, k, e.status)
            arm(L, c2, kind, 0)
            assert not built, "glp_circuit_build: event %d (kind %d) did not fail it" % (k, kind)
            fired += 1
            assert counters(L, c2)[2] == base, "glp_circuit_build kind %d k %d leaked pool blocks" % (kind, k)
            pc2 = pkg.ProverCircuit(c2, cd)  # the same context builds and proves afterwards
            assert (pc2.constants_sigmas_cap.elements == cs["cap"]).all()
            assert (pc2.prove(wit[0][0], pis[0]) == want[0]).all(), ("after a failed build", kind, k)
            pc2.free()
            c2.close()
        report["cold glp_circuit_build kind %d" % kind] = (n_build[kind], fired)
        fired = 0
        for k in range(1, n_first[kind] + 1, cold_stride):  # failures inside the FIRST proof of a context (first-use paths)
            c2 = pkg.Context(0, use_torch_stream=False)
            pc2 = pkg.ProverCircuit(c2, cd)
            base = counters(L, c2)[2]
            arm(L, c2, kind, k)
            failed = expect_failure(lambda: pc2.prove(wit[0][0], pis[0]), kind, "first proof kind %d k %d" % (kind, k))
            arm(L, c2, kind, 0)
            assert failed, "first proof: event %d (kind %d) did not fail it" % (k, kind)
            fired += 1
            assert counters(L, c2)[2] == base, "first proof kind %d k %d leaked pool blocks" % (kind, k)
            assert (pc2.prove(wit[0][0], pis[0]) == want[0]).all(), ("after a failed first proof", kind, k)
            assert (pc2.prove_batch(batch, pis) == want).all(), ("batch after a failed first proof", kind, k)
            pc2.free()
            c2.close()
        report["cold first proof kind %d" % kind] = (n_first[kind], fired)
    # the handle out-parameter of a failed build is untouched
    c3 = pkg.Context(0, use_torch_stream=False)
    pc_ok = pkg.ProverCircuit(c3, cd)
    sentinel = ctypes.c_void_p(0x1234)
    vals = np.ascontiguousarray(cd.constants_sigmas_values())
    arm(L, c3, POOL, 1)
    st = L.glp_circuit_build(c3._h, ctypes.byref(pc_ok._c), vals.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ctypes.byref(sentinel),
                             None, None)
    arm(L, c3, POOL, 0)
    assert st == -4 and sentinel.value == 0x1234, (st, sentinel.value)
    pc_ok.free()
    c3.close()
    setup.close()
    for k, v in report.items():
        print("FAULT %-40s events %4d  injected %4d" % (k, v[0], v[1]))
    print("FAULT SWEEP OK")


def trees_main():
    """the same sweep over the native tree entry points (host-array arguments: nothing but the context's pool and HIP calls can
    fail): glp_smt_root_of_set, glp_smt_proofs_of_set, glp_smt_apply_updates, glp_smt_roots_of_sets — every pool allocation and every
    checked HIP call of one call fails once; status, pool baseline and the next call's result are checked"""
    import __graft_entry__ as g
    from oracle import oracle as orc

    pkg = g.load_package()
    trees = importlib.import_module(pkg.__name__ + ".trees")
    L = pkg.lib()
    ctx = pkg.Context(0, use_torch_stream=False)
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 2**63, (3000, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (3000, 4), dtype=np.uint64)
    q = np.concatenate([keys[:200], rng.integers(0, 2**63, (100, 4), dtype=np.uint64)])
    uk = np.concatenate([keys[100:160], rng.integers(0, 2**63, (40, 4), dtype=np.uint64)])
    uv = rng.integers(0, 2**63, (100, 4), dtype=np.uint64)
    uv[::4] = 0
    sets = [(rng.integers(0, 2**63, (5, 4), dtype=np.uint64), rng.integers(1, 2**63, (5, 4), dtype=np.uint64)) for _ in range(300)]

    def find():
        r = trees.smt_find_batch(ctx, keys, vals, q, max_siblings=40)
        return np.concatenate([r.padded_siblings().ravel(), r.n_siblings.astype(np.uint64), r.value.ravel(), r.root])

    def upd():
        proofs, roots = trees.smt_apply_updates(ctx, keys, vals, uk, uv)
        return np.concatenate([roots.ravel(), np.array([p["fnc"] for p in proofs], dtype=np.uint64)])

    eps = [("glp_smt_root_of_set", lambda: trees.smt_root_of_set(ctx, keys, vals)),
           ("glp_smt_proofs_of_set", find),
           ("glp_smt_apply_updates", upd),
           ("glp_smt_roots_of_sets", lambda: trees.smt_roots_of_sets(ctx, sets).ravel())]
    assert (eps[0][1]() == orc.smt_root_of_set(keys, vals)).all()
    # the resident tree (glp_smt_open / _update / _root / _find / _entries / _close in ONE armed region): a failure anywhere surfaces as
    # an error and frees everything; a failed UPDATE must leave the tree as it was (root, size)
    root_base = np.asarray(orc.smt_root_of_set(keys, vals))

    def resident():
        rt = trees.ResidentSparseMerkleTree(ctx, keys, vals)
        try:
            try:
                proofs, roots = rt.update(uk, uv)
            except pkg.GlproverError:
                assert (rt.root() == root_base).all() and len(rt) == keys.shape[0], "a failed glp_smt_update changed the resident tree"
                raise
            pr = rt.find(q[:60])
            ek, ev = rt.entries()
            return np.concatenate([roots.ravel(), rt.root(), pr._sib.ravel(), pr.n_siblings.astype(np.uint64), ek.ravel(), ev.ravel()])
        finally:
            rt.close()

    eps.append(("glp_smt_open/update/find/entries", resident))
    for name, fn in eps:
        want = fn().copy()
        a0, h0, base, _ = counters(L, ctx)
        assert (fn() == want).all()
        a1, h1, b1, _ = counters(L, ctx)
        assert b1 == base, "%s leaves pool blocks behind on success" % name
        for kind, n_events in ((POOL, a1 - a0), (HIP, h1 - h0)):
            fired = 0
            for k in range(1, n_events + 1):
                arm(L, ctx, kind, k)
                try:
                    fn()
                    failed = False
                except pkg.GlproverError as e:
                    failed = True
                    assert e.status == WANT[kind], (name, kind, k, e.status)
                arm(L, ctx, kind, 0)
                # (trees.smt_apply_updates retries with a lengths-only call after GLP_ERR_INVALID only: an injected OOM / HIP error surfaces)
                assert failed, "%s: event %d of %d (kind %d) did not fail the call" % (name, k, n_events, kind)
                fired += 1
                assert counters(L, ctx)[2] == base, "%s kind %d k %d leaked pool blocks" % (name, kind, k)
                if k % 4 == 0 or k == n_events:
                    assert (fn() == want).all(), "%s kind %d k %d: the next call differs" % (name, kind, k)
            print("FAULT trees %-28s kind %d  events %4d  injected %4d" % (name, kind, n_events, fired))
    ctx.close()
    print("FAULT SWEEP OK")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trees":
        trees_main()
    else:
        main(*(int(x) for x in sys.argv[1:]))
