"""Fault injection on the GPU (SURVEY.md §5 "GPU OOM / hipError -> Err, caller may fall back"): tests/fault_sweep.py in a child
process — every pool allocation and every checked HIP call of glp_circuit_build / glp_prove_dev / glp_prove /
glp_prove_batch_dev / glp_prove_stream is made to fail once; each must come back as GLP_ERR_OOM / GLP_ERR_HIP with nothing leaked,
and the next call on the same context must give the oracle's proof."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_injected_failure_surfaces_as_an_error_and_the_context_recovers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fault_sweep.py")], capture_output=True, text=True, timeout=1800,
                       cwd=ROOT)
    assert r.returncode == 0 and "FAULT SWEEP OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("FAULT ") and "events" in ln]
    assert len(rows) == 14, rows  # 5 entry points x 2 kinds warm, (build + first proof) x 2 kinds cold
    for ln in rows:
        events = int(ln.split("events")[1].split()[0])
        assert events > 0, ln


def test_failures_inside_the_chunked_witness_upload():
    """2^16 rows: the host witness (70 MB) travels in column chunks on the copy stream while the wires commitment already runs
    (glp_prove, first witness of glp_prove_stream, glp_prove_cols through the pinned ring, the leaf sponge following the chunks) — every
    fifth event of those three entry points fails once"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fault_sweep.py"), "16", "5", "5", "1"], capture_output=True,
                       text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0 and "FAULT SWEEP OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("FAULT warm")]) == 6


def test_failures_inside_the_native_tree_entry_points():
    """glp_smt_root_of_set, glp_smt_proofs_of_set, glp_smt_apply_updates, glp_smt_roots_of_sets and the resident tree (glp_smt_open /
    _update / _find / _entries / _root / _close in one armed region; a failed update must leave the tree as it was): every pool
    allocation and every checked HIP call fails once"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fault_sweep.py"), "trees"], capture_output=True, text=True,
                       timeout=1800, cwd=ROOT)
    assert r.returncode == 0 and "FAULT SWEEP OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("FAULT trees")]
    assert len(rows) == 10 and all(int(ln.split("events")[1].split()[0]) > 0 for ln in rows), rows


def test_fault_switches_from_the_environment():
    """GLP_POOL_FAIL_AT / GLP_HIP_FAIL_AT arm the same counters when a context is created"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import __graft_entry__ as g\n"
            "pkg = g.load_package(); import numpy as np\n"
            "ctx = pkg.Context(0, use_torch_stream=False)\n"
            "import torch; v = torch.from_numpy(pkg.splitmix_columns(3, 1 << 10).view(np.int64)).to('cuda:0')  # (not through the library: its calls are counted)\n"
            "torch.cuda.synchronize()\n"
            "try:\n"
            "    pkg.PolynomialBatch.from_values(ctx, v, 3, False, 4); print('STATUS 0')\n"
            "except pkg.GlproverError as e:\n"
            "    print('STATUS', e.status)\n"
            "b = pkg.PolynomialBatch.from_values(ctx, v, 3, False, 4); print('THEN ok', b.cap.elements.shape[0])\n") % ROOT
    for var, want in (("GLP_POOL_FAIL_AT", "-4"), ("GLP_HIP_FAIL_AT", "-3")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **{var: "2"}))
        assert "STATUS " + want in r.stdout and "THEN ok 16" in r.stdout, (var, r.stdout[-500:], r.stderr[-1500:])


def test_a_real_out_of_memory_does_not_poison_the_next_call():
    """Not injected: an allocation larger than the GPU really fails inside hipMalloc.  The HIP runtime keeps a failed call as the
    thread's "last error" until it is read — the launch checks (`hipGetLastError()` after a kernel) of the NEXT call would report it
    as their own.  After GLP_ERR_OOM (and after an out-of-memory that the pool recovered from by dropping caches) the same context,
    on the same thread, builds a circuit and proves; the proof equals the one made before."""
    import ctypes
    import importlib

    import numpy as np
    import torch

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    L = pkg.lib()
    ctx = pkg.Context(0)
    lc = circ.MixedGateCircuit(10, 16)
    pc = pkg.ProverCircuit(ctx, lc.data)
    wires, pis = lc.witness_device(ctx, pkg, seed=1)
    want = np.array(pc.prove(wires, pis), copy=True)
    total = torch.cuda.get_device_properties(0).total_memory
    p = ctypes.c_void_p()
    assert L.glp_dev_alloc(ctx._h, ctypes.c_size_t(2 * total), ctypes.byref(p)) == -4 and not p.value  # GLP_ERR_OOM, out-param untouched
    assert (pc.prove(wires, pis) == want).all()
    pc2 = pkg.ProverCircuit(ctx, lc.data)  # the first launches after the failure that are checked with hipGetLastError
    assert (pc2.prove(wires, pis) == want).all()
    pc2.free()
    # an out-of-memory the pool recovers from: a second context's cache holds 30 % of the memory, this one asks for 80 %
    other = pkg.Context(0, use_torch_stream=False)
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info(0)
    q = ctypes.c_void_p()
    assert L.glp_dev_alloc(other._h, ctypes.c_size_t(int(free0 * 0.30)), ctypes.byref(q)) == 0 and L.glp_dev_free(other._h, q) == 0
    assert L.glp_dev_alloc(ctx._h, ctypes.c_size_t(int(free0 * 0.80)), ctypes.byref(q)) == 0 and L.glp_dev_free(ctx._h, q) == 0
    pc3 = pkg.ProverCircuit(ctx, lc.data)
    assert (pc3.prove(wires, pis) == want).all()
    pc3.free()
    pc.free()
    other.close()
    ctx.close()


def test_real_out_of_memory_in_the_middle_of_a_proof():
    """The injected failures above never reach the driver.  Here the GPU is REALLY short of memory: a fresh context (cold pool: every
    block is a hipMalloc) proves a 2^16-row circuit while a torch tensor occupies all of the HBM but L GB, for a ladder of L — the
    proof runs out of memory at a different point each time (or fits).  Every failure must be GLP_ERR_OOM with no block left handed
    out, and after the memory is given back the SAME context proves, word for word."""
    import ctypes
    import importlib

    import numpy as np
    import torch

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    L = pkg.lib()
    lc = circ.MixedGateCircuit(16, 16)
    base = pkg.Context(0)
    pc0 = pkg.ProverCircuit(base, lc.data)
    wires, pis = lc.witness_device(base, pkg, seed=2)
    torch.cuda.synchronize()
    want = np.array(pc0.prove(wires, pis), copy=True)
    pc0.free()
    base.close()

    def live(ctx):
        a, h, b, by = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        assert L.glp_ctx_debug_counters(ctx._h, ctypes.byref(a), ctypes.byref(h), ctypes.byref(b), ctypes.byref(by)) == 0
        return b.value

    def allocs(ctx):
        a = ctypes.c_uint64()
        assert L.glp_ctx_debug_counters(ctx._h, ctypes.byref(a), None, None, None) == 0
        return a.value

    failed = fitted = 0
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(0)
    room_gb = 4.0  # one large block takes all of the HBM but this once (a 285 GB hipMalloc takes seconds); the ladder squeezes the rest
    big = torch.empty(free - int(room_gb * 2**30), dtype=torch.uint8, device="cuda:0")
    for left_gb in (0.05, 0.3, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95, 1.0, 1.2, 2.0):
        ctx = pkg.Context(0, use_torch_stream=False)
        pc = pkg.ProverCircuit(ctx, lc.data)  # the circuit's own data first, then the squeeze
        blocks_before = live(ctx)
        torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info(0)
        hog = torch.empty(max(1, free - int(left_gb * 2**30)), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        allocs_before = allocs(ctx)
        try:
            got = pc.prove(wires, pis)
            assert (got == want).all(), "proof differs with %.2f GB free" % left_gb
            fitted += 1
            print("%.2f GB free: the proof fits" % left_gb)
        except pkg.GlproverError as e:
            assert e.status == -4, "with %.2f GB free: status %d, not GLP_ERR_OOM" % (left_gb, e.status)
            failed += 1
            print("%.2f GB free: GLP_ERR_OOM after %d pool allocations" % (left_gb, allocs(ctx) - allocs_before))
        assert live(ctx) == blocks_before, "blocks still handed out after the call with %.2f GB free" % left_gb
        del hog
        torch.cuda.empty_cache()
        assert (pc.prove(wires, pis) == want).all(), "the context does not prove after the out-of-memory at %.2f GB" % left_gb
        pc.free()
        ctx.close()
    del big
    torch.cuda.empty_cache()
    assert failed >= 3 and fitted >= 1, (failed, fitted)  # the ladder really straddles the proof's footprint


def test_real_out_of_memory_in_the_other_entry_points():
    """the same squeeze for the batched prover (4 proofs of 2^14 rows), the host-pointer prover (2^16 rows: chunked upload through
    pinned staging), `glp_circuit_build` (2^16 rows) and 512 sequential sparse-tree `set`s on a 2^18-entry set: a ladder of free memory per
    operation; every refusal is GLP_ERR_OOM with nothing left handed out, and the same context gives the unsqueezed result afterwards"""
    import ctypes
    import importlib

    import numpy as np
    import torch

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    trees = importlib.import_module(pkg.__name__ + ".trees")
    L = pkg.lib()
    P = 0xFFFFFFFF00000001

    def counters(ctx):
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        assert L.glp_ctx_debug_counters(ctx._h, ctypes.byref(a), None, ctypes.byref(b), None) == 0
        return a.value, b.value

    base = pkg.Context(0)
    small, big = circ.MixedGateCircuit(14, 16), circ.MixedGateCircuit(16, 16)
    ws = [small.witness_device(base, pkg, seed=s) for s in range(4)]
    wb, pb = big.witness_device(base, pkg, seed=5)
    wb_host = base.to_host(wb)
    batch = torch.stack([w for w, _ in ws]).contiguous()
    torch.cuda.synchronize()
    rng = np.random.default_rng(9)
    n = 1 << 18
    keys, vals = rng.integers(0, P, (n, 4), dtype=np.uint64), rng.integers(1, P, (n, 4), dtype=np.uint64)
    uk, uv = keys[rng.choice(n, 512, replace=False)].copy(), rng.integers(1, P, (512, 4), dtype=np.uint64)

    def flat(x):  # results of the tree helpers: nested tuples / lists / arrays -> one comparable array
        if isinstance(x, np.ndarray):
            return x.astype(np.uint64).ravel()
        if isinstance(x, (list, tuple)):
            return np.concatenate([flat(y) for y in x]) if len(x) else np.zeros(0, np.uint64)
        if isinstance(x, dict):
            return flat([x[k] for k in sorted(x)])
        return np.array([int(x)], dtype=np.uint64)

    ops = {
        # name: (setup(ctx) -> state, run(ctx, state) -> result, ladder of GB left free)
        "prove_batch 4 x 2^14": (lambda c: pkg.ProverCircuit(c, small.data), lambda c, pc: pc.prove_batch(batch, [p for _, p in ws]),
                                 (0.02, 0.2, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.5)),
        "prove from a host pointer, 2^16": (lambda c: pkg.ProverCircuit(c, big.data), lambda c, pc: pc.prove(wb_host, pb),
                                            (0.02, 0.2, 0.4, 0.6, 0.7, 0.8, 0.9, 1.0, 1.2, 2.0)),
        "circuit build 2^16": (lambda c: None, lambda c, _: pkg.ProverCircuit(c, big.data).constants_sigmas_cap.elements,
                               (0.01, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.8, 1.5)),
        "smt_apply_updates 2^18 + 512": (lambda c: None, lambda c, _: flat(trees.smt_apply_updates(c, keys, vals, uk, uv)),
                                         (0.002, 0.01, 0.02, 0.03, 0.04, 0.06, 0.08, 0.12, 0.3)),
    }
    want = {}
    for name, (setup, run, _) in ops.items():
        st = setup(base)
        want[name] = np.array(run(base, st), copy=True)
        if hasattr(st, "free"):
            st.free()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(0)
    hold = torch.empty(free - int(4.0 * 2**30), dtype=torch.uint8, device="cuda:0")  # all of the HBM but 4 GB, once
    for name, (setup, run, ladder) in ops.items():
        failed = fitted = 0
        for left_gb in ladder:
            ctx = pkg.Context(0, use_torch_stream=False)
            st = setup(ctx)
            live_before = counters(ctx)[1]
            torch.cuda.empty_cache()
            free, _ = torch.cuda.mem_get_info(0)
            hog = torch.empty(max(1, free - int(left_gb * 2**30)), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            a0 = counters(ctx)[0]
            try:
                got = run(ctx, st)
                assert (np.asarray(got) == want[name]).all(), "%s differs with %.3f GB free" % (name, left_gb)
                fitted += 1
                print("%s, %.3f GB free: fits" % (name, left_gb))
            except pkg.GlproverError as e:
                assert e.status == -4, "%s with %.3f GB free: status %d, not GLP_ERR_OOM" % (name, left_gb, e.status)
                failed += 1
                print("%s, %.3f GB free: GLP_ERR_OOM after %d pool allocations" % (name, left_gb, counters(ctx)[0] - a0))
            got = None
            import gc

            gc.collect()  # (a half-built ProverCircuit wrapper of the failed build)
            assert counters(ctx)[1] == live_before, "%s: blocks still handed out after the call with %.3f GB free" % (name, left_gb)
            del hog
            torch.cuda.empty_cache()
            again = run(ctx, st)
            assert (np.asarray(again) == want[name]).all(), "%s: wrong result after the out-of-memory at %.3f GB" % (name, left_gb)
            again = None
            gc.collect()
            if hasattr(st, "free"):
                st.free()
            ctx.close()
        assert failed >= 1 and fitted >= 1, (name, failed, fitted)
    del hold
    torch.cuda.empty_cache()  # (torch would keep the 284 GB block in its own cache: the tests after this one need the memory)
    base.close()
