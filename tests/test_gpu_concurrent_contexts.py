"""SURVEY.md §8b threading contract on the GPU: eight host threads, eight contexts, different circuits, sizes and entry points at the same
time (tests/concurrent_mix.py) — every concurrent result equals the sequential one word for word, in this process and again
in a child process under GLP_POOL_CANARY=2 (every pool block guarded; the first overrun aborts the child)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_contexts_on_eight_threads_give_the_sequential_results():
    from tests import concurrent_mix

    errors, mismatches, canary = concurrent_mix.run(rounds=20, verify=True)
    assert not errors, errors
    assert not mismatches, mismatches
    assert canary == 0


def test_the_same_mix_with_every_pool_block_guarded():
    env = dict(os.environ, GLP_POOL_CANARY="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "concurrent_mix.py"), "20"], capture_output=True, text=True,
                       env=env, timeout=1200, cwd=ROOT)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CONCURRENT")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-3000:]
    assert line[-1].endswith("errors=0 mismatches=0 canary=0"), line[-1]
    assert "[glprover canary]" not in r.stderr, r.stderr[-2000:]


def test_a_context_made_on_one_thread_is_driven_from_others():
    """HIP's current device is per host thread and starts at 0; every entry point makes the context's device current for the call
    and puts the caller's back (csrc/internal.h DeviceScope).  A context is created on the main thread — on the LAST device of the box,
    which on a multi-GPU machine differs from the side threads' device 0 — and a fresh thread builds a circuit, proves, builds a tree
    and frees everything on it: same bytes as from the main thread, and the calling threads' current device is what it was before."""
    import ctypes
    import importlib
    import threading

    import numpy as np
    import torch

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    trees = importlib.import_module(pkg.__name__ + ".trees")
    hip = ctypes.CDLL("libamdhip64.so")

    def current_device():
        d = ctypes.c_int(-1)
        assert hip.hipGetDevice(ctypes.byref(d)) == 0
        return d.value

    dev = torch.cuda.device_count() - 1
    before = current_device()
    ctx = pkg.Context(dev)  # on torch's stream of that device: the witness is filled by torch and by the library in turn
    assert current_device() == before  # glp_ctx_create does not leave the caller on another device
    lc = circ.MixedGateCircuit(10, 16)
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 0xFFFFFFFF00000001, (500, 4), dtype=np.uint64)
    vals = rng.integers(1, 0xFFFFFFFF00000001, (500, 4), dtype=np.uint64)

    def work(out):
        try:
            d0 = current_device()
            pc = pkg.ProverCircuit(ctx, lc.data)
            wires, pis = lc.witness_device(ctx, pkg, seed=3)
            torch.cuda.synchronize(dev)
            proof = np.array(pc.prove(wires, pis), copy=True)
            root = np.array(trees.smt_root_of_set(ctx, keys, vals), copy=True)
            pc.free()
            out.update(proof=proof, root=root, device_before=d0, device_after=current_device())
        except Exception as e:  # noqa: BLE001
            out["error"] = repr(e)

    main = {}
    work(main)
    assert "error" not in main, main
    for _ in range(2):  # two fresh threads, one after the other (a context is not driven by two threads at once)
        side = {}
        th = threading.Thread(target=work, args=(side,))
        th.start()
        th.join()
        assert "error" not in side, side
        assert side["device_before"] == side["device_after"]
        assert (side["proof"] == main["proof"]).all() and (side["root"] == main["root"]).all()
    assert current_device() == before
    ctx.close()


def test_out_of_memory_takes_the_other_contexts_caches_back():
    """A freed pool block stays cached in ITS context (up to GLP_POOL_CACHE_GB).  A host that keeps one context per proving thread
    (integration/rust/gpu.rs) can therefore have most of the HBM parked in the caches of idle contexts: when a context's hipMalloc
    says out-of-memory after its own cache is gone, the other contexts' caches on the same GPU are returned to the driver and the
    allocation is retried.  Context B caches a block of 30 % of the free memory; context A then gets 80 % of it in one piece."""
    import ctypes

    import torch

    import __graft_entry__ as g

    pkg = g.load_package()
    L = pkg.lib()
    a, b = pkg.Context(0, use_torch_stream=False), pkg.Context(0, use_torch_stream=False)
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info(0)
    pb = ctypes.c_void_p()
    assert L.glp_dev_alloc(b._h, ctypes.c_size_t(int(free0 * 0.30)), ctypes.byref(pb)) == 0
    assert L.glp_dev_free(b._h, pb) == 0  # stays in B's cache (30 % of 288 GB is inside the 96 GB budget)
    free1, _ = torch.cuda.mem_get_info(0)
    if free1 > free0 * 0.85:
        pytest.skip("the block went back to the driver (cache budget below 30 % of this GPU's memory)")
    assert b.memory()[0] == 0 and b.memory()[1] >= int(free0 * 0.30)  # nothing handed out, the block cached
    pa = ctypes.c_void_p()
    assert L.glp_dev_alloc(a._h, ctypes.c_size_t(int(free0 * 0.80)), ctypes.byref(pa)) == 0, "OOM although the memory is only cached"
    assert b.memory() == (0, 0)  # B's cache was returned to the driver for A
    assert a.memory()[0] >= int(free0 * 0.80)
    assert L.glp_dev_free(a._h, pa) == 0
    # the explicit form: glp_ctx_trim_cache
    pc = ctypes.c_void_p()
    assert L.glp_dev_alloc(b._h, ctypes.c_size_t(1 << 30), ctypes.byref(pc)) == 0 and L.glp_dev_free(b._h, pc) == 0
    assert b.memory()[1] >= 1 << 30
    before, _ = torch.cuda.mem_get_info(0)
    b.trim_cache()
    assert b.memory() == (0, 0) and torch.cuda.mem_get_info(0)[0] >= before + (1 << 30) - (64 << 20)
    # B still works after its cache was taken away
    pb2 = ctypes.c_void_p()
    assert L.glp_dev_alloc(b._h, ctypes.c_size_t(1 << 20), ctypes.byref(pb2)) == 0 and L.glp_dev_free(b._h, pb2) == 0
    a.close()
    b.close()


def test_one_circuit_is_proven_through_other_contexts_at_the_same_time():
    """INTEGRATION.md's call-site patch builds the circuit's device data once (`glp_circuit_build` on whichever thread runs
    `CircuitBuilder::build`) and proves with the calling thread's context (`gpu::ctx()`): the circuit data of context A are read by
    three other contexts on three threads at the same time — single proofs and batches — and every proof equals the one context A
    makes itself.  The circuit outlives context A's handle (it holds a reference) and is freed from yet another thread."""
    import importlib
    import threading

    import numpy as np
    import torch

    import __graft_entry__ as g

    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    a = pkg.Context(0)
    lc = circ.MixedGateCircuit(12, 16)
    pc = pkg.ProverCircuit(a, lc.data)
    ws = [lc.witness_device(a, pkg, seed=s) for s in range(3)]
    torch.cuda.synchronize()
    want = [np.array(pc.prove(w, p), copy=True) for w, p in ws]
    a.close()  # the creator's handle goes away; the circuit keeps the context alive
    others = [pkg.Context(0, use_torch_stream=False) for _ in range(3)]
    errors = []

    def work(i):
        try:
            c = others[i]
            for r in range(6):
                k = (i + r) % 3
                got = pc.prove(ws[k][0], ws[k][1], ctx=c)
                if not (got == want[k]).all():
                    errors.append("thread %d round %d: single proof differs" % (i, r))
                pair = torch.stack([ws[k][0], ws[(k + 1) % 3][0]]).contiguous()
                torch.cuda.current_stream().synchronize()  # torch built the pair on ITS stream; the context proves on its own
                both = pc.prove_batch(pair, [ws[k][1], ws[(k + 1) % 3][1]], ctx=c)
                if not ((both[0] == want[k]).all() and (both[1] == want[(k + 1) % 3]).all()):
                    errors.append("thread %d round %d: batched proofs differ" % (i, r))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    t = threading.Thread(target=pc.free)
    t.start()
    t.join()
    for c in others:
        c.close()
