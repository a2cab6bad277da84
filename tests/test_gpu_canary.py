"""GLP_POOL_CANARY=1 — the debug mode that stands in for a compute sanitizer ("canary-padded device buffers"): every
device block of a context's pool carries a 256-byte guard behind the bytes that were asked for, checked when the block is
returned.  A representative slice of the hot path (commitments of several shapes, single / batched / streamed proofs of the
14-gate circuit, FRI, the native tree hashing) must leave every guard intact; an upload planted 8 bytes past a block must be
reported.  Runs in a child process (the switch is read once per process)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes, importlib, sys
    import numpy as np
    sys.path.insert(0, %r)
    import __graft_entry__ as g
    pkg = g.load_package()
    circ = importlib.import_module(pkg.__name__ + ".circuit")
    trees = importlib.import_module(pkg.__name__ + ".trees")
    import torch
    L = pkg.lib()
    ctx = pkg.Context(0)

    def violations():
        n = ctypes.c_uint64()
        assert L.glp_ctx_canary_violations(ctx._h, ctypes.byref(n)) == 0
        return n.value

    for C, log_n in ((135, 12), (20, 9), (3, 14), (16, 5)):
        b = pkg.PolynomialBatch.from_values(ctx, ctx.to_device(pkg.splitmix_columns(C, 1 << log_n)), 3, False, 4)
        b.leaf(5)
        b.free()
    work = ctx.to_device(pkg.splitmix_columns(5, 1 << 13))
    ctx.ntt(work, 13)
    ctx.ntt(work, 13, inverse=True, shift=7)
    for bits, layers in ((8, 3), (12, 8)):
        mc = circ.MixedGateCircuit(bits, layers)
        pc = pkg.ProverCircuit(ctx, mc.data)
        ws = [mc.witness_device(ctx, pkg, seed=k) for k in range(3)]
        p0 = pc.prove(*ws[0])
        batch = torch.stack([w for w, _ in ws]).contiguous()
        pb = pc.prove_batch(batch, [p for _, p in ws])
        assert (pb[0] == p0).all()
        ps = pc.prove_stream([w.cpu().pin_memory() for w, _ in ws], [p for _, p in ws])
        assert (ps == pb).all()
        assert (pc.prove(ctx.to_host(ws[1][0]), ws[1][1]) == pb[1]).all()
        pc.free()
    rng = np.random.default_rng(1)
    keys = rng.integers(0, 2**63, (500, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (500, 4), dtype=np.uint64)
    trees.smt_root_of_set(ctx, keys, vals)
    ctx.synchronize()
    clean = violations()
    # planted overrun: 1000 bytes asked for, 1008 uploaded
    ptr = ctypes.c_void_p()
    assert L.glp_dev_alloc(ctx._h, ctypes.c_size_t(1000), ctypes.byref(ptr)) == 0
    src = (ctypes.c_uint8 * 1008)(*([7] * 1008))
    assert L.glp_dev_upload(ctx._h, ptr, src, ctypes.c_size_t(1008)) == 0
    assert L.glp_dev_free(ctx._h, ptr) == 0
    planted = violations() - clean
    print("CANARY clean=%%d planted=%%d" %% (clean, planted))
    ctx.close()
""") % ROOT


def test_hot_path_leaves_every_guard_intact_and_an_overrun_is_caught():
    env = dict(os.environ, GLP_POOL_CANARY="1")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CANARY")][-1]
    assert line == "CANARY clean=0 planted=1", line + "\n" + r.stderr[-2000:]
    assert "[glprover canary]" in r.stderr and "1000 bytes were asked for" in r.stderr


def test_without_the_switch_nothing_is_counted():
    import ctypes

    import __graft_entry__ as g

    pkg = g.load_package()
    ctx = pkg.Context(0)
    n = ctypes.c_uint64(5)
    assert pkg.lib().glp_ctx_canary_violations(ctx._h, ctypes.byref(n)) == 0 and n.value == 0
    ctx.close()
