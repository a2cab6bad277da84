#!/usr/bin/env python3
"""Cost of the quotient kernel as a function of the circuit's GATE SET (every gate's constraints are evaluated at every
LDE point, filtered by its selector): a 2^13-row hash-chain circuit with 4 gate types next to one of the same size that
also holds the six recursion gates, BaseSum and the extension-arithmetic gates (14 types, what a circuit containing
builder.verify_proof has)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import importlib  # noqa: E402

circ = importlib.import_module(pkg.__name__ + ".circuit")

P = 0xFFFFFFFF00000001
ctx = pkg.Context(0)
DB = int(sys.argv[1]) if len(sys.argv) > 1 else 13


def chain(b, n_hashes, width=512):
    """`width` independent hash chains, built level by level so that witness generation batches a level per GPU call"""
    x = [b.virtual() for _ in range(4)]
    hs = [x] * width
    for _ in range(max(1, n_hashes // width)):
        hs = [b.two_to_one(h, x) for h in hs]
    acc = hs[0]
    for h in hs[1:8]:
        acc = b.two_to_one(acc, h)
    return x, acc


def run(name, cd, inputs):
    wires, pis = cd.generate_witness(inputs, lambda ins: pkg.poseidon_gate_rows(ctx, ins))
    pc = pkg.ProverCircuit(ctx, cd)
    wd = ctx.to_device(wires)
    for _ in range(2):
        pc.prove(wd, pis)
    ctx.timing_enable(True)
    ctx.timing_reset()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        pc.prove(wd, pis)
    t = ctx.timing_read()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.timing_enable(False)
    print("| %s | %d | %d | %d | %d | %.2f | %.2f |" % (name, cd.degree_bits, len(cd.gates), cd.num_selectors, cd.num_constants,
                                                        t["plonk"]["ms"] / reps, wall))
    pc.free()


print("| circuit | degree_bits | gate types | selectors | constants columns | plonk kernels ms/proof | wall ms/proof |")
print("|---|---|---|---|---|---|---|")
n_hashes = (1 << DB) - 200
b = circ.CircuitBuilder()
x, h = chain(b, n_hashes)
b.register_public_inputs(h)
cd = b.build()
run("4 gate types", cd, {t: 3 + i for i, t in enumerate(x)})

b = circ.CircuitBuilder()
x, h = chain(b, n_hashes)
# the recursion-gate block and the base-sum / extension-arithmetic block of the test circuits, in the same builder
idx = b.virtual()
items = [b.virtual() for _ in range(16)]
picked = b.random_access(idx, items)
bits = b.split_le(idx, 4)
alpha = [b.virtual(), b.virtual()]
zero = b.constant(0)
red = b.reduce_base(items + [picked], alpha, [zero, zero])
vals = [[b.virtual(), b.virtual()] for _ in range(16)]
red2 = b.reduce_extension(vals[:5] + [red], alpha, red)
xpow = b.exp_from_bits(b.constant(circ.root_of_unity(4)), bits)
shift = b.virtual()
ev = b.interpolate_coset(shift, vals, red2)
mds = b.poseidon_mds(vals[:11] + [ev])
uv = b.mul_extension(vals[0], vals[1])
w = b.arithmetic_extension(3, 5, uv, vals[2], vals[3])
s = b.add(picked, xpow)
b.register_public_inputs(h + [s] + w + mds[0])
cd = b.build()
rng = np.random.default_rng(1)
inputs = {t: 3 + i for i, t in enumerate(x)}
inputs.update({idx: 11, shift: 12345, alpha[0]: 7, alpha[1]: 9})
for t in items + [t for p in vals for t in p]:
    inputs[t] = int(rng.integers(1, P, dtype=np.uint64))
run("14 gate types", cd, inputs)
