"""Times of the reference's block pipeline with its REAL circuits (restated: circuit.UserTransactionCircuit, SimpleSignatureCircuit,
rollup.BlockProductionCircuit, recursion.BlockBatchCircuit) on one GPU: circuit builds, host witness generation, glp_prove per proof.
Run from the repository root on a GPU box: python intmax-zkp-core_amd/tools/refblock_probe.py"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
circ, rec, rollup, wl = (importlib.import_module(pkg.__name__ + "." + m) for m in ("circuit", "recursion", "rollup", "workloads"))
ctx = pkg.Context(0)
out = wl.reference_block_pipeline(pkg, circ, rec, rollup, ctx, reps=5)
print(json.dumps(out))
