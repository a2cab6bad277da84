"""Single-proof latency from the HOST at BASELINE's size: the 2^20-row 14-gate witness as (a) a device-resident tensor, (b) 135 separate
pageable numpy columns through glp_prove_cols, (c) one pageable array and (d) one pinned array through glp_prove.  The chunk plan and
the chunked sponge are read from the environment once per process (GLP_ARRIVAL_COLS, GLP_CHUNKED_SPONGE, GLP_HOST_THREADS), so a
sweep is one process per setting:  python host_cols_probe.py [log_n]"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
circ = importlib.import_module(pkg.__name__ + ".circuit")
import torch  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = pkg.Context(0)
mc = circ.MixedGateCircuit(log_n, 16)
pc = pkg.ProverCircuit(ctx, mc.data)
w, pis = mc.witness_device(ctx, pkg, seed=0)


def timed(fn, reps=4):
    fn()
    ts = []
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        r = fn()
        ts.append(1e3 * (time.perf_counter() - t0))
    return r, ts


proof, t_dev = timed(lambda: pc.prove(w, pis))
host = np.array(ctx.to_host(w), copy=True)
cols = [np.array(host[j], copy=True) for j in range(host.shape[0])]
if len(sys.argv) > 2 and sys.argv[2] == "cols":  # for the profiler: three column-witness proofs and nothing after them
    for _ in range(3):
        pc.prove_cols(cols, pis)
    ctx.synchronize()
    sys.exit(0)
p_cols, t_cols = timed(lambda: pc.prove_cols(cols, pis))
p_flat, t_flat = timed(lambda: pc.prove(host, pis))
pinned = torch.from_numpy(host.view(np.int64)).pin_memory()
p_pin, t_pin = timed(lambda: pc.prove(pinned.numpy().view(np.uint64), pis))
out = {"log_n": log_n, "settings": {k: os.environ[k] for k in ("GLP_ARRIVAL_COLS", "GLP_CHUNKED_SPONGE", "GLP_HOST_THREADS", "GLP_CHUNKED_UPLOAD") if k in os.environ},
       "host_threads": pkg.host_threads() if hasattr(pkg, "host_threads") else None,
       "device_resident_ms": [round(t, 2) for t in t_dev], "pageable_columns_ms": [round(t, 2) for t in t_cols],
       "pageable_flat_ms": [round(t, 2) for t in t_flat], "pinned_flat_ms": [round(t, 2) for t in t_pin],
       "equal": bool((p_cols == proof).all() and (p_flat == proof).all() and (p_pin == proof).all())}
out["columns_minus_device_ms"] = round(min(t_cols) - min(t_dev), 2)
print(json.dumps(out))
