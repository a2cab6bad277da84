"""world_size-2 CPU (gloo) coverage of the multi-GPU path: unit sharding with no data-path collective, the
max-over-ranks timing / sum-of-units reduction bench.py relies on, and the proof-blob gather of the aggregation step."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g

    pkg = g.load_package()
    import importlib

    dist_mod = importlib.import_module("intmax_zkp_core_amd.dist")
    r, lr, w = dist_mod.init(backend="gloo")
    units = dist_mod.shard_units(9, r, w)
    dist_mod.barrier()
    elapsed = 1.0 + r  # rank 1 is the slow one
    t, total = dist_mod.reduce_max_sum(elapsed, len(units))
    blobs = dist_mod.gather_bytes(bytes([r + 1]) * (3 + 5 * r))
    # aggregation exchange: 5 units sharded round-robin, every rank ends up with all of them in unit order
    mine_units = {j: bytes([j]) * (j + 1) for j in dist_mod.shard_units(5, r, world)}
    level = dist_mod.gather_indexed(mine_units, 5)
    assert level == [bytes([j]) * (j + 1) for j in range(5)], level
    # work-queue variant: rank 1 is slow, so rank 0 ends up with most of the 12 units; every unit is taken exactly once
    import time

    wq = dist_mod.WorkQueue(12)
    mine = []
    for u in wq:
        mine.append(u)
        time.sleep(0.05 if r == 1 else 0.002)
    dist_mod.barrier()
    # a second queue under another name shares the process's TCPStore (no second server on the same port)
    wq2 = dist_mod.WorkQueue(4, name="second")
    mine2 = list(wq2)
    dist_mod.barrier()
    assert all(u < 4 for u in mine2)
    t2, n2 = dist_mod.reduce_max_sum(0.0, len(mine2))
    assert n2 == 4, n2
    # a second queue under the SAME name is a fresh queue (its own counter), not the tail of the first one
    wq3 = dist_mod.WorkQueue(6)
    mine3 = list(wq3)
    dist_mod.barrier()
    _, n3 = dist_mod.reduce_max_sum(0.0, len(mine3))
    assert n3 == 6 and all(u < 6 for u in mine3), (n3, mine3)
    q.put((r, units, t, total, [len(b) for b in blobs], [b[0] for b in blobs], mine))
    import torch.distributed as dist

    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, u0, t0, tot0, l0, f0, q0), (r1, u1, t1, tot1, l1, f1, q1) = res
    assert sorted(q0 + q1) == list(range(12)) and len(q0) > len(q1)  # dynamic split, nothing lost or duplicated
    assert u0 == [0, 2, 4, 6, 8] and u1 == [1, 3, 5, 7]
    assert sorted(u0 + u1) == list(range(9))  # every unit exactly once, no overlap
    assert t0 == t1 == 2.0  # max over ranks
    assert tot0 == tot1 == 9  # whole-job unit count
    assert l0 == l1 == [3, 8] and f0 == f1 == [1, 2]


def test_single_process_fallbacks():
    import importlib
    import __graft_entry__ as g

    g.load_package()
    d = importlib.import_module("intmax_zkp_core_amd.dist")
    assert d.shard_units(5, 0, 1) == [0, 1, 2, 3, 4]
    assert d.reduce_max_sum(1.5, 7) == (1.5, 7)
    assert d.gather_bytes(b"abc") == [b"abc"]
    assert d.gather_indexed({1: b"b", 0: b"aa"}, 2) == [b"aa", b"b"]
    with pytest.raises(ValueError):
        d.gather_indexed({0: b"x"}, 2)
    assert list(d.WorkQueue(3)) == [0, 1, 2]


def test_gather_indexed_rejects_truncated_records(monkeypatch):
    """a record whose declared length runs past the received buffer (or a torn header) raises instead of handing a short
    child proof to the aggregation step"""
    import importlib
    import __graft_entry__ as g

    g.load_package()
    d = importlib.import_module("intmax_zkp_core_amd.dist")
    good = (0).to_bytes(4, "little") + (5).to_bytes(4, "little") + b"abcde"
    for bad in (good[:-2], good + b"\x01\x00\x00"):
        monkeypatch.setattr(d, "gather_bytes", lambda blob, device=None, bad=bad: [bad])
        with pytest.raises(ValueError):
            d.gather_indexed({}, 1)
    monkeypatch.setattr(d, "gather_bytes", lambda blob, device=None: [good])
    assert d.gather_indexed({}, 1) == [b"abcde"]


def test_bench_starts_its_own_ranks_and_relays_one_line():
    """`python bench.py --gpus 2` run plainly (no torchrun around it) starts two ranks as a child process and relays rank 0's
    JSON line: n_gpus = 2, both ranks seen by the all-reduce, the aggregation exchange delivered every unit (gloo on CPU)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GLP_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "dist-selftest", "--steps", "2"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # exactly ONE line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dist_backend"] == "gloo" and out["exchange_ok"] is True
    # what the benchmark ranks do before they create a context: each rank pinned to its own half of the CPUs, the library's host
    # worker pool sized to that half, per-rank values gathered in rank order
    assert out["per_rank_probe"] == [1.0, 2.0]
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 2:
        assert out["pinned_cpus_per_rank"][0] == out["pinned_cpus_per_rank"][1] > 0
        assert sum(out["pinned_cpus_per_rank"]) <= ncpu
        assert out["host_threads_per_rank_all"] == [min(32.0, c) for c in out["pinned_cpus_per_rank"]]


def test_bench_refuses_more_ranks_than_gpus():
    """on a box with fewer GPUs than --gpus the bench exits non-zero with a message and prints NO JSON line (it must never
    label a 1-GPU number as an N-GPU one)"""
    import subprocess
    import sys

    import torch

    want = torch.cuda.device_count() + 1
    if want < 2:
        want = 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(want), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "GPU" in r.stderr and "refusing" in r.stderr
    # under an external launcher with the wrong rank count it refuses as well
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999", GLP_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "dist-selftest"],
                       capture_output=True, text=True, env=env2, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "--gpus 2" in r.stderr


def test_rank_cpu_slices_keep_smt_siblings_together():
    """dist.rank_cpu_slice: ranks get disjoint slices of PHYSICAL cores; on the usual Linux numbering (threads 0..63 = first hardware
    thread of every core, 64..127 = their siblings) a plain slice of the id list would give ranks 0 and 4 the same cores"""
    import importlib

    import __graft_entry__ as g

    d = importlib.import_module(g.load_package().__name__ + ".dist")
    sib = lambda c: [c % 64, c % 64 + 64]  # noqa: E731 — a 64-core, 128-thread socket
    slices = [d.rank_cpu_slice(list(range(128)), r, 8, siblings=sib) for r in range(8)]
    assert slices[0] == list(range(0, 8)) + list(range(64, 72))
    assert sorted(c for s in slices for c in s) == list(range(128))
    assert all(len(s) == 16 for s in slices)
    for s in slices:  # every CPU's sibling is in the same slice
        assert all((c + 64) % 128 in s for c in s)
    # no SMT: plain contiguous slices; a restricted mask is respected; more ranks than cores -> nothing to pin
    assert d.rank_cpu_slice(list(range(8)), 1, 4, siblings=lambda c: [c]) == [2, 3]
    assert d.rank_cpu_slice([4, 5, 6, 7, 68, 69, 70, 71], 1, 2, siblings=sib) == [6, 7, 70, 71]
    assert d.rank_cpu_slice([0, 1], 0, 4, siblings=lambda c: [c]) == []
    assert d.rank_cpu_slice(sorted(__import__("os").sched_getaffinity(0)), 0, 1)  # the real sysfs reader works here


def _prep_worker(rank, world, port, q, cache, key, mode):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      GLP_BENCH_CACHE_DIR=cache)
    import sys
    import time

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g

    g.load_package()
    import importlib

    import numpy as np

    d = importlib.import_module("intmax_zkp_core_amd.dist")
    r, _, w = d.init(backend="gloo")
    calls = []

    def build():
        calls.append(1)
        time.sleep(1.0)  # rank 0 is busy for a while: the others must wait on the file, not time out in a collective
        if mode == "fail":
            raise ValueError("cannot build this")
        return dict(meta=np.array('{"degree_bits": 5}'), wires=np.arange(12, dtype=np.uint64).reshape(3, 4), pis=np.array([7, 8], dtype=np.uint64))

    t0 = time.monotonic()
    try:
        z, source = d.prepare_shared_arrays(key, build, r, w, timeout_s=60)
        out = ("ok", source, len(calls), str(z["meta"]), z["wires"].tolist(), z["pis"].tolist(), time.monotonic() - t0)
    except Exception as e:  # noqa: BLE001
        out = ("error", type(e).__name__, len(calls), str(e), None, None, time.monotonic() - t0)
    # every rank took the same branch: the agreement bench.py makes before it goes on
    _, n_ok = d.reduce_max_sum(0.0, 1 if out[0] == "ok" else 0)
    q.put((r,) + out + (n_ok,))
    import torch.distributed as dist

    dist.destroy_process_group()


def _run_prep(cache, key, mode):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prep_worker, args=(r, 2, port, q, cache, key, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_tables_prepared_once_per_host_and_shared_by_every_rank(tmp_path):
    """dist.prepare_shared_arrays (bench.py's block circuit): the first run builds on rank 0 only and rank 1 loads rank 0's file;
    a second run with the same key builds nothing and starts at once (the cached path); a failure on rank 0 reaches rank 1 as an error
    too — both ranks take the same branch; nothing in the cache needs pickling"""
    import stat

    cache = str(tmp_path / "cache")
    first = _run_prep(cache, "k1", "ok")
    assert [r[1] for r in first] == ["ok", "ok"] and [r[2] for r in first] == ["built", "from rank 0"]
    assert [r[3] for r in first] == [1, 0]  # build() ran on rank 0 only
    for r in first:
        assert r[4] == '{"degree_bits": 5}' and r[5] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]] and r[6] == [7, 8] and r[8] == 2
    assert stat.S_IMODE(os.stat(cache).st_mode) == 0o700 and sorted(os.listdir(cache)) == ["k1.npz"]
    second = _run_prep(cache, "k1", "ok")
    assert [r[2] for r in second] == ["cache", "from rank 0"] and [r[3] for r in second] == [0, 0]
    assert all(r[7] < 0.9 for r in second), second  # nobody waited for a build
    failed = _run_prep(cache, "k2", "fail")
    assert [r[1] for r in failed] == ["error", "error"] and failed[0][2] == "ValueError" and failed[1][2] == "RuntimeError"
    assert "cannot build this" in failed[1][4] and [r[8] for r in failed] == [0, 0]
    # a later run under the same key is not poisoned by the old failure marker of another rendezvous
    again = _run_prep(cache, "k2", "ok")
    assert [r[1] for r in again] == ["ok", "ok"]


def test_cache_directory_must_be_private(tmp_path, monkeypatch):
    import importlib

    import __graft_entry__ as g

    g.load_package()
    d = importlib.import_module("intmax_zkp_core_amd.dist")
    shared = tmp_path / "shared"
    shared.mkdir(mode=0o777)
    os.chmod(shared, 0o777)
    monkeypatch.setenv("GLP_BENCH_CACHE_DIR", str(shared))
    with pytest.raises(RuntimeError):
        d.cache_dir()
    k1 = d.content_key([__file__], "log_n_txs", 7)
    assert k1 == d.content_key([__file__], "log_n_txs", 7) != d.content_key([__file__], "log_n_txs", 4) and len(k1) == 24
