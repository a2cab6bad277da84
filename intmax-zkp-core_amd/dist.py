"""Multi-GPU sharding of independent proving units: one process per GPU, proof-level data
parallelism, NO data-path collective — block proofs / per-tx proofs / polynomial batches of different proofs are
independent (reference src/bin/block_circuit.rs:293-361,410-434; src/rollup/circuits/mod.rs:550-568).
torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU) is used only for the rendezvous barrier and to
reduce the timing (max over ranks) and unit counts (sum over ranks)."""
import os


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_units(n_units, rank, world):
    """Static round-robin partition of unit ids 0..n_units-1 (config 4: 64 proofs over 8 GPUs -> 8 each)."""
    return list(range(rank, n_units, world))


def init(backend=None, device_index=None):
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("GLP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {}
        if backend == "nccl" and device_index is not None:
            kw["device_id"] = torch.device("cuda", device_index)  # one rank = one GPU (RCCL over xGMI)
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:  # older torch without device_id
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_max_sum(elapsed_s, units, device=None):
    """(max over ranks of elapsed, sum over ranks of units)"""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), int(units)
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=dev)
    u = torch.tensor([float(units)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(u.item()))


def gather_floats(x, device=None):
    """every rank's value of one float, in rank order (per-rank step times of an N > 1 bench line: an imbalanced run is then
    diagnosable from the line itself)"""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [float(x)]
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def pin_rank_to_cores(local_rank, local_world):
    """One process per GPU on one host: give rank r the r-th slice of the CPUs this process may run on, BEFORE the library's
    host worker pool is created (csrc/host_pool.h sizes itself from the affinity mask), so that eight ranks neither start
    8 x 32 worker threads nor migrate over each other's cores.  No-op for a single rank or where the OS has no affinity call.
    Returns the list of CPUs the rank now owns (None when nothing was changed)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    mine = rank_cpu_slice(sorted(os.sched_getaffinity(0)), local_rank, local_world)
    if not mine:
        return None  # more ranks than cores: leave the mask alone (the pool falls back to hardware threads / LOCAL_WORLD_SIZE)
    os.sched_setaffinity(0, mine)
    return mine


def _thread_siblings(cpu):
    """the logical CPUs sharing a physical core with `cpu` (sysfs), or just [cpu]"""
    try:
        with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as f:
            out = []
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                out += list(range(int(lo), int(hi or lo) + 1))
            return out
    except (OSError, ValueError):
        return [cpu]


def rank_cpu_slice(cpus, local_rank, local_world, siblings=_thread_siblings):
    """The CPUs of rank `local_rank`: the allowed CPUs are grouped by PHYSICAL core (SMT siblings stay together — Linux numbers the
    second hardware thread of every core after all first ones, so slicing the plain id list would hand two ranks the two threads
    of the same cores), the cores are dealt in contiguous slices, the rank gets every allowed thread of its cores."""
    allowed = set(cpus)
    cores, seen = [], set()
    for c in sorted(allowed):
        if c in seen:
            continue
        grp = sorted(set(siblings(c)) & allowed) or [c]
        seen.update(grp)
        cores.append(grp)
    per = len(cores) // local_world
    if per < 1:
        return []
    return sorted(c for grp in cores[local_rank * per:(local_rank + 1) * per] for c in grp)


def gather_bytes(blob, device=None):
    """All-gather of variable-length byte strings (serialized child proofs for the aggregation step, SURVEY.md §8e:
    ~100-200 kB each — latency-bound, fixed-size padded uint8 buffers). Returns the list of every rank's blob."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [bytes(blob)]
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    world = dist.get_world_size()
    ln = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(ln) for _ in range(world)]
    dist.all_gather(lens, ln)
    mx = max(int(x.item()) for x in lens)
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
    if len(blob):
        buf[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [bytes(o[: int(l.item())].cpu().numpy().tobytes()) for o, l in zip(outs, lens)]


def gather_indexed(blobs, n_units, device=None):
    """One exchange step of the aggregation tree (BASELINE configs[4]): every rank contributes the serialized proofs it made
    ({unit id: bytes}); every rank receives the complete list [bytes] * n_units.  One gather_bytes of a length-prefixed
    concatenation per rank."""
    packed = b"".join(int(j).to_bytes(4, "little") + len(b).to_bytes(4, "little") + bytes(b) for j, b in sorted(blobs.items()))
    out = [None] * n_units
    for buf in gather_bytes(packed, device=device):
        pos = 0
        while pos < len(buf):
            if len(buf) - pos < 8:
                raise ValueError("truncated record header (%d trailing bytes)" % (len(buf) - pos))
            j = int.from_bytes(buf[pos:pos + 4], "little")
            ln = int.from_bytes(buf[pos + 4:pos + 8], "little")
            if pos + 8 + ln > len(buf):  # a short slice would hand a truncated child proof to the aggregation step
                raise ValueError("unit %d declares %d bytes, %d received" % (j, ln, len(buf) - pos - 8))
            if j >= n_units or out[j] is not None:
                raise ValueError("unit %d delivered twice or out of range" % j)
            out[j] = buf[pos + 8:pos + 8 + ln]
            pos += 8 + ln
    missing = [j for j, b in enumerate(out) if b is None]
    if missing:
        raise ValueError("units %s were not delivered" % missing[:8])
    return out


class WorkQueue:
    """Dynamic distribution of independent proving units (BASELINE config 4, "work-queue variant"): every rank pulls the
    next unit id from one shared atomic counter, so ranks whose proofs are cheaper simply prove more of them.  The counter
    lives in a TCPStore next to the rendezvous (host side, a few bytes per unit) — still no data-path collective.
    Single process: a plain local counter."""

    _stores = {}  # (host, port) -> TCPStore: one store per process serves every queue; queues are told apart by their key
    _seq = {}     # name -> queues of that name constructed so far in this process

    def __init__(self, n_units, name="units", port_offset=1):
        import torch.distributed as dist

        # The counter lives as long as the store, so every queue needs a key of its own: the name plus the number of queues
        # of that name constructed before it.  Queues are constructed collectively (every rank makes the same queues in the
        # same order, like a process group), so all ranks derive the same key, and a second queue of the same name starts
        # from zero instead of from the first one's final count.
        seq = WorkQueue._seq.get(name, 0)
        WorkQueue._seq[name] = seq + 1
        self.n_units, self.key, self._local, self.store = int(n_units), "glp_queue_%s#%d" % (name, seq), 0, None
        rank, _, world = env_world()
        if world > 1:
            # the counter lives on the rendezvous store the process group was created from (keys prefixed): no second port to
            # reserve.  Only without a process group (a queue used on its own) a TCPStore is opened at MASTER_PORT + port_offset.
            base = None
            if dist.is_available() and dist.is_initialized():
                try:
                    base = dist.distributed_c10d._get_default_store()
                except Exception:  # noqa: BLE001 — private accessor: fall back to a store of our own
                    base = None
            if base is not None:
                self.store = dist.PrefixStore("glp_queue", base)
            else:
                from datetime import timedelta

                host = os.environ.get("MASTER_ADDR", "127.0.0.1")
                port = int(os.environ.get("MASTER_PORT", "29531")) + port_offset
                if (host, port) not in WorkQueue._stores:
                    WorkQueue._stores[(host, port)] = dist.TCPStore(host, port, world, rank == 0, timeout=timedelta(seconds=120))
                self.store = WorkQueue._stores[(host, port)]

    def next(self):
        """-> the next unit id, or None when the queue is drained"""
        if self.store is None:
            i, self._local = self._local, self._local + 1
        else:
            i = self.store.add(self.key, 1) - 1
        return i if i < self.n_units else None

    def __iter__(self):
        while True:
            i = self.next()
            if i is None:
                return
            yield i


# ---- tables prepared once per host: built by rank 0, cached under a content key, loaded by every rank ----
def cache_dir():
    """A directory private to this user (mode 0700, ownership checked): GLP_BENCH_CACHE_DIR, else /dev/shm/glp_cache_<uid> (memory-backed:
    the 2 GB of a prepared 2^20-row circuit load in about a second), else the system's temporary directory."""
    import tempfile

    d = os.environ.get("GLP_BENCH_CACHE_DIR")
    if not d:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        d = os.path.join(base, "glp_cache_%d" % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("cache directory %s is not private to this user (owner %d, mode %o)" % (d, st.st_uid, st.st_mode & 0o777))
    return d


def content_key(paths, *extra):
    """sha256 over the bytes of `paths` (the sources a prepared table depends on) and the extra parameters, 24 hex digits"""
    import hashlib

    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for x in extra:
        h.update(repr(x).encode() + b"\0")
    return h.hexdigest()[:24]


def prepare_shared_arrays(key, build, rank, world, timeout_s=3600.0, poll_s=0.2, log=None):
    """Numeric tables that are expensive to make on the host (bench.py: the reference's block circuit with 128 slots, ~95 s of Python)
    exist ONCE per host: rank 0 builds them with `build()` — a dict of numpy arrays, strings as 0-d unicode arrays, nothing that needs
    pickling — and leaves them in the private cache directory under `key` (written to a temporary name, renamed when complete); a later
    run with the same key, and every other rank of this run, loads the file.  Ranks != 0 WAIT ON THE FILE (or on rank 0's failure
    marker), not inside a collective: an RCCL barrier held for a minute and a half while one rank runs Python is where watchdogs bite.
    Every rank takes the same branch: rank 0's exception travels through the marker and is raised on all of them.
    Returns (a mapping name -> array: the NpzFile, or rank 0's own dict right after a build, "built" | "cache" | "from rank 0")."""
    import tempfile
    import time

    import numpy as np

    try:
        d = cache_dir()
    except (OSError, RuntimeError) as e:
        if world > 1:
            raise
        (log or (lambda *_: None))("no cache directory (%r): built in memory, not kept" % (e,))
        return build(), "built (no cache directory)"
    path = os.path.join(d, key + ".npz")
    run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "_" + os.environ.get("MASTER_PORT", "0")
    failed = os.path.join(d, "%s.failed.%s" % (key, run))
    say = log or (lambda *_: None)
    if rank == 0 and os.path.exists(failed):
        os.remove(failed)  # (a marker of an earlier run under the same rendezvous: gone before anybody looks)
    if world > 1:
        barrier()  # short: every rank has just arrived
    source = "cache"
    if rank == 0:
        if not os.path.exists(path):
            try:
                arrays = build()
            except BaseException as e:
                with open(failed, "w") as f:
                    f.write(repr(e)[:500])
                raise
            tmp = None
            try:  # leaving the tables for later runs / the other ranks: a full or read-only cache directory must not cost a one-rank run its circuit
                fd, tmp = tempfile.mkstemp(dir=d, suffix=".tmp")
                with os.fdopen(fd, "wb") as f:
                    np.savez(f, **arrays)
                os.replace(tmp, path)
            except OSError as e:
                if tmp and os.path.exists(tmp):
                    os.remove(tmp)
                if world > 1:  # the other ranks have nothing to load
                    with open(failed, "w") as f:
                        f.write("the prepared tables could not be written to %s: %r" % (d, e))
                    raise
                say("prepared tables not cached (%r): this run keeps them in memory" % (e,))
                return arrays, "built (not cached: %s)" % type(e).__name__
            return arrays, "built"  # (rank 0 keeps what it has just made: no second copy read back)
        else:
            say("prepared tables found under %s" % path)
    else:
        source = "from rank 0"
        t0 = time.monotonic()
        while not os.path.exists(path):
            if os.path.exists(failed):
                with open(failed) as f:
                    raise RuntimeError("rank 0 could not prepare the shared tables: " + f.read())
            if time.monotonic() - t0 > timeout_s:
                raise RuntimeError("no prepared tables from rank 0 after %.0f s" % timeout_s)
            time.sleep(poll_s)
    return np.load(path, allow_pickle=False), source
