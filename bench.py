#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on MI355X: whole proofs of a 2^20-row circuit.

Default workload (BASELINE.json: "block proofs/sec + single-proof latency, 2^20-row rollup circuit"):
one STEP = one complete `CircuitData::prove` of a 2^20-row circuit under `standard_recursion_config`
(135 wires, 80 routed, 2 challenges, quotient degree factor 8, FRI rate 3 / cap 4 / 16 PoW bits / 28 queries):
    wires commitment (135 cols) -> betas/gammas -> Z + partial products (20 cols) commitment -> alphas ->
    quotient evaluation on the 8n coset + coset iNTT -> quotient commitment (16 cols) -> zeta -> openings (257 polys) ->
    FRI (combine, 4 fold/commit rounds, proof-of-work grind, 28 query rounds)
with the Fiat-Shamir transcript on the host, exactly the step order of plonky2's prover.

THE CIRCUIT (default since the end of round 5, `--circuit block`): the reference's OWN block production circuit —
`make_block_proof_circuit` (reference src/rollup/circuits/mod.rs:389-624) restated on this repository's host builder
(intmax_zkp_core_amd.rollup.BlockProductionCircuit) with 128 transaction slots, each verifying one user-transaction proof and one
signature proof through the restated recursive verifier: 736 359 rows used = a 2^20-row circuit with all 14 gate types — filled
with the block of src/bin/block_circuit.rs (its trees and process proofs from the library's tree calls, its inner proofs by
glp_prove).  Preparing it is host Python above the boundary (~1.5 minutes) and happens ONCE PER HOST: rank 0 builds it and leaves tables
and witness in a private cache directory under a content key of the sources they are made from (dist.prepare_shared_arrays); the other
ranks of an N > 1 run — waiting on the file, not inside a collective — and every later run on the host load them in about a second
(`block_circuit_preparation` says which); if the preparation fails every rank falls back to the synthetic trace together and says so
(`circuit_fallback`).
`--circuit mixed` (the default through round 4, and the default for every `--log-n` other than 20): a synthetic trace of the same
shape — ~65 % PoseidonGate rows in 16 layers tied by copy constraints, the other 13 gates' rows with valid random
witnesses, a public-input hash; timed beside the block circuit in the default line (`synthetic_14_gate_trace_same_size`).
`--circuit poseidon`: the 4-gate PoseidonGate-only trace of round 1.  Witness generation and circuit building are outside the timed
region (host side of the boundary in the reference); the witness is resident in HBM when timing starts.  After timing, the proof is
checked with the restated verifier (oracle) and the cpu_baseline leg proves the same witness on the host — outside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload proof|batch64|aggtree] [--circuit mixed|poseidon|block]
`--circuit block`: the timed circuit is the reference's OWN block production circuit (make_block_proof_circuit restated on this
repository's builder, 128 transaction slots = 736 359 rows = a 2^20-row circuit) filled with the block of src/bin/block_circuit.rs —
~2.5 minutes of host preparation, then the same timed region and the same cpu_baseline leg (profiles/r05_block128.md).  The default run
proves it too, as the extra `reference_block_circuit_128_slots`.
`--workload batch64`: BASELINE config 4 — 64 independent 2^16-row block proofs per step, sharded over the ranks, no collective.
`--workload aggtree`: BASELINE config 5 — the aggregation tree over 64 block proofs (arity 8) per step; serialized child proofs
are gathered between levels with RCCL (all_gather of padded uint8 buffers); time in the exchange is reported.
N > 1: one process per GPU, independent proofs per rank, barrier + max-over-ranks timing, ONE JSON line from
rank 0.  Either launch it as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's way), or
run `python bench.py --gpus N` plainly: with no WORLD_SIZE in the environment it starts those N ranks itself as a child process
and relays the line; it exits non-zero, without a line, when the machine has fewer than N GPUs.  Every N > 1 line carries
`ranks_seen` (an all-reduce of ones over the RCCL group) and `dist_backend`.
"""
import argparse
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable by a streaming copy


def leaf_hash_algorithmic_bytes(n_cols, lde_bits):
    """compulsory traffic of ONE leaf-hash launch: read the N x C LDE matrix once, write N 32-byte digests"""
    N = 1 << lde_bits
    return 8 * N * n_cols + 32 * N


def quotient_algorithmic_bytes(cd):
    """every column of the three LDE matrices once, two output columns: 64 n (cs + wires + zs) + 2 * 64 n"""
    cols = cd.oracle_cols()
    return 64 * cd.n * (cols[0] + cols[1] + cols[2]) + 2 * 64 * cd.n


def perms_per_proof(cd):
    """Poseidon permutations in the three per-proof commitments (leaf sponges + tree levels)"""
    N = 1 << (cd.degree_bits + cd.cfg.rate_bits)
    cols = cd.oracle_cols()[1:]
    return sum(N * (-(-c // 8)) + N - (1 << cd.cfg.cap_height) for c in cols)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cpu_counts():
    """(logical CPUs this process may run on, physical cores among them, CPUs its cgroup's quota allows — None without one)"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                cores.add(f.read().strip())
        except OSError:
            cores.add(str(c))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return len(cpus), max(1, len(cores)), quota


def cpu_baseline(cd, kind, log_n, wires_host, pis, check=None, reps=2):
    """The only leg of bench.py that touches oracle/ (as the checker and as the CPU baseline):
    (1) `check` = (constants_sigmas cap of the GPU build, GPU proof): the restated verifier must accept the GPU proof made in
        the timed region;
    (2) the oracle's TIMED LEG (oracle/fast*.{c,h}: the same restatement with packed-Goldilocks AVX-512 / AVX2 arithmetic, VL
        leaves per Poseidon permutation, cache-blocked NTTs, a base-field vector quotient evaluator; kind "port" — NOT plonky2,
        which cannot be built here) proves THE SAME circuit with THE SAME witness at THE SAME size as the GPU's timed region, on
        this host's cores: `value` is MEASURED on the complete proof (no extrapolation), with one thread per physical core and
        with every hardware thread, the faster of the two; its proof must equal the GPU's word for word (a full-size parity
        check inside the bench) — and the scalar checker is what tests/test_oracle_fast.py holds the timed leg to."""
    import numpy as np

    from oracle import oracle as orc

    verified = None
    if check is not None:
        cap_gpu, proof_gpu = check
        orc.set_fast(False)
        verified = orc.plonk_verify(cd, cap_gpu, pis, proof_gpu) == 0
    logical, physical, quota = host_cpu_counts()
    # The CPUs the host really grants: the GPU boxes of this pool show 256 logical CPUs (2 x EPYC 9575F) but their cgroup allows 16
    # CPUs of run time (cpu.max "1600000 100000"); threads beyond the quota are throttled, not run (profiles/r05_cpu_baseline.md).
    usable = logical if quota is None else max(1, min(logical, int(quota + 0.999)))
    usable_cores = min(physical, usable)
    # per-core permutation rates: the timed leg and the scalar checker, one thread
    rng = np.random.default_rng(0)
    st = rng.integers(0, 2**63, (1 << 16, 12), dtype=np.uint64)
    isa = orc.set_fast(True)
    orc.permute_many(st[:4096])
    t0 = time.perf_counter()
    orc.permute_many(st)
    ns_fast = (time.perf_counter() - t0) / len(st) * 1e9
    t0 = time.perf_counter()
    orc.permute_many(st[:8192], fast=False)
    ns_scalar = (time.perf_counter() - t0) / 8192 * 1e9

    orc.set_threads(usable)
    t0 = time.perf_counter()
    cs = orc.batch_from_values(cd.constants_sigmas_values(), cd.cfg.rate_bits, cd.cfg.cap_height)  # CircuitBuilder::build, untimed
    t_build = time.perf_counter() - t0
    cd.circuit_digest = orc.circuit_digest(cs["cap"])
    build_equal = None if check is None else bool((np.asarray(cs["cap"]).reshape(-1) == np.asarray(check[0]).reshape(-1)).all())
    runs, best, proof_cpu = [], None, None
    for threads in ([usable_cores, usable] if usable > usable_cores else [usable]):
        orc.set_threads(threads)
        ts, phases = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            proof_cpu = orc.plonk_prove(cd, cs, wires_host, pis)
            ts.append(time.perf_counter() - t0)
            if phases is None or ts[-1] <= min(ts):
                phases = orc.last_phase_seconds()
            if ts[-1] > 90:  # a small host: one repetition is the sample
                break
        run = {"threads": threads, "reps_s": [round(t, 3) for t in ts], "best_s": min(ts), "phases_s_of_best": {k: round(v, 3) for k, v in phases.items()}}
        runs.append(run)
        if best is None or run["best_s"] < best["best_s"]:
            best = run
        if min(ts) > 90:
            break
    equal = None if check is None else bool(proof_cpu.size == check[1].size and (np.asarray(proof_cpu) == np.asarray(check[1])).all())
    orc.set_fast(False)
    return {
        "gpu_proof_accepted_by_restated_verifier": verified,
        "value": 1.0 / best["best_s"],
        "value_basis": "measured: the complete 2^%d-row proof of the same circuit and witness, best of %d, T = %d threads" % (
            log_n, len(best["reps_s"]), best["threads"]),
        "unit": "proofs/s",
        "cores": usable_cores,
        "threads": best["threads"],
        "logical_cpus": logical,
        "physical_cores": physical,
        "cgroup_cpu_quota": quota,
        "cores_note": "cores = CPUs this process can actually use: min(affinity mask, cgroup CPU quota); the host shows %d logical CPUs "
                      "on %d cores, its cgroup grants %s" % (logical, physical, "no quota" if quota is None else "%.0f CPUs of run time" % quota),
        "kind": "port",
        "isa": isa,
        "cpu_model": cpu_model(),
        "seconds_per_proof": best["best_s"],
        "ns_per_permutation_per_core": ns_fast,
        "ns_per_permutation_per_core_scalar_checker": ns_scalar,
        "runs": runs,
        "constants_sigmas_commitment_s": t_build,
        "constants_sigmas_cap_equals_gpu": build_equal,
        "proof_equals_gpu_proof_word_for_word": equal,
        "sample": "the oracle's timed leg (oracle/fast_impl.h: packed-Goldilocks %s, the restatement's algorithms; NOT the reference's "
                  "Rust prover, which cannot be built here) proving the SAME 2^%d-row %s circuit with the SAME witness as the GPU's timed "
                  "region, complete proofs: %s; %.0f ns per Poseidon permutation per core (scalar checker: %.0f ns)" % (
                      isa, log_n, kind, "; ".join("T = %d: %s s" % (r["threads"], "/".join("%.2f" % t for t in r["reps_s"])) for r in runs),
                      ns_fast, ns_scalar),
    }


def small_circuit_batches(pkg, circ, ctx, sizes=((4, 256), (12, 64), (16, 16))):
    """the reference's real circuit sizes (2^4 .. 2^16 rows): proofs/s one at a time vs glp_prove_batch_dev"""
    import torch

    out = {}
    for bits, K in sizes:
        lc = circ.LayeredPoseidonCircuit(bits, min(16, max(1, (1 << bits) // 8)))
        pc = pkg.ProverCircuit(ctx, lc.data)
        ws = [lc.witness_device(ctx, pkg, seed=k) for k in range(K)]  # K DISTINCT witnesses: K different proof-of-work grinds
        for _ in range(2):
            pc.prove(*ws[0])
        ctx.synchronize()
        reps = max(4, 64 >> max(0, bits - 10))
        t0 = time.perf_counter()
        for i in range(reps):
            pc.prove(*ws[i % K])
        ctx.synchronize()
        single = reps / (time.perf_counter() - t0)
        batch = torch.stack([w[0] for w in ws]).contiguous()
        pis = [w[1] for w in ws]
        proofs = pc.prove_batch(batch, pis)
        ctx.synchronize()
        nrep = 3
        t0 = time.perf_counter()
        for _ in range(nrep):
            pc.prove_batch(batch, pis, out=proofs)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / nrep
        out["2^%d rows" % bits] = {"single_proofs_per_s": single, "batch_size": K, "batched_proofs_per_s": K / dt,
                                   "speedup": K / dt / single}
        del batch
        pc.free()
    return out


class _SharedBlockCircuit:
    """the prepared block circuit as bench.py needs it: the tables `ProverCircuit` takes and the witness — loaded from the host's cache
    (dist.prepare_shared_arrays), by every rank, whoever built it"""

    def __init__(self, circ, z):
        meta = json.loads(str(z["meta"]))
        self.data = circ.CircuitData(circ.CircuitConfig, meta["degree_bits"], meta["gates"], meta["num_selectors"], z["constants"], z["sigmas"],
                                     [int(x) for x in z["k_is"]], None)
        self._wires, self._pis = z["wires"], [int(x) for x in z["pis"]]
        self.n_slots, self.rows_used, self.poseidon_rows = meta["n_slots"], meta["rows_used"], meta["poseidon_rows"]
        self.build_s, self.witness_s = meta["build_s"], meta["witness_s"]

    def witness_host(self):
        return self._wires, self._pis

    def witness_device(self, ctx, pkg, seed=0):
        return ctx.to_device(self._wires), self._pis


BLOCK_CIRCUIT_SOURCES = ("circuit.py", "recursion.py", "rollup.py", "workloads.py", "state.py", "trees.py", "codec.py", "__init__.py")


def prepare_block_circuit(pkg, circ, ctx, dist_mod, rank, world, log_n_txs):
    """The reference's block circuit and its witness (workloads.ReferenceBlockCircuit), prepared ONCE PER HOST: rank 0 builds it (~95 s of
    Python at 128 slots) unless the host's cache already holds it under the content key of the sources it is made from; every rank —
    and every later run on this host: the driver's N = 1, 2, 4, 8 runs back to back — loads the cached tables in about a second.  Ranks
    other than 0 wait on the file, not in a collective.  Returns (circuit, how it was obtained)."""
    import numpy as np

    key = "block_circuit_" + dist_mod.content_key([os.path.join(ROOT, "intmax-zkp-core_amd", n) for n in BLOCK_CIRCUIT_SOURCES],
                                                 "log_n_txs", log_n_txs, "abi", int(pkg.lib().glp_abi_version()))

    def build():
        if os.environ.get("GLP_BENCH_FAIL_BLOCK") == "1":  # test hook: the fallback path (tests/test_gpu_workloads.py)
            raise RuntimeError("GLP_BENCH_FAIL_BLOCK")
        mods = [importlib.import_module("intmax_zkp_core_amd." + n) for n in ("recursion", "rollup", "workloads")]
        lc = mods[2].ReferenceBlockCircuit(pkg, circ, mods[0], mods[1], ctx, log_n_txs, log=lambda m: print("[bench] " + m, file=sys.stderr))
        wires, pis = lc.witness_host()
        meta = dict(degree_bits=lc.data.degree_bits, gates=lc.data.gates, num_selectors=lc.data.num_selectors, n_slots=lc.n_slots,
                    rows_used=lc.rows_used, poseidon_rows=sum(1 for g_, _ in lc.data.builder.rows if g_ == circ.GATE_POSEIDON),
                    build_s=lc.build_s, witness_s=lc.witness_s)
        return dict(meta=np.array(json.dumps(meta)), constants=lc.data.constants, sigmas=lc.data.sigmas,
                    k_is=np.array(lc.data.k_is, dtype=np.uint64), wires=wires, pis=np.array(pis, dtype=np.uint64))

    t0 = time.perf_counter()
    lc, err = None, None
    try:
        z, source = dist_mod.prepare_shared_arrays(key, build, rank, world, log=lambda m: print("[bench] " + m, file=sys.stderr))
        lc = _SharedBlockCircuit(circ, z)
        lc.preparation = {"source": source, "seconds": round(time.perf_counter() - t0, 2), "cache_key": key}
    except Exception as e:  # noqa: BLE001
        err = e
    if world > 1:  # every rank takes the same branch: one rank that could not load makes all of them fall back together
        _, n_ok = dist_mod.reduce_max_sum(0.0, 0 if err is not None else 1, device="cuda:%d" % ctx.device)
        if n_ok != world and err is None:
            err = RuntimeError("%d of %d ranks could not load the prepared block circuit" % (world - n_ok, world))
    if err is not None:
        raise err
    return lc


def reference_gadget_chain(pkg, circ, ctx, n_entries=1 << 16, k=128):
    """Two of the reference's OWN circuits through the whole chain on the GPU: k sequential SparseMerkleTree::set calls on
    a tree of n_entries as ONE glp_smt_apply_updates call -> k SparseMerkleProcessProofs -> k witnesses of the reference's processor
    gadget (circuit.SmtProcessCircuit = verify_smt_process_proof, src/sparse_merkle_tree/gadgets/process/process_smt.rs:117-341; host
    witness generation, PoseidonGate rows from glp_poseidon_gate_witness_dev) -> ONE glp_prove_batch_dev call; and the same for k
    (non-)inclusion proofs out of the device-side tree with the reference's verifier gadget (circuit.SmtInclusionCircuit).  Timed:
    the tree call and the proving call; witness generation is host code above the boundary and is not."""
    import importlib

    import numpy as np
    import torch

    trees = importlib.import_module(pkg.__name__ + ".trees")
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 2**63, (n_entries, 4), dtype=np.uint64)
    vals = rng.integers(1, 2**63, (n_entries, 4), dtype=np.uint64)
    uk = np.concatenate([keys[rng.integers(0, n_entries, k // 2)], rng.integers(0, 2**63, (k - k // 2, 4), dtype=np.uint64)])
    rng.shuffle(uk)
    uv = rng.integers(1, 2**63, (k, 4), dtype=np.uint64)
    uv[rng.random(k) < 0.25] = 0
    rows_fn = lambda ins: pkg.poseidon_gate_rows(ctx, ins)  # noqa: E731
    out = {}
    trees.smt_apply_updates(ctx, keys, vals, uk, uv)
    t0 = time.perf_counter()
    proofs, roots = trees.smt_apply_updates(ctx, keys, vals, uk, uv)
    t_tree = time.perf_counter() - t0
    t0 = time.perf_counter()
    found = trees.smt_find_batch(ctx, keys, vals, uk)
    t_find = time.perf_counter() - t0
    for name, c, items in (
            ("smt_process_proof_gadget", circ.SmtProcessCircuit(32, public_roots=True), proofs),
            ("smt_inclusion_proof_gadget", circ.SmtInclusionCircuit(32, public_root=True),
             [dict(root=found.root, key=found.keys[i], value=found.value[i], siblings=[tuple(int(x) for x in s) for s in found.siblings(i)],
                   found=bool(found.found[i]), is_old0=bool(found.is_old0[i]), not_found_key=found.not_found_key[i],
                   not_found_value=found.not_found_value[i]) for i in range(k)])):
        t0 = time.perf_counter()
        ws, ps = zip(*[c.witness(p, rows_fn) for p in items])
        t_wit = time.perf_counter() - t0
        pc = pkg.ProverCircuit(ctx, c.data)
        batch = torch.stack([ctx.to_device(w) for w in ws]).contiguous()
        flats = pc.prove_batch(batch, list(ps))
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pc.prove_batch(batch, list(ps), out=flats)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 3
        rows = {}
        for gid, _ in c.data.builder.rows:
            rows[gid] = rows.get(gid, 0) + 1
        out[name] = {"degree_bits": c.data.degree_bits, "proofs_per_call": k, "proofs_per_s": k / dt, "ms_per_call": 1e3 * dt,
                     "arithmetic_rows": rows.get(circ.GATE_ARITHMETIC, 0), "poseidon_rows": rows.get(circ.GATE_POSEIDON, 0),
                     "host_witness_generation_ms_per_proof": 1e3 * t_wit / k}
        pc.free()
        del batch
    out["tree_entries"] = n_entries
    out["smt_apply_updates_ms"] = 1e3 * t_tree
    out["smt_find_batch_ms"] = 1e3 * t_find
    out["note"] = ("the reference's own gadgets restated on this repository's CircuitBuilder (row counts are this builder's: it does not fold "
                   "constants like plonky2's); every proof of this chain is checked against the oracle in tests/test_gpu_smt_circuit.py")
    return out


# The contract is ONE JSON line on stdout.  Libraries write there too (Gloo announces its ranks on std::cout, the OpenMP
# runtime of the oracle may warn), so the process's stdout is pointed at stderr for the whole run and the line goes to the
# saved descriptor.
_REAL_STDOUT = None


def _claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(out):
    sys.stdout.flush()
    (_REAL_STDOUT or sys.stdout).write(json.dumps(out) + "\n")
    (_REAL_STDOUT or sys.stdout).flush()


def launch_ranks(n_ranks, argv, check_devices=True):
    """`python bench.py --gpus N` run plainly (no WORLD_SIZE in the environment): start the N ranks ourselves, as a CHILD
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` (never an exec; this process has not touched the GPU:
    torch.cuda.device_count() does not initialise it), relay rank 0's JSON line to our stdout and return the child's exit code.
    Fails loudly when the box has fewer GPUs than ranks instead of printing a 1-GPU number."""
    import socket
    import subprocess

    if check_devices and os.environ.get("GLP_BENCH_SHARE_GPU") != "1":  # (tests run two gloo ranks on one GPU with that switch)
        import torch

        have = torch.cuda.device_count()
        if have < n_ranks:
            sys.stderr.write("bench.py: --gpus %d asked for, but this machine exposes %d GPU(s): refusing to run fewer ranks "
                             "than requested (no JSON line printed)\n" % (n_ranks, have))
            return 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t  # the ONE line of rank 0
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no JSON line\n")
        rc = 3
    if line is not None and rc == 0:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    return rc


def ranks_seen(dist_mod, device):
    """all-reduce of ones over the process group (on the device when the group runs on RCCL): how many ranks really took part"""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 1
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def dist_selftest(args):
    """`--workload dist-selftest`: launcher + process group only (gloo on a CPU-only box, RCCL on GPUs) — what
    tests/test_dist_gloo.py runs through `python bench.py --gpus 2` to cover the relay without a GPU.  Not a benchmark."""
    import torch

    dist_mod = importlib.import_module("intmax_zkp_core_amd.dist")
    rank, local_rank, world = dist_mod.env_world()
    pinned = dist_mod.pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))  # as the benchmark ranks do
    dev = None
    if torch.cuda.is_available() and os.environ.get("GLP_DIST_BACKEND") != "gloo":
        dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev)
    rank, local_rank, world = dist_mod.init(device_index=dev)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    device = "cuda:%d" % dev if dev is not None else None
    dist_mod.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        blobs = dist_mod.gather_indexed({j: bytes([j]) * 1000 for j in dist_mod.shard_units(8, rank, world)}, 8, device=device)
    dist_mod.barrier()
    elapsed, units = dist_mod.reduce_max_sum(time.perf_counter() - t0, args.steps * len(dist_mod.shard_units(8, rank, world)), device=device)
    seen = ranks_seen(dist_mod, device)
    per_rank = dist_mod.gather_floats(float(rank + 1), device=device)
    import ctypes

    import __graft_entry__ as g

    L = g.load_package().lib()
    L.glp_host_threads.restype = ctypes.c_uint32
    widths = dist_mod.gather_floats(float(L.glp_host_threads()), device=device)
    cpus = dist_mod.gather_floats(float(len(pinned)) if pinned else -1.0, device=device)
    if rank == 0:
        emit({"per_rank_probe": per_rank, "host_threads_per_rank_all": widths, "pinned_cpus_per_rank": cpus,
              "metric": "dist selftest (launcher + process group + the aggregation exchange; not a benchmark)", "value": units / elapsed,
              "unit": "units/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
              "config": {"workload": "dist-selftest", "dist_backend": torch.distributed.get_backend() if world > 1 else "none"},
              "ranks_seen": seen, "dist_backend": torch.distributed.get_backend() if world > 1 else "none",
              "exchange_ok": blobs == [bytes([j]) * 1000 for j in range(8)], "roofline": None, "cpu_baseline": None})
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("proof", "batch64", "aggtree", "block", "realagg", "dist-selftest"), default="proof")
    ap.add_argument("--blocks", type=int, default=16, help="block: blocks per step (a multiple of 2: one batch proof per 2 blocks; "
                                                           "16 = eight batches: whole batches for 1, 2, 4 or 8 ranks)")
    ap.add_argument("--block-sizes", default="13,16,15", help="block: log2 rows of the user-tx, block and batch circuits")
    ap.add_argument("--no-interleave", action="store_true", help="block: skip the interleaved pass")
    ap.add_argument("--no-wave", action="store_true", help="block: skip the level-by-level (throughput) pass")
    ap.add_argument("--pipeline", action="store_true", help="aggtree: two trees in flight (level 0 of tree t + 1 on a side thread / "
                                                            "context while tree t finishes its exchanges and upper levels)")
    ap.add_argument("--log-n", type=int, default=None, help="rows of the circuit (default 20 for proof, 16 for batch64 / aggtree)")
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--arity", type=int, default=2, help="realagg: children per parent")
    ap.add_argument("--real-circuits", action="store_true", help="block: the reference's real circuits (restated) with the prepared witnesses "
                    "of its test block instead of the synthetic traces")
    ap.add_argument("--no-big-block", action="store_true", help="skip the default run's 128-slot block circuit extra (minutes of host preparation)")
    ap.add_argument("--log-n-txs", type=int, default=7, help="--circuit block: log2 of the block's transaction slots (7: a 2^20-row circuit)")
    ap.add_argument("--circuit", choices=("mixed", "poseidon", "block"), default=None,
                    help="mixed: all 14 gate types of a recursive circuit (default); poseidon: the 4-gate PoseidonGate trace")
    ap.add_argument("--proofs", type=int, default=64, help="batch64 / aggtree: block proofs per step")
    ap.add_argument("--max-batch", type=int, default=32, help="batch64 / aggtree: proofs per glp_prove_batch_dev call (2^16 rows: 147 "
                    "proofs/s at 16, 150 at 32, 151.5 at 64; the wires forest of 32 proofs is 18 GB)")
    ap.add_argument("--queue", action="store_true", help="batch64: ranks pull proofs from one shared work queue instead of the static "
                                                         "round-robin split")
    ap.add_argument("--host-witness", action="store_true", help="block / batch64: ALSO time the PCIe-inclusive form — witnesses in "
                    "pinned host memory, proven through glp_prove_stream — reported under `host_resident_witnesses`, never `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra keys (4-gate circuit, small-circuit batches, host witness)")
    args = ap.parse_args()
    if args.circuit is None:
        # the default 2^20-row proof runs on the reference's OWN block circuit (128 slots, profiles/r05_block128.md); any other size or
        # workload, and GLP_BENCH_SYNTHETIC=1, take the synthetic 14-gate trace
        args.circuit = "block" if (args.workload == "proof" and args.log_n in (None, 20) and os.environ.get("GLP_BENCH_SYNTHETIC") != "1") else "mixed"
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # run plainly: become the launcher (nothing below has run, the GPU is untouched) and relay the ranks' line
        cpu_ranks = args.workload == "dist-selftest" and os.environ.get("GLP_DIST_BACKEND") == "gloo"
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], check_devices=not cpu_ranks))
    _claim_stdout()

    import numpy as np
    import torch

    import __graft_entry__ as g

    pkg = g.load_package()
    if args.workload == "dist-selftest":
        return dist_selftest(args)
    circ = importlib.import_module("intmax_zkp_core_amd.circuit")
    dist_mod = importlib.import_module("intmax_zkp_core_amd.dist")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libglprover has no CPU fallback")
    rank, local_rank, world = dist_mod.env_world()
    # one process per GPU on one host: rank r gets the r-th slice of the host's CPUs BEFORE the library creates its worker pool
    # (csrc/host_pool.h sizes itself from the affinity mask; unpinned it divides the hardware threads by LOCAL_WORLD_SIZE)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    pinned_cpus = None if os.environ.get("GLP_BENCH_NO_PIN") == "1" else dist_mod.pin_rank_to_cores(local_rank, local_world)
    dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)  # before the process group exists: RCCL binds each rank to its own GPU
    rank, local_rank, world = dist_mod.init(device_index=dev)  # backend "nccl" (= RCCL) on GPUs
    if world != args.gpus:  # a launcher that started another number of ranks than --gpus says: never print a mislabelled line
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE = %d (launch with --nproc-per-node = --gpus, or run bench.py plainly "
                         "and let it start its ranks itself)" % (args.gpus, world))
    share_gpu = os.environ.get("GLP_BENCH_SHARE_GPU") == "1"  # test switch: several ranks on one GPU (flagged in the line)
    if world > torch.cuda.device_count() and not share_gpu:
        raise SystemExit("bench.py: %d ranks for %d GPU(s): one rank per GPU" % (world, torch.cuda.device_count()))
    device = "cuda:%d" % dev if world > 1 else None
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 64))  # the oracle's per-column / per-leaf split stops scaling beyond this
    pkg.lib().glp_host_threads.restype = __import__("ctypes").c_uint32
    common = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "u64 (Goldilocks field, integer modular)",
              "host_threads_per_rank": int(pkg.lib().glp_host_threads()),
              "host_cpus_per_rank": len(pinned_cpus) if pinned_cpus else threads}
    if world > 1:  # evidence that N ranks really ran, on which backend (every rank takes part in the all-reduce)
        common["ranks_seen"] = ranks_seen(dist_mod, device)
        common["dist_backend"] = torch.distributed.get_backend()
        common["time_in_collective_ms_per_step"] = 0.0  # no data-path collective in this workload (aggtree overrides it)
        if world > torch.cuda.device_count():
            common["ranks_share_gpus"] = True  # NOT a multi-GPU measurement

    stream = torch.cuda.Stream(device=dev)
    if args.workload == "realagg":
        # BASELINE config 5 with nothing synthetic in it (workloads.run_real_aggregation): the reference's block proofs sharded over the
        # ranks, exchanged as bytes, aggregated by BlockBatchTarget circuits that verify their children.  A correctness run of the whole
        # distributed path on real circuits, not a throughput figure (parent witnesses are generated in Python inside the loop).
        workloads = importlib.import_module("intmax_zkp_core_amd.workloads")
        mods = [importlib.import_module("intmax_zkp_core_amd." + n) for n in ("recursion", "rollup", "codec")]
        with torch.cuda.stream(stream):
            ctx = pkg.Context(dev)
            t0 = time.perf_counter()
            res = workloads.run_real_aggregation(pkg, circ, mods[0], mods[1], mods[2], dist_mod, ctx, rank, world, n_blocks=args.blocks,
                                                 arity=args.arity, device=device)
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        if rank == 0:
            out = dict(common)
            ok = None
            if not args.no_verify:
                from oracle import oracle as orc

                ok = orc.plonk_verify(res["cd"], res["pc_cap"], res["root_pis"], res["root_flat"]) == 0
            out.update({"metric": "aggregation of %d real block proofs (arity %d), seconds per tree incl. host witness generation" % (args.blocks, args.arity),
                        "value": elapsed, "unit": "s", "higher_is_better": False, "steps": 1, "warmup": 0, "ms_per_step": 1e3 * elapsed,
                        "data": "synthetic transactions: the test block of the reference's src/bin/block_circuit.rs in every leaf",
                        "config": {"workload": "recursive aggregation tree over %d proofs of the reference's block circuit (make_block_proof_circuit at its "
                                               "test constants), parents = BlockBatchTarget circuits verifying %d children each, one byte exchange "
                                               "per level" % (args.blocks, args.arity),
                                   "dist_backend": (torch.distributed.get_backend() if world > 1 else "none (single process)")},
                        "levels": [{"level": r[0], "degree_bits": r[1], "proofs": r[2], "prove_ms_this_rank": r[3], "host_witness_ms_this_rank": r[4],
                                    "exchange_ms": r[5], "bytes": r[6]} for r in res["levels"]],
                        "root_public_inputs": res["root_pis"], "leaf_public_inputs": res["leaf_public_inputs"],
                        "proof_verified_by_restated_verifier": ok, "roofline": None, "cpu_baseline": None})
            emit(out)
        ctx.close()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if args.workload == "block":
        workloads = importlib.import_module("intmax_zkp_core_amd.workloads")
        codec = importlib.import_module("intmax_zkp_core_amd.codec")
        ul, bl, tl = (int(x) for x in args.block_sizes.split(","))
        with torch.cuda.stream(stream):
            res = workloads.run_block(pkg, circ, codec, dist_mod, dev, rank, world, n_blocks=args.blocks, user_log_n=ul, block_log_n=bl,
                                      batch_log_n=tl, steps=args.steps, warmup=args.warmup, interleave=not args.no_interleave,
                                      device=device, wave=not args.no_wave, real=args.real_circuits)
            res_h = None
            if args.host_witness:  # the PCIe-inclusive form: every witness starts in pinned host memory (where the reference's witness
                # generators leave it, src/bin/block_circuit.rs:293-361,410-434) and crosses PCIe inside the timed region
                res_h = workloads.run_block(pkg, circ, codec, dist_mod, dev, rank, world, n_blocks=args.blocks, user_log_n=ul,
                                            block_log_n=bl, batch_log_n=tl, steps=args.steps, warmup=args.warmup, interleave=False,
                                            device=device, wave=not args.no_wave, host_witness=True, real=args.real_circuits)
        if rank == 0:
            out = dict(common)
            n_blocks = args.blocks * args.steps
            out.update({
                "metric": "blocks/s, the reference's per-block proof set (3 user-tx + 2 signature + 1 block proof, 1 batch proof per 2 "
                          "blocks; 4 circuits resident, dependency chain respected)",
                "value": n_blocks / res["elapsed_s"], "unit": "blocks/s", "ms_per_step": 1e3 * res["elapsed_s"] / args.steps,
                "scaling": "strong", "data": "synthetic (witnesses cached in HBM before timing; a parent's witness is chosen by a digest "
                                             "of its children's serialized proof bytes)",
                "config": {"workload": "reference block production (src/bin/block_circuit.rs:293-442,665-710, src/rollup/circuits/mod.rs:"
                                       "1272-1326): per block 3 user-tx proofs and 2 signature proofs (two glp_prove_batch_dev calls), then "
                                       "the block proof, a batch proof every 2 blocks; " + (
                                           "ALL FOUR CIRCUITS ARE THE REFERENCE'S OWN (restated on this repository's builder: make_user_proof_circuit, "
                                           "zkdsa, make_block_proof_circuit, BlockBatchTarget at the constants of its tests), every witness is one of those "
                                           "of the block of src/bin/block_circuit.rs, prepared once; " if args.real_circuits else
                                           "user-tx / block / batch circuits are SYNTHETIC traces of the estimated sizes, the signature circuit is the "
                                           "reference's own; ") + "whole batches round-robin over the ranks, no exchange",
                           "blocks_per_step": args.blocks, "circuits": res["circuits"], "proofs_per_block": res["proofs_per_block"],
                           "mode": "`value` = the fastest of the forms run: sequential (one block after the other: the latency figures), "
                                   "interleaved (leaf proofs of block k+1 on a second context / stream / host thread under block k's "
                                   "block proof), wave (the step's independent blocks level by level, one batched call per circuit "
                                   "and level)"},
                "sequential_blocks_per_s": n_blocks / res["sequential_elapsed_s"],
                "interleaved_blocks_per_s": (n_blocks / res["interleaved_elapsed_s"]) if "interleaved_elapsed_s" in res else None,
                "wave_blocks_per_s": (n_blocks / res["wave_elapsed_s"]) if "wave_elapsed_s" in res else None,
                "value_form": res.get("value_form"),  # which of the forms produced `value` (the fastest that ran)
                "stage_ms_per_block": res["stage_ms_per_block"], "block_latency_ms": res["block_latency_ms"],
                "proofs_per_s": res["proofs_per_block"] * n_blocks / res["elapsed_s"],
                "per_rank_ms": res["per_rank_ms"], "roofline": None, "cpu_baseline": None,
            })
            if res_h is not None:
                out["host_resident_witnesses"] = {
                    "blocks_per_s": n_blocks / res_h["elapsed_s"], "form": res_h.get("value_form"),
                    "sequential_blocks_per_s": n_blocks / res_h["sequential_elapsed_s"],
                    "wave_blocks_per_s": (n_blocks / res_h["wave_elapsed_s"]) if "wave_elapsed_s" in res_h else None,
                    "block_latency_ms": res_h["block_latency_ms"], "stage_ms_per_block": res_h["stage_ms_per_block"],
                    "over_device_resident": res["elapsed_s"] / res_h["elapsed_s"],
                    "same_sample_bytes": bool(res["sample"] == res_h["sample"]),
                    "note": "PCIe-inclusive, never `value`: every witness starts in pinned host memory and is proven through "
                            "glp_prove_stream (groups of witnesses uploaded on the copy stream while the previous group proves)"}
            ok = None
            if not args.no_verify and res["sample"] is not None:
                from oracle import oracle as orc

                flat, pis = codec.flat_proof_from_bytes(res["sample"], res["sample_layout"], res.get("sample_n_pis", 8))
                ok = orc.plonk_verify(res["sample_cd"], res["sample_cap"], pis, flat) == 0
            out["proof_verified_by_restated_verifier"] = ok
            emit(out)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if args.workload != "proof":
        workloads = importlib.import_module("intmax_zkp_core_amd.workloads")
        codec = importlib.import_module("intmax_zkp_core_amd.codec")
        log_n = args.log_n or 16
        with torch.cuda.stream(stream):
            ctx = pkg.Context(dev)
            real_lc = None
            if args.workload == "batch64" and args.real_circuits:  # the 64 block proofs are proofs of the reference's REAL block circuit
                mods = [importlib.import_module("intmax_zkp_core_amd." + n) for n in ("recursion", "rollup")]
                real_lc = workloads.real_block_set(pkg, circ, mods[0], mods[1], ctx)["block"]
                log_n = real_lc.data.degree_bits
            if args.workload == "batch64":
                res = workloads.run_batch64(pkg, circ, dist_mod, ctx, rank, world, args.proofs, log_n, args.steps, args.warmup,
                                            args.circuit, max_batch=args.max_batch, device=device, dynamic=args.queue, lc=real_lc)
                res_h = None
                if args.host_witness:  # PCIe-inclusive: pinned host witnesses through glp_prove_stream (never `value`)
                    res_h = workloads.run_batch64(pkg, circ, dist_mod, ctx, rank, world, args.proofs, log_n, args.steps, args.warmup,
                                                  args.circuit, max_batch=args.max_batch, device=device, dynamic=args.queue,
                                                  host_witness=True, lc=real_lc)
            else:
                res_h = None
                real_levels = None
                if args.real_circuits:  # every level a real circuit: the reference's block circuit, then BlockBatchTarget circuits verifying 8 children
                    mods = [importlib.import_module("intmax_zkp_core_amd." + n) for n in ("recursion", "rollup")]
                    real_levels = workloads.real_aggregation_set(pkg, circ, mods[0], mods[1], ctx, args.proofs, 8)
                    log_n = real_levels[0].data.degree_bits
                res = workloads.run_aggtree(pkg, circ, codec, dist_mod, ctx, rank, world, args.proofs, 8, log_n, log_n, args.steps,
                                            args.warmup, args.circuit, max_batch=args.max_batch, device=device, pipeline=args.pipeline,
                                            real_levels=real_levels)
        if rank == 0:
            n_blocks = args.proofs * args.steps
            if args.workload == "batch64" and res["proofs"] != n_blocks:
                raise SystemExit("bench.py: %d proofs were proven, %d expected" % (res["proofs"], n_blocks))
            out = dict(common)
            out.update({
                "metric": "block proofs/s, %d independent 2^%d-row proofs per step%s" % (
                    args.proofs, log_n, "" if args.workload == "batch64" else " aggregated by a tree of arity 8 (whole tree per step)"),
                "value": n_blocks / res["elapsed_s"], "unit": "proofs/s", "ms_per_step": 1e3 * res["elapsed_s"] / args.steps,
                "scaling": "strong",
                "data": ("synthetic (%s circuit, random witnesses cached in HBM before timing)" % args.circuit) if not args.real_circuits
                        else "synthetic transactions: the reference's block circuit (make_block_proof_circuit at its test constants, restated) with the "
                             "witness of its test block, cached in HBM before timing",
                "real_circuits": bool(args.real_circuits),
                "config": {"workload": "BASELINE configs[%d]: %s" % (
                    3 if args.workload == "batch64" else 4,
                    ("64 independent block proofs pulled by the ranks from one shared work queue (TCPStore counter), no data-path "
                     "collective" if args.queue else
                     "64 independent block proofs sharded round-robin over the GPUs, no collective") if args.workload == "batch64" else
                    "aggregation tree over 64 block proofs, arity 8, SYNTHETIC PARENTS (a parent's witness is chosen by a digest of "
                    "its children's proof bytes — a data dependency, not plonky2's set_proof_with_pis_target filling, which is host "
                    "Rust): per level one batch of proofs per rank, proofs serialised in plonky2's byte format, one all_gather of "
                    "padded uint8 buffers"),
                    "degree_bits": log_n, "circuit": args.circuit, "proofs_per_step": args.proofs,
                    "prover": "glp_prove_batch_dev, up to %d proofs per call" % args.max_batch,
                    "dist_backend": (torch.distributed.get_backend() if world > 1 else "none (single process)")},
            })
            orc_ok = None
            if not args.no_verify:
                from oracle import oracle as orc

                if args.workload == "batch64" and res["sample"] is not None:
                    orc_ok = orc.plonk_verify(res["cd"], res["pc_cap"], res["sample"][1], res["sample"][0]) == 0
                elif args.workload == "aggtree":
                    orc_ok = orc.plonk_verify(res["cd"], res["pc_cap"], res["root_pis"], res["root_flat"]) == 0
            out["proof_verified_by_restated_verifier"] = orc_ok
            out["per_rank_ms"] = res["per_rank_ms"]  # each rank's own work per step (before the closing barrier), rank order
            if "per_rank_proofs" in res:
                out["per_rank_proofs"] = res["per_rank_proofs"]
            if args.workload == "aggtree":
                out["levels"] = [{"level": r[0], "circuit": r[1], "proofs": r[2], "prove_ms_max_over_ranks": r[3], "exchange_ms": r[4],
                                  "bytes_exchanged": r[5], "wait_for_slowest_rank_ms": r[6]} for r in res["levels"]]
                out["time_in_collective_ms_per_step"] = sum(r[4] for r in res["levels"])
                out["idle_at_level_barriers_ms_per_step"] = sum(r[6] for r in res["levels"])  # rank 0's view, last tree
                out["config"]["pipelined"] = bool(args.pipeline)
                out["root_proof_bytes"] = res["root_bytes"]
            if args.workload == "batch64" and res_h is not None:
                import numpy as np

                out["host_resident_witnesses"] = {
                    "proofs_per_s": n_blocks / res_h["elapsed_s"], "ms_per_step": 1e3 * res_h["elapsed_s"] / args.steps,
                    "over_device_resident": res["elapsed_s"] / res_h["elapsed_s"],
                    "same_sample_proof": bool(res["sample"] is not None and res_h["sample"] is not None
                                              and np.array_equal(res["sample"][0], res_h["sample"][0])),
                    "witness_bytes_per_proof": 8 * 135 << log_n,
                    "note": "PCIe-inclusive, never `value`: the %d witnesses of a step start in pinned host memory and are proven "
                            "through glp_prove_stream (groups of witnesses uploaded on the copy stream while the previous group proves)"
                            % args.proofs}
            out["roofline"] = None
            out["cpu_baseline"] = None
            emit(out)
        ctx.close()
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    log_n = args.log_n or 20
    with torch.cuda.stream(stream):
        ctx = pkg.Context(dev)  # binds the context to this torch stream
        circuit_fallback = None
        if args.circuit == "block":
            # the reference's OWN block production circuit (restated: rollup.BlockProductionCircuit) with 2^log_n_txs transaction slots and
            # the block of src/bin/block_circuit.rs as its witness; ~1.5 minutes of host preparation ONCE PER HOST (prepare_block_circuit: rank 0
            # builds, the host's cache keeps the tables under a content key, every rank and every later run loads them), then the same timed
            # region.  If the preparation fails every rank falls back to the synthetic trace of the same size together and says so in
            # `config.workload` — the driver never loses its line to a host-side problem.
            try:
                lc = prepare_block_circuit(pkg, circ, ctx, dist_mod, rank, world, args.log_n_txs)
                log_n = lc.data.degree_bits
            except Exception as e:  # noqa: BLE001
                circuit_fallback = repr(e)[:300]
                print("[bench] block circuit preparation failed (%s): falling back to the synthetic trace" % circuit_fallback, file=sys.stderr)
                args.circuit = "mixed"
                log_n = args.log_n or 20
        if args.circuit != "block":
            mk = circ.MixedGateCircuit if args.circuit == "mixed" else circ.LayeredPoseidonCircuit
            lc = mk(log_n, args.layers)
        cd = lc.data
        pc = pkg.ProverCircuit(ctx, cd)  # CircuitBuilder::build's commitment (once per circuit, not per proof)
        wires, pis = lc.witness_device(ctx, pkg, seed=rank)  # a different synthetic witness per rank

        for _ in range(args.warmup):
            pc.prove(wires, pis)
        torch.cuda.synchronize()
        dist_mod.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            proof = pc.prove(wires, pis)
        torch.cuda.synchronize()
        busy = time.perf_counter() - t0
        dist_mod.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0

        # per-phase GPU time of one more proof, outside the timed region (two hipEvents around each of its ~220 launches)
        ctx.timing_enable(True)
        ctx.timing_reset()
        pc.prove(wires, pis)
        timing = ctx.timing_read()
        ctx.timing_enable(False)

        # dominant kernel, isolated: the C = 135 leaf-hash launch of the wires commitment, timed live with HIP events on
        # the context's stream
        ctx.timing_enable(True)
        ctx.timing_reset()
        wb = pkg.PolynomialBatch.from_values(ctx, wires, cd.cfg.rate_bits, False, cd.cfg.cap_height)
        t135 = ctx.timing_read()
        ctx.timing_enable(False)

        # the quotient evaluation, isolated (its kernels are the only "plonk" launches of compute_quotient_polys)
        tq = None
        if rank == 0:
            rng = np.random.default_rng(1)
            betas, gammas, alphas = (rng.integers(0, pkg.P, 2, dtype=np.uint64) for _ in range(3))
            zs_vals = pc.partial_products(wires, betas, gammas)
            zb = pkg.PolynomialBatch.from_values(ctx, zs_vals, cd.cfg.rate_bits, False, cd.cfg.cap_height)
            pc.quotient_chunks(wb, zb, betas, gammas, alphas, np.zeros(4, dtype=np.uint64))
            ctx.timing_enable(True)
            ctx.timing_reset()
            pc.quotient_chunks(wb, zb, betas, gammas, alphas, np.zeros(4, dtype=np.uint64))
            tq = ctx.timing_read()
            ctx.timing_enable(False)
            zb.free()
            del zs_vals
        wb.free()

        # the HBM-side kernels, isolated: forward NTT of the 135 wire columns (16 n bytes per column)
        work = wires.clone()
        ctx.ntt(work, cd.degree_bits)
        ctx.timing_enable(True)
        tntt = None
        for _ in range(5):  # the median repetition, presented as three of it (the consumer below divides by three)
            ctx.timing_reset()
            ctx.ntt(work, cd.degree_bits)
            one = ctx.timing_read()
            tot = one.get("ntt", {"ms": 0})["ms"] + one.get("reorder", {"ms": 0})["ms"]
            tntt = (tntt or []) + [(tot, one)]
        tntt = sorted(tntt, key=lambda x: x[0])[2][1]
        for v in tntt.values():
            v["ms"] *= 3.0
        ctx.timing_enable(False)
        del work

        extras = {}
        big_block_check = None
        if rank == 0 and world == 1 and not args.no_extras:
            # (a) host-resident witnesses: pinned memory, upload of witness k + 1 overlapping proof k (glp_prove_stream)
            pinned = wires.cpu().pin_memory()
            pc.prove_stream([pinned, pinned], [pis, pis])
            ctx.synchronize()
            t0 = time.perf_counter()
            ps = pc.prove_stream([pinned] * 4, [pis] * 4)
            ctx.synchronize()
            t_stream = (time.perf_counter() - t0) / 4
            assert (ps[3] == proof).all()
            t0 = time.perf_counter()
            pc.prove(pinned.numpy().view(np.uint64), pis)  # glp_prove from ONE pinned array: column chunks straight off the caller's buffer
            t_sync = time.perf_counter() - t0
            # the witness as plonky2 holds it when witness generation ends (witness.wire_values: Vec<Vec<F>>): 135 SEPARATE PAGEABLE
            # columns, through glp_prove_cols — the PCIe-inclusive single-proof latency of the boundary as the reference would call it
            cols = [np.array(c_, copy=True) for c_ in pinned.numpy().view(np.uint64)]
            p_cols = pc.prove_cols(cols, pis)
            t_cols = []
            for _ in range(3):
                ctx.synchronize()
                t0 = time.perf_counter()
                p_cols = pc.prove_cols(cols, pis)
                t_cols.append(time.perf_counter() - t0)
            dev_ms = 1e3 * elapsed / args.steps
            extras["single_proof_latency_ms_from_host"] = 1e3 * min(t_cols)
            extras["host_resident_witness"] = {
                "pageable_columns_single_proof_ms": [round(1e3 * t, 2) for t in t_cols],
                "pageable_columns_proof_equals_device_resident": bool((p_cols == proof).all()),
                "pageable_columns_minus_device_resident_ms": 1e3 * min(t_cols) - dev_ms,
                "pinned_pipelined_ms_per_proof": 1e3 * t_stream, "single_proof_from_host_pointer_ms": 1e3 * t_sync,
                "device_resident_ms_per_proof": dev_ms,
                "note": "single_proof_latency_ms_from_host = glp_prove_cols from %d separate pageable numpy columns (plonky2's "
                        "witness.wire_values layout, 1.13 GB at 2^20 rows): the library's worker pool copies the columns through a ring of "
                        "pinned slots while the previous slot is on the bus, and the wires commitment — inverse NTT, LDE and the leaf sponge "
                        "in 8-column blocks — follows the chunks as they land (DESIGN 3.8).  single_proof_from_host_pointer_ms: glp_prove from "
                        "one pinned array.  pinned_pipelined: glp_prove_stream over 4 pinned witnesses, witness k + 1 uploaded by the copy "
                        "engine while proof k runs.  Never part of `value`" % len(cols)}
            del cols
            del pinned
        cs_cap = pc.constants_sigmas_cap.elements
        pc.free()
        wires_host = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            wires_host = wires.cpu().numpy().view(np.uint64)  # the CPU baseline proves the SAME witness
        del wires
        if rank == 0 and world == 1 and not args.no_extras:
            # (b) the 4-gate PoseidonGate-only circuit of round 1, same size
            if args.circuit in ("mixed", "block"):
                lp = circ.LayeredPoseidonCircuit(log_n, args.layers)
                pp = pkg.ProverCircuit(ctx, lp.data)
                wp, pip = lp.witness_device(ctx, pkg, seed=0)
                pp.prove(wp, pip)
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    pp.prove(wp, pip)
                ctx.synchronize()
                tp = (time.perf_counter() - t0) / 3
                extras["poseidon_only_circuit"] = {"ms_per_step": 1e3 * tp, "proofs_per_s": 1 / tp, "gates": 4,
                                                   "note": "the 4-gate layered PoseidonGate trace (round 1's headline workload)"}
                pp.free()
                del wp
            # (c) the reference's real circuit sizes: one proof at a time vs batched proving
            extras["small_circuit_batches"] = small_circuit_batches(pkg, circ, ctx)
            # (c2) two of the reference's own gadget circuits, fed by the device-side tree, through the batched prover
            extras["reference_gadget_chain"] = reference_gadget_chain(pkg, circ, ctx)
            # (c3) one block of the reference through its REAL circuits (user transactions, signatures, the block circuit with its eight
            # recursive verifications, the batch circuit): DESIGN 3.10
            try:
                import importlib as _il

                _m = [_il.import_module(pkg.__name__ + "." + n) for n in ("recursion", "rollup", "workloads")]
                extras["reference_block_pipeline"] = _m[2].reference_block_pipeline(pkg, circ, _m[0], _m[1], ctx)
            except Exception as e:  # an extra never costs the driver its line
                extras["reference_block_pipeline"] = {"error": repr(e)[:300]}
            # (c3b) when the timed circuit is the reference's block circuit: the synthetic 14-gate trace of rounds 1-4 beside it
            if args.circuit == "block":
                try:
                    lm0 = circ.MixedGateCircuit(log_n, args.layers)
                    pm0 = pkg.ProverCircuit(ctx, lm0.data)
                    wm0, pim0 = lm0.witness_device(ctx, pkg, seed=0)
                    pm0.prove(wm0, pim0)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        pm0.prove(wm0, pim0)
                    torch.cuda.synchronize()
                    tm0 = (time.perf_counter() - t0) / 3
                    extras["synthetic_14_gate_trace_same_size"] = {
                        "ms_per_step": 1e3 * tm0, "proofs_per_s": 1 / tm0,
                        "note": "circuit.MixedGateCircuit: the synthetic trace `value` was quoted on through round 4 (~65 % PoseidonGate rows in "
                                "layers + the other 13 gates' rows with random valid witnesses); `--circuit mixed` times it as the main workload"}
                    pm0.free()
                    del wm0
                except Exception as e:  # noqa: BLE001
                    extras["synthetic_14_gate_trace_same_size"] = {"error": repr(e)[:300]}
            # (c4) BASELINE's headline size on the reference's OWN circuit: make_block_proof_circuit with 128 transaction slots is a
            # 2^20-row circuit (this builder); built and filled on the host (~2 minutes of Python), proven on the GPU, checked by the
            # restated verifier in the cpu_baseline leg below.  N = 1 default runs only; --no-big-block skips it.
            if args.circuit == "mixed" and log_n == 20 and world == 1 and not args.no_big_block:
                try:
                    rb = _m[2].ReferenceBlockCircuit(pkg, circ, _m[0], _m[1], ctx, 7)
                    pb = pkg.ProverCircuit(ctx, rb.data)
                    wb, pib = rb.witness_device(ctx, pkg)
                    proof_b = pb.prove(wb, pib)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        proof_b = pb.prove(wb, pib)
                    torch.cuda.synchronize()
                    tb = (time.perf_counter() - t0) / 3
                    extras["reference_block_circuit_128_slots"] = {
                        "degree_bits": rb.data.degree_bits, "rows_used": rb.rows_used, "gate_types": len(rb.data.gates),
                        "transaction_slots": rb.n_slots, "ms_per_proof": 1e3 * tb, "proofs_per_s": 1 / tb,
                        "host_circuit_build_s": round(rb.build_s, 1), "host_witness_generation_s": round(rb.witness_s, 1),
                        "note": "the reference's make_block_proof_circuit (src/rollup/circuits/mod.rs:389-624) restated on this repository's "
                                "builder, 128 slots each verifying one user-transaction proof and one signature proof, filled with the block of "
                                "src/bin/block_circuit.rs: BASELINE's 2^20-row rollup circuit as a real circuit of the reference; never `value`"}
                    big_block_check = (rb.data, np.array(pb.constants_sigmas_cap.elements, dtype=np.uint64), pib, proof_b)
                    pb.free()
                    del wb
                except Exception as e:
                    extras["reference_block_circuit_128_slots"] = {"error": repr(e)[:300]}
            # (d) the OPT-IN matrix-core MDS variant (glp_ctx_set_mds_variant(2); never the default: north_star = non-MFMA path),
            # measured beside the default so that the trade can be judged (profiles/r03_poseidon_alternatives.md)
            if args.circuit in ("mixed", "block"):
                lm = circ.MixedGateCircuit(log_n, args.layers)
                pm = pkg.ProverCircuit(ctx, lm.data)
                wm, pim = lm.witness_device(ctx, pkg, seed=rank)
                ctx.set_mds_variant(2)
                try:
                    pm2 = pm.prove(wm, pim)
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        pm.prove(wm, pim)
                    ctx.synchronize()
                    tm = (time.perf_counter() - t0) / 3
                finally:
                    ctx.set_mds_variant(0)
                extras["opt_in_matrix_core_mds_variant"] = {
                    "ms_per_step": 1e3 * tm, "proofs_per_s": 1 / tm, "proof_identical_to_default": bool((pm2 == proof).all()),
                    "note": "NOT the measured configuration: the six full MDS layers of every permutation as int8 MFMA products on byte "
                            "planes in the wide lane-per-state hashing launches; everything else unchanged"}
                # (e) three proofs in flight (three contexts on their own streams, three host threads): what the phases that do not
                # saturate the VALU (tree tops, transcript round trips, openings) leave on the table for a node that proves a
                # stream of large proofs; `value` stays the one-proof-at-a-time rate
                import threading

                provers = []
                for _ in range(3):
                    c3 = pkg.Context(dev, use_torch_stream=False)
                    provers.append((c3, pkg.ProverCircuit(c3, lm.data)))
                torch.cuda.synchronize()
                for c3, p3 in provers:
                    p3.prove(wm, pim)
                reps = 3

                def stream_of_proofs(p3):
                    torch.cuda.set_device(dev)  # per thread, starts at 0
                    for _ in range(reps):
                        p3.prove(wm, pim)

                t0 = time.perf_counter()
                ths = [threading.Thread(target=stream_of_proofs, args=(p3,)) for _, p3 in provers]
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                for c3, _ in provers:
                    c3.synchronize()
                t3 = time.perf_counter() - t0
                extras["three_proofs_in_flight"] = {"proofs_per_s": 3 * reps / t3, "ms_per_proof_of_throughput": 1e3 * t3 / (3 * reps),
                                                    "note": "three contexts / streams / host threads proving the same 2^20-row circuit "
                                                            "concurrently (60 GB of HBM in use); never `value`"}
                for c3, p3 in provers:
                    p3.free()
                    c3.close()
                pm.free()
                del wm

    total_elapsed, total_steps = dist_mod.reduce_max_sum(elapsed, args.steps, device="cuda:%d" % dev if world > 1 else None)
    per_rank_ms = [1e3 * b / max(1, args.steps) for b in dist_mod.gather_floats(busy, device="cuda:%d" % dev if world > 1 else None)]

    if rank == 0:
        lde_bits = cd.degree_bits + cd.cfg.rate_bits
        leaf_ms = t135["leaf_hash"]["ms"] / max(1, t135["leaf_hash"]["launches"])
        alg = leaf_hash_algorithmic_bytes(cd.cfg.num_wires, lde_bits)
        achieved = alg / (leaf_ms * 1e-3) / 1e9
        perms_leaf = (1 << lde_bits) * 17
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "leaf_hash_traffic.json")
        if os.path.exists(tpath) and log_n == 20:
            try:
                rec = json.load(open(tpath))
                traffic, traffic_source = rec.get("hbm_bytes_per_launch"), rec.get("source")
            except Exception:
                traffic = None
        n_gates = len(cd.gates)
        out = dict(common)
        out.update({
            "metric": ("proofs/s, 2^%d-row circuit (CircuitData::prove after witness generation, SURVEY 8a rows a2-a10: 3 commitments, "
                       "quotient, openings, FRI; synthetic trace of the block circuit's shape, %d gate types)" % (log_n, n_gates)) if args.circuit != "block" else
                      ("proofs/s, 2^%d-row circuit (CircuitData::prove after witness generation, SURVEY 8a rows a2-a10): the reference's block "
                       "production circuit with %d transaction slots, restated; %d gate types" % (log_n, lc.n_slots, n_gates)),
            "value": total_steps / total_elapsed,
            "unit": "proofs/s",
            "ms_per_step": 1e3 * total_elapsed / args.steps,
            "data": ("synthetic (%s, random witness, HBM-resident before timing)" % (
                "14-gate recursive-circuit trace: ~65 % PoseidonGate rows in layers + the other 13 gates' rows" if args.circuit == "mixed"
                else "layered PoseidonGate circuit")) if args.circuit != "block" else
                    ("synthetic transactions: the test block of the reference's src/bin/block_circuit.rs (two user transactions, one signature, "
                     "one deposit; the other slots hold the default transaction) in make_block_proof_circuit with %d slots; witness "
                     "HBM-resident before timing" % lc.n_slots),
            "config": {"workload": ("one 2^%d-row proof under standard_recursion_config (BASELINE configs[2] shape; the Rust block "
                                    "circuit is replaced by a synthetic trace with the gate set of a circuit containing verify_proof, "
                                    "SURVEY.md 8d)" % log_n) if args.circuit != "block" else
                                   ("one 2^%d-row proof under standard_recursion_config: the reference's make_block_proof_circuit "
                                    "(src/rollup/circuits/mod.rs:389-624) restated on this repository's builder, %d transaction slots each "
                                    "verifying a user-transaction proof and a signature proof, %d of 2^%d rows used" % (log_n, lc.n_slots, lc.rows_used, log_n)),
                       "degree_bits": cd.degree_bits, "num_wires": 135, "rate_bits": 3, "cap_height": 4,
                       "proof_of_work_bits": 16, "num_query_rounds": 28, "gate_types": n_gates,
                       "poseidon_rows": (lc.layers * lc.width + 1) if args.circuit != "block" else lc.poseidon_rows,
                       "sharding": "independent proofs per GPU, no collective"},
            "circuit_fallback": circuit_fallback,  # not None: the block circuit could not be prepared, the synthetic trace was timed instead
            "block_circuit_preparation": getattr(lc, "preparation", None),  # built by rank 0 | found in the host's cache | loaded from rank 0's file
            "proof_verified_by_restated_verifier": None,
            "per_rank_ms": per_rank_ms,  # each rank's own time per proof (before the closing barrier), rank order
            "proof_bytes": int(proof.size * 8),
            "roofline": {"bound": "hbm", "kernel": "merkle::%s (C=135, 2^%d leaves)" % ("leaf_hash_thru_kernel" if lde_bits >= 19 else "leaf_hash_kernel", lde_bits),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": leaf_ms,
                         "traffic_source": (("NOT measured in this run (counters need rocprofv3 around the process): the committed PMC record "
                                             "of the same kernel and launch shape — %s; `avg_launch_ms` beside it is THIS run's live timing on "
                                             "box %s" % (traffic_source, os.uname().nodename)) if traffic is not None else None),
                         "note": "Poseidon hashing is integer-VALU-bound, not HBM-bound (17 permutations per 1080-byte "
                                 "leaf); perms/s is the meaningful rate: 13.1 k VALU wave-instructions per permutation at 3.95 "
                                 "cycles each (VALUBusy 100 %: the attached `valu` object and its `valu_source`) bound it at "
                                 "~2.9 G perm/s at the 2.28 GHz the chip holds",
                         "poseidon_gperms_per_s": perms_leaf / (leaf_ms * 1e-3) / 1e9},
            "roofline_ntt": (lambda ms, by: {"bound": "hbm", "kernel": "ntt_fwd, 2^%d points x %d columns (three DIF passes, the "
                                             "last one storing natural order itself)" % (cd.degree_bits, cd.cfg.num_wires),
                                             "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                             "algorithmic_bytes": by, "ms": ms,
                                             "note": "three sweeps of 16 n C bytes each (the separate bit-reversal sweep of rounds "
                                                     "1-2 is gone: 2.13 -> 1.58 ms, profiles/r03_microbench.md); the two strided "
                                                     "passes are VALU-bound (profiles/r03n_valu.md; the LDE pass since its loads are issued together, profiles/r03k_lde_pmc.md)"})(
                (tntt.get("ntt", {"ms": 0})["ms"] + tntt.get("reorder", {"ms": 0})["ms"]) / 3.0,
                16 * cd.n * cd.cfg.num_wires),
            "phase_ms_per_step": {k: v["ms"] for k, v in timing.items() if v["launches"]},
            "perms_per_proof_commitments": perms_per_proof(cd),
        })
        if tq is not None:
            qms, qb = tq["plonk"]["ms"], quotient_algorithmic_bytes(cd)
            out["roofline_quotient"] = {
                "bound": "hbm", "kernel": "quotient evaluation on the 8n coset (%s)" % (
                    "plonk::quotient_fused_kernel: every column read once through an LDS tile" if n_gates >= 9 else
                    "plonk::poseidon_gate_kernel + plonk::quotient_kernel"),
                "achieved": qb / (qms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": qb / (qms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes": qb, "ms": qms,
                "note": "VALU-bound: ~81 k VALU instructions per 64 points for the 14 gate types (profiles/r03n_quotient_pmc.md, r03q_quotient_waits.md: "
                        "16.7 GB read, 0.13 GB written, VALUBusy 86 %)"}
        out.update(extras)
        # the VALU side of the same kernels from the committed PMC passes (profiles/valu_busy.json, tools/summarize_valu.py):
        # the path is integer-VALU-bound, so the HBM fractions above are small by construction and these say how full the
        # unit that does bound it was.  Only for the configuration the passes were taken on.
        vpath = os.path.join(ROOT, "profiles", "valu_busy.json")
        if os.path.exists(vpath) and log_n == 20 and args.circuit in ("mixed", "block"):
            try:
                vb = json.load(open(vpath))
                out["roofline"]["valu"] = vb.get("leaf_hash")
                if "roofline_quotient" in out:
                    out["roofline_quotient"]["valu"] = vb.get("quotient")
                out["roofline_ntt"]["valu_lde_passes"] = {k: vb.get(k) for k in ("lde_pass", "ntt_pass_ab", "ntt_pass_a", "ntt_pass_b") if vb.get(k)}
                out["roofline"]["valu_source"] = vb.get("source")
            except Exception:
                pass
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (the contract): N > 1 runs report null
            chk = None if args.no_verify else (cs_cap, proof)
            out["cpu_baseline"] = cpu_baseline(cd, args.circuit, log_n, wires_host, pis, chk)
            out["proof_verified_by_restated_verifier"] = out["cpu_baseline"].pop("gpu_proof_accepted_by_restated_verifier")
            if big_block_check is not None and not args.no_verify:  # the 128-slot block proof, by the same restated verifier
                from oracle import oracle as _orc

                _cd, _cap, _pis, _pf = big_block_check
                _orc.set_fast(False)
                if isinstance(out.get("reference_block_circuit_128_slots"), dict):
                    out["reference_block_circuit_128_slots"]["proof_verified_by_restated_verifier"] = _orc.plonk_verify(_cd, _cap, _pis, _pf) == 0
        else:
            out["cpu_baseline"] = None
        emit(out)
    ctx.close()
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
